"""Weight-only static symmetric per-channel int8 quantization (reference: quant/Weight_Only)."""
from .Qop import Qop  # noqa: F401
from .Quantizer import Quantizer  # noqa: F401
from .Chunker import Chunker  # noqa: F401
