"""Weight + activation quantization with calibrated output scaling (reference: quant/Weight_Activation)."""
from .Qop import Qop  # noqa: F401
from .Quantizer import Quantizer, QuantizedMaxPool2d, QuantizedAvgPool2d  # noqa: F401
from .Chunker import Chunker  # noqa: F401
