"""MI355X-native int8 quantization hot path (HIP kernels behind a C ABI) with the Python
surface of satabios/quantization's quant/ package."""
__version__ = "0.1.0"
