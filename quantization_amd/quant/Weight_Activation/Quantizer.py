"""Quantizer and the int8 pooling wrappers of quant/Weight_Activation/Quantizer.py."""
from ...qlayers import QuantizedAvgPool2d, QuantizedMaxPool2d, WeightActivationQuantizer  # noqa: F401


class Quantizer(WeightActivationQuantizer):
    pass
