"""Quantizer of quant/Weight_Only/Quantizer.py."""
from ...qlayers import WeightOnlyQuantizer


class Quantizer(WeightOnlyQuantizer):
    pass
