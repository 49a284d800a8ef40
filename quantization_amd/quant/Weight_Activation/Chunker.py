"""Chunker of quant/Weight_Activation/Chunker.py."""
from ...qlayers import WeightActivationChunker
from .Quantizer import Quantizer


class Chunker(WeightActivationChunker):
    quantizer_cls = Quantizer
