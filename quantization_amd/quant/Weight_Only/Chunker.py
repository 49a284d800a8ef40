"""Chunker of quant/Weight_Only/Chunker.py."""
from ...qlayers import WeightOnlyChunker
from .Quantizer import Quantizer


class Chunker(WeightOnlyChunker):
    quantizer_cls = Quantizer
