"""Qop with the constructor of quant/Weight_Only/Qop.py:12."""
from ...qop import QopWeightOnly


class Qop(QopWeightOnly):
    pass
