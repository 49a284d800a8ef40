"""Qop with the constructor of quant/Weight_Activation/Qop.py:8."""
from ...qop import QopWeightActivation


class Qop(QopWeightActivation):
    pass
