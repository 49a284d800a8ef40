"""The reference's CPU path restated with the same torch CPU operators it uses, kept as slow
as the reference is (a Python loop over output channels, five full-tensor passes, a per-call
int8 -> float cast of the whole weight).  This is the CPU BASELINE that bench.py times beside
the GPU path; it is also checked against the golden fixtures (tests/test_oracle_golden.py).

TEST INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline leg, never by
quantization_amd/.  Parity status: PINNED by tests/golden/*.npz (produced by the reference).
Paths cited are relative to the reference checkout.
"""
import torch
import torch.nn.functional as F

EPS = torch.tensor(torch.finfo(torch.float32).eps)


@torch.no_grad()
def quantize_channel_sym_int8(w):
    """Qop.quantize, channel / symmetric / int8 (quant/Weight_Only/Qop.py:74-102, 37-54, 139-162)."""
    maxima = [torch.amax(w[i]) for i in range(w.shape[0])]           # Qop.py:81-87: one dispatch per row
    max_val = torch.tensor(maxima)                                    # CPU tensor, Qop.py:89
    min_val = torch.zeros_like(max_val)
    lo = torch.min(min_val, torch.tensor(0, dtype=min_val.dtype))
    hi = torch.max(max_val, torch.tensor(0, dtype=max_val.dtype))
    hi = torch.max(-lo, hi)
    scale = torch.max(hi / (255.0 / 2), EPS.clone())                  # Qop.py:50-51
    zp = torch.zeros(lo.size(), dtype=torch.int64)
    shape = [w.shape[0]] + [1] * (w.dim() - 1)
    scale, zp = scale.view(shape), zp.view(shape)
    t = w.detach().clone()                                            # Qop.py:147
    q = torch.round(t / scale.to(w.device)).clamp(-128, 127)          # Qop.py:155
    return q.type(torch.int8), scale, zp                              # Qop.py:162


@torch.no_grad()
def wo_linear_forward(x, wq, scale, zp, bias=None):
    """Weight-only Quantizer.forward, Linear (quant/Weight_Only/Quantizer.py:79-86, Qop.py:175-179)."""
    y = F.linear(x, wq.to(x.dtype))                                   # per-call cast of the whole weight
    if bias is not None:
        y = y + bias.view(1, -1)
    return scale.view(1, -1) * (y.float() - zp.view(1, -1))


@torch.no_grad()
def wo_conv2d_forward(x, wq, scale, zp, bias=None, stride=1, padding=0):
    """Weight-only Quantizer.forward, Conv2d (quant/Weight_Only/Quantizer.py:72-77,86)."""
    y = F.conv2d(x, wq.to(x.dtype), stride=stride, padding=padding)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return scale.view(1, -1, 1, 1) * (y.float() - zp.view(1, -1, 1, 1))
