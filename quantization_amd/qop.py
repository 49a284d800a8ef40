"""Qop: the quantization operator of the reference (quant/Weight_Only/Qop.py,
quant/Weight_Activation/Qop.py) with its tensor work done by HIP kernels.

Same constructor arguments, attributes and methods as the reference class.  What differs is
where the work happens:

* per-channel symmetric int8 (`affine='channel', affine_dim=0, symmetric=True`), the only
  configuration the reference's quantize() completes for, runs as ONE fused kernel
  (row maximum -> scale -> divide / round / clamp, a single pass over HBM);
* quantize() with parameters computed earlier, dequantize() and the row-maximum reduction
  of compute_scales_zero_point() each run as one kernel;
* calculate_params() is a handful of scalar/vector operations on N values and stays on the
  host, exactly where the reference runs it (its max_val tensor is built on the CPU).

Tensors that live in host memory are staged to the GPU and the results returned to the host;
there is no CPU implementation.  Configurations the reference itself cannot run (per-tensor
or group quantize(), asymmetric per-channel) raise the same class of error here.
"""
import torch
import torch.nn.functional as F

from . import _native


def _to_gpu(t):
    """Device copy of `t` for the kernels; raises if there is no GPU to run them on."""
    if t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise RuntimeError("quantization_amd needs a GPU: the quantization kernels have no CPU implementation")
    return t.cuda()


class QopBase:
    """Shared implementation; the two public classes differ only in constructor signature."""

    def _setup(self, dtype, w_a, affine, affine_dim, group_size, symmetric):
        self.symmetric = symmetric
        if dtype is not None:
            self.dtype = dtype
            info = torch.finfo if dtype.is_floating_point else torch.iinfo
            self.q_min = info(dtype).min
            self.q_max = info(dtype).max
        # torch.finfo(float32).eps as a 0-dim tensor (Weight_Only/Qop.py:22).
        self.eps = torch.tensor(torch.finfo(torch.float32).eps)
        self.min_val = None
        self._max_val = None
        self._max_val_dev = None
        self.w_a = w_a
        self.q_group_size = group_size
        self.affine = affine
        self.affine_dim = affine_dim
        self.scales = None
        self.zero_point = None
        self._q_int = None

    # max_val is produced on the GPU by the fused kernel; the host copy the reference keeps
    # (a CPU tensor) is made on first access so that quantize() never synchronises.
    @property
    def max_val(self):
        if self._max_val is None and self._max_val_dev is not None:
            self._max_val = self._max_val_dev.cpu()
        return self._max_val

    @max_val.setter
    def max_val(self, value):
        self._max_val = value
        self._max_val_dev = None

    # The reference keeps the rounded, clamped tensor in floating point next to the int8 result;
    # here the float view is materialised only if somebody reads it.
    @property
    def quantized_tensor(self):
        if self._q_int is None:
            return None
        return self._q_int.to(self.tensor.dtype if self.tensor.dtype.is_floating_point else torch.float32)

    @quantized_tensor.setter
    def quantized_tensor(self, value):
        self._q_int = value

    # ------------------------------------------------------------------ parameters (host)
    @torch.no_grad()
    def calculate_params(self):
        """Scale and zero point from self.min_val / self.max_val (Weight_Only/Qop.py:37-72)."""
        lo, hi = self.min_val, self.max_val
        zero_lo = torch.tensor(0, device=lo.device, dtype=lo.dtype)
        zero_hi = torch.tensor(0, device=hi.device, dtype=hi.dtype)
        lo = torch.min(lo, zero_lo)
        hi = torch.max(hi, zero_hi)
        dev = lo.device
        span = float(self.q_max - self.q_min)
        eps = self.eps.clone().detach().to(dev)
        zero_point = torch.zeros(lo.size(), dtype=torch.int64, device=dev)
        if self.symmetric:
            scale = torch.max(-lo, hi) / (span / 2)
            scale = torch.max(scale, eps)
            if self.dtype in (torch.quint8, torch.uint8):
                zero_point = zero_point.new_full(zero_point.size(), (self.q_min + self.q_max) // 2)
        elif self.affine == "channel" and self.affine_dim is not None:
            scale = (hi - lo) / span
            scale = torch.where(scale > self.eps, scale, torch.ones_like(scale))
            zero_point = (-1 * lo / scale).to(torch.int)
        else:
            scale = (hi - lo) / span
            scale = torch.max(scale, eps)
            zero_point = (self.q_min - torch.round(lo / scale)).to(torch.int)
            zero_point = torch.clamp(zero_point, self.q_min, self.q_max)
        if scale.dim() == 0:
            scale = torch.tensor([float(scale)], dtype=scale.dtype, device=dev)
        if zero_point.dim() == 0:
            zero_point = torch.tensor([int(zero_point)], dtype=zero_point.dtype, device=dev)
        return scale, zero_point

    @torch.no_grad()
    def compute_scales_zero_point(self, tensor=None):
        """Min/max statistics of `tensor`, then calculate_params() (Weight_Only/Qop.py:74-102)."""
        if self.min_val is None and self.max_val is None:
            if self.affine == "channel":
                if not self.symmetric:
                    # The reference reaches torch.tensor(None) here (Qop.py:92).
                    raise RuntimeError("Could not infer dtype of NoneType")
                if self.affine_dim != 0:
                    tensor[self.affine_dim]  # same indexing error as the reference for a bad dim
                    raise RuntimeError("per-channel statistics are taken over dim 0 only")
                t = _to_gpu(tensor.detach()).contiguous()
                # signed row maximum on the GPU; the reference's max_val is a CPU tensor (Qop.py:89)
                self.max_val = _native.row_max(t).to(tensor.dtype).cpu()
                self.min_val = torch.zeros_like(self.max_val)
            elif self.symmetric:
                self.max_val = tensor.abs().max()
                self.min_val = torch.tensor(0, device=tensor.device)
            else:
                self.max_val = tensor.max()
                self.min_val = tensor.min()
        return self.calculate_params()

    @torch.no_grad()
    def compute_scales_zero_point_dimension(self, tensor, dim=-1):
        """Parameters shaped for broadcasting along `dim` (Weight_Only/Qop.py:104-113)."""
        self.scales, self.zero_point = self.compute_scales_zero_point(tensor)
        shape = [1] * tensor.dim()
        shape[dim] = tensor.shape[dim]
        self.scales = self.scales.view(shape)
        self.zero_point = self.zero_point.view(shape)

    def compute_scale_zero_pointer(self):
        """Dispatch on the granularity (Weight_Only/Qop.py:115-125)."""
        if self.affine in ("tensor", "channel"):
            self.compute_scales_zero_point_dimension(self.tensor, dim=self.affine_dim)
        elif self.affine == "group":
            assert self.tensor.shape[1] % self.q_group_size == 0
            assert self.tensor.dim() == 2
            self.compute_scales_zero_point_dimension(self.tensor.view(-1, self.q_group_size), dim=0)

    def push_to_tensor_device(self, tensor_device):
        """Move scales and zero point to `tensor_device` (Weight_Only/Qop.py:127-137)."""
        for name in ("zero_point", "scales"):
            v = getattr(self, name)
            v = v.clone().detach() if isinstance(v, torch.Tensor) else torch.tensor(v)
            setattr(self, name, v.to(tensor_device))

    # ------------------------------------------------------------------------ tensor work
    def _fused_config(self):
        return (self.affine == "channel" and self.affine_dim == 0 and self.symmetric
                and getattr(self, "dtype", None) == torch.int8
                and self.min_val is None and self.max_val is None)

    def quantize(self, tensor):
        """int8 copy of `tensor` (Weight_Only/Qop.py:139-162)."""
        self.tensor = tensor
        self.tensor_shape = tensor.shape
        src = _to_gpu(tensor.detach()).contiguous()
        if (self.zero_point is None or self.scales is None) and self._fused_config() and src.dim() >= 1:
            q, scale, zp, row_max = _native.quantize_rows_sym_i8(src, want_row_max=True)
            shape = [src.shape[0]] + [1] * (src.dim() - 1)
            # 16-bit weights keep their dtype in the scales (the kernel returns them widened, exactly)
            self.scales = scale.to(tensor.dtype).view(shape).to(tensor.device)
            self.zero_point = zp.view(shape).to(tensor.device)
            self._max_val, self._max_val_dev = None, row_max.to(tensor.dtype)
            self.min_val = torch.zeros(src.shape[0], dtype=tensor.dtype)
        else:
            if getattr(self, "dtype", None) != torch.int8:
                raise NotImplementedError("only torch.int8 targets are implemented")
            if self.zero_point is None or self.scales is None:
                self.compute_scale_zero_pointer()
            self.push_to_tensor_device(tensor.device)
            if self.affine == "group":
                raise RuntimeError("group quantization is not implemented (the reference's quantize() fails for it too)")
            n_par = self.scales.numel()
            if n_par == 1:
                rows2d = src.reshape(1, -1)
            elif src.dim() >= 1 and self.scales.dim() == src.dim() and self.scales.shape[0] == n_par == src.shape[0]:
                rows2d = src.reshape(n_par, -1)
            else:
                raise RuntimeError(f"scales of shape {tuple(self.scales.shape)} do not broadcast along dim 0 of a "
                                   f"tensor of shape {tuple(src.shape)}")
            zp = None if self.symmetric else self.zero_point
            q = _native.quantize_rows_given_i8(rows2d, self.scales, zp).view(self.tensor_shape)
        q = q.to(tensor.device)
        self._q_int = q
        return q

    @torch.no_grad()
    def dequantize(self, quantized_tensor, activation=False):
        """scales * (float(q) - zero_point) (Weight_Only/Qop.py:164-181)."""
        self.push_to_tensor_device(quantized_tensor.device)
        if self.affine == "group":
            raise RuntimeError("group quantization is not reachable through quantize() and is not implemented")
        if activation and self.scales.shape[1] == 1:
            # a layer output carries the channel on axis 1: [N,1,..] -> [1,N,..] (kept, as the reference does)
            self.zero_point = self.zero_point.view([1, self.zero_point.shape[0], *self.zero_point.shape[2:]])
            self.scales = self.scales.view([1, self.scales.shape[0], *self.scales.shape[2:]])
        src = _to_gpu(quantized_tensor.detach()).contiguous()
        live = [i for i, n in enumerate(self.scales.shape) if n != 1]
        if len(live) > 1 or (live and (self.scales.dim() != src.dim() or self.scales.shape[live[0]] != src.shape[live[0]])):
            raise RuntimeError(f"scales of shape {tuple(self.scales.shape)} do not broadcast against "
                               f"{tuple(src.shape)} along one axis")
        if live == [0] and src.dtype == torch.int8:
            out = _native.dequantize_rows_i8(src, self.scales, self.zero_point)
        else:
            axis = live[0] if live else 0
            channels = src.shape[axis] if live else 1
            outer = 1
            for n in src.shape[:axis]:
                outer *= n
            y = src.to(torch.float32).reshape(outer if live else 1, channels, -1)
            out = _native.dequantize_channels_f32(y, self.scales, self.zero_point).view(src.shape)
        self.dequantized_tensor = out.to(quantized_tensor.device)
        return self.dequantized_tensor

    @torch.no_grad()
    def compute_dequantization_error(self, original_tensor, dequantized_tensor):
        """MSE after normalising by the dequantized tensor's max (Weight_Only/Qop.py:183-196)."""
        if torch.isinf(original_tensor).any() or torch.isinf(dequantized_tensor).any():
            print("Inf values detected")
        if torch.isnan(original_tensor).any() or torch.isnan(dequantized_tensor).any():
            print("NaN values detected")
        peak = dequantized_tensor.abs().max()
        if peak > 0:
            dequantized_tensor /= peak
        return F.mse_loss(original_tensor, dequantized_tensor)


class QopWeightOnly(QopBase):
    """Signature of quant/Weight_Only/Qop.py:12."""

    def __init__(self, dtype, w_a=None, affine='tensor', affine_dim=None, group_size=-1, symmetric=False):
        self._setup(dtype, w_a, affine, affine_dim, group_size, symmetric)


class QopWeightActivation(QopBase):
    """Signature of quant/Weight_Activation/Qop.py:8 (dtype optional, extra `bits`)."""

    def __init__(self, dtype=None, bits=None, w_a=None, affine='tensor', affine_dim=None, group_size=-1,
                 symmetric=False):
        self._setup(dtype, w_a, affine, affine_dim, group_size, symmetric)
