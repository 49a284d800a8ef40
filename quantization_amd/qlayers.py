"""Quantizer layers and Chunker model rewriters with the reference's interface
(quant/Weight_Only/{Quantizer,Chunker}.py, quant/Weight_Activation/{Quantizer,Chunker}.py).

forward():
  * int8 activations take the int8 MFMA path: ONE kernel computes the integer product and the
    layer's whole floating-point tail (bias, per-channel scale, output zero point);
  * floating-point activations (what the reference's layers are fed) multiply by the int8 weight
    as it is stored (widened in registers; the reference casts the whole weight per call,
    quant/Weight_Only/Quantizer.py:73,79), with the same tail fused;
  * Conv2d layers with dilation, groups, a stride or padding per axis, or padding='valid' / 'same'
    take the direct convolution kernel (miq_conv2d_general), int8 and floating-point alike.
Host tensors are staged to the GPU and results returned to the caller's device.
"""
import functools
import warnings

import torch
import torch.nn as nn

from . import _native
from .qop import QopWeightActivation, QopWeightOnly, _to_gpu


def _single(v):
    """Conv2d hyper-parameters arrive as (a, a) tuples; the int8 kernel takes one integer."""
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError(f"the int8 convolution needs square stride/padding, got {v}")
        return int(v[0])
    return int(v)


def _uniform(v):
    """True for an integer or a tuple whose entries are all equal."""
    return not isinstance(v, (tuple, list)) or len(set(v)) == 1


def _plain_conv(stride, padding, dilation, groups):
    """What the MFMA convolution kernels take: dilation 1, groups 1, one stride and one numeric padding for both axes."""
    return (not isinstance(padding, str) and _uniform(dilation) and _single(dilation) == 1 and groups == 1
            and _uniform(stride) and _uniform(padding))


class _QuantizerBase(nn.Module):
    def _init_common(self, in_features, out_features, kernel_size, stride, padding, dilation, groups, cnn,
                     data_metrics):
        self.in_features = in_features
        self.out_features = out_features
        self.cnn = cnn
        self.data_metrics = data_metrics
        self.kernel_size = kernel_size if cnn else None
        self.stride = stride if cnn else None
        self.padding = padding if cnn else None
        self.dilation = dilation if cnn else None
        self.groups = groups if cnn else None
        self.weight_shape = ((out_features, in_features, *self.kernel_size) if cnn
                             else (out_features, in_features))
        # Placeholder until from_float() installs the quantized weight (the reference registers a
        # random fp16 tensor of this shape, Quantizer.py:34).
        self.register_buffer("weight", torch.zeros(self.weight_shape, dtype=torch.int8))
        self._gpu_cache = {}      # device copies of host-resident weight / bias / scales, keyed by role
        self._beta_cache = None   # (output_zero_point tensor, its version, the value as a Python float)
        self._packed_cache = {}   # activation dtype -> (weight tensor, its version, packed filter): see _packed()
        self._params_stale = None # set to a message when a checkpoint without quantization parameters was loaded

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.weight = self.weight.to(*args, **kwargs)
        if self.bias is not None:
            self.bias = self.bias.to(*args, **kwargs)
        self._invalidate()
        return self

    def _invalidate(self):
        self._gpu_cache = {}
        self._beta_cache = None
        self._packed_cache = {}

    def _packed(self, w, act_dtype):
        """The Conv2d filter in the form the MFMA kernels stage from (miq_conv2d_i8_pack_weight /
        miq_conv2d_w8_pack_weight), made once per weight and activation type: the weight changes only in
        from_float (quant/Weight_Only/Quantizer.py:93-112), so no forward re-lays it.  `w` is the weight on the GPU."""
        cache = self.__dict__.setdefault("_packed_cache", {})
        ent = cache.get(act_dtype)
        if ent is None or ent[0] is not w or ent[1] != w._version:
            packed = (_native.conv2d_i8_pack_weight(w) if act_dtype == torch.int8
                      else _native.conv2d_w8_pack_weight(w, act_dtype))
            ent = cache[act_dtype] = (w, w._version, packed)
        return ent[2]

    def _prepack(self):
        """from_float's last step for plain Conv2d layers whose weight is already on the GPU: the packed filters for int8
        and fp32 activations (other activation types are packed on their first forward)."""
        if self.cnn and self.weight.is_cuda and _plain_conv(self.stride, self.padding, self.dilation, self.groups):
            self._packed(self.weight, torch.int8)
            self._packed(self.weight, torch.float32)

    def _on_gpu(self, role, t):
        """`t` on the GPU.  A module kept on the host would otherwise upload its whole int8 weight (and bias,
        scales) on every forward: the device copy is kept until the tensor is replaced or written in place."""
        if t is None or t.is_cuda:
            return t
        ent = self._gpu_cache.get(role)
        if ent is None or ent[0] is not t or ent[1] != t._version:
            ent = (t, t._version, _to_gpu(t.detach()))
            self._gpu_cache[role] = ent
        return ent[2]

    # ---- persisted quantization parameters ------------------------------------------------
    # The reference's state_dict carries the int8 `weight` buffer only: its scales live on a plain
    # Python Qop (Weight_Only/Quantizer.py:34,49) or on plain attributes (Weight_Activation/
    # Quantizer.py:97) and are lost on save.  Here they travel as the module's extra state, under
    # `<name>._extra_state`, next to the same `weight` key, so a saved quantized model reloads
    # without re-quantizing.  A state_dict without that key (one the reference wrote) still loads.
    def _quant_state(self):
        raise NotImplementedError

    def _set_quant_state(self, state):
        raise NotImplementedError

    def get_extra_state(self):
        def cpu(v):
            return v.detach().to("cpu").clone() if isinstance(v, torch.Tensor) else v
        return {k: cpu(v) for k, v in self._quant_state().items()}

    def set_extra_state(self, state):
        dev = self.weight.device
        self._set_quant_state({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in state.items()})
        self._params_stale = None
        self._invalidate()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # A checkpoint the reference wrote holds the int8 weight without its scales.  The reference's own flow is Chunker
        # on the pretrained float model, then load_state_dict of such a checkpoint of the SAME model: the incoming weight
        # then equals the one this module's scales were computed for, and they stay valid.
        incoming = state_dict.get(prefix + "weight")
        same_weight = (isinstance(incoming, torch.Tensor) and incoming.shape == self.weight.shape
                       and incoming.dtype == self.weight.dtype
                       and bool(torch.equal(incoming.to(self.weight.device), self.weight)))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        self._invalidate()
        key = prefix + "_extra_state"
        if key in missing_keys:
            # It loads (the keys it has are the reference's); when the weight is a different one, the scales this module
            # holds belong to OTHER weights: refuse to run with them until the caller supplies the parameters
            # (set_extra_state / accept_current_quant_params / re-quantize / re-calibrate).
            missing_keys.remove(key)
            if prefix + "weight" in state_dict and not same_weight:
                self._params_stale = (f"{prefix or 'layer'}: the loaded state_dict carries an int8 weight but no "
                                      "quantization parameters (no '_extra_state' entry); the scales held by this "
                                      "module were computed from different weights")
                warnings.warn(self._params_stale + " -- forward() will raise until they are set", stacklevel=2)

    def accept_current_quant_params(self):
        """Declare that the quantization parameters this module holds belong to the weight it now holds (after loading
        a checkpoint that carries the int8 weight only): forward() runs again."""
        self._params_stale = None
        return self

    # ---- the two products -------------------------------------------------------------
    def _int8_product(self, x, bias, alpha, beta):
        w = self._on_gpu("weight", self.weight)
        if self.cnn:
            if _plain_conv(self.stride, self.padding, self.dilation, self.groups):
                return _native.conv2d_i8(x, w, _single(self.stride), _single(self.padding), bias=bias, alpha=alpha,
                                         beta=beta, packed=self._packed(w, torch.int8))
            return _native.conv2d_general(x, w, self.stride, self.padding, self.dilation, self.groups, bias=bias,
                                          alpha=alpha, beta=beta)
        lead = x.shape[:-1]
        y = _native.gemm_i8(x.reshape(-1, x.shape[-1]).contiguous(), w, bias=bias, alpha=alpha, beta=beta)
        return y.view(*lead, y.shape[-1])

    def _float_product(self, x, bias, alpha, beta):
        w = self._on_gpu("weight", self.weight)
        if not self.cnn:
            # Linear: the int8 weight goes to the matrix cores as it is stored (no float copy of it), with the
            # bias / scale epilogue fused (miq_gemm_w8).
            lead = x.shape[:-1]
            if x.dtype == torch.float64:
                x = x.to(torch.float32)
            out = _native.gemm_w8(x.reshape(-1, x.shape[-1]).contiguous(), w, bias=bias, alpha=alpha, beta=beta)
            return out.view(*lead, out.shape[-1])
        # Conv2d: the same product as an implicit GEMM (miq_conv2d_w8) for what that kernel covers ...
        if x.dtype == torch.float64:
            x = x.to(torch.float32)
        batched = x.dim() == 4
        if not batched:   # F.conv2d also takes one image [C, H, W]
            x = x.unsqueeze(0)
        if _plain_conv(self.stride, self.padding, self.dilation, self.groups):
            y = _native.conv2d_w8(x, w, _single(self.stride), _single(self.padding), bias=bias, alpha=alpha, beta=beta,
                                  packed=self._packed(w, x.dtype))
        else:
            # ... and otherwise (dilation, groups, a stride / padding per axis, string padding) the direct convolution
            y = _native.conv2d_general(x, w, self.stride, self.padding, self.dilation, self.groups, bias=bias,
                                       alpha=alpha, beta=beta)
        return y if batched else y.squeeze(0)

    def _run(self, x, bias, alpha, beta):
        if self._params_stale:
            raise RuntimeError(self._params_stale)
        home = x.device
        xd = _to_gpu(x).contiguous()
        bias, alpha = self._on_gpu("bias", bias), self._on_gpu("alpha", alpha)
        if xd.dtype == torch.int8:
            y = self._int8_product(xd, bias, alpha, beta)
        elif xd.dtype.is_floating_point:
            y = self._float_product(xd, bias, alpha, beta)
        else:
            raise TypeError(f"activations must be int8 or floating point, got {xd.dtype}")
        return y.to(home)

    def _weight_details(self):
        dm = self.data_metrics
        if dm and 'weights' in dm and dm['weights']:
            sym = "sym" if dm['weights'].get('symentric', True) else "asym"
            return f"weightq={sym}/{str(dm['weights'].get('dtype', '')).split('.')[-1]}/{dm['weights'].get('affine', '')}"
        return "weightq=None"

    def _repr(self, details):
        if self.cnn:
            return (f"QConv2d({self.in_features}, {self.out_features}, kernel_size={self.kernel_size}, "
                    f"stride={self.stride}, padding={self.padding}, dilation={self.dilation}, "
                    f"groups={self.groups}, bias={self.bias is not None} {details})")
        return f"QLinear({self.in_features}, {self.out_features}, bias={self.bias is not None} {details})"


def _layer_kwargs(module):
    conv = isinstance(module, nn.Conv2d)
    return dict(
        in_features=module.in_channels if conv else module.in_features,
        out_features=module.out_channels if conv else module.out_features,
        kernel_size=module.kernel_size if conv else None,
        stride=module.stride if conv else None,
        padding=module.padding if conv else None,
        dilation=module.dilation if conv else None,
        groups=module.groups if conv else None,
        cnn=conv,
    )


# =====================================================================================
# Weight-only
# =====================================================================================
class WeightOnlyQuantizer(_QuantizerBase):
    """quant/Weight_Only/Quantizer.py: y = scale[n] * ((x . Wq^T + b[n]) - zero_point[n])."""

    def __init__(self, in_features, out_features, kernel_size=None, stride=None, padding=None, dilation=None,
                 groups=None, bias=None, quantize_output=False, cnn=False, data_metrics=None, activations=None):
        super().__init__()
        self._init_common(in_features, out_features, kernel_size, stride, padding, dilation, groups, cnn,
                          data_metrics)
        self.bias = None if bias is None else bias
        self.input_quant = False
        self.input_quantizer = QopWeightOnly(dtype=torch.uint8, symmetric=False, affine='tensor', affine_dim=None)
        self.input_quantizer.max_val = 127
        # Part of the reference's attribute surface (Quantizer.py:45-47); its forward never calls it.
        self.input_observer = torch.ao.quantization.observer.MinMaxObserver(dtype=torch.quint8,
                                                                            qscheme=torch.per_tensor_affine)
        self.weight_quant = QopWeightOnly(
            dtype=data_metrics['weights']['dtype'],
            symmetric=data_metrics['weights']['symmetric'],
            affine=data_metrics['weights']['affine'],
            affine_dim=data_metrics['weights']['affine_dim'],
        )
        self.quantize_output = quantize_output

    @torch.no_grad()
    def forward(self, x):
        wq = self.weight_quant
        if not (wq.symmetric and wq.dtype == torch.int8):
            # zero_point is identically 0 for symmetric int8, the only form quantize() produces
            raise NotImplementedError("only symmetric int8 weights are implemented")
        grp = self.__dict__.get("_sibling_group")
        if grp is not None and not self.input_quant and grp[0].usable(x):
            return grp[0].output(x, grp[1])   # this layer's columns of the one product its siblings share
        if self.input_quant and x.dtype.is_floating_point:
            # The step the reference leaves commented out (Quantizer.py:69-70): quantize the activation per
            # tensor and take the int8 product (see quantization_amd/activation.py; a convolution's padding
            # is filled with the activation's zero point there).
            from .activation import conv2d_dynamic_w8a8, linear_dynamic_w8a8, weight_rowsum
            if self._params_stale:
                raise RuntimeError(self._params_stale)
            home = x.device
            w = self._on_gpu("weight", self.weight)
            if getattr(self, "_rowsum_src", None) is not self.weight:
                self._rowsum, self._rowsum_src = weight_rowsum(w), self.weight
            bias = None if self.bias is None else _to_gpu(self.bias.detach())
            xg = _to_gpu(x).to(torch.float32)
            if self.cnn:
                plain = _plain_conv(self.stride, self.padding, self.dilation, self.groups)
                y = conv2d_dynamic_w8a8(xg, w, _to_gpu(wq.scales), bias, self._rowsum, self.stride, self.padding,
                                        dilation=self.dilation, groups=self.groups,
                                        packed=self._packed(w, torch.int8) if plain else None)
            else:
                y = linear_dynamic_w8a8(xg, w, _to_gpu(wq.scales), bias, self._rowsum)
            return y.to(home)
        y = self._run(x, self.bias, wq.scales, 0.0)
        # the reference's dequantize(activation=True) leaves the parameters re-viewed [1,N,...] (Qop.py:175-177)
        if wq.affine == "channel" and wq.scales.dim() > 1 and wq.scales.shape[1] == 1:
            wq.zero_point = wq.zero_point.view([1, wq.zero_point.shape[0], *wq.zero_point.shape[2:]])
            wq.scales = wq.scales.view([1, wq.scales.shape[0], *wq.scales.shape[2:]])
        return y

    def _quant_state(self):
        return {"scales": self.weight_quant.scales, "zero_point": self.weight_quant.zero_point}

    def _set_quant_state(self, state):
        self.weight_quant.scales, self.weight_quant.zero_point = state["scales"], state["zero_point"]

    @classmethod
    def from_float(cls, module, quantize_output=False, activations=None, data_metry=None):
        new_module = cls(bias=module.bias, quantize_output=quantize_output, data_metrics=data_metry,
                                         activations=activations, **_layer_kwargs(module))
        new_module.weight = new_module.weight_quant.quantize(module.weight)
        new_module._prepack()
        return new_module

    def __repr__(self):
        dm = self.data_metrics
        if dm and 'outputs' in dm and dm['outputs']:
            sym = "sym" if dm['outputs'].get('symentric', True) else "asym"
            out = (f"outputq={sym}/{str(dm['outputs'].get('dtype', '')).split('.')[-1]}/"
                   f"{dm['outputs'].get('affine', '')}")
        else:
            out = "outputq=None"
        details = self._weight_details()
        if details != "weightq=None":
            details = f"{details}, {out}"
        return self._repr(details)



class SiblingGroup:
    """Weight-only quantized Linear layers that multiply the SAME activation (the gate and up projections of an MLP, the
    q / k / v projections of an attention block), stored as ONE [sum N_i, K] int8 weight and run as ONE product: at decode
    sizes a layer's time is its weight read plus a fixed few microseconds per launch, so n layers in one launch save n - 1
    of those (Llama-3-8B gate + up at one token: 2 x 14.4 us -> 24.6 us).  Per-channel scales and biases make the fused
    product the concatenation of the separate ones bit for bit: no accumulation order changes.

    The member layers keep their place in the model and their interface.  Their `weight` buffers become row blocks
    (views) of the fused weight, so nothing is stored twice and state_dict() is unchanged.  The first member called on an
    activation computes the whole product; the others, called on the same tensor, take their columns of it.  A member whose
    weight has been moved or replaced since (so that it is no longer a view of the fused weight) runs on its own again."""

    def __init__(self, layers):
        layers = list(layers)
        if len(layers) < 2:
            raise ValueError("a group needs at least two layers")
        for m in layers:
            if not isinstance(m, WeightOnlyQuantizer) or m.cnn:
                raise TypeError("only weight-only quantized Linear layers can be fused")
            if m._params_stale:
                raise RuntimeError(m._params_stale)
            if m.in_features != layers[0].in_features:
                raise ValueError("fused layers must have the same in_features")
            if m.__dict__.get("_sibling_group") is not None:
                raise ValueError("a layer can belong to one group only")
        self.layers = layers
        self.splits = [m.out_features for m in layers]
        weight = torch.cat([_to_gpu(m.weight.detach()) for m in layers], dim=0).contiguous()
        self.weight = weight
        self.alpha = torch.cat([_to_gpu(m.weight_quant.scales.detach()).reshape(-1).to(torch.float32) for m in layers]).contiguous()
        if any(m.bias is not None for m in layers):
            self.bias = torch.cat([(_to_gpu(m.bias.detach()).reshape(-1).to(torch.float32) if m.bias is not None
                                    else torch.zeros(m.out_features, dtype=torch.float32, device=weight.device))
                                   for m in layers]).contiguous()
        else:
            self.bias = None
        lo = 0
        for i, m in enumerate(layers):
            m.weight = weight[lo:lo + m.out_features]          # a view: the fused weight is the only copy
            m._invalidate()
            m.__dict__["_sibling_group"] = (self, i)
            lo += m.out_features
        self._ptrs = [m.weight.data_ptr() for m in layers]
        self._last = None     # (activation, its version, outputs, members still to collect)

    def usable(self, x):
        """The fused weight still is what every member holds (none was moved, replaced or re-quantized), and x is one
        the fused kernels take."""
        return (all(m.weight.data_ptr() == p and not m._params_stale for m, p in zip(self.layers, self._ptrs))
                and (x.dtype == torch.int8 or x.dtype.is_floating_point))

    @torch.no_grad()
    def output(self, x, index):
        c = self._last
        if c is None or c[0] is not x or c[1] != x._version or index not in c[3]:
            home = x.device
            xd = _to_gpu(x).contiguous()
            lead = xd.shape[:-1]
            x2 = xd.reshape(-1, xd.shape[-1])
            if xd.dtype == torch.int8:
                y = _native.gemm_i8(x2, self.weight, bias=self.bias, alpha=self.alpha, beta=0.0)
            else:
                if xd.dtype == torch.float64:
                    x2 = x2.to(torch.float32)
                y = _native.gemm_w8(x2.contiguous(), self.weight, bias=self.bias, alpha=self.alpha, beta=0.0)
            outs = torch.split(y.view(*lead, y.shape[-1]).to(home), self.splits, dim=-1)
            c = self._last = (x, x._version, outs, set(range(len(self.layers))))
        c[3].discard(index)
        out = c[2][index]
        if not c[3]:
            self._last = None     # every member has its columns: let go of the activation and the product
        return out


def fuse_linear_siblings(parent, names):
    """Fuse the weight-only quantized Linear layers `names` of `parent` (they must read the same activation: the caller
    says so) into one stored weight and one product per forward.  Returns the SiblingGroup."""
    return SiblingGroup([getattr(parent, n) for n in names])


def fuse_shared_inputs(model, *example_inputs, **example_kwargs):
    """Find the weight-only quantized Linear layers of `model` that are called on the very same tensor during one forward
    pass on the example inputs, and fuse each such set.  Returns the list of groups."""
    seen, hooks = {}, []

    def hook(module, inputs):
        if inputs and isinstance(inputs[0], torch.Tensor):
            seen.setdefault(id(inputs[0]), (inputs[0], []))[1].append(module)

    for m in model.modules():
        if isinstance(m, WeightOnlyQuantizer) and not m.cnn and m.__dict__.get("_sibling_group") is None:
            hooks.append(m.register_forward_pre_hook(hook))
    try:
        with torch.no_grad():
            model(*example_inputs, **example_kwargs)
    finally:
        for h in hooks:
            h.remove()
    groups = []
    for _, mods in seen.values():
        uniq = list(dict.fromkeys(mods))
        if len(uniq) >= 2 and len({m.in_features for m in uniq}) == 1:
            groups.append(SiblingGroup(uniq))
    return groups


class WeightOnlyChunker:
    """quant/Weight_Only/Chunker.py: swap every Conv2d / Linear for a quantized layer.

    The reference also runs a ModelAnalyzer pass whose result it never uses (Chunker.py:13);
    that pass is not reproduced and `mapped_layers` is None.
    """
    quantizer_cls = WeightOnlyQuantizer

    def __init__(self, model, calibiration_data):
        self.model = model
        self.calibiration_data = calibiration_data
        self.mapped_layers = None
        self.interested_layers = []
        self.chunk()

    def replace_modules(self, module, target_class, look_out_for, module_name_to_exclude=""):
        for name, child in module.named_children():
            if isinstance(child, look_out_for) and not any(x == name for x in module_name_to_exclude):
                if target_class == 'weights':
                    q_params = {'weights': {'dtype': torch.int8, 'symmetric': True, 'affine': 'channel',
                                            'affine_dim': 0}}
                    setattr(module, name, self.quantizer_cls.from_float(module=child, data_metry=q_params,
                                                                       quantize_output=False))
            else:
                self.replace_modules(child, target_class, look_out_for, module_name_to_exclude)

    def weight_quantize(self):
        self.replace_modules(module=self.model, target_class='weights', look_out_for=(nn.Conv2d, nn.Linear))

    def chunk(self):
        self.weight_quantize()

    def fuse_siblings(self, *example_inputs, **example_kwargs):
        """After the rewrite: fuse the quantized Linear layers that one forward pass on the example inputs calls on the
        same tensor (see SiblingGroup).  Has no counterpart in the reference, whose layers each run their own product."""
        return fuse_shared_inputs(self.model, *example_inputs, **example_kwargs)


# =====================================================================================
# Weight + activation
# =====================================================================================
class WeightActivationQuantizer(_QuantizerBase):
    """quant/Weight_Activation/Quantizer.py: y = (x . Wq^T + bias') * scalers + output_zero_point."""

    def __init__(self, in_features, out_features, kernel_size=None, stride=None, padding=None, dilation=None,
                 groups=None, bias=None, quantize_output=False, cnn=False, data_metrics=None, activations=None):
        super().__init__()
        self._init_common(in_features, out_features, kernel_size, stride, padding, dilation, groups, cnn,
                          data_metrics)
        self.bias = bias
        self.weight_qscale, self.weight_zero_point = None, None
        self.output_zero_point = None, None
        self.scalers = None

    @torch.no_grad()
    def forward(self, x):
        if self.scalers is None:
            raise TypeError("scalers are not set: run Chunker.compute_qstats() (the reference fails here too)")
        n = self.out_features
        bias = self.bias
        if bias is not None and bias.numel() == 1 and n != 1:
            bias = bias.reshape(1).expand(n)  # `torch.tensor(0)` stands for "no bias" (Quantizer.py:81)
        return self._run(x, bias, self.scalers, self._beta())

    def _beta(self):
        """output_zero_point as a Python float.  Reading it from a GPU tensor blocks the host, so it is read once
        per value (when the tensor is replaced or written) and not on every forward: a layer's forward then
        only enqueues work, and captures into a HIP graph."""
        ozp = self.output_zero_point
        if not isinstance(ozp, torch.Tensor):
            return float(ozp)
        c = self._beta_cache
        if c is None or c[0] is not ozp or c[1] != ozp._version:
            c = self._beta_cache = (ozp, ozp._version, float(ozp.reshape(-1)[0]))
        return c[2]

    _STATE = ("bias", "weight_qscale", "weight_zero_point", "scalers", "output_zero_point")

    def _quant_state(self):
        return {k: getattr(self, k) for k in self._STATE}

    def _set_quant_state(self, state):
        for k in self._STATE:
            setattr(self, k, state[k])

    @classmethod
    def from_float(cls, module, quantize_output=False, activations=None, data_metry=None):
        if module.bias is not None:
            bias = module.bias.view(1, -1, 1, 1) if isinstance(module, nn.Conv2d) else module.bias.view(1, -1)
        else:
            bias = torch.tensor(0)
        new_module = cls(bias=bias, quantize_output=quantize_output, data_metrics=data_metry,
                                               activations=activations, **_layer_kwargs(module))
        weight_quant = QopWeightActivation(
            dtype=data_metry['weights']['dtype'],
            symmetric=data_metry['weights']['symmetric'],
            affine=data_metry['weights']['affine'],
            affine_dim=data_metry['weights']['affine_dim'],
        )
        new_module.weight = weight_quant.quantize(module.weight)
        new_module.weight_qscale, new_module.weight_zero_point = weight_quant.scales, weight_quant.zero_point
        new_module._prepack()
        return new_module

    def __repr__(self):
        return self._repr(self._weight_details())


def _pool_to_i8(module, x, mode):
    """One fused kernel for `pool(x.float()).to(torch.int8)`.  The reference builds these modules with
    kernel_size, stride and padding only (Chunker.py:75-77); other settings are not implemented."""
    if getattr(module, "ceil_mode", False) or getattr(module, "dilation", 1) not in (1, (1, 1)) \
            or getattr(module, "return_indices", False) or not getattr(module, "count_include_pad", True) \
            or getattr(module, "divisor_override", None) is not None:
        raise NotImplementedError("int8 pooling supports kernel_size, stride and padding only")
    if x.dim() == 3:
        return _pool_to_i8(module, x.unsqueeze(0), mode).squeeze(0)
    home = x.device
    xd = _to_gpu(x).contiguous()
    return _native.pool2d_to_i8(xd, module.kernel_size, module.stride, module.padding, mode).to(home)


class QuantizedMaxPool2d(nn.MaxPool2d):
    """quant/Weight_Activation/Quantizer.py:122-125."""

    def forward(self, x):
        return _pool_to_i8(self, x, "max")


class QuantizedAvgPool2d(nn.AvgPool2d):
    """quant/Weight_Activation/Quantizer.py:127-130."""

    def forward(self, x):
        return _pool_to_i8(self, x, "avg")


class WeightActivationChunker:
    """quant/Weight_Activation/Chunker.py: observe activations, quantize weights, fold the
    statistics into each layer's `bias` / `scalers` / `output_zero_point`.

    The calibration arithmetic is reproduced as the reference computes it, including what it
    names oddly: `input_zero_point` and `output_zero_point` both hold the OUTPUT scale, and the
    running minimum of the outputs is fed with output.max() (Chunker.py:47,95-96).
    """
    quantizer_cls = WeightActivationQuantizer

    def __init__(self, model, calibiration_data):
        self.model = model
        self.calibiration_data = calibiration_data
        self.mapped_layers = None
        self.interested_layers = []
        self.hooks = {}
        self.attached_hooks = []
        self.chunk()

    def _stat_observer(self, module, input, output, module_name):
        x = input[0]
        h = self.hooks.setdefault(module_name, {'input_stats': [float('inf'), float('-inf')],
                                                'output_stats': [float('inf'), float('-inf')]})
        h['type'] = type(module)
        h['input_stats'] = [min(h['input_stats'][0], x.min()), max(h['input_stats'][1], x.max())]
        h['output_stats'] = [min(h['output_stats'][0], output.max()), max(h['output_stats'][1], output.max())]

    def _compute_qstats(self, name, child):
        qin = QopWeightActivation(dtype=torch.int8, symmetric=False, affine='tensor', affine_dim=None)
        qout = QopWeightActivation(dtype=torch.int8, symmetric=False, affine='tensor', affine_dim=None)
        qin.min_val, qin.max_val = self.hooks[name]['input_stats']
        qout.min_val, qout.max_val = self.hooks[name]['output_stats']
        child.input_qscale = qin.calculate_params()[0].item()
        child.input_zero_point = qout.calculate_params()[0].item()
        child.output_qscale = qout.calculate_params()[0].item()
        child.output_zero_point = qout.calculate_params()[0]
        shape = (1, -1, 1, 1) if child.weight.dim() == 4 else (1, -1)
        dims = (1, 2, 3) if child.weight.dim() == 4 else 1
        qweight = child.weight.sum(dims)
        child.bias = (child.bias.squeeze() - qweight.to(torch.int32) * child.input_zero_point).squeeze().view(shape)
        child.scalers = ((child.input_qscale * child.weight_qscale) / child.output_qscale).squeeze().view(shape)

    def replace_modules(self, module, target_class, look_out_for, module_name_to_exclude=""):
        for name, child in module.named_children():
            if isinstance(child, look_out_for) and not any(x == name for x in module_name_to_exclude):
                if target_class == 'weights':
                    q_params = {'weights': {'dtype': torch.int8, 'symmetric': True, 'affine': 'channel',
                                            'affine_dim': 0}}
                    setattr(module, name, self.quantizer_cls.from_float(module=child, data_metry=q_params,
                                                                       quantize_output=False))
                elif target_class == 'attach_observers':
                    self.attached_hooks.append(child.register_forward_hook(
                        functools.partial(self._stat_observer, module_name=name)))
                elif target_class == 'misc_layers':
                    if isinstance(child, nn.MaxPool2d):
                        setattr(module, name, QuantizedMaxPool2d(kernel_size=child.kernel_size, stride=child.stride,
                                                                 padding=child.padding))
                    elif isinstance(child, nn.AvgPool2d):
                        setattr(module, name, QuantizedAvgPool2d(kernel_size=child.kernel_size, stride=child.stride,
                                                                 padding=child.padding))
                elif target_class == 'compute_qstats':
                    self._compute_qstats(name, child)
            else:
                self.replace_modules(child, target_class, look_out_for, module_name_to_exclude)

    def weight_quantize(self):
        self.replace_modules(module=self.model, target_class='weights', look_out_for=(nn.Conv2d, nn.Linear))

    def calibirate_model(self):
        device = next(self.model.parameters()).device
        with torch.no_grad():
            for input_data, _ in self.calibiration_data:
                self.model(input_data.to(device))

    def attach_observers_observe(self):
        self.replace_modules(module=self.model, target_class='attach_observers', look_out_for=(nn.Conv2d, nn.Linear))
        self.calibirate_model()
        for hook in self.attached_hooks:
            hook.remove()

    def compute_qstats(self):
        self.replace_modules(module=self.model, target_class='compute_qstats', look_out_for=self.quantizer_cls)

    def misc_layers(self):
        self.replace_modules(module=self.model, target_class='misc_layers', look_out_for=(nn.MaxPool2d, nn.AvgPool2d))

    def chunk(self):
        self.attach_observers_observe()
        self.weight_quantize()
        self.compute_qstats()
