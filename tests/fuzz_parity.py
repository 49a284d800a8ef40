#!/usr/bin/env python3
"""Seeded random differential campaign of the int8 kernels on one GPU, against exact references computed by torch on the
same device (float64 products of int8 values are exact; the epilogue is torch's own fp32 add / mul / add, one rounding
each): every launch plan the size ranges reach (skinny, ring tiles, split-K, 256 x 256, whole rounds + tail, the
transposed / column-block / int8-requantised forms; im2col-order, pipelined, patch (staged from NCHW or from the padded
copy, small images, pixel-by-pixel stores), stem, strided 1 x 1 and channels-last convolutions).  Prints every mismatch
with the arguments that reproduce it and exits non-zero if there was one.

Further families: the per-channel weight quantizer against the numpy oracle (every dtype, ties, wide-range rows), the
weight-only Linear / Conv2d on floating-point activations and the general convolution (groups, dilation, a stride and a
padding per axis) against float64 products, to the bound the weight-only tests state (5e-5 of the sum of magnitudes).
Test infrastructure: tests/test_gpu_fuzz.py runs a few hundred cases per family; for a longer campaign

    python tests/fuzz_parity.py [cases per family, default 400] [seed, default 1] [families, e.g. gemm,conv]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as O  # noqa: E402
from quantization_amd import _native as N  # noqa: E402

dev = torch.device("cuda", 0)
seed = 1
rng = np.random.default_rng(seed)
bad = 0


def reseed(s):
    global seed, rng, bad
    seed, rng, bad = s, np.random.default_rng(s), 0


def ri8(shape, g):
    return torch.randint(-128, 128, shape, dtype=torch.int32, device=dev, generator=g).to(torch.int8)


def epilogue(acc, bias, alpha, beta, channel_dim):
    shape = [1] * acc.dim()
    shape[channel_dim] = -1
    y = acc.to(torch.float32)
    y = y + (bias.view(shape) if bias is not None else 0.0)
    y = y * alpha.view(shape)
    return y + beta


def report(family, what, args):
    global bad
    bad += 1
    print(f"MISMATCH {family}: {what}  args={args}", flush=True)


def pick(options):
    return options[int(rng.integers(0, len(options)))]


def fuzz_gemm(n):
    g = torch.Generator(device=dev).manual_seed(seed)
    for it in range(n):
        kind = int(rng.integers(0, 6))
        M = int(pick([1, 2, 7, 16, 17, 31, 33, 64, 100, 128, 129, 255, 256, 300, 512, 700, 1024, 1500]))
        Nn = int(pick([1, 5, 16, 48, 64, 100, 128, 200, 256, 384, 500, 1024, 2500, 4096, 6000]))
        K = int(pick([16, 32, 48, 64, 100, 128, 144, 256, 500, 512, 1000, 1024, 2048, 4096, 9000]))
        if kind == 5:   # large enough for whole rounds of 256 x 256 tiles plus a tail
            M, Nn, K = int(pick([2048, 2304, 2100, 4100])), int(pick([4096, 8192, 5000])), int(pick([256, 512, 1040]))
        a, w = ri8((M, K), g), ri8((Nn, K), g)
        alpha = torch.rand(Nn, device=dev, generator=g) * 0.01 + 1e-4
        bias = torch.randn(Nn, device=dev, generator=g) * 50 if rng.integers(0, 2) else None
        beta = float(pick([0.0, -1.25, 3.0]))
        acc = (a.double() @ w.double().t()).to(torch.int32)
        args = (kind, M, Nn, K, bias is not None, beta)
        out = N.gemm_i8(a, w, out_dtype=torch.int32)
        if not torch.equal(out, acc):
            report("gemm", "int32", args)
            continue
        ref = epilogue(acc, bias, alpha, beta, 1)
        if kind in (0, 5):
            y = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=beta)
            ok = torch.equal(y, ref)
        elif kind == 1:
            y = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=beta, out_dtype=torch.bfloat16)
            ok = torch.equal(y, ref.to(torch.bfloat16))
        elif kind == 2:   # the [K, N] weight of linear_w8a8
            y = N.gemm_i8(a, w.t().contiguous(), w_layout=N.MIQ_W_KN, bias=bias, alpha=alpha, beta=beta)
            ok = torch.equal(y, ref)
        elif kind == 3:   # transposed result, as a column-sharded layer gathers it
            y = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=beta, transpose_out=True)
            ok = torch.equal(y, ref.t())
        else:             # a column block of a wider matrix, int8 requantised in the epilogue
            wide = torch.zeros(M, Nn + 37, device=dev)
            N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=beta, out=wide[:, 5:5 + Nn])
            ok = torch.equal(wide[:, 5:5 + Nn], ref) and not wide[:, :5].any() and not wide[:, 5 + Nn:].any()
            s = float(ref.abs().max()) / 127 + 1e-6
            q = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=beta, out_dtype=torch.int8, requant_scale=s)
            qr = torch.clamp(torch.round(ref / torch.tensor(s, device=dev)), -128, 127).to(torch.int8)
            ok = ok and torch.equal(q, qr)
        if not ok:
            report("gemm", "epilogue form", args)


def fuzz_conv(n):
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    for it in range(n):
        kind = int(rng.integers(0, 6))
        if kind == 0:     # 3 x 3 / stride 1 on whole channel groups: the patch kernels and what they decline
            C, k, s = 64 * int(rng.integers(1, 5)), 3, 1
            H, W = int(rng.integers(5, 60)), int(rng.integers(5, 60))
            p = int(rng.integers(0, 3))
            Co = int(rng.integers(1, 200))
            B = int(rng.integers(1, 4)) + 2000 // max(1, (H + 2 * p - 2) * (W + 2 * p - 2))
        elif kind == 1:   # strided 1 x 1
            C, k, s, p = int(rng.integers(1, 300)), 1, int(rng.integers(1, 4)), 0
            H, W = int(rng.integers(1, 60)), int(rng.integers(1, 60))
            Co, B = int(rng.integers(1, 300)), int(rng.integers(1, 6))
        elif kind == 2:   # stems
            C, k, s, p = int(rng.integers(1, 5)), int(pick([5, 7, 8])), 2, int(rng.integers(0, 4))
            H, W = int(rng.integers(16, 120)), int(rng.integers(16, 240))
            Co, B = int(rng.integers(1, 140)), int(rng.integers(1, 4))
        elif kind == 3:   # layers large enough for the pipelined 256 x 256 kernel
            C, k, s, p = int(pick([64, 128, 256])), 3, int(pick([1, 2])), 1
            H = W = int(pick([14, 16, 28]))
            Co, B = int(pick([256, 512])), int(pick([96, 128]))
        else:             # anything
            C, k = int(rng.integers(1, 140)), int(rng.integers(1, 6))
            s, p = int(rng.integers(1, 4)), int(rng.integers(0, 4))
            H, W = int(rng.integers(k, 50)), int(rng.integers(k, 50))
            Co, B = int(rng.integers(1, 150)), int(rng.integers(1, 6))
        if H + 2 * p < k or W + 2 * p < k:
            continue
        x, w = ri8((B, C, H, W), g), ri8((Co, C, k, k), g)
        args = (kind, B, C, H, W, Co, k, s, p)
        acc = F.conv2d(x.double(), w.double(), stride=s, padding=p).to(torch.int32)
        packed = N.conv2d_i8_prepack(w, (B, C, H, W), s, p) if rng.integers(0, 2) else None
        out = N.conv2d_i8(x, w, s, p, out_dtype=torch.int32, packed=packed)
        if out.shape != acc.shape or not torch.equal(out, acc):
            report("conv", "int32", args)
            continue
        alpha = torch.rand(Co, device=dev, generator=g) * 0.01 + 1e-4
        bias = torch.randn(Co, device=dev, generator=g) * 50 if rng.integers(0, 2) else None
        beta = float(pick([0.0, -1.25]))
        ref = epilogue(acc, bias, alpha, beta, 1)
        od = pick([torch.float32, torch.bfloat16])
        y = N.conv2d_i8(x, w, s, p, bias=bias, alpha=alpha, beta=beta, out_dtype=od, packed=packed)
        if not torch.equal(y, ref.to(od)):
            report("conv", f"epilogue {od}", args)
            continue
        if kind in (0, 3) and C % 4 == 0 and Co % 4 == 0:   # the channels-last entry on the same layer: int32 and int8 out
            xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, p, p, p, p)).contiguous()
            wp = N.conv2d_i8_pack_weight(w)
            y = N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, out_dtype=torch.int32)
            if not torch.equal(y.permute(0, 3, 1, 2), acc):
                report("conv channels-last", "int32", args)
                continue
            small = alpha * 0.02   # (so that most outputs fall inside the int8 range)
            q = N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, bias=bias, alpha=small, beta=beta, out_dtype=torch.int8)
            qr = torch.clamp(torch.round(epilogue(acc, bias, small, beta, 1)), -128, 127).to(torch.int8)
            if not torch.equal(q.permute(0, 3, 1, 2), qr):
                report("conv channels-last", "int8", args)


def within(y, ref, mag, out_dtype):
    tol = 5e-5 * mag + 1e-30
    if out_dtype == torch.bfloat16:
        tol = tol + ref.abs() * 2.0 ** -8
    return bool(((y.double() - ref).abs() <= tol).all())


def fuzz_gemm_w8(n):
    g = torch.Generator(device=dev).manual_seed(seed + 2)
    for it in range(n):
        M = int(pick([1, 3, 16, 17, 40, 64, 100, 128, 200, 256, 300, 600, 1100]))
        Nn = int(pick([1, 7, 16, 64, 100, 128, 256, 500, 1024, 3000]))
        K = int(pick([8, 16, 32, 64, 100, 128, 256, 520, 1024, 2048, 4100]))
        xdt = pick([torch.float32, torch.float16, torch.bfloat16])
        od = pick([torch.float32, torch.float32, torch.bfloat16])
        x = torch.randn(M, K, device=dev, generator=g)
        if rng.integers(0, 3) == 0:
            x = x * torch.exp(3 * torch.randn(M, K, device=dev, generator=g))
        x = x.clamp(-60000.0, 60000.0).to(xdt)
        w = ri8((Nn, K), g)
        alpha = torch.rand(Nn, device=dev, generator=g) * 0.02 + 1e-4
        bias = torch.randn(Nn, device=dev, generator=g) * 10 if rng.integers(0, 2) else None
        beta = float(pick([0.0, 0.5]))
        y = N.gemm_w8(x, w, bias=bias, alpha=alpha, beta=beta, out_dtype=od)
        xd, wd = x.double(), w.double()
        bd = bias.double() if bias is not None else torch.zeros(Nn, device=dev, dtype=torch.float64)
        ref = (xd @ wd.t() + bd) * alpha.double() + beta
        mag = (xd.abs() @ wd.abs().t() + bd.abs()) * alpha.double() + abs(beta)
        if y.dtype != od or not within(y, ref, mag, od):
            report("gemm_w8", f"{xdt} -> {od}", (M, Nn, K, bias is not None, beta))


def conv_ref(x, w, bias, alpha, beta, **kw):
    xd, wd = x.double(), w.double()
    Co = w.shape[0]
    bd = (bias.double() if bias is not None else torch.zeros(Co, device=dev, dtype=torch.float64)).view(1, -1, 1, 1)
    ad = alpha.double().view(1, -1, 1, 1)
    ref = (F.conv2d(xd, wd, **kw) + bd) * ad + beta
    mag = (F.conv2d(xd.abs(), wd.abs(), **kw) + bd.abs()) * ad + abs(beta)
    return ref, mag


def fuzz_conv_w8(n):
    g = torch.Generator(device=dev).manual_seed(seed + 3)
    for it in range(n):
        kind = int(rng.integers(0, 4))
        if kind == 0:     # 3 x 3 / stride 1 on whole 32-channel groups: the patch kernels and what they decline
            C, k, s = 32 * int(rng.integers(1, 7)), 3, 1
            H, W = int(rng.integers(5, 50)), int(rng.integers(5, 50))
            p = int(rng.integers(0, 3))
            Co = int(rng.integers(1, 150))
            B = int(rng.integers(1, 4)) + 2000 // max(1, (H + 2 * p - 2) * (W + 2 * p - 2))
        elif kind == 1:   # stems
            C, k, s, p = int(rng.integers(1, 5)), int(pick([5, 7, 8])), 2, int(rng.integers(0, 4))
            H, W = int(rng.integers(16, 100)), int(rng.integers(16, 200))
            Co, B = int(rng.integers(1, 140)), int(rng.integers(1, 4))
        else:             # anything
            C, k = int(rng.integers(1, 140)), int(rng.integers(1, 6))
            s, p = int(rng.integers(1, 4)), int(rng.integers(0, 4))
            H, W = int(rng.integers(k, 40)), int(rng.integers(k, 40))
            Co, B = int(rng.integers(1, 150)), int(rng.integers(1, 6))
        if H + 2 * p < k or W + 2 * p < k:
            continue
        xdt = pick([torch.float32, torch.float16, torch.bfloat16])
        od = pick([torch.float32, torch.float32, torch.bfloat16])
        x = torch.randn(B, C, H, W, device=dev, generator=g).to(xdt)
        w = ri8((Co, C, k, k), g)
        alpha = torch.rand(Co, device=dev, generator=g) * 0.02 + 1e-4
        bias = torch.randn(Co, device=dev, generator=g) * 10 if rng.integers(0, 2) else None
        beta = float(pick([0.0, 0.5]))
        packed = N.conv2d_w8_prepack(w, xdt, (B, C, H, W), s, p) if rng.integers(0, 2) else None
        y = N.conv2d_w8(x, w, s, p, bias=bias, alpha=alpha, beta=beta, out_dtype=od, packed=packed)
        ref, mag = conv_ref(x, w, bias, alpha, beta, stride=s, padding=p)
        if y.shape != ref.shape or not within(y, ref, mag, od):
            report("conv_w8", f"{xdt} -> {od}", (kind, B, C, H, W, Co, k, s, p, packed is not None))


def fuzz_conv_general(n):
    g = torch.Generator(device=dev).manual_seed(seed + 4)
    for it in range(n):
        groups = int(pick([1, 1, 2, 3, 4, 8, 0]))
        cg, cog = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if groups == 0:   # depthwise
            groups, cg, cog = int(rng.integers(1, 200)), 1, int(pick([1, 1, 2]))
        C, Co = groups * cg, groups * cog
        kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        st = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        dl = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        pd = (int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        B = int(rng.integers(1, 5))
        if H + 2 * pd[0] < dl[0] * (kh - 1) + 1 or W + 2 * pd[1] < dl[1] * (kw - 1) + 1:
            continue
        w = ri8((Co, cg, kh, kw), g)
        alpha = torch.rand(Co, device=dev, generator=g) * 0.02 + 1e-4
        bias = torch.randn(Co, device=dev, generator=g) * 10 if rng.integers(0, 2) else None
        beta = float(pick([0.0, 0.5]))
        args = (B, C, H, W, Co, kh, kw, st, pd, dl, groups)
        kw_ = dict(stride=st, padding=pd, dilation=dl, groups=groups)
        if rng.integers(0, 2):   # int8 activations: exact
            x = ri8((B, C, H, W), g)
            acc = F.conv2d(x.double(), w.double(), **kw_).to(torch.int32)
            out = N.conv2d_general(x, w, st, pd, dl, groups, out_dtype=torch.int32)
            if out.shape != acc.shape or not torch.equal(out, acc):
                report("conv_general", "int8 -> int32", args)
                continue
            y = N.conv2d_general(x, w, st, pd, dl, groups, bias=bias, alpha=alpha, beta=beta)
            if not torch.equal(y, epilogue(acc, bias, alpha, beta, 1)):
                report("conv_general", "int8 -> fp32", args)
        else:
            xdt = pick([torch.float32, torch.float16, torch.bfloat16])
            x = torch.randn(B, C, H, W, device=dev, generator=g).to(xdt)
            y = N.conv2d_general(x, w, st, pd, dl, groups, bias=bias, alpha=alpha, beta=beta)
            ref, mag = conv_ref(x, w, bias, alpha, beta, **kw_)
            if y.shape != ref.shape or not within(y, ref, mag, torch.float32):
                report("conv_general", f"{xdt} -> fp32", args)


def fuzz_quantizer(n):
    kinds = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}
    for it in range(n):
        rows = int(pick([1, 2, 5, 16, 33, 64, 100, 257]))
        cols = int(pick([1, 3, 4, 7, 16, 31, 64, 100, 128, 255, 256, 1000, 1024, 4096, 5000, 8192, 12288, 40000]))
        dt = pick(list(kinds))
        w = rng.standard_normal((rows, cols)).astype(np.float32)
        mode = int(rng.integers(0, 5))
        if mode == 1:
            w *= np.exp(4 * rng.standard_normal((rows, 1))).astype(np.float32)   # rows over many binades
        elif mode == 2:
            w = -np.abs(w)                                                         # negative rows: scale = the floor
        elif mode == 3:   # quotients on rounding boundaries: multiples of half a step of the row's own scale
            top = np.abs(w).max(axis=1, keepdims=True) + 1.0
            w = np.rint(w / top * 255.0) * (top / 255.0)
            w[:, 0] = top[:, 0]
        elif mode == 4:
            w[rng.random((rows, cols)) < 0.3] = 0.0
        if dt == torch.float16:
            w = np.clip(w, -60000.0, 60000.0)   # (infinite weights are outside the path: the reference divides inf by inf)
        t = torch.from_numpy(w).to(dt)
        w32 = t.float().numpy()
        q, s, zp = N.quantize_rows_sym_i8(t.to(dev))
        rq, rs, rzp = O.quantize_channel_sym_int8(w32, kinds[dt])
        if not (np.array_equal(q.cpu().numpy(), rq) and np.array_equal(s.cpu().numpy(), rs.reshape(-1))
                and np.array_equal(zp.cpu().numpy(), rzp.reshape(-1))):
            report("quantizer", str(dt), (rows, cols, mode))


def fuzz_layers(n):
    """Random nn.Linear / nn.Conv2d modules through the weight-only Quantizer (from_float + forward) on int8 and
    floating-point activations: the stored weight against the oracle, the output against the float64 product of the
    stored weight (Weight_Only/Quantizer.py:73-91: (x . Wq + b) * scale)."""
    from quantization_amd.quant.Weight_Only import Quantizer
    qp = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}
    g = torch.Generator(device=dev).manual_seed(seed + 5)
    for it in range(n):
        torch.manual_seed(seed * 100003 + it)
        has_bias = bool(rng.integers(0, 2))
        if rng.integers(0, 2):
            K, Nn = int(rng.integers(1, 600)), int(rng.integers(1, 400))
            mod = torch.nn.Linear(K, Nn, bias=has_bias)
            lead = tuple(int(v) for v in rng.integers(1, 6, size=int(rng.integers(1, 3))))
            xshape, args, kw = lead + (K,), ("linear", K, Nn, lead, has_bias), None
        else:
            groups = int(pick([1, 1, 1, 2, 4]))
            C, Co = groups * int(rng.integers(1, 40)), groups * int(rng.integers(1, 40))
            ks = (int(rng.integers(1, 5)), int(rng.integers(1, 5))) if rng.integers(0, 3) == 0 else int(rng.integers(1, 5))
            st = int(rng.integers(1, 3))
            pd = pick([0, 1, 2, (1, 2), "same", "valid"])
            dl = int(pick([1, 1, 2]))
            if pd == "same" and st != 1:
                st = 1
            mod = torch.nn.Conv2d(C, Co, ks, stride=st, padding=pd, dilation=dl, groups=groups, bias=has_bias)
            H, W = int(rng.integers(8, 30)), int(rng.integers(8, 30))
            xshape, args = (int(rng.integers(1, 4)), C, H, W), ("conv", C, Co, ks, st, pd, dl, groups, H, W, has_bias)
            kw = dict(stride=mod.stride, padding=mod.padding, dilation=mod.dilation, groups=groups)
        mod = mod.to(dev)
        ql = Quantizer.from_float(module=mod, data_metry=qp)
        rq, rs, _ = O.quantize_channel_sym_int8(mod.weight.detach().cpu().numpy())
        if not np.array_equal(ql.weight.cpu().numpy(), rq):
            report("layers", "stored weight", args)
            continue
        wq, sc = ql.weight.double(), torch.from_numpy(rs.reshape(-1)).to(dev).double()
        bd = mod.bias.detach().double() if has_bias else torch.zeros(wq.shape[0], device=dev, dtype=torch.float64)
        for xdt in (torch.int8, pick([torch.float32, torch.float16, torch.bfloat16])):
            x = ri8(xshape, g) if xdt == torch.int8 else torch.randn(xshape, device=dev, generator=g).to(xdt)
            y = ql(x)
            xd = x.double()
            if kw is None:
                ref, mag = (xd @ wq.t() + bd) * sc, (xd.abs() @ wq.abs().t() + bd.abs()) * sc
            else:
                v = (1, -1, 1, 1)
                ref = (F.conv2d(xd, wq, **kw) + bd.view(v)) * sc.view(v)
                mag = (F.conv2d(xd.abs(), wq.abs(), **kw) + bd.abs().view(v)) * sc.view(v)
            if y.shape != ref.shape or not within(y, ref, mag, torch.float32):
                report("layers", f"forward on {xdt}", args)
                break


def fuzz_layers_wa(n):
    """Random modules through the weight + activation Quantizer with calibration results set as Chunker.compute_qstats
    leaves them (quant/Weight_Activation/Chunker.py:79-106: bias' [1,N(,1,1)], scalers of the same shape, a one-element
    output_zero_point): forward on int8 activations equals (acc + bias') * scalers + output_zero_point evaluated by
    torch in fp32, one rounding per step (Weight_Activation/Quantizer.py:58-60), bit for bit."""
    from quantization_amd.quant.Weight_Activation import Quantizer
    qp = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}
    g = torch.Generator(device=dev).manual_seed(seed + 7)
    for it in range(n):
        torch.manual_seed(seed * 100003 + it)
        has_bias = bool(rng.integers(0, 2))
        if rng.integers(0, 2):
            K, Nn = int(rng.integers(1, 600)), int(rng.integers(1, 400))
            mod, kw = torch.nn.Linear(K, Nn, bias=has_bias), None
            xshape, shape, args = (int(rng.integers(1, 40)), K), (1, -1), ("linear", K, Nn, has_bias)
        else:
            C, Co, k = int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.integers(1, 5))
            st, pd = int(rng.integers(1, 3)), int(rng.integers(0, 3))
            mod = torch.nn.Conv2d(C, Co, k, stride=st, padding=pd, bias=has_bias)
            H, W = int(rng.integers(k, 30)), int(rng.integers(k, 30))
            xshape, shape, args = (int(rng.integers(1, 4)), C, H, W), (1, -1, 1, 1), ("conv", C, Co, k, st, pd, H, W, has_bias)
            kw, Nn = dict(stride=st, padding=pd), Co
        ql = Quantizer.from_float(module=mod.to(dev), data_metry=qp)
        rq, _, _ = O.quantize_channel_sym_int8(mod.weight.detach().cpu().numpy())
        if not np.array_equal(ql.weight.cpu().numpy(), rq):
            report("layers_wa", "stored weight", args)
            continue
        ql.bias = (torch.randn(Nn, device=dev, generator=g) * 200).view(shape)
        ql.scalers = (torch.rand(Nn, device=dev, generator=g) * 0.01 + 1e-5).view(shape)
        ql.output_zero_point = torch.tensor([float(rng.integers(-100, 100))], device=dev)
        x = ri8(xshape, g)
        y = ql(x)
        wq = ql.weight.double()
        acc = (x.double() @ wq.t() if kw is None else F.conv2d(x.double(), wq, **kw)).to(torch.int32)
        ref = (acc.to(torch.float32) + ql.bias) * ql.scalers + ql.output_zero_point
        if y.shape != ref.shape or not torch.equal(y, ref):
            report("layers_wa", "forward", args)


def fuzz_pool(n):
    """miq_pool2d_to_i8 (QuantizedMaxPool2d / QuantizedAvgPool2d, quant/Weight_Activation/Quantizer.py:122-130) against
    the numpy oracle, bit for bit: int8 and floating-point inputs, windows / strides / paddings per axis."""
    for it in range(n):
        kh, kw = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        if rng.integers(0, 2):
            kw = kh
        stride = None if rng.integers(0, 3) == 0 else (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        ph, pw = int(rng.integers(0, kh // 2 + 1)), int(rng.integers(0, kw // 2 + 1))
        B, C = int(rng.integers(1, 4)), int(rng.integers(1, 70))
        H, W = int(rng.integers(kh, 40)), int(rng.integers(kw, 40))
        mode = pick(["max", "avg"])
        dt = pick([torch.int8, torch.float32, torch.float16, torch.bfloat16])
        if dt == torch.int8:
            x = torch.from_numpy(rng.integers(-128, 128, size=(B, C, H, W)).astype(np.int8))
        else:   # values whose cast to int8 truncates, wraps (|v| > 128) and keeps its sign
            x = torch.from_numpy((rng.standard_normal((B, C, H, W)) * pick([1.0, 40.0, 300.0])).astype(np.float32)).to(dt)
        y = N.pool2d_to_i8(x.to(dev), (kh, kw), stride, (ph, pw), mode)
        ref = O.pool2d_to_int8(x.float().numpy(), (kh, kw), stride, (ph, pw), mode)
        if tuple(y.shape) != ref.shape or not np.array_equal(y.cpu().numpy(), ref):
            report("pool", f"{mode} {dt}", (B, C, H, W, kh, kw, stride, ph, pw))


def fuzz_activation(n):
    """The dynamic per-tensor activation quantizer: parameters and int8 values against the numpy oracle bit for bit
    (Qop.py:62-65,157), the Linear / Conv2d built on it against the float64 value of the expression they state."""
    from quantization_amd.activation import conv2d_dynamic_w8a8, linear_dynamic_w8a8, quantize_tensor_i8
    g = torch.Generator(device=dev).manual_seed(seed + 6)
    for it in range(n):
        kind = int(rng.integers(0, 3))
        shift, spread = float(rng.normal() * 2), float(np.exp(rng.normal()))
        if kind == 0:
            shape = tuple(int(v) for v in rng.integers(1, pick([8, 40, 700]), size=int(rng.integers(1, 4))))
            x = torch.randn(shape, device=dev, generator=g) * spread + shift
            q, sc, zp = quantize_tensor_i8(x)
            rq, rs, rzp = O.quantize_tensor_affine_int8(x.cpu().numpy())
            if not (np.array_equal(sc.cpu().numpy(), rs) and np.array_equal(zp.cpu().numpy(), rzp)
                    and np.array_equal(q.cpu().numpy(), rq)):
                report("activation", "quantize_tensor", (shape, shift, spread))
            continue
        has_bias = bool(rng.integers(0, 2))
        if kind == 1:
            M, Nn, K = int(rng.integers(1, 300)), int(rng.integers(1, 500)), 16 * int(rng.integers(1, 80))
            x = torch.randn(M, K, device=dev, generator=g) * spread + shift
            wq, kw = ri8((Nn, K), g), None
            args = ("linear", M, Nn, K, has_bias)
        else:
            C, Co, k = int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.integers(1, 5))
            st, pd = int(rng.integers(1, 3)), int(rng.integers(0, 3))
            H, W = int(rng.integers(k, 30)), int(rng.integers(k, 30))
            x = torch.randn(int(rng.integers(1, 4)), C, H, W, device=dev, generator=g) * spread + shift
            wq, kw = ri8((Co, C, k, k), g), dict(stride=st, padding=pd)
            Nn, args = Co, ("conv", tuple(x.shape), Co, k, st, pd, has_bias)
        ws = torch.rand(Nn, device=dev, generator=g) * 0.01 + 1e-4
        b = torch.randn(Nn, device=dev, generator=g) if has_bias else None
        xq, sx, zx = quantize_tensor_i8(x)
        x_hat = (xq.double() - zx.double()) * sx.double()   # real zeros in a convolution's padding
        bd = b.double() if has_bias else torch.zeros(Nn, device=dev, dtype=torch.float64)
        if kw is None:
            y = linear_dynamic_w8a8(x, wq, ws, b)
            ref = (x_hat @ wq.double().t() + bd) * ws.double()
        else:
            y = conv2d_dynamic_w8a8(x, wq, ws, b, stride=st, padding=pd)
            ref = (F.conv2d(x_hat, wq.double(), **kw) + bd.view(1, -1, 1, 1)) * ws.double().view(1, -1, 1, 1)
        # (the zero-point term the epilogue subtracts is part of the scale of the sum: outputs that read padding only are
        # what is left of it after cancellation)
        zterm = float((zx.double().abs() * sx.double() * wq.double().abs().flatten(1).sum(1) * ws.double()).max())
        if y.shape != ref.shape or float((y.double() - ref).abs().max()) > 1e-5 * max(float(ref.abs().max()), zterm) + 1e-30:
            report("activation", "layer", args)


def fuzz_small_kernels(n):
    """The method-by-method pieces behind Qop (row maximum, quantize with given parameters, dequantize by row / by channel,
    the float epilogue) and the byte transpose, against torch's own fp32 operators (one rounding per step) bit for bit."""
    g = torch.Generator(device=dev).manual_seed(seed + 8)
    for it in range(n):
        kind = int(rng.integers(0, 6))
        rows = int(pick([1, 2, 3, 17, 64, 100, 255, 1000]))
        cols = int(pick([1, 2, 5, 16, 63, 64, 129, 1000, 4096, 5001]))
        dt = pick([torch.float32, torch.float16, torch.bfloat16])
        args = (kind, rows, cols, dt)
        if kind == 0:
            w = (torch.randn(rows, cols, device=dev, generator=g) * 3).to(dt)
            ok = torch.equal(N.row_max(w), w.float().max(dim=1).values)
        elif kind == 1:   # Qop.py:157: round(w / scale + zp), clamp; per row and per tensor
            w = (torch.randn(rows, cols, device=dev, generator=g) * 3).to(torch.float32)
            per_row = bool(rng.integers(0, 2))
            sc = torch.rand(rows if per_row else 1, device=dev, generator=g) * 0.05 + 1e-3
            zp = torch.randint(-20, 20, (rows if per_row else 1,), device=dev, generator=g).float() if rng.integers(0, 2) else None
            q = N.quantize_rows_given_i8(w, sc, zp)
            v = w / sc.view(-1, 1)
            if zp is not None:
                v = v + zp.view(-1, 1)
            ok = torch.equal(q, torch.clamp(torch.round(v), -128, 127).to(torch.int8))
        elif kind == 2:
            q = ri8((rows, cols), g)
            sc = torch.rand(rows, device=dev, generator=g) * 0.05 + 1e-3
            zp = torch.randint(-20, 20, (rows,), device=dev, generator=g) if rng.integers(0, 2) else None
            ref = sc.view(-1, 1) * (q.float() - (zp.float().view(-1, 1) if zp is not None else 0.0))
            ok = torch.equal(N.dequantize_rows_i8(q, sc, zp), ref)
        elif kind in (3, 4):
            C, inner = int(pick([1, 3, 64, 130])), int(pick([1, 7, 49, 196, 1024]))
            y = torch.randn(min(rows, 9), C, inner, device=dev, generator=g) * 100
            a = torch.rand(C, device=dev, generator=g) * 0.05 + 1e-3
            b = torch.randn(C, device=dev, generator=g) * 5 if rng.integers(0, 2) else None
            if kind == 3:   # Qop.dequantize with activation=True: scale[c] * (y - zp[c])
                ref = a.view(1, -1, 1) * (y - (b.view(1, -1, 1) if b is not None else 0.0))
                ok = torch.equal(N.dequantize_channels_f32(y, a, b), ref)
            else:
                ref = (y + (b.view(1, -1, 1) if b is not None else 0.0)) * a.view(1, -1, 1) + 0.75
                ok = torch.equal(N.epilogue_channels_f32(y, b, a, 0.75), ref)
            args = (kind, tuple(y.shape), b is not None)
        else:
            src = ri8((rows, cols), g)
            ok = torch.equal(N.transpose_i8(src), src.t().contiguous())
        if not ok:
            report("small_kernels", "mismatch", args)


FAMILIES = {"gemm": fuzz_gemm, "conv": fuzz_conv, "gemm_w8": fuzz_gemm_w8, "conv_w8": fuzz_conv_w8,
            "conv_general": fuzz_conv_general, "quantizer": fuzz_quantizer, "layers": fuzz_layers,
            "layers_wa": fuzz_layers_wa, "pool": fuzz_pool, "activation": fuzz_activation,
            "small_kernels": fuzz_small_kernels}


def run(families, cases, s):
    """Runs `cases` random cases of each named family with seed `s`; returns the number of mismatches (each one printed)."""
    reseed(s)
    t0 = time.time()
    for name in families:
        FAMILIES[name](cases)
        print(f"{name}: {cases} cases, {bad} mismatches so far, {time.time() - t0:.0f} s", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    s_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fams = sys.argv[3].split(",") if len(sys.argv) > 3 else list(FAMILIES)
    sys.exit(1 if run(fams, n_cases, s_arg) else 0)
