#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference's own Python on CPU.

Run in the build container only (it needs /root/reference); the GPU box never sees the
reference, only the small fixtures this script writes.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden.py            # rewrites tests/golden/

What is imported, unmodified, from the reference checkout:
  quant/Weight_Only/{Qop,Quantizer,Chunker,ModelAnalyzer,models}.py
  quant/Weight_Activation/{Qop,Quantizer,Chunker,ModelAnalyzer,models}.py
quant/Weight_Only/Qop.py imports seaborn (unused, Qop.py:6), which is not installed here;
an empty module object named 'seaborn' is placed in sys.modules so the import statement
succeeds.  Nothing else is stubbed.
"""
import hashlib
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get("MIQ_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

Q_PARAMS = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}


def load_ref(pkg):
    """Import the flat-module package quant/<pkg> and return its modules."""
    for name in ("Qop", "Quantizer", "Chunker", "ModelAnalyzer", "models"):
        sys.modules.pop(name, None)
    path = os.path.join(REF, "quant", pkg)
    sys.path.insert(0, path)
    try:
        if "seaborn" not in sys.modules:
            sys.modules["seaborn"] = types.ModuleType("seaborn")
        mods = {n: importlib.import_module(n) for n in ("Qop", "Quantizer", "Chunker", "models")}
    finally:
        sys.path.remove(path)
    return mods


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def as_f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


EDGE = np.array([[-1, -2, -.5, -3], [0, 0, 0, 0], [1.275, .005, .015, -.025], [2, -4, 1, .5]],
                dtype=np.float32)


def quantizer_cases(Qop, full=True):
    """(i)-(iii): channel-symmetric int8 quantization of fp32 / fp16 / bf16 weights."""
    out = {}

    def run(name, w):
        q = Qop(dtype=torch.int8, symmetric=True, affine="channel", affine_dim=0)
        qt = q.quantize(w)
        out[name + "/w"] = as_f32(w)
        out[name + "/q"] = qt.numpy()
        out[name + "/scales"] = as_f32(q.scales)
        out[name + "/zp"] = q.zero_point.numpy()
        assert qt.dtype == torch.int8 and q.zero_point.dtype == torch.int64
        assert q.scales.dtype == w.dtype, (q.scales.dtype, w.dtype)

    run("edge", torch.from_numpy(EDGE.copy()))
    g = torch.Generator().manual_seed(1234)
    shapes = [(4, 6), (64, 27), (10, 512), (257, 1000), (9, 4096), (5, 1), (2, 8200)]
    for (n, k) in (shapes if full else shapes[:3]):
        w = torch.randn(n, k, generator=g)
        if n >= 10:
            w[1] = -w[1].abs() - 0.01          # all-negative row -> scale = eps, all -128
            w[2] = 0                           # all-zero row
            w[3, 0] = 1e-9                     # tiny positive max -> floor at eps
        run(f"f32_{n}x{k}", w)
    # ties: multiples of scale/2 so that round-half-even matters
    w = (torch.arange(-300, 300, dtype=torch.float32).view(4, 150) * 0.5)
    w[:, 0] = 127.5                            # scale = 1.0 exactly
    run("ties", w)
    # conv-shaped weights
    cshapes = [(5, 3, 3, 3), (16, 8, 3, 3), (8, 4, 1, 1), (6, 3, 7, 7)]
    for shape in (cshapes if full else cshapes[:1]):
        run("conv_" + "x".join(map(str, shape)), torch.randn(*shape, generator=g))
    # reduced precision inputs keep their dtype in the scales (F7)
    for dt, nm in [(torch.float16, "f16"), (torch.bfloat16, "bf16")]:
        for (n, k) in ([(8, 64), (17, 300)] if full else [(8, 64)]):
            w = torch.randn(n, k, generator=g).to(dt)
            w[1] = -w[1].abs() - 0.01
            run(f"{nm}_{n}x{k}", w)
        run(f"{nm}_conv", torch.randn(6, 4, 3, 3, generator=g).to(dt))
    return out


def wo_forward_cases(mods):
    """(iv): weight-only Quantizer.from_float + forward on integer-valued activations."""
    Quantizer = mods["Quantizer"].Quantizer
    out = {}
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(99)
    cases = {
        "lin_6_4": (torch.nn.Linear(6, 4), (3, 6)),
        "lin_64_48_nobias": (torch.nn.Linear(64, 48, bias=False), (5, 64)),
        "lin_300_130": (torch.nn.Linear(300, 130), (37, 300)),
        "conv_3_5_p1": (torch.nn.Conv2d(3, 5, 3, padding=1), (2, 3, 8, 8)),
        "conv_8_16_s2": (torch.nn.Conv2d(8, 16, 3, stride=2, padding=1), (2, 8, 9, 11)),
        "conv_4_6_1x1_nobias": (torch.nn.Conv2d(4, 6, 1, bias=False), (1, 4, 5, 5)),
        # strided 1 x 1 layers (appended last: the cases above keep their random draws)
        "conv_12_20_1x1_s2_nobias": (torch.nn.Conv2d(12, 20, 1, stride=2, bias=False), (2, 12, 9, 16)),
        "conv_5_7_1x1_s3": (torch.nn.Conv2d(5, 7, 1, stride=3), (2, 5, 10, 11)),
    }
    for name, (mod, xshape) in cases.items():
        ql = Quantizer.from_float(module=mod, data_metry=Q_PARAMS, quantize_output=False)
        x = torch.randint(-128, 128, xshape, generator=g, dtype=torch.int32).to(torch.int8)
        y = ql(x.float())
        out[name + "/w"] = as_f32(mod.weight)
        if mod.bias is not None:
            out[name + "/bias"] = as_f32(mod.bias)
        out[name + "/x"] = x.numpy()
        out[name + "/q"] = ql.weight.numpy()
        out[name + "/scales"] = as_f32(ql.weight_quant.scales).reshape(-1)
        out[name + "/y"] = as_f32(y)
        if isinstance(mod, torch.nn.Conv2d):
            out[name + "/stride_pad"] = np.array([mod.stride[0], mod.padding[0]], dtype=np.int64)
        out[name + "/repr"] = np.frombuffer(repr(ql).encode(), dtype=np.uint8)
    return out


def loader(batches, bs, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(bs, 3, 32, 32, generator=g), torch.randint(0, 10, (bs,), generator=g))
            for _ in range(batches)]


def wa_chunker_case(mods):
    """(v): Weight_Activation Chunker on seeded SimpleCNN: folded epilogue parameters."""
    torch.manual_seed(0)
    model = mods["models"].SimpleCNN().eval()
    fp_weights = {n: as_f32(p) for n, p in model.named_parameters()}
    data = loader(2, 4, 5)
    qmodel = mods["Chunker"].Chunker(model, data).model
    out = {}
    for k, v in fp_weights.items():
        out["fp_sha/" + k] = np.frombuffer(sha(v).encode(), dtype=np.uint8)
        if v.size <= 40000:
            out["fp/" + k] = v
    out["calib_x"] = np.stack([as_f32(x) for x, _ in data])
    for name in ("conv1", "conv2", "fc1", "fc2"):
        layer = getattr(qmodel, name)
        out[f"{name}/sha_weight"] = np.frombuffer(sha(layer.weight.numpy()).encode(), dtype=np.uint8)
        if layer.weight.numel() <= 40000:
            out[f"{name}/weight"] = layer.weight.numpy()
        out[f"{name}/weight_qscale"] = as_f32(layer.weight_qscale).reshape(-1)
        out[f"{name}/bias"] = as_f32(layer.bias).reshape(-1)
        out[f"{name}/scalers"] = as_f32(layer.scalers).reshape(-1)
        out[f"{name}/output_zero_point"] = as_f32(layer.output_zero_point).reshape(-1)
        out[f"{name}/input_qscale"] = np.array([layer.input_qscale], dtype=np.float32)
        out[f"{name}/input_zero_point"] = np.array([layer.input_zero_point], dtype=np.float32)
    # one layer forward on integer-valued activations: the epilogue formula in isolation
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-128, 128, (2, 3, 8, 8), generator=g, dtype=torch.int32).to(torch.int8)
    out["conv1/x"] = x.numpy()
    out["conv1/y"] = as_f32(qmodel.conv1(x.float()))
    xf = torch.randint(-128, 128, (3, 512), generator=g, dtype=torch.int32).to(torch.int8)
    out["fc2/x"] = xf.numpy()
    out["fc2/y"] = as_f32(qmodel.fc2(xf.float()))
    return out


def config1_models(mods):
    """Config 1 of BASELINE.json: the sample models, seeded default init, weight-only Chunker.

    Weights are too large for a fixture; digests of the fp32 weights (to prove the test
    regenerated the same input) and of the reference's int8 weights and scales are stored,
    plus the full tensors of each model's smallest layers.
    """
    m = mods["models"]
    out = {}
    for mname, ctor in [("SimpleCNN", m.SimpleCNN), ("LinearNet", lambda: m.LinearNet(3072, 10)),
                        ("VGG", m.VGG)]:
        torch.manual_seed(0)
        model = ctor().eval()
        fp = {n: mod.weight.detach().clone() for n, mod in model.named_modules()
              if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear))}
        x = loader(1, 2, 11)[0][0]
        with torch.no_grad():
            out[mname + "/y_fp"] = as_f32(model(x))            # the float model, before the rewrite
        qmodel = mods["Chunker"].Chunker(model, loader(1, 2, 11)).model
        with torch.no_grad():
            out[mname + "/y_quant"] = as_f32(qmodel(x))        # the reference flow: Chunker(model).model, then a forward
        names = []
        for n, mod in qmodel.named_modules():
            if type(mod).__name__ != "Quantizer":
                continue
            names.append(n)
            key = f"{mname}/{n}"
            out[key + "/shape"] = np.array(mod.weight.shape, dtype=np.int64)
            out[key + "/sha_w"] = np.frombuffer(sha(as_f32(fp[n])).encode(), dtype=np.uint8)
            out[key + "/sha_q"] = np.frombuffer(sha(mod.weight.numpy()).encode(), dtype=np.uint8)
            out[key + "/sha_scales"] = np.frombuffer(
                sha(as_f32(mod.weight_quant.scales)).encode(), dtype=np.uint8)
            if mod.weight.numel() <= 40000:
                out[key + "/w"] = as_f32(fp[n])
                out[key + "/q"] = mod.weight.numpy()
                out[key + "/scales"] = as_f32(mod.weight_quant.scales).reshape(-1)
        out[mname + "/layers"] = np.frombuffer(",".join(names).encode(), dtype=np.uint8)
    return out


POOL_CASES = [
    # name, mode, x dtype, shape, kernel, stride, padding, value scale
    ("max_k2", "max", "f32", (2, 3, 8, 8), 2, 2, 0, 60.0),
    ("max_k3s2p1", "max", "f32", (2, 4, 9, 11), 3, 2, 1, 200.0),      # values past int8: the cast wraps
    ("max_rect", "max", "f32", (1, 2, 10, 13), (2, 3), (2, 1), (1, 1), 50.0),
    ("avg_k2", "avg", "f32", (2, 3, 8, 8), 2, 2, 0, 60.0),
    ("avg_k3s1p1", "avg", "f32", (1, 2, 7, 10), 3, 1, 1, 100.0),      # padding counts in the divisor
    ("avg_rect", "avg", "f32", (2, 2, 9, 12), (3, 2), (1, 2), (1, 0), 300.0),
    ("max_i8", "max", "i8", (2, 3, 12, 12), 2, 2, 0, 0.0),
    ("avg_i8", "avg", "i8", (2, 3, 11, 14), 3, 2, 1, 0.0),
    ("max_f16", "max", "f16", (1, 3, 8, 16), 2, 2, 0, 40.0),
    ("avg_bf16", "avg", "bf16", (1, 3, 8, 16), 2, 2, 0, 40.0),
]


def pool_input(name, kind, shape, scale):
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    if kind == "i8":
        return torch.randint(-128, 128, shape, generator=g, dtype=torch.int32).to(torch.int8)
    x = torch.randn(shape, generator=g) * scale
    return x.to({"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[kind])


def pool_cases(mods):
    """QuantizedMaxPool2d / QuantizedAvgPool2d (quant/Weight_Activation/Quantizer.py:122-130), constructed with
    the arguments the Chunker forwards (kernel_size, stride, padding: Chunker.py:75-77)."""
    Q = mods["Quantizer"]
    out = {}
    for name, mode, kind, shape, k, s, p, scale in POOL_CASES:
        x = pool_input(name, kind, shape, scale)
        cls = Q.QuantizedMaxPool2d if mode == "max" else Q.QuantizedAvgPool2d
        y = cls(kernel_size=k, stride=s, padding=p)(x)
        assert y.dtype == torch.int8
        out[name + "/x"] = x.numpy() if kind == "i8" else as_f32(x)
        out[name + "/y"] = y.numpy()
    return out


def wo_forward_float_cases(mods):
    """Weight-only Quantizer.from_float + forward of nn.Linear on REAL-valued activations (the reference's fp32
    product; bf16 activations for one case, which the reference multiplies in bf16)."""
    Quantizer = mods["Quantizer"].Quantizer
    out = {}
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(123)
    cases = {
        "f32_lin_300_130": (torch.nn.Linear(300, 130), (37, 300), torch.float32, 1.0),
        "f32_lin_1024_40_nobias": (torch.nn.Linear(1024, 40, bias=False), (8, 1024), torch.float32, 1.0),
        "f32_lin_48_200_wide": (torch.nn.Linear(48, 200), (65, 48), torch.float32, 50.0),
        "f32_lin_2048_32": (torch.nn.Linear(2048, 32), (2, 2048), torch.float32, 0.1),
        "bf16_lin_256_72": (torch.nn.Linear(256, 72), (19, 256), torch.bfloat16, 1.0),
        "f32_conv_3_5_p1": (torch.nn.Conv2d(3, 5, 3, padding=1), (2, 3, 8, 8), torch.float32, 1.0),
        "f32_conv_8_16_s2": (torch.nn.Conv2d(8, 16, 3, stride=2, padding=1), (2, 8, 9, 11), torch.float32, 3.0),
        "f32_conv_20_40_1x1_nobias": (torch.nn.Conv2d(20, 40, 1, bias=False), (3, 20, 5, 5), torch.float32, 1.0),
        "f32_conv_16_24_5x5": (torch.nn.Conv2d(16, 24, 5, padding=2), (2, 16, 12, 12), torch.float32, 0.2),
        # strided 1 x 1 layers (appended last: the cases above keep their random draws)
        "f32_conv_24_40_1x1_s2": (torch.nn.Conv2d(24, 40, 1, stride=2), (2, 24, 9, 16), torch.float32, 1.0),
        "f32_conv_20_9_1x1_s3_nobias": (torch.nn.Conv2d(20, 9, 1, stride=3, bias=False), (2, 20, 10, 11), torch.float32, 2.0),
    }
    for name, (mod, xshape, dt, mag) in cases.items():
        ql = Quantizer.from_float(module=mod, data_metry=Q_PARAMS, quantize_output=False)
        x = (torch.randn(xshape, generator=g) * mag).to(dt)
        if dt != torch.float32:
            ql.bias = None if ql.bias is None else torch.nn.Parameter(ql.bias.to(dt))
        y = ql(x)
        out[name + "/w"] = as_f32(mod.weight)
        if mod.bias is not None:
            out[name + "/bias"] = as_f32(mod.bias)
        out[name + "/x"] = as_f32(x)
        out[name + "/q"] = ql.weight.numpy()
        out[name + "/scales"] = as_f32(ql.weight_quant.scales).reshape(-1)
        out[name + "/y"] = as_f32(y)
        if isinstance(mod, torch.nn.Conv2d):
            out[name + "/stride_pad"] = np.array([mod.stride[0], mod.padding[0]], dtype=np.int64)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if sys.argv[1:] == ["wo_float"]:
        wo = load_ref("Weight_Only")
        np.savez_compressed(os.path.join(OUT, "wo_forward_float.npz"), **wo_forward_float_cases(wo))
        print("wo_forward_float.npz", os.path.getsize(os.path.join(OUT, "wo_forward_float.npz")))
        return
    if sys.argv[1:] == ["wo_forward"]:
        wo = load_ref("Weight_Only")
        np.savez_compressed(os.path.join(OUT, "wo_forward.npz"), **wo_forward_cases(wo))
        print("wo_forward.npz", os.path.getsize(os.path.join(OUT, "wo_forward.npz")))
        return
    if sys.argv[1:] == ["config1"]:
        wo = load_ref("Weight_Only")
        np.savez_compressed(os.path.join(OUT, "config1_models.npz"), **config1_models(wo))
        print("wrote config1_models.npz")
        return
    if sys.argv[1:] == ["pools"]:
        wa = load_ref("Weight_Activation")
        np.savez_compressed(os.path.join(OUT, "pools.npz"), **pool_cases(wa))
        print("pools.npz", os.path.getsize(os.path.join(OUT, "pools.npz")))
        return
    wo = load_ref("Weight_Only")
    np.savez_compressed(os.path.join(OUT, "quantizer_wo.npz"), **quantizer_cases(wo["Qop"].Qop))
    np.savez_compressed(os.path.join(OUT, "wo_forward.npz"), **wo_forward_cases(wo))
    np.savez_compressed(os.path.join(OUT, "wo_forward_float.npz"), **wo_forward_float_cases(wo))
    np.savez_compressed(os.path.join(OUT, "config1_models.npz"), **config1_models(wo))
    wa = load_ref("Weight_Activation")
    np.savez_compressed(os.path.join(OUT, "quantizer_wa.npz"), **quantizer_cases(wa["Qop"].Qop, full=False))
    np.savez_compressed(os.path.join(OUT, "wa_chunker.npz"), **wa_chunker_case(wa))
    np.savez_compressed(os.path.join(OUT, "pools.npz"), **pool_cases(wa))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
