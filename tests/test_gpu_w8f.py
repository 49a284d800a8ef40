"""Weight-only Linear on floating-point activations (miq_gemm_w8, through the C ABI) against an fp64 product.

Bar: the reference computes this product in fp32 (`F.linear(x, weight.to(x.dtype))`,
quant/Weight_Only/Quantizer.py:79-91) and the dequantised output must agree within 1e-3 relative.  The kernel
carries fp32 activations as two bf16 planes (residual below 2^-17 |x|) with exact bf16 x int8 products and fp32
accumulation, so the test holds it to a far tighter bound than 1e-3: per element
    |y - ref| <= 5e-5 * alpha[n] * (sum_k |x[m,k] * w[n,k]| + |bias[n]|)
(the term a cancelling sum is measured against), and 1e-4 of the output's norm overall.
"""
import numpy as np
import pytest
import torch

from conftest import Golden
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

FF = Golden("wo_forward_float.npz")
Q_PARAMS = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}

DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def make(M, Nn, K, xdt, seed, bias=True, wide=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    if wide:
        x = x * torch.exp(4 * torch.randn(M, K, generator=g))   # magnitudes over many binades
        if xdt == "f16":
            x = x.clamp(-60000.0, 60000.0)
    x = x.to(DT[xdt])
    w = torch.randint(-128, 128, (Nn, K), generator=g, dtype=torch.int32).to(torch.int8)
    alpha = torch.rand(Nn, generator=g) * 0.02 + 1e-4
    b = torch.randn(Nn, generator=g) * 10 if bias else None
    return x, w, b, alpha


def check(N, x, w, b, alpha, beta=0.0, out_dtype=torch.float32):
    y = N.gemm_w8(x.cuda(), w.cuda(), bias=None if b is None else b.cuda(), alpha=alpha.cuda(), beta=beta,
                  out_dtype=out_dtype)
    torch.cuda.synchronize()
    assert y.dtype == out_dtype and tuple(y.shape) == (x.shape[0], w.shape[0])
    xd, wd = x.double(), w.double()
    bd = torch.zeros(w.shape[0], dtype=torch.float64) if b is None else b.double()
    ref = (xd @ wd.t() + bd) * alpha.double() + beta
    mag = (xd.abs() @ wd.abs().t() + bd.abs()) * alpha.double() + abs(beta)
    err = (y.cpu().double() - ref).abs()
    tol = 5e-5 * mag + 1e-30
    if out_dtype == torch.bfloat16:
        tol = tol + ref.abs() * 2.0 ** -8
    worst = float((err / tol).max()) if err.numel() else 0.0
    assert worst <= 1.0, f"worst error / tolerance = {worst:.3f}"
    if err.numel() and out_dtype == torch.float32:
        assert float(err.norm() / ref.norm().clamp_min(1e-30)) <= 1e-4
    return y


@pytest.mark.parametrize("case", [c for c in FF.cases() if "conv" not in c])
def test_w8f_golden_reference_outputs(N, case):
    """The reference's own Quantizer.forward outputs on real-valued activations (fixture made by
    oracle/gen_golden.py from the reference's Python), through the C ABI and through the drop-in layer."""
    x, q, s, ref = FF.get(case, "x"), FF.get(case, "q"), FF.get(case, "scales"), FF.get(case, "y")
    bias = FF.get(case, "bias")
    xdt = torch.bfloat16 if case.startswith("bf16") else torch.float32
    xt = torch.from_numpy(x).to(xdt)
    y = N.gemm_w8(xt.cuda(), torch.from_numpy(q).cuda(), bias=None if bias is None else torch.from_numpy(bias).cuda(),
                  alpha=torch.from_numpy(s).cuda()).cpu().numpy()
    mag = (np.abs(x).astype(np.float64) @ np.abs(q.astype(np.float64)).T + (0 if bias is None else np.abs(bias))) * s
    # against the oracle: 5e-5 of the sum of magnitudes; against the reference: 1e-3 (bf16 activations: the
    # reference rounds its product to bf16, so bf16 resolution)
    np.testing.assert_array_less(np.abs(y - O.wo_linear_forward_float(x, q, s, bias)), 5e-5 * mag + 1e-30)
    np.testing.assert_array_less(np.abs(y - ref), (2.0 ** -6 if xdt == torch.bfloat16 else 1e-3) * np.maximum(np.abs(ref), 1e-2 * mag) + 1e-30)

    from quantization_amd.quant.Weight_Only import Quantizer
    lin = torch.nn.Linear(q.shape[1], q.shape[0], bias=bias is not None)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(FF.get(case, "w")))
        if bias is not None:
            lin.bias.copy_(torch.from_numpy(bias))
    ql = Quantizer.from_float(module=lin, data_metry=Q_PARAMS, quantize_output=False)
    np.testing.assert_array_equal(ql.weight.cpu().numpy(), q)
    y2 = ql(xt.cuda()).float().cpu().numpy()
    np.testing.assert_array_equal(y2, y)
    y3 = ql(xt)          # host activations are staged to the GPU and the result comes home
    assert y3.device.type == "cpu"
    np.testing.assert_array_equal(y3.float().numpy(), y)


SHAPES = [
    (1, 64, 128), (3, 10, 16), (16, 64, 256), (17, 100, 48), (33, 257, 400), (64, 64, 4096 + 16),
    (65, 130, 1040), (129, 96, 2064), (257, 300, 528), (300, 1000, 640), (5, 7, 16), (1, 1, 16),
]


@pytest.mark.parametrize("xdt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("M,Nn,K", SHAPES)
def test_w8f_shapes(N, M, Nn, K, xdt):
    x, w, b, alpha = make(M, Nn, K, xdt, seed=M * 7 + Nn * 3 + K)
    check(N, x, w, b, alpha)


@pytest.mark.parametrize("xdt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("M,Nn,K", [(4, 33, 24), (9, 12, 7), (2, 5, 1), (40, 17, 100)])
def test_w8f_unaligned_k_takes_the_scalar_kernel(N, M, Nn, K, xdt):
    x, w, b, alpha = make(M, Nn, K, xdt, seed=K)
    check(N, x, w, b, alpha, beta=0.25)


@pytest.mark.parametrize("xdt", ["f32", "bf16"])
@pytest.mark.parametrize("M,Nn,K", [(1, 512, 8192), (8, 1024, 4096), (32, 256, 14336), (64, 1000, 6144), (200, 512, 4096)])
def test_w8f_split_k(N, M, Nn, K, xdt):
    from quantization_amd import _native
    assert _native.lib().miq_gemm_w8_workspace(0, M, Nn, K) > 2 * ((M * K * 2 + 255) // 256 * 256), "expected a split-K plan"
    x, w, b, alpha = make(M, Nn, K, xdt, seed=K + M)
    check(N, x, w, b, alpha)


@pytest.mark.parametrize("xdt", ["f32", "f16"])
def test_w8f_wide_range_activations(N, xdt):
    x, w, b, alpha = make(96, 200, 1024, xdt, seed=5, wide=True)
    check(N, x, w, None, alpha)


def test_w8f_f16_planes_are_exact(N):
    """An fp16 value is the sum of its two bf16 planes: integer-valued fp16 activations give the exact integer sums."""
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-2048, 2049, (40, 512), generator=g).to(torch.float16)
    w = torch.randint(-128, 128, (72, 512), generator=g, dtype=torch.int32).to(torch.int8)
    one = torch.ones(72)
    y = N.gemm_w8(x.cuda(), w.cuda(), alpha=one.cuda()).cpu()
    ref = x.double() @ w.double().t()     # |sum| < 2^24: exact in fp32 whatever the order
    assert float(ref.abs().max()) < 2 ** 24
    assert torch.equal(y.double(), ref)


@pytest.mark.parametrize("xdt", ["f32", "bf16"])
def test_w8f_bf16_out_and_no_bias(N, xdt):
    x, w, b, alpha = make(70, 136, 768, xdt, seed=11, bias=False)
    check(N, x, w, None, alpha, beta=-0.5, out_dtype=torch.bfloat16)


def test_w8f_specials_propagate(N):
    x, w, b, alpha = make(8, 64, 256, "f32", seed=2)
    x[1, 5] = float("inf")
    x[2, 7] = float("nan")
    x[3, 9] = 3.4e38          # rounds past the largest bf16: must stay finite
    w[:, 9] = 0
    y = N.gemm_w8(x.cuda(), w.cuda(), bias=b.cuda(), alpha=alpha.cuda()).cpu()
    assert torch.isnan(y[2]).all() and not torch.isfinite(y[1]).any()
    assert torch.isfinite(y[0]).all() and torch.isfinite(y[3]).all()


def test_w8f_llama_mlp_shape(N):
    x, w, b, alpha = make(256, 14336, 4096, "f32", seed=0)
    check(N, x, w, b, alpha)
    x, w, b, alpha = make(16, 4096, 14336, "bf16", seed=1)
    check(N, x, w, b, alpha)


@pytest.mark.parametrize("xdt", ["bf16", "f32"])
def test_w8f_large_m_tiles(N, xdt):
    """M > 128 with enough 128 x 256 tiles to fill the chip (bf16 takes that tile), ragged in all three dimensions."""
    x, w, b, alpha = make(1300, 5000, 272, xdt, seed=21)
    check(N, x, w, b, alpha)
    x, w, b, alpha = make(4100, 4100, 144, xdt, seed=22)   # enough 256 x 256 tiles for the pipelined kernel (bf16)
    check(N, x, w, b, alpha)


@pytest.mark.parametrize("M,Nn,K", [(256, 256, 64), (300, 520, 80), (1000, 777 * 4, 1040), (513, 260, 4096 + 48),
                                    (2048, 2048, 2048), (256, 512, 16)])
@pytest.mark.parametrize("xdt", ["bf16", "f32", "f16"])
def test_w8f_pipelined_256_tile(N, M, Nn, K, xdt, monkeypatch):
    """The software-pipelined 256 x 256 kernel (one or two activation planes), forced through the developer switches so that
    ragged edges in all three dimensions and one-tile problems reach it; each case runs three times on fresh
    data (a staging race would show as a wrong tile in some run).  The switches exist only in the developer
    build of the library (same sources, same kernels), which this test loads in place of the product."""
    from quantization_amd import build as miq_build
    monkeypatch.setattr(N, "_lib", N.load(miq_build.DEV_LIB))
    monkeypatch.setenv("MIQ_W8F_BM", "256")
    monkeypatch.setenv("MIQ_W8F_BN", "256")
    for rep in range(3):
        x, w, b, alpha = make(M, Nn, K, xdt, seed=31 * rep + M)
        check(N, x, w, b, alpha, out_dtype=torch.float32 if rep < 2 else torch.bfloat16)


def test_sharded_layer_takes_float_and_int8_activations(N):
    """ColumnShardedLinear on one GPU (world size 1: the layer owns every row): int8 activations give the int8
    product, floating-point activations the weight-only product, both equal to the unsharded calls."""
    from quantization_amd.sharded import ColumnShardedLinear
    g = torch.Generator().manual_seed(4)
    w = (torch.randn(200, 320, generator=g) * 0.05).cuda()
    b = torch.randn(200, generator=g).cuda()
    layer = ColumnShardedLinear.from_float(w, b)
    q, scale, _ = N.quantize_rows_sym_i8(w)
    xi = torch.randint(-128, 128, (9, 320), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    assert torch.equal(layer(xi), N.gemm_i8(xi, q, bias=b, alpha=scale))
    for dt in (torch.float32, torch.bfloat16):
        xf = torch.randn(9, 320, generator=g).to(dt).cuda()
        assert torch.equal(layer(xf), N.gemm_w8(xf, q, bias=b, alpha=scale))


def test_launches_capture_into_a_hip_graph(N):
    """Nothing in the library allocates or synchronises: an int8 GEMM and a weight-only GEMM captured into one
    graph replay to the eager results, also after the inputs change in place."""
    g = torch.Generator().manual_seed(8)
    w = torch.randint(-128, 128, (384, 512), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    alpha = (torch.rand(384, generator=g) * 0.01).cuda()
    xi = torch.randint(-128, 128, (24, 512), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    xf = torch.randn(24, 512, generator=g).cuda()
    oi, of = torch.empty(24, 384).cuda(), torch.empty(24, 384).cuda()

    def step():
        N.gemm_i8(xi, w, alpha=alpha, out=oi)
        N.gemm_w8(xf, w, alpha=alpha, out=of)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    xi.copy_(torch.randint(-128, 128, (24, 512), generator=g, dtype=torch.int32).to(torch.int8))
    xf.copy_(torch.randn(24, 512, generator=g))
    graph.replay()
    torch.cuda.synchronize()
    gi, gf = oi.clone(), of.clone()
    step()
    torch.cuda.synchronize()
    assert torch.equal(gi, oi) and torch.equal(gf, of)


def test_w8f_random_shapes(N):
    """120 random (M, N, K, dtype, bias, out dtype) problems across every tile the plan can pick."""
    import random
    rnd = random.Random(2024)
    for i in range(120):
        M = rnd.choice([rnd.randint(1, 16), rnd.randint(17, 64), rnd.randint(65, 128), rnd.randint(129, 700)])
        Nn = rnd.choice([rnd.randint(1, 63), rnd.randint(64, 600), rnd.randint(601, 1800)])
        K = 16 * rnd.randint(1, 180)
        if rnd.random() < 0.1:
            K += rnd.randint(1, 15)      # scalar kernel
        xdt = rnd.choice(["f32", "f16", "bf16"])
        x, w, b, alpha = make(M, Nn, K, xdt, seed=1000 + i, bias=rnd.random() < 0.7)
        check(N, x, w, b, alpha, beta=rnd.choice([0.0, 1.5]),
              out_dtype=torch.bfloat16 if rnd.random() < 0.25 else torch.float32)


@pytest.mark.parametrize("xdt", ["bf16", "f32", "f16"])
@pytest.mark.parametrize("M,Nn,K", [(8192, 8192, 8192), (256, 14336, 4096), (16, 4096, 14336)])
def test_w8f_full_size_equals_the_int8_product(N, M, Nn, K, xdt):
    """Full BASELINE sizes, every element: on small integer activations (|x| <= 8, so every partial sum stays below
    2^24 and fp32 accumulation is exact in any order) the weight-only kernel must reproduce the int8 GEMM's exact
    int32 sums, epilogue included."""
    g = torch.Generator(device="cuda").manual_seed(5)
    xi = torch.randint(-8, 9, (M, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Nn, generator=g, device="cuda") * 0.01 + 1e-4
    bias = torch.randn(Nn, generator=g, device="cuda")
    ref = N.gemm_i8(xi, w, bias=bias, alpha=alpha)
    y = N.gemm_w8(xi.to(DT[xdt]), w, bias=bias, alpha=alpha)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("xdt", ["bf16", "f32", "f16"])
@pytest.mark.parametrize("M,Nn,K", [(2304, 8192, 80), (2100, 8192, 144), (4400, 8190, 64)])
def test_w8f_whole_rounds_plus_tail(N, M, Nn, K, xdt):
    """A count of 256 x 256 tiles a little over whole rounds of the CUs runs as the whole rounds plus a second launch
    for the remaining rows (both into the same output)."""
    assert N.lib().miq_gemm_w8_workspace(2, M, Nn, K) >= 0
    x, w, b, alpha = make(M, Nn, K, xdt, seed=M)
    check(N, x, w, b, alpha)


def test_w8f_argument_checks(N):
    x = torch.randn(4, 32).cuda()
    w = torch.zeros(8, 32, dtype=torch.int8).cuda()
    a = torch.ones(8).cuda()
    with pytest.raises(RuntimeError):
        N.gemm_w8(x.cpu(), w, alpha=a)
    with pytest.raises(RuntimeError):
        N.gemm_w8(x, w[:, :16].contiguous(), alpha=a)
    with pytest.raises(RuntimeError):
        N.gemm_w8(x.to(torch.int8), w, alpha=a)
    with pytest.raises(RuntimeError):
        N.gemm_w8(x, w, alpha=None)
    with pytest.raises(RuntimeError):
        N.gemm_w8(x, w, alpha=a, out_dtype=torch.int32)
    L = N.lib()
    assert L.miq_gemm_w8(x.data_ptr(), 0, w.data_ptr(), 4, 8, 32, None, a.data_ptr(), 0.0, None, 0, None, 0, None) == -1
    assert L.miq_gemm_w8(x.data_ptr(), 5, w.data_ptr(), 4, 8, 32, None, a.data_ptr(), 0.0, x.data_ptr(), 0, None, 0, None) == -1
    out = torch.empty(4, 8).cuda()
    assert L.miq_gemm_w8(x.data_ptr(), 0, w.data_ptr(), 4, 8, 32, None, a.data_ptr(), 0.0, out.data_ptr(), 0, None, 0, None) == -3
    assert N.gemm_w8(torch.empty(0, 32).cuda(), w, alpha=a).shape == (0, 8)
