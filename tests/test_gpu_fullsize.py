"""BASELINE.json's full-size configurations on the GPU, checked through size-independent
properties (the CPU oracle would take minutes to hours at these sizes):

* quantizer: scales equal max(amax, 0) / 127.5 floored at 2^-23 for EVERY row (torch.amax is exact,
  the N divisions are done on the host in numpy), and the int8 rows of a random sample of rows equal
  the oracle's bit for bit;
* int32 GEMM: column checksum  sum_m C[m, :] == (sum_m A[m, :]) . W^T  and row checksum
  C . 1 == A . (W^T 1), both exact in float64 / int64, plus sampled rows recomputed exactly;
* fused epilogue: sampled rows against the oracle's epilogue on exact accumulators;
* conv: the same checksum idea over the batch for a ResNet-18 layer at batch 256, and a sampled
  image against F.conv2d in float64.
"""
import numpy as np
import pytest
import torch

from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def rand_i8(shape, seed, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randint(-128, 128, shape, generator=g, dtype=torch.int32, device=device).to(torch.int8)


@pytest.mark.parametrize("rows,cols,std", [(14336, 4096, 0.02), (4096, 14336, 0.02), (28672, 8192, 0.02),
                                            (8192, 28672, 0.02), (8192, 8192, 1.0)])
def test_quantizer_full_size(N, rows, cols, std):
    g = torch.Generator(device="cuda").manual_seed(rows + cols)
    w = torch.randn(rows, cols, generator=g, device="cuda") * std
    w[5] = -w[5].abs() - 1.0           # an all-negative row: scale floors at eps, all -128
    w[6] = 0
    q, scale, zp, rmax = N.quantize_rows_sym_i8(w, want_row_max=True)
    amax = torch.amax(w, dim=1)
    assert torch.equal(rmax, amax)
    ref_scale, _ = O.calculate_params_symmetric(amax.cpu().numpy())
    np.testing.assert_array_equal(scale.cpu().numpy(), ref_scale)
    assert not zp.any()
    rows_checked = np.random.default_rng(0).choice(rows, 48, replace=False).tolist() + [5, 6, 0, rows - 1]
    sub = w[rows_checked].cpu().numpy()
    rq, rs, _ = O.quantize_channel_sym_int8(sub)
    np.testing.assert_array_equal(q[rows_checked].cpu().numpy(), rq)
    # every element: |w - q * scale| <= scale / 2 unless clamped, and clamped values sit at the rails
    deq = q.float() * scale[:, None]
    err = (w - deq).abs()
    inside = (q > -128) & (q < 127)
    assert bool((err[inside] <= scale[:, None].expand_as(w)[inside] * 0.50002).all())
    assert int((q[5] != -128).sum()) == 0 and int(q[6].abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", ["normal", "near_ties", "wide"])
def test_quantizer_every_element_against_true_division(N, dtype, mode):
    """The kernel multiplies by 1/scale and falls back to the true division only near a rounding
    boundary.  Every element of a 32M-element matrix against torch's division on the GPU (IEEE,
    computed in fp32 and rounded to the storage type as on the CPU), including rows built to sit on
    the boundaries: quotients at and next to half-integers and storage-format midpoints.  A sample
    of rows is also checked against the CPU oracle, which pins torch's GPU division itself."""
    rows, cols = 4096, 8192
    g = torch.Generator(device="cuda").manual_seed(11)
    if mode == "normal":
        w = (torch.randn(rows, cols, generator=g, device="cuda") * 0.02).to(dtype)
    elif mode == "wide":     # many clamped elements, tiny and huge magnitudes in one row
        w = (torch.randn(rows, cols, generator=g, device="cuda") *
             torch.logspace(-6, 3, cols, device="cuda")[None, :]).to(dtype)
    else:
        # row maximum 127.5 * s makes the scale s; the other elements are (k + 1/2) * s nudged by a few ulps
        s = (torch.rand(rows, 1, generator=g, device="cuda") + 0.5) * 0.01
        k = torch.randint(-130, 130, (rows, cols), generator=g, device="cuda").float()
        nudge = torch.randint(-3, 4, (rows, cols), generator=g, device="cuda").float() * 2.0 ** -22
        w = ((k + 0.5) * (1.0 + nudge) * s)
        w[:, 0] = (127.5 * s)[:, 0]
        w = w.to(dtype)
    q, scale, zp = N.quantize_rows_sym_i8(w)
    amax = torch.amax(w, dim=1)
    # tensor / tensor: torch's GPU division by a Python scalar multiplies by the reciprocal instead
    ref_scale = torch.maximum(torch.clamp(amax, min=0) / torch.full_like(amax, 127.5),
                              torch.tensor(2.0 ** -23, device="cuda").to(dtype))
    assert torch.equal(scale, ref_scale.float()) or torch.equal(scale.to(dtype), ref_scale)
    ref_q = torch.clamp(torch.round(w / ref_scale[:, None]), -128, 127).to(torch.int8)
    bad = (q != ref_q)
    assert not bool(bad.any()), f"{int(bad.sum())} elements differ from the true division"
    pick = np.random.default_rng(2).choice(rows, 24, replace=False).tolist()
    kind = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}[dtype]
    sub = w[pick].float().cpu().numpy() if dtype != torch.float32 else w[pick].cpu().numpy()
    rq, _, _ = O.quantize_channel_sym_int8(sub, kind=kind)
    np.testing.assert_array_equal(q[pick].cpu().numpy(), rq)


@pytest.mark.parametrize("M,Nn,K", [(4096, 4096, 4096), (8192, 8192, 8192), (256, 14336, 4096), (256, 4096, 14336),
                                     (2048, 28672, 8192), (16, 14336, 4096), (1, 4096, 14336),
                                     # 128-row ring tiles: 128 x 128 (one round of tiles) and 128 x 256 (deep K)
                                     (1000, 4096, 4096), (384, 12288, 8192), (256, 28672, 8192),
                                     # the K = 28672 down projection of the Llama-3-70B MLP, and the token counts in between
                                     # on the Llama-3-8B gate / up projection
                                     (256, 8192, 28672), (2048, 8192, 28672),
                                     (2, 14336, 4096), (4, 14336, 4096), (8, 14336, 4096), (32, 14336, 4096),
                                     (64, 14336, 4096), (128, 14336, 4096)])
def test_gemm_full_size_checksums(N, M, Nn, K):
    a, w = rand_i8((M, K), M + K), rand_i8((Nn, K), Nn + 7 * K)
    c = N.gemm_i8(a, w, out_dtype=torch.int32)
    col = c.sum(dim=0, dtype=torch.int64)
    ref_col = (a.sum(dim=0, dtype=torch.int64).double() @ w.double().T).to(torch.int64)   # < 2^53: exact
    assert torch.equal(col, ref_col)
    row = c.sum(dim=1, dtype=torch.int64)
    ref_row = (a.double() @ w.sum(dim=0, dtype=torch.int64).double()).to(torch.int64)
    assert torch.equal(row, ref_row)
    ms = torch.from_numpy(np.random.default_rng(1).choice(M, min(M, 8), replace=False)).cuda()
    exact = (a[ms].double() @ w.double().T).to(torch.int32)
    assert torch.equal(c[ms], exact)
    # fused epilogue on the same operands, sampled rows against the oracle's epilogue
    g = torch.Generator(device="cuda").manual_seed(3)
    alpha = torch.rand(Nn, generator=g, device="cuda") * 1e-3 + 1e-5
    bias = torch.randn(Nn, generator=g, device="cuda") * 100
    y = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=0.25)
    ref = OC.epilogue_f32(exact.cpu().numpy(), bias.cpu().numpy(), alpha.cpu().numpy(), 0.25)
    np.testing.assert_array_equal(y[ms].cpu().numpy(), ref)
    yb = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=0.25, out_dtype=torch.bfloat16)
    np.testing.assert_array_equal(yb[ms].float().cpu().numpy(), O.round_f32_to_bf16(ref))
    assert torch.equal(yb, y.to(torch.bfloat16))   # every element: the fp32 result rounded once (round-half-even)


# every distinct convolution of ResNet-18 (eleven shapes) at batch 256
@pytest.mark.parametrize("C,H,Co,k,s,p", [(64, 56, 64, 3, 1, 1), (3, 224, 64, 7, 2, 3), (128, 28, 256, 3, 2, 1),
                                          (256, 14, 256, 3, 1, 1), (512, 7, 512, 3, 1, 1), (64, 56, 128, 1, 2, 0),
                                          (64, 56, 128, 3, 2, 1), (128, 28, 128, 3, 1, 1), (256, 14, 512, 3, 2, 1),
                                          (128, 28, 256, 1, 2, 0), (256, 14, 512, 1, 2, 0)])
def test_conv_resnet18_batch256(N, C, H, Co, k, s, p):
    B = 256
    x, w = rand_i8((B, C, H, H), C + H), rand_i8((Co, C, k, k), Co + k)
    out = N.conv2d_i8(x, w, s, p, out_dtype=torch.int32)
    # linearity over the batch: conv(sum_b x_b) == sum_b conv(x_b)  (int64 sums, float64 conv is exact here)
    xs = x.sum(dim=0, keepdim=True, dtype=torch.int64).double()
    ref_sum = torch.nn.functional.conv2d(xs, w.double(), stride=s, padding=p).to(torch.int64)[0]
    assert torch.equal(out.sum(dim=0, dtype=torch.int64), ref_sum)
    for b in (0, 137, 255):
        ref = torch.nn.functional.conv2d(x[b:b + 1].double(), w.double(), stride=s, padding=p).to(torch.int32)
        assert torch.equal(out[b:b + 1], ref)
    # dequantised outputs: fp32 against the oracle's epilogue on a sampled image, bf16 = that fp32 rounded once
    g = torch.Generator(device="cuda").manual_seed(5)
    alpha = torch.rand(Co, generator=g, device="cuda") * 1e-3 + 1e-5
    bias = torch.randn(Co, generator=g, device="cuda") * 100
    y = N.conv2d_i8(x, w, s, p, bias=bias, alpha=alpha, beta=-0.5)
    ref = OC.epilogue_f32(out[137].cpu().numpy()[None], bias.cpu().numpy(), alpha.cpu().numpy(), -0.5)
    np.testing.assert_array_equal(y[137:138].cpu().numpy(), ref)
    del out
    yb = N.conv2d_i8(x, w, s, p, bias=bias, alpha=alpha, beta=-0.5, out_dtype=torch.bfloat16)
    assert torch.equal(yb, y.to(torch.bfloat16))


@pytest.mark.parametrize("mode,dtype,shape,k,s,p", [("max", torch.int8, (256, 64, 112, 112), 3, 2, 1),
                                                     ("max", torch.float32, (256, 64, 112, 112), 3, 2, 1),
                                                     ("avg", torch.int8, (256, 512, 7, 7), 7, 1, 0),
                                                     ("avg", torch.float32, (64, 128, 56, 56), 2, 2, 0)])
def test_pool_resnet_sizes(N, mode, dtype, shape, k, s, p):
    """ResNet-18's pooling layers at batch 256 against torch's pooling of x.float() on the GPU (int8-valued
    inputs: max is exact and every partial sum of the average is an integer below 2^24, so any summation
    order gives the same float) followed by the truncating cast; sampled planes against the oracle."""
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randint(-128, 128, shape, generator=g, dtype=torch.int32, device="cuda").to(dtype)
    y = N.pool2d_to_i8(x, k, s, p, mode)
    f = torch.nn.functional
    ref = f.max_pool2d(x.float(), k, s, p) if mode == "max" else f.avg_pool2d(x.float(), k, s, p)
    assert torch.equal(y, ref.to(torch.int8))
    sub = x[3:5, 7:9].cpu().numpy()
    np.testing.assert_array_equal(y[3:5, 7:9].cpu().numpy(), O.pool2d_to_int8(sub, k, s, p, mode))
