"""Parity of the HIP kernels, called through the C ABI (quantization_amd._native -> libmiq.so),
against the CPU oracle and the golden fixtures produced by the reference.

Bars: bit-exact for int8 weights, int64 zero points, fp32 scales and int32 accumulators;
fp32 dequantised outputs bit-exact against the oracle's epilogue (same operation order) and
within 1e-3 relative of the reference's own float outputs; bf16 outputs within one bf16 ulp.
"""
import numpy as np
import pytest
import torch

from conftest import Golden, kind_of
from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

QW = Golden("quantizer_wo.npz")
FW = Golden("wo_forward.npz")
WA = Golden("wa_chunker.npz")
TORCH_KIND = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


# ------------------------------------------------------------------------------ quantizer
@pytest.mark.parametrize("case", QW.cases())
def test_quantizer_golden(N, case):
    kind = kind_of(case)
    w = dev(QW.get(case, "w"), TORCH_KIND[kind])
    q, s, zp, rmax = N.quantize_rows_sym_i8(w, want_row_max=True)
    ref_q, ref_s, ref_zp = QW.get(case, "q"), QW.get(case, "scales"), QW.get(case, "zp")
    np.testing.assert_array_equal(q.cpu().numpy(), ref_q)
    np.testing.assert_array_equal(s.cpu().numpy(), ref_s.reshape(-1))
    np.testing.assert_array_equal(zp.cpu().numpy(), ref_zp.reshape(-1))
    assert zp.dtype == torch.int64
    np.testing.assert_array_equal(rmax.cpu().numpy(), O.row_signed_max(QW.get(case, "w")))


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (7, 64), (5, 256), (130, 260), (9, 1024), (6, 2052),
                                        (4, 4096), (3, 8192), (2, 16384), (2, 28672), (2, 40000), (2, 40001),
                                        (1000, 36), (2049, 128)])
@pytest.mark.parametrize("kind", ["f32", "f16", "bf16"])
def test_quantizer_vs_oracle_shapes(N, rows, cols, kind):
    """Every dispatch branch (wave-per-row, workgroup-per-row at each register depth, the
    streaming fallbacks, unaligned rows) in all three input precisions."""
    g = torch.Generator().manual_seed(rows * 100003 + cols)
    w = torch.randn(rows, cols, generator=g)
    if rows > 2:
        w[1] = -w[1].abs() - 1e-3
        w[2] = 0
    w = w.to(TORCH_KIND[kind])
    q, s, zp = N.quantize_rows_sym_i8(w.cuda())
    with np.errstate(over="ignore"):
        rq, rs, rzp = O.quantize_channel_sym_int8(w.float().numpy(), kind=kind)
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    np.testing.assert_array_equal(s.cpu().numpy(), rs.reshape(-1))
    assert int(zp.abs().sum()) == 0


def test_quantizer_unaligned_base(N):
    """A view that starts 4 bytes into an allocation takes the element-wise kernel."""
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(1 + 12 * 64, generator=g).cuda()
    w = buf[1:].view(12, 64)
    q, s, _ = N.quantize_rows_sym_i8(w)
    rq, rs, _ = O.quantize_channel_sym_int8(w.cpu().numpy())
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    np.testing.assert_array_equal(s.cpu().numpy(), rs.reshape(-1))


def test_quantizer_ties_and_saturation(N):
    """Half-way quotients (round-half-even), +-inf-sized quotients against the eps floor."""
    w = torch.zeros(4, 256)
    w[0] = torch.arange(-128, 128) * 0.5 + 0.0
    w[0, 0] = 127.5                       # scale = 1 -> quotients are k/2: ties everywhere
    w[1] = -torch.rand(256) * 1e30 - 1.0  # all negative: scale = eps, everything saturates at -128
    w[2, 5] = 3e38                        # scale ~ 2.35e36
    w[2, 6] = -3e38
    w[3] = torch.rand(256) * 1e-12        # max below eps*127.5: floor applies, all round to 0
    q, s, _ = N.quantize_rows_sym_i8(w.cuda())
    rq, rs, _ = O.quantize_channel_sym_int8(w.numpy())
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    np.testing.assert_array_equal(s.cpu().numpy(), rs.reshape(-1))
    assert (q[1] == -128).all() and float(s[1]) == 2.0 ** -23


def test_quantizer_empty_and_errors(N):
    q, s, zp = N.quantize_rows_sym_i8(torch.empty(0, 16).cuda())
    assert q.shape == (0, 16) and s.numel() == 0
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        N.quantize_rows_sym_i8(torch.zeros(2, 2))
    with pytest.raises(RuntimeError, match="contiguous"):
        N.quantize_rows_sym_i8(torch.zeros(4, 4).cuda().t())
    with pytest.raises(RuntimeError, match="zero length"):
        N.quantize_rows_sym_i8(torch.empty(3, 0).cuda())


def test_row_max_given_and_dequant(N):
    g = torch.Generator().manual_seed(11)
    w = torch.randn(37, 301, generator=g)
    wd = w.cuda()
    np.testing.assert_array_equal(N.row_max(wd).cpu().numpy(), w.numpy().max(axis=1))
    q, s, zp = N.quantize_rows_sym_i8(wd)
    q2 = N.quantize_rows_given_i8(wd, s)
    assert torch.equal(q, q2)
    # per-tensor affine form: q = clamp(rint(w / s + zp))
    sc, z = O.calculate_params_affine_tensor(w.min().item(), w.max().item())
    q3 = N.quantize_rows_given_i8(wd, torch.from_numpy(sc), torch.from_numpy(z))
    ref = np.clip(np.rint(w.numpy() / sc[0] + np.float32(z[0])), -128, 127).astype(np.int8)
    np.testing.assert_array_equal(q3.cpu().numpy(), ref)
    d = N.dequantize_rows_i8(q, s, zp)
    np.testing.assert_array_equal(d.cpu().numpy(), O.dequantize(q.cpu().numpy(), s.cpu().numpy()[:, None], 0))
    y = torch.randn(3, 37, 5, generator=g)
    dz = N.dequantize_channels_f32(y.cuda(), s, None)
    np.testing.assert_array_equal(dz.cpu().numpy(), (s.cpu().numpy()[None, :, None] * y.numpy()))


# ------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [
    (1, 64, 64), (3, 4, 16), (16, 64, 128), (17, 130, 48), (33, 100, 272), (64, 256, 512), (65, 72, 1040),
    (128, 128, 128), (129, 257, 400), (200, 520, 384), (256, 256, 256), (300, 4096, 256), (512, 768, 1024),
    (1024, 4096, 2048),  # the reference's own test shape (linear_int8/test.py:5-6)
    (37, 130, 300), (3, 4, 6), (5, 48, 64),  # K not a multiple of 16 -> element-wise kernel
    # narrow weights with many rows: kept off the 256-column tiles (128 x 64 tiles up to 64 channels)
    (60000, 64, 64), (50000, 100, 48), (200, 40, 4096), (50000, 200, 32),
    # thin problems: split along K into int32 slabs + reduce kernel
    (1, 256, 4096), (16, 1024, 14336), (40, 520, 8192), (100, 300, 2064), (256, 512, 8192), (200, 128, 16384),
]


def rand_i8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-128, 128, shape, generator=g, dtype=torch.int32).to(torch.int8)


@pytest.mark.parametrize("M,N_,K", GEMM_SHAPES)
def test_gemm_i32_bit_exact(N, M, N_, K):
    a, w = rand_i8((M, K), M * 7 + K), rand_i8((N_, K), N_ * 13 + K)
    ref = OC.linear_i8_nt(a.numpy(), w.numpy()) if M * N_ * K <= 2 ** 28 else \
        O.linear_w8a8(a.numpy(), np.ascontiguousarray(w.numpy().T))
    out = N.gemm_i8(a.cuda(), w.cuda(), out_dtype=torch.int32)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_gemm_extreme_values(N):
    """All -128 x -128 at K = 8192: the largest accumulator the path can see."""
    M, N_, K = 32, 64, 8192
    a = torch.full((M, K), -128, dtype=torch.int8)
    w = torch.full((N_, K), -128, dtype=torch.int8)
    out = N.gemm_i8(a.cuda(), w.cuda(), out_dtype=torch.int32)
    assert (out == 128 * 128 * K).all()
    w[:, ::2] = 127
    out = N.gemm_i8(a.cuda(), w.cuda(), out_dtype=torch.int32)
    assert (out == (128 * 128 - 128 * 127) * (K // 2)).all()


def test_gemm_lane_map_asymmetric(N):
    """A = identity-like, W asymmetric: catches a transposed or permuted output tile."""
    K = 256
    a = torch.zeros(K, K, dtype=torch.int8)
    a[torch.arange(K), torch.arange(K)] = 1
    w = ((torch.arange(300)[:, None] * 7 + torch.arange(K)[None, :] * 3) % 251 - 125).to(torch.int8)
    out = N.gemm_i8(a.cuda(), w.cuda(), out_dtype=torch.int32)
    np.testing.assert_array_equal(out.cpu().numpy(), w.numpy().T.astype(np.int32))


@pytest.mark.parametrize("M,N_,K", [(5, 64, 64), (37, 130, 304), (256, 512, 1024), (300, 1027, 512), (8, 4, 6)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_gemm_epilogue_f32_bit_exact_vs_oracle(N, M, N_, K, with_bias):
    a, w = rand_i8((M, K), 1 + M), rand_i8((N_, K), 2 + N_)
    g = torch.Generator().manual_seed(3)
    alpha = torch.rand(N_, generator=g) * 0.01 + 1e-4
    bias = torch.randn(N_, generator=g) * 50 if with_bias else None
    beta = 0.375
    acc = O.linear_w8a8(a.numpy(), np.ascontiguousarray(w.numpy().T))
    ref = OC.epilogue_f32(acc, None if bias is None else bias.numpy(), alpha.numpy(), beta)
    out = N.gemm_i8(a.cuda(), w.cuda(), bias=None if bias is None else bias.cuda(), alpha=alpha.cuda(), beta=beta)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # bf16 output: the same value rounded once to bf16
    outb = N.gemm_i8(a.cuda(), w.cuda(), bias=None if bias is None else bias.cuda(), alpha=alpha.cuda(), beta=beta,
                     out_dtype=torch.bfloat16)
    np.testing.assert_array_equal(outb.float().cpu().numpy(), O.round_f32_to_bf16(ref))


@pytest.mark.parametrize("M,N_,K", [(3584, 3600, 64), (3500, 3604, 80), (3584, 3592, 48)])
def test_gemm_big_tile_ragged_edges(N, M, N_, K):
    """Shapes that take the 256 x 256 kernel with partial last tiles; N % 8 == 0 (paired 16-byte bf16
    stores, also next to a ragged last 32-channel group) and N % 8 == 4 (8-byte bf16 stores)."""
    a, w = rand_i8((M, K), 11 + M), rand_i8((N_, K), 12 + N_)
    g = torch.Generator().manual_seed(13)
    alpha, bias = torch.rand(N_, generator=g) * 0.01 + 1e-4, torch.randn(N_, generator=g) * 50
    acc = OC.linear_i8_nt(a.numpy(), w.numpy())
    np.testing.assert_array_equal(N.gemm_i8(a.cuda(), w.cuda(), out_dtype=torch.int32).cpu().numpy(), acc)
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), 0.5)
    out = N.gemm_i8(a.cuda(), w.cuda(), bias=bias.cuda(), alpha=alpha.cuda(), beta=0.5)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    outb = N.gemm_i8(a.cuda(), w.cuda(), bias=bias.cuda(), alpha=alpha.cuda(), beta=0.5, out_dtype=torch.bfloat16)
    np.testing.assert_array_equal(outb.float().cpu().numpy(), O.round_f32_to_bf16(ref))


def test_gemm_kn_layout_is_linear_w8a8(N):
    """[K,N] weights, raw int32 out: the linear_w8a8 contract."""
    a, b = rand_i8((70, 96), 5), rand_i8((96, 200), 6)
    out = N.gemm_i8(a.cuda(), b.cuda(), w_layout=N.MIQ_W_KN, out_dtype=torch.int32)
    np.testing.assert_array_equal(out.cpu().numpy(), OC.linear_w8a8(a.numpy(), b.numpy()))
    t = N.transpose_i8(b.cuda())
    np.testing.assert_array_equal(t.cpu().numpy(), b.numpy().T)
    for shape in [(67, 129), (100, 130), (2048, 4096), (72, 64), (4, 4), (1, 7), (260, 1)]:
        m = rand_i8(shape, 8)
        np.testing.assert_array_equal(N.transpose_i8(m.cuda()).cpu().numpy(), m.numpy().T, err_msg=str(shape))


def test_gemm_errors(N):
    a, w = rand_i8((4, 16), 1).cuda(), rand_i8((8, 32), 2).cuda()
    with pytest.raises(RuntimeError, match="shape mismatch"):
        N.gemm_i8(a, w)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        N.gemm_i8(a.cpu(), w)
    with pytest.raises(RuntimeError, match="contiguous"):
        N.gemm_i8(a, rand_i8((16, 8), 3).cuda().t())
    with pytest.raises(RuntimeError, match="alpha"):
        N.gemm_i8(a, rand_i8((8, 16), 3).cuda())
    with pytest.raises(RuntimeError, match="int8"):
        N.gemm_i8(a.float(), w)
    out = N.gemm_i8(torch.empty(0, 16, dtype=torch.int8).cuda(), rand_i8((8, 16), 3).cuda(), out_dtype=torch.int32)
    assert out.shape == (0, 8)


@pytest.mark.parametrize("case", [c for c in FW.cases() if c.startswith("lin")])
def test_wo_linear_forward_golden(N, case):
    """Quantize + int8 GEMM + fused weight-only epilogue against the reference's outputs."""
    w, x, bias = FW.get(case, "w"), FW.get(case, "x"), FW.get(case, "bias")
    q, s, _ = N.quantize_rows_sym_i8(dev(w))
    np.testing.assert_array_equal(q.cpu().numpy(), FW.get(case, "q"))
    y = N.gemm_i8(dev(x), q, bias=None if bias is None else dev(bias), alpha=s, beta=0.0)
    ref = FW.get(case, "y")
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    np.testing.assert_array_equal(y.cpu().numpy(), O.wo_linear_forward(x, q.cpu().numpy(), s.cpu().numpy(), bias))


# ------------------------------------------------------------------------------------ conv
CONV_SHAPES = [
    # B, Cin, H, W, Cout, k, stride, pad
    (2, 3, 8, 8, 5, 3, 1, 1), (2, 8, 9, 11, 16, 3, 2, 1), (1, 4, 5, 5, 6, 1, 1, 0), (1, 3, 12, 12, 2, 7, 2, 3),
    (2, 3, 32, 32, 4, 3, 1, 0),      # the reference test's channel counts (conv2d_int8/test.py:17-25), smaller image
    (3, 16, 14, 14, 32, 3, 1, 1), (2, 64, 14, 14, 64, 3, 1, 1), (2, 64, 15, 15, 128, 3, 2, 1),
    (2, 64, 14, 14, 128, 1, 2, 0), (4, 128, 7, 7, 256, 3, 1, 1), (1, 5, 6, 7, 3, 2, 1, 0), (1, 1, 4, 4, 1, 3, 1, 2),
    (2, 3, 30, 30, 64, 7, 2, 3),
    # W % 4 == 0: the dword-transposing re-layout kernels (channel counts that are / are not multiples of 16,
    # more than one 64-channel block, more than one 64-pixel tile per row)
    (2, 8, 12, 12, 16, 3, 1, 1), (1, 70, 8, 72, 32, 3, 1, 1), (2, 20, 16, 16, 24, 3, 2, 1),
    (1, 130, 8, 8, 16, 1, 1, 0), (2, 4, 8, 132, 8, 3, 1, 2), (1, 2, 16, 16, 4, 5, 1, 2),
    # 3 x 3 / stride 1, whole 64-channel groups, Ho*Wo % 4 == 0 and images that fill 448-pixel tiles: the
    # LDS-resident-patch kernel (a partial last tile, ragged output channels, two and three channel groups,
    # pad 0 / 1 / 2, an odd width); the last two stay on the im2col-order kernel
    (3, 128, 28, 28, 70, 3, 1, 1), (5, 64, 20, 20, 64, 3, 1, 1), (5, 64, 22, 22, 32, 3, 1, 0),
    (2, 192, 42, 42, 70, 3, 1, 1), (5, 64, 18, 18, 64, 3, 1, 2), (2, 64, 56, 56, 8, 3, 1, 1),
    (5, 64, 20, 21, 16, 3, 1, 1), (6, 64, 16, 16, 64, 3, 1, 1), (40, 64, 8, 8, 64, 3, 1, 1),
    # 1 x 1 filters with a stride and no padding: the re-layout keeps only the pixels the filter visits (stride 2
    # from aligned 8-byte runs; odd sizes, ragged channels, stride 3 and a 3-channel image by single bytes)
    (2, 64, 16, 16, 128, 1, 2, 0), (3, 70, 24, 40, 20, 1, 2, 0), (2, 16, 15, 17, 8, 1, 2, 0),
    (2, 32, 13, 13, 8, 1, 3, 0), (1, 3, 9, 9, 4, 1, 2, 0), (2, 8, 12, 20, 8, 1, 2, 0), (2, 128, 56, 56, 64, 1, 2, 0),
    # the patch kernel: staging the NCHW activation itself (3 x 3 / 1, 64 input channels, W % 4 == 0; one and three channel
    # tiles, a ragged last tile) and from the padded copy (two channel groups; a patch that only fits that form)
    (3, 64, 56, 56, 64, 3, 1, 1), (4, 128, 28, 28, 128, 3, 1, 1), (5, 64, 30, 28, 192, 3, 1, 1), (4, 64, 40, 40, 64, 3, 1, 0),
    # the patch kernel with several small images per tile (a last tile with one image; two channel groups and tiles)
    (11, 64, 14, 14, 64, 3, 1, 1), (12, 128, 14, 14, 192, 3, 1, 1),
    # 3 x 3 / stride 1 on whole channel groups with a pixel count that is not a multiple of 4: the patch kernel storing pixel
    # by pixel (eight 7 x 7 images a tile, rows of Wp slots; four channel groups; one 21 x 21 image a tile; tiles of a
    # 27 x 33 and a 45 x 45 image whose last pixel quads straddle the image's end)
    (40, 64, 7, 7, 70, 3, 1, 1), (37, 128, 7, 7, 64, 3, 1, 1), (40, 64, 9, 9, 64, 3, 1, 0), (4, 64, 27, 33, 16, 3, 1, 1),
    (37, 256, 7, 7, 130, 3, 1, 1), (10, 64, 21, 21, 64, 3, 1, 1), (3, 64, 45, 45, 32, 3, 1, 1),
    # ... and rows of whole dwords whose padding leaves such a pixel count (not for the kernel that stages the NCHW activation)
    (3, 64, 37, 20, 76, 3, 1, 2), (3, 64, 49, 28, 70, 3, 1, 0),
    # stems (Cin <= 4, 5-8 filter columns, stride 2): ragged tiles, ragged channel tiles, 8 x 8 filters
    (2, 3, 32, 32, 64, 7, 2, 3), (3, 3, 45, 40, 80, 7, 2, 3), (2, 2, 33, 48, 130, 5, 2, 2), (1, 4, 64, 64, 64, 8, 2, 3),
    (2, 3, 30, 224, 64, 7, 2, 3),    # two 64-pixel blocks a row, the second ragged
]


@pytest.mark.parametrize("B,C,H,W,Co,k,s,p", CONV_SHAPES)
def test_conv_i32_bit_exact(N, B, C, H, W, Co, k, s, p):
    x, w = rand_i8((B, C, H, W), B + C * 3 + H), rand_i8((Co, C, k, k), Co + k)
    ref = OC.conv2d_i8(x.numpy(), w.numpy(), s, p)
    out = N.conv2d_i8(x.cuda(), w.cuda(), s, p, out_dtype=torch.int32)
    assert out.shape == ref.shape
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # the same layer with its filter packed once (miq_conv2d_i8_pack_weight + miq_conv2d_i8_prepacked), against the oracle
    wd = w.cuda()
    packed = N.conv2d_i8_pack_weight(wd)
    out2 = N.conv2d_i8(x.cuda(), wd, s, p, out_dtype=torch.int32, packed=packed)
    np.testing.assert_array_equal(out2.cpu().numpy(), ref)


@pytest.mark.parametrize("B,C,H,W,Co,pad,shift", [(3, 64, 56, 56, 64, 1, 4), (8, 64, 30, 28, 64, 0, 4), (8, 64, 30, 28, 70, 1, 12),
                                                  (3, 64, 56, 56, 64, 1, 1)])
def test_conv_3x3_on_an_activation_that_is_not_16_byte_aligned(N, B, C, H, W, Co, pad, shift):
    """3 x 3 / stride 1 layers with 64 input channels stage the NCHW activation themselves in 16-byte chunks when its base
    is at least dword-aligned (shift 4, 12: the chunks then start inside a channel's first row), and go through the padded
    copy otherwise (shift 1); with and without padding."""
    flat = torch.zeros(B * C * H * W + 64, dtype=torch.int8, device="cuda")
    x = flat[shift:shift + B * C * H * W].view(B, C, H, W)
    x.copy_(rand_i8((B, C, H, W), B + H + shift).cuda())
    assert x.data_ptr() % 16 == shift % 16
    w = rand_i8((Co, C, 3, 3), Co + shift).cuda()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=1, padding=pad).to(torch.int32)
    out = N.conv2d_i8(x, w, 1, pad, out_dtype=torch.int32)
    assert torch.equal(out, ref)


def test_conv_3x3_random_shapes_vs_torch(N):
    """Seeded random 3 x 3 / stride 1 layers with whole 64-channel groups (most take the LDS-resident-patch
    kernel, the small ones the im2col-order kernel) against F.conv2d in float64, exact on these integers."""
    rng = np.random.default_rng(2024)
    for _ in range(16):
        C = 64 * int(rng.integers(1, 4))
        H, W = int(rng.integers(6, 61)), int(rng.integers(6, 61))
        pad = int(rng.integers(0, 3))
        Co = int(rng.integers(1, 140))
        B = int(rng.integers(1, 4)) + 2200 // ((H + 2 * pad - 2) * (W + 2 * pad - 2))
        x, w = rand_i8((B, C, H, W), B + H), rand_i8((Co, C, 3, 3), Co + W)
        ref = torch.nn.functional.conv2d(x.cuda().double(), w.cuda().double(), stride=1, padding=pad).to(torch.int32)
        out = N.conv2d_i8(x.cuda(), w.cuda(), 1, pad, out_dtype=torch.int32)
        assert torch.equal(out, ref), (B, C, H, W, Co, pad)


def test_conv_1x1_strided_random_shapes_vs_torch(N):
    """Seeded random 1 x 1 layers with stride 2..4 and no padding (the subsampling re-layout: aligned 8-byte runs,
    single bytes, ragged channels and pixel tiles) and with padding (the whole image re-laid) against F.conv2d in
    float64, exact on these integers."""
    rng = np.random.default_rng(77)
    for i in range(24):
        C, Co = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        H, W = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        if i % 3 == 0:
            H, W = 8 * int(rng.integers(1, 7)), 8 * int(rng.integers(1, 7))
        s, pad, B = int(rng.integers(2, 5)), int(i % 8 == 7), int(rng.integers(1, 5))
        x, w = rand_i8((B, C, H, W), 3 * i + 1), rand_i8((Co, C, 1, 1), 3 * i + 2)
        ref = torch.nn.functional.conv2d(x.cuda().double(), w.cuda().double(), stride=s, padding=pad).to(torch.int32)
        out = N.conv2d_i8(x.cuda(), w.cuda(), s, pad, out_dtype=torch.int32)
        assert out.shape == ref.shape and torch.equal(out, ref), (B, C, H, W, Co, s, pad)


def test_conv_resnet_layer_vs_torch(N):
    """A ResNet-18 layer1 shape at a small batch, checked against F.conv2d in float64."""
    x, w = rand_i8((4, 64, 56, 56), 1), rand_i8((64, 64, 3, 3), 2)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=1, padding=1).to(torch.int32)
    out = N.conv2d_i8(x.cuda(), w.cuda(), 1, 1, out_dtype=torch.int32)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("B,C,H,W,Co,k,s,p", [(2, 3, 8, 8, 5, 3, 1, 1), (2, 16, 9, 9, 70, 3, 2, 1), (2, 8, 7, 7, 12, 3, 1, 1),
                                              # Ho*Wo % 8 == 0: paired 16-byte bf16 stores (ragged channels / pixels)
                                              (3, 16, 16, 16, 70, 3, 1, 1), (1, 8, 8, 12, 16, 3, 1, 1),
                                              (5, 32, 12, 12, 200, 3, 1, 1), (2, 4, 18, 18, 9, 3, 2, 0),
                                              # the LDS-resident-patch kernel
                                              (5, 64, 20, 20, 70, 3, 1, 1), (3, 128, 28, 28, 24, 3, 1, 1),
                                              (5, 64, 20, 21, 16, 3, 1, 1), (40, 64, 7, 7, 70, 3, 1, 1),
                                              (4, 64, 27, 33, 16, 3, 1, 1), (37, 128, 7, 7, 64, 3, 1, 1), (10, 64, 21, 21, 64, 3, 1, 1),
                                              # strided 1 x 1: the subsampling re-layout
                                              (2, 64, 16, 16, 70, 1, 2, 0), (2, 24, 9, 11, 12, 1, 2, 0),
                                              # stems: the row-block kernel (fp32) beside the im2col-order one (bf16)
                                              (2, 3, 45, 40, 80, 7, 2, 3), (1, 3, 18, 224, 64, 7, 2, 3)])
def test_conv_epilogue_bit_exact_vs_oracle(N, B, C, H, W, Co, k, s, p):
    x, w = rand_i8((B, C, H, W), 9), rand_i8((Co, C, k, k), 10)
    g = torch.Generator().manual_seed(4)
    alpha, bias = torch.rand(Co, generator=g) * 0.02, torch.randn(Co, generator=g) * 30
    acc = OC.conv2d_i8(x.numpy(), w.numpy(), s, p)
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), -1.25)
    out = N.conv2d_i8(x.cuda(), w.cuda(), s, p, bias=bias.cuda(), alpha=alpha.cuda(), beta=-1.25)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    outp = N.conv2d_i8(x.cuda(), w.cuda(), s, p, bias=bias.cuda(), alpha=alpha.cuda(), beta=-1.25,
                       packed=N.conv2d_i8_pack_weight(w.cuda()))
    np.testing.assert_array_equal(outp.cpu().numpy(), ref)
    outb = N.conv2d_i8(x.cuda(), w.cuda(), s, p, bias=bias.cuda(), alpha=alpha.cuda(), beta=-1.25,
                       out_dtype=torch.bfloat16)
    np.testing.assert_array_equal(outb.float().cpu().numpy(), O.round_f32_to_bf16(ref))


@pytest.mark.parametrize("case", [c for c in FW.cases() if c.startswith("conv")])
def test_wo_conv_forward_golden(N, case):
    w, x, bias = FW.get(case, "w"), FW.get(case, "x"), FW.get(case, "bias")
    stride, pad = (int(v) for v in FW.get(case, "stride_pad"))
    q, s, _ = N.quantize_rows_sym_i8(dev(w))
    np.testing.assert_array_equal(q.cpu().numpy(), FW.get(case, "q"))
    y = N.conv2d_i8(dev(x), q, stride, pad, bias=None if bias is None else dev(bias), alpha=s, beta=0.0)
    ref = FW.get(case, "y")
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


def test_wa_epilogue_golden(N):
    """Weight+activation epilogue (acc + bias') * scalers + out_zp on the reference's folded params."""
    for name, conv in (("conv1", True), ("fc2", False)):
        x, wq = dev(WA.get(name, "x")), dev(WA.get(name, "weight"))
        bias, scalers = dev(WA.get(name, "bias")), dev(WA.get(name, "scalers"))
        ozp = float(WA.get(name, "output_zero_point")[0])
        if conv:
            y = N.conv2d_i8(x, wq, 1, 1, bias=bias, alpha=scalers, beta=ozp)
        else:
            y = N.gemm_i8(x, wq, bias=bias, alpha=scalers, beta=ozp)
        ref = WA.get(name, "y")
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


# ---- int8 pooling (QuantizedMaxPool2d / QuantizedAvgPool2d) ------------------------------------
def _pool_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "pools.npz"))


@pytest.mark.gpu
def test_pool_matches_reference_fixture():
    from oracle.gen_golden import POOL_CASES
    from quantization_amd import _native as N
    fix = _pool_fixture()
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "i8": torch.int8}
    for name, mode, kind, shape, k, s, p, _ in POOL_CASES:
        x = torch.from_numpy(fix[name + "/x"]).to(tdt[kind]).cuda()
        y = N.pool2d_to_i8(x, k, s, p, mode)
        np.testing.assert_array_equal(y.cpu().numpy(), fix[name + "/y"], err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["max", "avg"])
@pytest.mark.parametrize("shape,k,s,p", [((3, 5, 17, 19), 3, 2, 1), ((2, 4, 16, 16), 2, 2, 0), ((1, 1, 5, 3), (2, 3), (1, 1), (1, 1)),
                                          ((2, 3, 9, 33), (3, 2), (2, 3), (0, 1)), ((1, 7, 1, 1), 1, 1, 0), ((0, 4, 8, 8), 2, 2, 0),
                                          ((2, 3, 32, 48), 3, 2, 1), ((2, 3, 10, 64), 3, 1, 1), ((1, 5, 64, 32), 2, 2, 0), ((1, 2, 9, 40), 3, 2, 1)])
@pytest.mark.parametrize("kind", ["f32", "i8", "bf16"])
def test_pool_against_oracle(mode, shape, k, s, p, kind):
    from quantization_amd import _native as N
    g = torch.Generator().manual_seed(5)
    if kind == "i8":
        x = torch.randint(-128, 128, shape, generator=g, dtype=torch.int32).to(torch.int8)
    else:
        x = (torch.randn(shape, generator=g) * 90).to(torch.float32 if kind == "f32" else torch.bfloat16)
    y = N.pool2d_to_i8(x.cuda(), k, s, p, mode)
    ref = O.pool2d_to_int8(x.float().numpy() if kind != "i8" else x.numpy(), k, s, p, mode)
    np.testing.assert_array_equal(y.cpu().numpy(), ref)


@pytest.mark.gpu
def test_pool_layers_drop_in():
    """The layer classes the weight+activation Chunker installs, on CPU and GPU tensors."""
    from quantization_amd.quant.Weight_Activation.Quantizer import QuantizedAvgPool2d, QuantizedMaxPool2d
    x = torch.randn(2, 3, 8, 8) * 50
    for cls, mode in ((QuantizedMaxPool2d, "max"), (QuantizedAvgPool2d, "avg")):
        m = cls(kernel_size=2, stride=2, padding=0)
        ref = O.pool2d_to_int8(x.numpy(), 2, 2, 0, mode)
        y = m(x)
        assert y.dtype == torch.int8 and y.device.type == "cpu"
        np.testing.assert_array_equal(y.numpy(), ref)
        np.testing.assert_array_equal(m(x.cuda()).cpu().numpy(), ref)
    with pytest.raises(NotImplementedError):
        QuantizedMaxPool2d(kernel_size=2, ceil_mode=True)(x)


@pytest.mark.gpu
def test_pool_rejects_bad_arguments():
    from quantization_amd import _native as N
    x = torch.zeros(1, 1, 4, 4, device="cuda")
    with pytest.raises(RuntimeError, match="pad must be at most half"):
        N.pool2d_to_i8(x, 2, 2, 2, "max")
    with pytest.raises(RuntimeError, match="window larger"):
        N.pool2d_to_i8(x, 7, 1, 0, "max")
    with pytest.raises(RuntimeError, match="must be float32"):
        N.pool2d_to_i8(x.double(), 2, 2, 0, "max")


# ------------------------------------------------------------------------------ GEMM cut along M
@pytest.mark.parametrize("M,Nn,K", [(2304, 8192, 272), (2100, 8192, 64), (4352, 8192, 128), (1280, 14336, 144),
                                    (2304, 8190, 96),
                                    # 3.5 rounds of tiles (stay whole: a half-empty round costs half a round), 7 x 42 tiles
                                    (2048, 28672, 80), (2048, 28500, 64), (1700, 10752, 48)])
@pytest.mark.parametrize("od", [torch.int32, torch.float32, torch.bfloat16])
def test_gemm_whole_rounds_plus_tail(N, M, Nn, K, od):
    """Tile counts a little over a multiple of the 256 CUs run as whole rounds of 256 x 256 tiles plus a second
    launch for the remaining rows: every element must equal the exact product (int32) and the oracle's epilogue
    of it (fp32 bit for bit; bf16 = that value rounded once)."""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    w = torch.randint(-128, 128, (Nn, K), generator=g, dtype=torch.int32, device="cuda").to(torch.int8)
    alpha = torch.rand(Nn, generator=g, device="cuda") * 0.01 + 1e-4
    bias = torch.randn(Nn, generator=g, device="cuda")
    kw = {} if od == torch.int32 else {"alpha": alpha, "bias": bias}
    y = N.gemm_i8(a, w, out_dtype=od, **kw)
    exact = (a.double() @ w.double().t()).to(torch.int32)      # |sum| < 2^53: exact
    if od == torch.int32:
        assert torch.equal(y, exact)
    else:
        ref = OC.epilogue_f32(exact.cpu().numpy(), bias.cpu().numpy(), alpha.cpu().numpy(), 0.0)
        if od == torch.bfloat16:
            ref = O.round_f32_to_bf16(ref)
        np.testing.assert_array_equal(y.float().cpu().numpy(), ref)


def test_developer_switches_do_not_reach_the_product(N, monkeypatch):
    """The environment variables of the developer library (a timing-only GEMM build that skips its DMA, a forced
    tile, the wide tile) leave the product's results bit-equal."""
    a, w = rand_i8((512, 512), 5).cuda(), rand_i8((8192, 512), 6).cuda()   # 64 tiles of 256 x 256 ... the big-tile kernel
    a2, w2 = rand_i8((2048, 256), 7).cuda(), rand_i8((8192, 256), 8).cuda()
    ref, ref2 = N.gemm_i8(a, w, out_dtype=torch.int32), N.gemm_i8(a2, w2, out_dtype=torch.int32)
    exact = (a2.double() @ w2.double().T).to(torch.int32)
    assert torch.equal(ref2, exact)
    for name, val in (("MIQ_GEMM_ABL", "1"), ("MIQ_GEMM_ABL", "3"), ("MIQ_GEMM_VARIANT", "0"), ("MIQ_GEMM_TILE", "g"),
                      ("MIQ_GEMM_WIDE", "2"), ("MIQ_GEMM_PLAN256", "0")):
        monkeypatch.setenv(name, val)
        assert torch.equal(N.gemm_i8(a, w, out_dtype=torch.int32), ref), name
        assert torch.equal(N.gemm_i8(a2, w2, out_dtype=torch.int32), ref2), name
        monkeypatch.delenv(name)


@pytest.mark.parametrize("sched", ["0", "1", "2", "3", "4"])
def test_wide_tile_kernel_of_the_developer_library(N, sched, monkeypatch):
    """The 256 x 384 kernel (kept in the developer library as the measured alternative to the 256 x 256 tile):
    a column block of whole wide tiles plus the remaining channels on the ordinary kernels, every DMA
    schedule, bit-equal to the exact product; twice on fresh data (a staging race would show as a wrong tile)."""
    from quantization_amd import build as miq_build
    monkeypatch.setattr(N, "_lib", N.load(miq_build.DEV_LIB))
    monkeypatch.setenv("MIQ_GEMM_WIDE_SCHED", sched)
    for rep, (M, Nn, K, cols) in enumerate([(512, 1152, 384, 3), (256, 1000, 128, 2), (768, 2000, 1024, 4)]):
        monkeypatch.setenv("MIQ_GEMM_WIDE", str(cols))
        a, w = rand_i8((M, K), 11 * rep + M).cuda(), rand_i8((Nn, K), 13 * rep + K).cuda()
        exact = (a.double() @ w.double().T).to(torch.int32)
        assert torch.equal(N.gemm_i8(a, w, out_dtype=torch.int32), exact)
        alpha = torch.rand(Nn, device="cuda") * 1e-3 + 1e-5
        bias = torch.randn(Nn, device="cuda")
        ref = OC.epilogue_f32(exact.cpu().numpy(), bias.cpu().numpy(), alpha.cpu().numpy(), 0.5)
        np.testing.assert_array_equal(N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=0.5).cpu().numpy(), ref)
        yb = N.gemm_i8(a, w, bias=bias, alpha=alpha, beta=0.5, out_dtype=torch.bfloat16)
        np.testing.assert_array_equal(yb.float().cpu().numpy(), O.round_f32_to_bf16(ref))
