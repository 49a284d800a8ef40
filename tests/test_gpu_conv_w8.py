"""Weight-only Conv2d on floating-point activations (miq_conv2d_w8, through the C ABI and the drop-in layer) against
an fp64 convolution and the reference's own outputs.

Bar as for the Linear (tests/test_gpu_w8f.py): the reference convolves in the activation's precision
(`F.conv2d(x, weight.to(x.dtype), ...)`, quant/Weight_Only/Quantizer.py:72-91) and outputs must agree within 1e-3
relative; the kernel is held to 5e-5 of the sum of magnitudes per element.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import Golden

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
FW = Golden("wo_forward.npz")
Q_PARAMS = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def check(N, B, C, H, W, Co, k, s, p, xdt, seed, bias=True, out_dtype=torch.float32, beta=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g).to(DT[xdt])
    w = torch.randint(-128, 128, (Co, C, k, k), generator=g, dtype=torch.int32).to(torch.int8)
    alpha = torch.rand(Co, generator=g) * 0.02 + 1e-4
    b = torch.randn(Co, generator=g) * 5 if bias else None
    y = N.conv2d_w8(x.cuda(), w.cuda(), s, p, bias=None if b is None else b.cuda(), alpha=alpha.cuda(), beta=beta,
                    out_dtype=out_dtype)
    # the same call with the filter packed once (miq_conv2d_w8_pack_weight + miq_conv2d_w8_prepacked): the same kernels
    # on the same operands, so the results are equal bit for bit (and y is checked against the fp64 convolution below)
    yp = N.conv2d_w8(x.cuda(), w.cuda(), s, p, bias=None if b is None else b.cuda(), alpha=alpha.cuda(), beta=beta,
                     out_dtype=out_dtype, packed=N.conv2d_w8_pack_weight(w.cuda(), DT[xdt]))
    torch.cuda.synchronize()
    assert torch.equal(y, yp), "prepacked filter: result differs from the call that packs its own"
    xd, wd = x.double(), w.double()
    bd = torch.zeros(Co, dtype=torch.float64) if b is None else b.double()
    a4 = alpha.double().view(1, -1, 1, 1)
    ref = (F.conv2d(xd, wd, stride=s, padding=p) + bd.view(1, -1, 1, 1)) * a4 + beta
    mag = (F.conv2d(xd.abs(), wd.abs(), stride=s, padding=p) + bd.abs().view(1, -1, 1, 1)) * a4 + abs(beta)
    assert tuple(y.shape) == tuple(ref.shape) and y.dtype == out_dtype
    err = (y.cpu().double() - ref).abs()
    tol = 5e-5 * mag + 1e-30
    if out_dtype == torch.bfloat16:
        tol = tol + ref.abs() * 2.0 ** -8
    worst = float((err / tol).max())
    assert worst <= 1.0, f"worst error / tolerance = {worst:.3f}"


CASES = [
    (2, 3, 8, 8, 5, 3, 1, 1), (2, 8, 9, 11, 16, 3, 2, 1), (1, 4, 5, 5, 6, 1, 1, 0), (3, 3, 32, 32, 64, 3, 1, 1),
    (2, 64, 14, 14, 128, 3, 1, 1), (2, 3, 30, 30, 64, 7, 2, 3), (4, 16, 12, 12, 130, 1, 2, 0), (1, 70, 9, 9, 33, 3, 1, 0),
    (2, 32, 16, 16, 64, 3, 2, 1), (5, 24, 7, 7, 200, 5, 1, 2), (1, 1, 4, 4, 1, 2, 1, 0), (2, 128, 7, 7, 256, 3, 1, 1),
    # strided 1 x 1 filters: the re-layout keeps only the pixels the filter visits (even and odd sizes, stride 3,
    # few channels, ragged channels)
    (2, 64, 16, 16, 128, 1, 2, 0), (3, 70, 15, 17, 20, 1, 2, 0), (2, 32, 13, 13, 8, 1, 3, 0), (1, 3, 9, 9, 4, 1, 2, 0),
    (2, 64, 56, 56, 128, 1, 2, 0),
    # channel planes that are whole 8-byte but not 16-byte vectors (fp32: H * W even, not a multiple of 4)
    (2, 24, 6, 7, 40, 3, 1, 1), (1, 40, 10, 9, 16, 3, 1, 1),
]


@pytest.mark.parametrize("xdt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("B,C,H,W,Co,k,s,p", CASES)
def test_conv_w8_shapes(N, B, C, H, W, Co, k, s, p, xdt):
    check(N, B, C, H, W, Co, k, s, p, xdt, seed=B * 100 + C + Co)


@pytest.mark.parametrize("xdt", ["f32", "bf16"])
def test_conv_w8_bf16_out_no_bias(N, xdt):
    check(N, 2, 16, 10, 10, 72, 3, 1, 1, xdt, seed=3, bias=False, out_dtype=torch.bfloat16, beta=0.5)


def test_conv_w8_resnet_layer(N):
    check(N, 32, 64, 56, 56, 64, 3, 1, 1, "f32", seed=9)
    check(N, 32, 128, 28, 28, 128, 3, 1, 1, "bf16", seed=10)


# 3 x 3 / stride 1 over whole 32-channel groups, Ho*Wo % 4 == 0 and images that fill 448-pixel tiles: the LDS-resident-patch
# kernel (a partial last tile, ragged output channels, one to six channel groups, pad 0 / 1 / 2, an odd width, several
# small images per tile with a last tile of one image, several channel tiles)
PATCH_CASES = [
    (3, 128, 28, 28, 70, 3, 1, 1), (5, 64, 20, 20, 64, 3, 1, 1), (5, 64, 22, 22, 32, 3, 1, 0), (2, 96, 42, 42, 70, 3, 1, 1),
    (5, 32, 18, 18, 64, 3, 1, 2), (2, 64, 56, 56, 8, 3, 1, 1), (5, 64, 20, 21, 16, 3, 1, 1), (11, 64, 14, 14, 64, 3, 1, 1),
    (12, 128, 14, 14, 192, 3, 1, 1), (40, 32, 8, 8, 64, 3, 1, 1), (5, 64, 30, 28, 192, 3, 1, 1),
    # images whose pixel count is not a multiple of 4 (pixels stored one by one): 7 x 7 (eight images a tile, a last tile
    # of three), 9 x 9 from a 7 x 7 input with pad 2, 21 x 23
    (43, 64, 7, 7, 70, 3, 1, 1), (40, 32, 7, 7, 64, 3, 1, 2), (5, 32, 21, 23, 64, 3, 1, 1),
]


@pytest.mark.parametrize("xdt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("B,C,H,W,Co,k,s,p", PATCH_CASES)
def test_conv_w8_patch_kernel_shapes(N, B, C, H, W, Co, k, s, p, xdt):
    check(N, B, C, H, W, Co, k, s, p, xdt, seed=B * 7 + C + Co)


@pytest.mark.parametrize("xdt", ["f32", "bf16"])
def test_conv_w8_patch_kernel_bf16_out_no_bias(N, xdt):
    check(N, 6, 64, 20, 20, 72, 3, 1, 1, xdt, seed=5, bias=False, out_dtype=torch.bfloat16, beta=0.5)
    check(N, 40, 64, 7, 7, 72, 3, 1, 1, xdt, seed=6, bias=False, out_dtype=torch.bfloat16, beta=0.5)


def test_conv_w8_patch_kernel_integer_activations_are_exact(N):
    """Integer-valued activations: every product and every partial sum is exact in fp32, so the patch kernel, whatever
    the order of its sums, must give the exact convolution and the oracle's epilogue of it bit for bit."""
    from oracle import oracle_c as OC
    g = torch.Generator().manual_seed(21)
    B, C, H, W, Co = 5, 64, 20, 20, 70
    x = torch.randint(-8, 9, (B, C, H, W), generator=g).float()
    w = torch.randint(-128, 128, (Co, C, 3, 3), generator=g, dtype=torch.int32).to(torch.int8)
    alpha, bias = torch.rand(Co, generator=g) * 0.02 + 1e-4, torch.randn(Co, generator=g) * 5
    acc = F.conv2d(x.double(), w.double(), padding=1).to(torch.int32).numpy()
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), 0.25)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        y = N.conv2d_w8(x.to(dt).cuda(), w.cuda(), 1, 1, bias=bias.cuda(), alpha=alpha.cuda(), beta=0.25)
        np.testing.assert_array_equal(y.cpu().numpy(), ref)


# stems (Cin <= 4, 5-7 filter rows and columns, stride 2, output rows of whole 16-byte vectors): ragged row and pixel blocks,
# ragged channel tiles, two pixel blocks a row, one to four input channels
STEM_CASES = [
    (4, 3, 32, 32, 64, 7, 2, 3), (3, 3, 45, 40, 80, 7, 2, 3), (3, 2, 33, 48, 130, 5, 2, 2), (2, 4, 64, 64, 64, 7, 2, 3),
    (1, 3, 30, 224, 64, 7, 2, 3), (3, 1, 40, 40, 16, 6, 2, 2),
]


@pytest.mark.parametrize("xdt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("B,C,H,W,Co,k,s,p", STEM_CASES)
def test_conv_w8_stem_kernel_shapes(N, B, C, H, W, Co, k, s, p, xdt):
    check(N, B, C, H, W, Co, k, s, p, xdt, seed=B * 11 + C + Co)


def test_conv_w8_stem_kernel_bf16_out_and_exact_integers(N):
    check(N, 3, 3, 45, 40, 80, 7, 2, 3, "f32", seed=8, bias=False, out_dtype=torch.bfloat16, beta=0.5)
    from oracle import oracle_c as OC
    g = torch.Generator().manual_seed(22)
    x = torch.randint(-8, 9, (3, 3, 40, 48), generator=g).float()
    w = torch.randint(-128, 128, (70, 3, 7, 7), generator=g, dtype=torch.int32).to(torch.int8)
    alpha, bias = torch.rand(70, generator=g) * 0.02 + 1e-4, torch.randn(70, generator=g) * 5
    acc = F.conv2d(x.double(), w.double(), stride=2, padding=3).to(torch.int32).numpy()
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), 0.25)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        y = N.conv2d_w8(x.to(dt).cuda(), w.cuda(), 2, 3, bias=bias.cuda(), alpha=alpha.cuda(), beta=0.25)
        np.testing.assert_array_equal(y.cpu().numpy(), ref)


def test_conv_w8_1x1_strided_random_shapes(N):
    """Seeded random 1 x 1 layers with stride 2..4 (the subsampling re-layout) on all three activation types."""
    rng = np.random.default_rng(78)
    for i in range(15):
        C, Co = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        check(N, int(rng.integers(1, 4)), C, H, W, Co, 1, int(rng.integers(2, 5)), 0, ("f32", "bf16", "f16")[i % 3], seed=100 + i)


def test_conv_w8_wide_tiles_one_plane(N):
    """bf16 activations, more than 128 output channels and tiles for most CUs: the 128 x 256 tiles (ragged channel
    counts, one and two tile columns, stride 1 and 2, fp32 and bf16 outputs); fp32 activations stay on 128 x 128."""
    check(N, 32, 128, 28, 28, 200, 3, 1, 1, "bf16", seed=11)
    check(N, 64, 64, 40, 40, 300, 3, 2, 1, "bf16", seed=12, out_dtype=torch.bfloat16, beta=0.25)
    check(N, 48, 32, 23, 23, 256, 1, 1, 0, "bf16", seed=13, bias=False)
    check(N, 32, 128, 28, 28, 200, 3, 1, 1, "f32", seed=14)


@pytest.mark.parametrize("case", [c for c in FW.cases() if c.startswith("conv")])
def test_conv_w8_golden_layer_forward(N, case):
    """The reference's Quantizer.forward outputs for Conv2d layers (integer-valued float activations: every product
    and sum is exact, so the drop-in layer must reproduce them to the last bit of the epilogue)."""
    from quantization_amd.quant.Weight_Only import Quantizer
    w, x, ref = FW.get(case, "w"), FW.get(case, "x"), FW.get(case, "y")
    bias = FW.get(case, "bias")
    stride, pad = (int(v) for v in FW.get(case, "stride_pad"))
    conv = torch.nn.Conv2d(w.shape[1], w.shape[0], w.shape[2], stride=stride, padding=pad, bias=bias is not None)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
        if bias is not None:
            conv.bias.copy_(torch.from_numpy(bias))
    ql = Quantizer.from_float(module=conv, data_metry=Q_PARAMS, quantize_output=False)
    y = ql(torch.from_numpy(x).float().cuda()).cpu().numpy()
    np.testing.assert_allclose(y, ref, rtol=1e-6, atol=1e-6)


FF = Golden("wo_forward_float.npz")


@pytest.mark.parametrize("case", [c for c in FF.cases() if "conv" in c])
def test_conv_w8_golden_real_valued_activations(N, case):
    """The reference's own Quantizer.forward outputs for Conv2d on real-valued fp32 activations (fixture made by
    oracle/gen_golden.py wo_float): within 1e-3 of the reference, 5e-5 of the sum of magnitudes of the oracle."""
    from oracle import oracle_np as O
    x, q, s, ref = FF.get(case, "x"), FF.get(case, "q"), FF.get(case, "scales"), FF.get(case, "y")
    bias = FF.get(case, "bias")
    stride, pad = (int(v) for v in FF.get(case, "stride_pad"))
    y = N.conv2d_w8(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), stride, pad,
                    bias=None if bias is None else torch.from_numpy(bias).cuda(), alpha=torch.from_numpy(s).cuda()).cpu().numpy()
    mag = O.wo_conv2d_forward_float(np.abs(x), np.abs(q.astype(np.int16)), s, None if bias is None else np.abs(bias), stride, pad)
    np.testing.assert_array_less(np.abs(y - O.wo_conv2d_forward_float(x, q, s, bias, stride, pad)), 5e-5 * mag + 1e-30)
    np.testing.assert_array_less(np.abs(y - ref), 1e-3 * np.maximum(np.abs(ref), 1e-2 * mag) + 1e-30)


def test_conv_w8_argument_checks(N):
    x = torch.randn(1, 4, 6, 6).cuda()
    w = torch.zeros(8, 4, 3, 3, dtype=torch.int8).cuda()
    a = torch.ones(8).cuda()
    with pytest.raises(RuntimeError):
        N.conv2d_w8(x.cpu(), w, alpha=a)
    with pytest.raises(RuntimeError):
        N.conv2d_w8(x, w[:, :2].contiguous(), alpha=a)
    with pytest.raises(RuntimeError):
        N.conv2d_w8(x, w, alpha=None)
    with pytest.raises(RuntimeError):
        N.conv2d_w8(x[:, :, :2, :2].contiguous(), w, alpha=a)
    assert N.conv2d_w8(x, w, 1, 1, alpha=a).shape == (1, 8, 6, 6)
