"""Conv2d with dilation, groups, a stride / padding per axis and string padding (miq_conv2d_general, through the C ABI and
the drop-in layers) against F.conv2d.

The reference's layers forward all of these arguments to F.conv2d (quant/Weight_Only/Quantizer.py:73,
quant/Weight_Activation/Quantizer.py:53-56).  int8 activations are held to bit-exactness against the same convolution
in float64 (exact on these integers: the reference's own acceptance check for its int8 kernel,
cuda_kernels/conv2d_int8/test.py:27-33) and, with an epilogue, against the oracle's (acc + bias) * alpha + beta;
floating-point activations to 5e-5 of the sum of magnitudes per element, far inside the 1e-3 relative of the
weight-only path.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
Q_PARAMS = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def rand_i8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-128, 128, shape, generator=g, dtype=torch.int32).to(torch.int8)


# B, Cin, H, W, Cout, (KH, KW), stride, padding, dilation, groups
CASES = [
    (2, 8, 12, 12, 16, (3, 3), 1, 2, 2, 1),              # dilation 2
    (2, 8, 13, 11, 6, (3, 3), (2, 1), (1, 2), 1, 1),     # a stride and a padding per axis
    (3, 16, 10, 10, 24, (3, 3), 1, 1, 1, 4),             # groups: 4 input, 6 output channels each (thin groups)
    (2, 32, 9, 9, 32, (3, 3), 1, 1, 1, 32),              # depthwise
    (2, 12, 9, 9, 36, (3, 3), 2, 1, 1, 12),              # depthwise with a channel multiplier of 3, stride 2
    (2, 64, 8, 8, 80, (3, 3), 1, 1, 1, 2),               # 40 output channels a group: one ragged 64-channel block
    (1, 40, 7, 9, 150, (3, 3), 1, "same", (2, 3), 1),    # 'same' under a dilation per axis; three channel blocks, 360 taps
    (2, 6, 10, 10, 10, (4, 2), 1, "same", 1, 2),         # 'same' on even filter extents: one more row / column behind
    (2, 4, 11, 13, 9, (5, 3), (1, 2), "valid", 1, 1),    # 'valid', a filter that is not square
    (1, 3, 40, 40, 8, (7, 7), (2, 3), (3, 0), (1, 2), 1),
    (2, 16, 20, 300, 16, (1, 5), 1, (0, 2), 1, 16),      # depthwise 1 x 5 over rows longer than a block of pixels
    (1, 1, 5, 5, 1, (3, 3), 1, 0, 1, 1),
    (70, 4, 6, 6, 4, (3, 3), 1, 1, 2, 2),                # many images
    # thin groups: the batch's pixels dealt over the lanes image after image (workgroups that span images, 1 / 8 / 4 channels
    # a lane), and with enough of them for four workgroups a CU four pixels a lane
    (2, 8, 50, 47, 8, (3, 3), 1, 1, 1, 8), (5, 8, 9, 11, 24, (3, 3), 1, 1, 1, 4), (3, 6, 20, 20, 9, (3, 3), 2, 1, 1, 3),
    (8, 64, 48, 47, 64, (3, 3), 1, 1, 1, 64), (8, 64, 64, 64, 96, (3, 3), 1, 1, 1, 32), (8, 32, 64, 63, 192, (3, 3), 1, 1, 2, 32),
    # depthwise with stride 1 / dilation 1 along x and 3, 5, 7 filter columns: four outputs of a row per lane (ragged rows,
    # a stride and a dilation along y, 'same')
    (2, 24, 13, 17, 24, (5, 5), 1, 2, 1, 24), (3, 10, 21, 18, 10, (7, 7), (2, 1), (3, 3), (2, 1), 10),
    (2, 6, 12, 10, 6, (2, 3), 1, "same", 1, 6), (1, 5, 8, 3, 5, (3, 3), 1, 1, 1, 5),
    # ... rows of whole vectors whose outputs reach past the image (padding 2 under a 3 x 3 filter), and stride 2 along x
    (2, 8, 10, 12, 8, (3, 3), 1, 2, 1, 8), (2, 8, 12, 16, 8, (5, 5), 1, (1, 4), 1, 8),
    (3, 8, 16, 16, 8, (3, 3), 2, 1, 1, 8), (2, 12, 15, 20, 12, (5, 5), (1, 2), 2, 1, 12), (2, 6, 9, 12, 6, (7, 7), 2, (3, 5), 1, 6),
    (4, 16, 30, 29, 16, (3, 3), 2, 1, 1, 16),
    # dense layers large enough for every tile shape of the implicit-GEMM kernels
    (4, 64, 28, 28, 64, (3, 3), 1, 2, 2, 1), (2, 128, 20, 20, 256, (3, 3), (2, 1), (1, 0), (1, 2), 1),
    (3, 32, 30, 30, 160, (3, 3), (1, 2), "valid", (2, 1), 1), (2, 70, 17, 19, 33, (2, 4), 1, "same", (3, 2), 1),
    (8, 256, 14, 14, 256, (3, 3), 1, 2, 2, 1),
    # a dilation named along an axis on which the filter has one tap (it then has none): dense and grouped
    (4, 39, 19, 39, 22, (2, 1), (2, 3), (2, 3), (1, 2), 1), (2, 20, 30, 31, 14, (4, 1), 3, 3, 2, 1),
    (2, 16, 12, 12, 16, (1, 3), 1, 1, (3, 1), 1), (2, 8, 9, 9, 8, (3, 1), 1, 1, (1, 2), 8), (2, 8, 9, 9, 16, (1, 1), 2, 0, 3, 2),
    # more channel blocks / images than a grid axis holds (65535): the excess rides on the grid's first axis
    (1, 70000, 2, 2, 70000, (2, 2), 1, 1, 1, 70000), (66000, 16, 2, 2, 16, (2, 2), 1, 1, 1, 1),
]


def torch_args(stride, padding, dilation, groups):
    return dict(stride=stride, padding=padding, dilation=dilation, groups=groups)


@pytest.mark.parametrize("B,C,H,W,Co,k,s,p,d,g", CASES)
def test_int8_raw_sums_are_exact(N, B, C, H, W, Co, k, s, p, d, g):
    x, w = rand_i8((B, C, H, W), B + C), rand_i8((Co, C // g, *k), Co + H)
    ref = F.conv2d(x.cuda().double(), w.cuda().double(), **torch_args(s, p, d, g)).to(torch.int32)
    out = N.conv2d_general(x.cuda(), w.cuda(), s, p, d, g, out_dtype=torch.int32)
    assert out.shape == ref.shape and out.dtype == torch.int32
    assert torch.equal(out, ref)


@pytest.mark.parametrize("B,C,H,W,Co,k,s,p,d,g", CASES[:8])
def test_int8_epilogue_bit_exact_vs_oracle(N, B, C, H, W, Co, k, s, p, d, g):
    x, w = rand_i8((B, C, H, W), 5), rand_i8((Co, C // g, *k), 6)
    gen = torch.Generator().manual_seed(4)
    alpha, bias = torch.rand(Co, generator=gen) * 0.02, torch.randn(Co, generator=gen) * 30
    acc = F.conv2d(x.double(), w.double(), **torch_args(s, p, d, g)).to(torch.int32).numpy()
    ref = OC.epilogue_f32(acc, bias.numpy(), alpha.numpy(), -1.25)
    out = N.conv2d_general(x.cuda(), w.cuda(), s, p, d, g, bias=bias.cuda(), alpha=alpha.cuda(), beta=-1.25)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    outb = N.conv2d_general(x.cuda(), w.cuda(), s, p, d, g, bias=bias.cuda(), alpha=alpha.cuda(), beta=-1.25,
                            out_dtype=torch.bfloat16)
    np.testing.assert_array_equal(outb.float().cpu().numpy(), O.round_f32_to_bf16(ref))
    # no bias
    ref0 = OC.epilogue_f32(acc, np.zeros(Co, np.float32), alpha.numpy(), 0.0)
    out0 = N.conv2d_general(x.cuda(), w.cuda(), s, p, d, g, alpha=alpha.cuda())
    np.testing.assert_array_equal(out0.cpu().numpy(), ref0)


@pytest.mark.parametrize("xdt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("B,C,H,W,Co,k,s,p,d,g", CASES)
def test_float_activations(N, B, C, H, W, Co, k, s, p, d, g, xdt):
    gen = torch.Generator().manual_seed(B * 100 + C + Co)
    x = torch.randn(B, C, H, W, generator=gen).to(DT[xdt])
    w = rand_i8((Co, C // g, *k), Co)
    alpha = torch.rand(Co, generator=gen) * 0.02 + 1e-4
    b = torch.randn(Co, generator=gen) * 5
    out_dtype = torch.bfloat16 if (B + Co) % 2 else torch.float32
    y = N.conv2d_general(x.cuda(), w.cuda(), s, p, d, g, bias=b.cuda(), alpha=alpha.cuda(), beta=0.25, out_dtype=out_dtype)
    a4 = alpha.double().view(1, -1, 1, 1)
    kw = torch_args(s, p, d, g)
    ref = (F.conv2d(x.double(), w.double(), **kw) + b.double().view(1, -1, 1, 1)) * a4 + 0.25
    mag = (F.conv2d(x.double().abs(), w.double().abs(), **kw) + b.double().abs().view(1, -1, 1, 1)) * a4 + 0.25
    assert tuple(y.shape) == tuple(ref.shape) and y.dtype == out_dtype
    tol = 5e-5 * mag + 1e-30
    if out_dtype == torch.bfloat16:
        tol = tol + ref.abs() * 2.0 ** -8
    worst = float(((y.cpu().double() - ref).abs() / tol).max())
    assert worst <= 1.0, f"worst error / tolerance = {worst:.3f}"


def test_rejects_what_conv2d_rejects(N):
    x, w = rand_i8((1, 4, 8, 8), 1).cuda(), rand_i8((4, 2, 3, 3), 2).cuda()
    a = torch.ones(4, device="cuda")
    with pytest.raises(RuntimeError, match="same"):
        N.conv2d_general(x, w, 2, "same", 1, 2, alpha=a)
    with pytest.raises(RuntimeError, match="shape mismatch"):
        N.conv2d_general(x, w, 1, 0, 1, 1, alpha=a)          # the kernel has Cin / 2 channels
    with pytest.raises(RuntimeError, match="larger than the padded input"):
        N.conv2d_general(x, w, 1, 0, 4, 2, alpha=a)          # dilated extent 9 > 8
    with pytest.raises(RuntimeError, match="alpha"):
        N.conv2d_general(x, w, 1, 0, 1, 2)
    with pytest.raises(RuntimeError, match="output dtype"):
        N.conv2d_general(x.float(), w, 1, 0, 1, 2, alpha=a, out_dtype=torch.int32)


LAYERS = [
    dict(in_channels=8, out_channels=12, kernel_size=3, stride=1, padding=2, dilation=2, groups=1),
    dict(in_channels=16, out_channels=16, kernel_size=3, stride=2, padding=1, groups=16),
    dict(in_channels=8, out_channels=8, kernel_size=(3, 5), stride=(1, 2), padding=(1, 2), groups=2),
    dict(in_channels=6, out_channels=10, kernel_size=4, padding="same", groups=2, bias=False),
]


@pytest.mark.parametrize("kw", LAYERS)
def test_weight_only_layer_forward(kw):
    """The drop-in layer on such a Conv2d: y = scale[n] * (conv(x, Wq) + b[n]) (quant/Weight_Only/Quantizer.py:73-86), for
    floating-point activations (within 1e-3 relative) and for int8 activations (the integers are exact, so the fp32
    epilogue is the only rounding)."""
    from quantization_amd.qlayers import WeightOnlyQuantizer
    torch.manual_seed(3)
    m = nn.Conv2d(**kw).cuda()
    q = WeightOnlyQuantizer.from_float(m, data_metry=Q_PARAMS)
    assert "QConv2d" in repr(q)
    x = torch.randn(3, kw["in_channels"], 14, 15, device="cuda")
    args = dict(stride=m.stride, padding=m.padding, dilation=m.dilation, groups=m.groups)
    scale = q.weight_quant.scales.reshape(1, -1, 1, 1).double()
    b = 0.0 if m.bias is None else m.bias.detach().double().view(1, -1, 1, 1)
    ref = (F.conv2d(x.double(), q.weight.double(), **args) + b) * scale
    y = q(x)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
    # one image without the batch axis, as F.conv2d takes it
    y1 = q(x[0])
    assert y1.shape == ref.shape[1:] and torch.equal(y1, y[0])
    xi = torch.randint(-128, 128, x.shape, device="cuda", dtype=torch.int32).to(torch.int8)
    refi = (F.conv2d(xi.double(), q.weight.double(), **args) + b) * scale
    yi = q(xi)
    np.testing.assert_allclose(yi.cpu().numpy(), refi.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(refi.abs().max()))


def test_weight_only_layer_with_input_quant_on_a_grouped_dilated_conv():
    """input_quant on such a layer: the activation quantised per tensor into a frame of its zero point, then the int8
    direct convolution: equal to convolving the dequantised activation in float64."""
    from quantization_amd.activation import quantize_tensor_i8
    from quantization_amd.qlayers import WeightOnlyQuantizer
    torch.manual_seed(5)
    m = nn.Conv2d(8, 12, 3, stride=(2, 1), padding=(2, 1), dilation=(2, 1), groups=4).cuda()
    q = WeightOnlyQuantizer.from_float(m, data_metry=Q_PARAMS)
    q.input_quant = True
    x = torch.randn(2, 8, 12, 12, device="cuda") + 0.3
    xq, sx, zp = quantize_tensor_i8(x)
    xhat = (xq.double() - zp.double()) * sx.double()
    ref = (F.conv2d(xhat, q.weight.double(), None, m.stride, m.padding, m.dilation, m.groups) +
           m.bias.detach().double().view(1, -1, 1, 1)) * q.weight_quant.scales.reshape(1, -1, 1, 1).double()
    y = q(x)
    np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("kw", LAYERS[:3])
def test_weight_activation_layer_forward(kw):
    """The weight+activation layer on such a Conv2d: y = (conv(x_q, Wq) + bias') * scalers + output_zero_point
    (quant/Weight_Activation/Quantizer.py:53-60) on int8 activations, bit-exact against the oracle's epilogue of the exact sums."""
    from quantization_amd.qlayers import WeightActivationQuantizer
    torch.manual_seed(7)
    m = nn.Conv2d(**kw).cuda()
    q = WeightActivationQuantizer.from_float(m, data_metry=Q_PARAMS)
    n = kw["out_channels"]
    gen = torch.Generator().manual_seed(11)
    q.bias = (torch.randn(n, generator=gen) * 20).cuda()
    q.scalers = (torch.rand(n, generator=gen) * 0.01 + 1e-4).cuda()
    q.output_zero_point = torch.tensor([3.0]).cuda()
    xi = torch.randint(-128, 128, (2, kw["in_channels"], 11, 13), dtype=torch.int32).to(torch.int8).cuda()
    acc = F.conv2d(xi.double(), q.weight.double(), None, m.stride, m.padding, m.dilation, m.groups).to(torch.int32)
    ref = OC.epilogue_f32(acc.cpu().numpy(), q.bias.cpu().numpy(), q.scalers.cpu().numpy(), 3.0)
    y = q(xi)
    np.testing.assert_array_equal(y.cpu().numpy(), ref)


def test_chunker_on_a_mobile_style_model():
    """Chunker(model).model, then a forward (the reference's flow, quant/Weight_Only/Main.py:40-42, Chunker.py:10-15), on a
    model whose convolutions are strided, depthwise, pointwise, dilated and 'same'-padded: the output equals the same network
    evaluated layer by layer as the reference's forward writes it, scale * (F.conv2d(x, Wq) + b) in float64, within 1e-3."""
    from quantization_amd.qlayers import WeightOnlyChunker, WeightOnlyQuantizer
    torch.manual_seed(21)
    net = nn.Sequential(
        nn.Conv2d(3, 16, 3, stride=2, padding=1), nn.ReLU(),
        nn.Conv2d(16, 16, 3, padding=1, groups=16), nn.ReLU(),
        nn.Conv2d(16, 32, 1), nn.ReLU(),
        nn.Conv2d(32, 32, 3, padding=2, dilation=2), nn.ReLU(),
        nn.Conv2d(32, 32, (3, 5), stride=(2, 1), padding=(1, 2), groups=4), nn.ReLU(),
        nn.Conv2d(32, 24, 4, padding="same"), nn.ReLU(),
        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(24, 10)).cuda()
    qnet = WeightOnlyChunker(net, None).model
    layers = [m for m in qnet if isinstance(m, WeightOnlyQuantizer)]
    assert len(layers) == 7 and sum(m.cnn for m in layers) == 6
    x = torch.randn(5, 3, 40, 36, device="cuda")
    y = qnet(x)
    h = x.double()
    for m in qnet:
        if isinstance(m, WeightOnlyQuantizer):
            s = m.weight_quant.scales.reshape(-1).double()
            b = 0.0 if m.bias is None else m.bias.detach().double()
            if m.cnn:
                h = (F.conv2d(h, m.weight.double(), None, m.stride, m.padding, m.dilation, m.groups) +
                     (b.view(1, -1, 1, 1) if m.bias is not None else 0.0)) * s.view(1, -1, 1, 1)
            else:
                h = (F.linear(h, m.weight.double()) + b) * s
        else:
            h = m(h)
    assert y.shape == h.shape
    np.testing.assert_allclose(y.cpu().numpy(), h.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(h.abs().max()))


def test_dense_layer_into_an_output_that_is_not_vector_aligned(N):
    """A dense layer whose output starts one element into its buffer cannot take the MFMA kernels' vector stores: the direct
    kernel computes it (same exact sums; the same fp32 epilogue)."""
    B, C, H, W, Co = 2, 24, 13, 14, 40
    x, w = rand_i8((B, C, H, W), 3).cuda(), rand_i8((Co, C, 3, 3), 4).cuda()
    ref = F.conv2d(x.double(), w.double(), stride=(1, 2), padding=(2, 1), dilation=(2, 1)).to(torch.int32)
    flat = torch.zeros(ref.numel() + 4, dtype=torch.int32, device="cuda")
    out = flat[1:1 + ref.numel()].view(ref.shape)
    assert out.data_ptr() % 16 == 4
    N.conv2d_general(x, w, (1, 2), (2, 1), (2, 1), 1, out_dtype=torch.int32, out=out)
    assert torch.equal(out, ref) and int(flat[0]) == 0 and int(flat[-1]) == 0
    alpha = torch.rand(Co, device="cuda") * 0.01
    yf = torch.zeros(ref.numel() + 4, device="cuda")
    y = yf[1:1 + ref.numel()].view(ref.shape)
    N.conv2d_general(x.float(), w, (1, 2), (2, 1), (2, 1), 1, alpha=alpha, out=y)
    want = ref.double() * alpha.double().view(1, -1, 1, 1)
    np.testing.assert_allclose(y.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_random_layers_vs_torch(N):
    """Seeded random layers over the whole argument space (groups, dilation, stride and padding per axis, 'same' / 'valid',
    ragged sizes): int8 activations exact against float64 F.conv2d, fp32 / fp16 / bf16 activations within the kernel's bar."""
    import random
    rnd = random.Random(77)
    for i in range(80):
        g = rnd.choice([1, 1, 1, 2, 3, 4, 8, 16])
        cin_g, cout_g = rnd.choice([1, 1, 2, 3, 5, 8, 16, 24]), rnd.choice([1, 1, 2, 3, 6, 8, 17, 40])
        if g == 1:
            cin_g, cout_g = rnd.randint(1, 70), rnd.randint(1, 90)
        C, Co = g * cin_g, g * cout_g
        k = (rnd.choice([1, 2, 3, 3, 5, 7]), rnd.choice([1, 2, 3, 3, 5, 7]))
        d = (rnd.choice([1, 1, 2, 3]), rnd.choice([1, 1, 1, 2]))
        s = (rnd.choice([1, 1, 2, 3]), rnd.choice([1, 1, 2]))
        mode = rnd.random()
        if mode < 0.15:
            p, s = "same", (1, 1)
        elif mode < 0.25:
            p = "valid"
        else:
            p = (rnd.randint(0, 3), rnd.randint(0, 3))
        eh, ew = d[0] * (k[0] - 1) + 1, d[1] * (k[1] - 1) + 1
        H, W = eh + rnd.randint(0, 30), ew + rnd.choice([rnd.randint(0, 30), 4 * rnd.randint(0, 8)])
        B = rnd.randint(1, 5)
        x, w = rand_i8((B, C, H, W), 5000 + i), rand_i8((Co, cin_g, *k), 6000 + i)
        kw = torch_args(s, p, d, g)
        ref = F.conv2d(x.cuda().double(), w.cuda().double(), **kw)
        out = N.conv2d_general(x.cuda(), w.cuda(), s, p, d, g, out_dtype=torch.int32)
        assert out.shape == ref.shape and torch.equal(out, ref.to(torch.int32)), (i, B, C, H, W, Co, k, s, p, d, g)
        xdt = rnd.choice(["f32", "f16", "bf16"])
        gen = torch.Generator().manual_seed(7000 + i)
        xf = torch.randn(B, C, H, W, generator=gen).to(DT[xdt])
        alpha = torch.rand(Co, generator=gen) * 0.02 + 1e-4
        b = torch.randn(Co, generator=gen) * 5
        y = N.conv2d_general(xf.cuda(), w.cuda(), s, p, d, g, bias=b.cuda(), alpha=alpha.cuda())
        a4 = alpha.double().view(1, -1, 1, 1)
        reff = (F.conv2d(xf.double(), w.double(), **kw) + b.double().view(1, -1, 1, 1)) * a4
        mag = (F.conv2d(xf.double().abs(), w.double().abs(), **kw) + b.double().abs().view(1, -1, 1, 1)) * a4
        worst = float(((y.cpu().double() - reff).abs() / (5e-5 * mag + 1e-30)).max())
        assert worst <= 1.0, (i, xdt, worst, B, C, H, W, Co, k, s, p, d, g)
