"""The channels-last int8 convolution (miq_conv2d_i8_nhwc) through the C ABI: raw int32 sums bit-equal to the exact
convolution, dequantised outputs bit-equal to the oracle's epilogue, the int8 output equal to that value requantised
(clamp(rint(.)), round-half-even), borders of a padded output left untouched, and two layers chained without any
re-layout in between.  The channels-last layout and the int8 output have no counterpart in the reference (parity
unpinned for them); the arithmetic they wrap is the NCHW path's, which the reference's fixtures pin."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    from quantization_amd import _native
    _native.lib()
    return _native


def rand_i8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-128, 128, shape, generator=g, dtype=torch.int32).to(torch.int8)


def to_padded_nhwc(x, pad):
    """NCHW int8 -> [B, H+2p, W+2p, C] with a zero border."""
    return F.pad(x.permute(0, 2, 3, 1), (0, 0, pad, pad, pad, pad)).contiguous()


SHAPES = [  # B, Cin, H, W, Cout, k, stride, pad
    (2, 4, 9, 11, 8, 3, 1, 1), (3, 16, 13, 13, 24, 3, 2, 1), (2, 64, 14, 14, 64, 3, 1, 1), (1, 8, 7, 7, 4, 1, 1, 0),
    (2, 32, 12, 10, 36, 1, 2, 0), (4, 128, 14, 14, 256, 3, 1, 1), (2, 12, 20, 20, 16, 5, 2, 2), (2, 64, 8, 8, 200, 3, 1, 0),
    (70, 256, 14, 14, 256, 3, 1, 1),   # enough 256 x 256 tiles for the pipelined kernel (54 x ... pixels)
    (256, 64, 28, 28, 64, 3, 1, 1),    # the LDS-resident-patch kernel, two tiles per image
    (3, 64, 56, 56, 64, 3, 1, 1), (4, 128, 28, 28, 128, 3, 1, 1),      # the same: one / two channel groups and channel tiles
    (5, 64, 30, 26, 192, 3, 1, 1), (4, 64, 40, 40, 64, 3, 1, 0),        # ... a ragged last tile, no padding
    # the patch kernel with several small images per tile: two 14 x 14 images (odd batch), eight 7 x 7 images (ragged batch)
    (11, 64, 14, 14, 64, 3, 1, 1), (43, 128, 7, 7, 128, 3, 1, 1),
    # the stem kernel (4-byte pixels, 5-8 filter columns, stride 2): ragged tiles, two channel tiles, 8 x 8 filters
    (2, 4, 32, 32, 64, 7, 2, 3), (3, 4, 45, 40, 80, 7, 2, 3), (2, 4, 33, 45, 32, 5, 2, 2), (1, 4, 64, 64, 64, 8, 2, 3),
]


@pytest.mark.parametrize("B,C,H,W,Co,k,s,p", SHAPES)
def test_conv_nhwc_matches_exact_convolution(N, B, C, H, W, Co, k, s, p):
    x, w = rand_i8((B, C, H, W), B + C + H), rand_i8((Co, C, k, k), Co + k)
    ref = F.conv2d(x.double(), w.double(), stride=s, padding=p).to(torch.int32)          # exact: |sum| < 2^53
    xp, wp = to_padded_nhwc(x, p).cuda(), N.conv2d_i8_pack_weight(w.cuda())
    acc = N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, out_dtype=torch.int32)
    assert torch.equal(acc.cpu(), ref.permute(0, 2, 3, 1))
    assert torch.equal(acc, N.conv2d_i8(x.cuda(), w.cuda(), s, p, out_dtype=torch.int32).permute(0, 2, 3, 1))
    g = torch.Generator().manual_seed(5)
    alpha = torch.rand(Co, generator=g) * 2e-4 + 1e-6
    bias = torch.randn(Co, generator=g) * 100
    yref = OC.epilogue_f32(ref.numpy(), bias.numpy(), alpha.numpy(), 0.25)            # NCHW, channel axis 1
    y = N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, bias=bias.cuda(), alpha=alpha.cuda(), beta=0.25, out_dtype=torch.float32)
    np.testing.assert_array_equal(y.cpu().numpy(), yref.transpose(0, 2, 3, 1))
    yb = N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, bias=bias.cuda(), alpha=alpha.cuda(), beta=0.25, out_dtype=torch.bfloat16)
    np.testing.assert_array_equal(yb.float().cpu().numpy(), O.round_f32_to_bf16(yref).transpose(0, 2, 3, 1))
    # int8 output into a buffer with a border of 1 that holds a sentinel: interior = requantised value, border untouched
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.full((B, Ho + 2, Wo + 2, Co), 77, dtype=torch.int8, device="cuda")
    N.conv2d_i8_nhwc(xp, wp, w.shape, s, p, bias=bias.cuda(), alpha=alpha.cuda(), beta=0.25, out_dtype=torch.int8, out_pad=1,
                     out=out)
    qref = np.clip(np.rint(yref), -128, 127).astype(np.int8).transpose(0, 2, 3, 1)
    o = out.cpu().numpy()
    np.testing.assert_array_equal(o[:, 1:-1, 1:-1, :], qref)
    border = np.ones(o.shape[:3], dtype=bool)
    border[:, 1:-1, 1:-1] = False
    assert (o[border] == 77).all()


def test_two_layers_chain_without_relayout(N):
    """conv 3x3 (16 -> 32, int8 out with the next layer's border) feeding conv 3x3 / 2 (32 -> 64): the second layer
    stages the first one's output buffer as it is."""
    B, H = 3, 18
    x, w1, w2 = rand_i8((B, 16, H, H), 1), rand_i8((32, 16, 3, 3), 2), rand_i8((64, 32, 3, 3), 3)
    a1 = torch.full((32,), 3e-5)
    wp1, wp2 = N.conv2d_i8_pack_weight(w1.cuda()), N.conv2d_i8_pack_weight(w2.cuda())
    h = N.conv2d_i8_nhwc(to_padded_nhwc(x, 1).cuda(), wp1, w1.shape, 1, 1, alpha=a1.cuda(), out_dtype=torch.int8, out_pad=1)
    y = N.conv2d_i8_nhwc(h, wp2, w2.shape, 2, 1, out_dtype=torch.int32)
    r1 = F.conv2d(x.double(), w1.double(), padding=1).to(torch.int32)
    h_ref = np.clip(np.rint(OC.epilogue_f32(r1.numpy(), None, a1.numpy(), 0.0)), -128, 127)
    r2 = F.conv2d(torch.from_numpy(h_ref).double(), w2.double(), stride=2, padding=1).to(torch.int32)
    assert torch.equal(y.cpu(), r2.permute(0, 2, 3, 1))


def test_conv_nhwc_rejects_bad_arguments(N):
    w = rand_i8((8, 6, 3, 3), 1).cuda()
    with pytest.raises(RuntimeError, match="multiples of 4"):
        N.conv2d_i8_nhwc(torch.zeros(1, 7, 7, 6, dtype=torch.int8, device="cuda"), N.conv2d_i8_pack_weight(w), w.shape, 1, 1)
    w = rand_i8((8, 8, 3, 3), 1).cuda()
    with pytest.raises(RuntimeError, match="does not match"):
        N.conv2d_i8_nhwc(torch.zeros(1, 7, 7, 4, dtype=torch.int8, device="cuda"), N.conv2d_i8_pack_weight(w), w.shape, 1, 1)
    with pytest.raises(RuntimeError, match="wrong shape"):
        N.conv2d_i8_nhwc(torch.zeros(1, 7, 7, 8, dtype=torch.int8, device="cuda"), N.conv2d_i8_pack_weight(w), w.shape, 1, 1,
                         out=torch.zeros(1, 5, 5, 4, dtype=torch.int8, device="cuda"))
