"""Dynamic per-tensor int8 activation quantization (quantization_amd/activation.py).

The reference never runs its activation quantizer (commented-out call sites, and Qop.quantize raises for
affine='tensor'), so no fixture of the whole step can exist: PARITY UNPINNED for the composition.  Its parts
are pinned: the scale / zero-point formula is Qop.calculate_params' per-tensor branch, whose restatement
(oracle_np.calculate_params_affine_tensor) is checked against reference-generated fixtures in
tests/test_oracle_golden.py, and the quantize formula is Qop.py:157.  Here the GPU path is compared bit for
bit with that restatement, and the layer forward with a float64 evaluation of the same expression.
"""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1,), (7,), (64, 300), (3, 5, 17), (1025, 4099), (4096, 8192)])
def test_tensor_minmax(dtype, shape):
    from quantization_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g, device="cuda") * 3 - 0.5).to(dtype)
    mm = N.tensor_minmax(x)
    assert mm.dtype == torch.float32
    assert float(mm[0]) == float(x.min()) and float(mm[1]) == float(x.max())
    if x.numel() > 9:        # a view that starts 4 bytes (or 2) into the allocation: the unaligned path
        y = torch.as_strided(x.reshape(-1), (x.numel() - 1,), (1,), 1)
        mm = N.tensor_minmax(y)
        assert float(mm[0]) == float(y.min()) and float(mm[1]) == float(y.max())


@pytest.mark.parametrize("shape,shift", [((64, 300), 0.0), ((1, 7), 2.0), ((33, 1000), -1.5), ((257, 4096), 0.3), ((5, 8), 100.0)])
def test_quantize_tensor_bit_exact_with_oracle(shape, shift):
    from quantization_amd.activation import quantize_tensor_i8
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g) * 2.5 + shift
    q, scale, zp = quantize_tensor_i8(x.cuda())
    rq, rs, rzp = O.quantize_tensor_affine_int8(x.numpy())
    np.testing.assert_array_equal(scale.cpu().numpy(), rs)
    np.testing.assert_array_equal(zp.cpu().numpy(), rzp)
    np.testing.assert_array_equal(q.cpu().numpy(), rq)


def test_quantize_tensor_degenerate_inputs():
    from quantization_amd.activation import quantize_tensor_i8
    q, scale, zp = quantize_tensor_i8(torch.zeros(4, 16, device="cuda"))       # scale floors at eps
    assert float(scale) == 2.0 ** -23 and int(zp) == -128 and bool((q == -128).all())
    rq, rs, rzp = O.quantize_tensor_affine_int8(np.zeros((4, 16), np.float32))
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    x = torch.full((3, 8), 5.0, device="cuda")                                 # all positive: min is clipped to 0
    q, scale, zp = quantize_tensor_i8(x)
    rq, rs, rzp = O.quantize_tensor_affine_int8(x.cpu().numpy())
    np.testing.assert_array_equal(q.cpu().numpy(), rq)
    assert int(zp) == int(rzp[0]) == -128


@pytest.mark.parametrize("M,N,K,bias", [(8, 64, 128, True), (100, 300, 512, True), (1, 4096, 4096, False), (256, 1024, 2048, True)])
def test_linear_dynamic_w8a8(M, N, K, bias):
    """The forward equals scale_w[n] * (x_hat . Wq^T + b[n]) with x_hat the dequantized activation, evaluated in
    float64, to 1e-5 relative to the output's scale; and it tracks the float layer within the quantization noise."""
    from quantization_amd import _native as Nv
    from quantization_amd.activation import linear_dynamic_w8a8, quantize_tensor_i8
    g = torch.Generator(device="cuda").manual_seed(M + N)
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    b = torch.randn(N, generator=g, device="cuda") if bias else None
    x = torch.randn(M, K, generator=g, device="cuda") + 0.25
    wq, ws, _ = Nv.quantize_rows_sym_i8(w)
    y = linear_dynamic_w8a8(x, wq, ws, b)
    xq, sx, zx = quantize_tensor_i8(x)
    x_hat = (xq.double() - zx.double()) * sx.double()
    ref = x_hat @ wq.double().T
    if bias:
        ref = ref + b.double()
    ref = ref * ws.double()[None, :]
    tol = 1e-5 * float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= tol
    # the weight-only forward scales the bias with the weight scale (Quantizer.py:79-86)
    dense = torch.nn.functional.linear(x, w) + (b * ws if bias else 0.0)
    rel = float((y - dense).norm() / dense.norm())
    assert rel < 0.05, rel


def test_quantizer_input_quant_flag():
    """`Quantizer.input_quant = True` routes a float activation of a Linear layer through the dynamic int8 path;
    the result stays within the activation's quantization noise of the float-activation forward."""
    from quantization_amd.quant.Weight_Only.Quantizer import Quantizer
    torch.manual_seed(3)
    lin = torch.nn.Linear(512, 256)
    qp = {'weights': {'dtype': torch.int8, 'symmetric': True, 'affine': 'channel', 'affine_dim': 0}}
    q = Quantizer.from_float(lin, data_metry=qp)
    x = torch.randn(32, 512)
    y_float = q(x)
    q.input_quant = True
    y_int8 = q(x)
    assert y_int8.shape == y_float.shape and y_int8.device == x.device
    rel = float((y_int8 - y_float).norm() / y_float.norm())
    assert 0.0 < rel < 0.03, rel


@pytest.mark.parametrize("B,C,H,Co,k,stride,pad,bias", [(2, 16, 12, 32, 3, 1, 1, True), (3, 3, 17, 8, 7, 2, 3, True),
                                                        (2, 64, 9, 64, 1, 2, 0, False), (1, 5, 8, 7, 3, 1, 2, True)])
def test_conv2d_dynamic_w8a8(B, C, H, Co, k, stride, pad, bias):
    """The convolution form: scale_w[n] * (conv(x_hat, Wq) + b[n]) with x_hat the dequantized activation and real
    zeros in the padding (which is what a frame of zero points dequantizes to), evaluated in float64."""
    from quantization_amd import _native as Nv
    from quantization_amd.activation import conv2d_dynamic_w8a8, quantize_tensor_i8
    g = torch.Generator(device="cuda").manual_seed(B + C + H)
    w = torch.randn(Co, C, k, k, generator=g, device="cuda") * 0.05
    b = torch.randn(Co, generator=g, device="cuda") if bias else None
    x = torch.randn(B, C, H, H, generator=g, device="cuda") * 1.5 + 0.7       # off-centre: a zero point far from 0
    wq, ws, _ = Nv.quantize_rows_sym_i8(w.view(Co, -1))
    wq = wq.view(Co, C, k, k)
    y = conv2d_dynamic_w8a8(x, wq, ws, b, stride=stride, padding=pad)
    xq, sx, zx = quantize_tensor_i8(x)
    assert int(zx) != 0
    x_hat = ((xq.double() - zx.double()) * sx.double()).cpu()
    ref = torch.nn.functional.conv2d(x_hat, wq.double().cpu(), stride=stride, padding=pad)
    if bias:
        ref = ref + b.double().cpu().view(1, -1, 1, 1)
    ref = ref * ws.double().cpu().view(1, -1, 1, 1)
    assert y.shape == ref.shape
    assert float((y.double().cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_quantizer_input_quant_flag_conv():
    """`input_quant = True` on a Conv2d layer: the dynamic int8 convolution, within the activation's quantization
    noise of the float-activation forward, borders included."""
    from quantization_amd.quant.Weight_Only.Quantizer import Quantizer
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(16, 32, 3, stride=1, padding=1)
    qp = {'weights': {'dtype': torch.int8, 'symmetric': True, 'affine': 'channel', 'affine_dim': 0}}
    q = Quantizer.from_float(conv, data_metry=qp)
    x = torch.randn(4, 16, 20, 20) + 0.5
    y_float = q(x)
    q.input_quant = True
    y_int8 = q(x)
    assert y_int8.shape == y_float.shape and y_int8.device == x.device
    rel = float((y_int8 - y_float).norm() / y_float.norm())
    assert 0.0 < rel < 0.03, rel
    border = torch.ones_like(y_float, dtype=torch.bool)
    border[:, :, 1:-1, 1:-1] = False
    rel_b = float((y_int8 - y_float)[border].norm() / y_float[border].norm())
    assert rel_b < 0.03, rel_b


def test_activation_entry_points_reject_bad_arguments():
    """Argument checks of the C entry points, through ctypes: every failure is reported before any launch."""
    import ctypes
    from quantization_amd import _native as N
    lib = N.lib()
    x = torch.zeros(1024, device="cuda")
    out = torch.zeros(2, device="cuda")
    ws = torch.zeros(64, dtype=torch.uint8, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    assert lib.miq_tensor_minmax(p(x), 0, 0, p(out), p(ws), 64, None) == -1            # empty tensor
    assert lib.miq_tensor_minmax(p(x), 7, 1024, p(out), p(ws), 64, None) == -1         # unknown dtype
    assert lib.miq_tensor_minmax(p(x), 0, 1024, p(out), None, 0, None) == -3           # no workspace
    assert b"workspace" in lib.miq_last_error()
    assert lib.miq_tensor_minmax(None, 0, 1024, p(out), p(ws), 64, None) == -1         # null input
    assert lib.miq_tensor_minmax_workspace(0) == 0 and lib.miq_tensor_minmax_workspace(1 << 30) == 2048 * 8
    assert lib.miq_dynamic_epilogue_i8(None, p(x), None, p(x), 4, p(out), p(x), p(x), None) == -1
    assert lib.miq_dynamic_epilogue_i8(p(out), p(x), None, p(x), -1, p(out), p(x), p(x), None) == -1
    assert lib.miq_pool2d_to_i8(p(x), 0, 1, 1, 4, 4, 2, 2, 2, 2, 0, 0, 5, p(ws), None) == -1   # unknown mode
    assert lib.miq_pool2d_to_i8(p(x), 3, 1, 1, 4, 4, 2, 2, 2, 2, 0, 0, 0, p(ws), None) == -1   # int32 input
    assert lib.miq_pool2d_to_i8(p(x), 0, 1, 1, 4, 4, 2, 2, 0, 2, 0, 0, 0, p(ws), None) == -1   # zero stride


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(7,), (64, 300), (1025, 4099), (4096, 8192)])
def test_one_launch_parameters_equal_the_separate_calls(dtype, shape):
    """miq_dynamic_quant_params_i8 (statistics, parameters and epilogue vectors in one launch, the last workgroup to
    arrive finishing the reduction) against the three-call form it replaces, bit for bit; the counter word it counts
    workgroups in is left zero, so the next call (here: repeated, and on an unaligned view) starts clean."""
    from quantization_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(sum(shape) + 1)
    x = (torch.randn(shape, generator=g, device="cuda") * 3 - 0.5).to(dtype)
    n = 300
    w_scale = torch.rand(n, generator=g, device="cuda") * 0.01 + 1e-4
    bias = torch.randn(n, generator=g, device="cuda")
    rowsum = torch.randint(-5000, 5000, (n,), generator=g, device="cuda").float()
    for rep in range(3):
        xx = x if rep < 2 or x.numel() < 10 else torch.as_strided(x.reshape(-1), (x.numel() - 1,), (1,), 1)
        ref_szp, ref_alpha, ref_corr = N.dynamic_epilogue_i8(N.tensor_minmax(xx), w_scale, bias, rowsum)
        szp, alpha, corr = N.dynamic_quant_params(xx, w_scale, bias, rowsum)
        assert torch.equal(szp[:2], ref_szp) and torch.equal(alpha, ref_alpha) and torch.equal(corr, ref_corr)
        assert int(szp[2:3].view(torch.int32)) == int(ref_szp[1])
        assert int(N._counter(x.device)) == 0
    szp2, a2, c2 = N.dynamic_quant_params(x)          # statistics and parameters only
    assert a2 is None and c2 is None and torch.equal(szp2[:2], N.dynamic_epilogue_i8(N.tensor_minmax(x), w_scale, None, rowsum)[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,pad", [((2, 3, 5, 7), (1, 1, 1, 1)), ((1, 4, 8, 8), (0, 2, 3, 0)), ((3, 16, 14, 14), (1, 1, 1, 1)),
                                       ((2, 5, 9, 13), (2, 1, 0, 3)), ((4, 8, 12, 12), (0, 0, 0, 0)), ((1, 1, 3, 1), (3, 3, 3, 3))])
def test_quantize_tensor_into_a_frame_of_zero_points(dtype, shape, pad):
    """miq_quantize_tensor_i8: the interior equals the row-parameter entry point (miq_quantize_rows_given_i8, itself
    pinned to the oracle), and for fp32 the oracle's formula directly; the frame holds the zero point."""
    from quantization_amd import _native as N
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g, device="cuda") * 2 + 0.8).to(dtype)
    szp, _, _ = N.dynamic_quant_params(x)
    q = N.quantize_tensor_i8(x, szp, pad)
    pt, pb, pl, pr = pad
    B, C, H, W = shape
    assert tuple(q.shape) == (B, C, H + pt + pb, W + pl + pr)
    inner = q[:, :, pt:pt + H, pl:pl + W]
    ref = N.quantize_rows_given_i8(x.reshape(B * C, H * W), szp[0:1], szp[1:2]).view(B, C, H, W)
    assert torch.equal(inner, ref)
    if dtype == torch.float32:
        rq, rs, rzp = O.quantize_tensor_affine_int8(x.cpu().numpy())
        np.testing.assert_array_equal(inner.cpu().numpy(), rq)
        assert float(szp[0]) == float(rs[0]) and int(szp[1]) == int(rzp[0])
    frame = torch.ones_like(q, dtype=torch.bool)
    frame[:, :, pt:pt + H, pl:pl + W] = False
    assert bool((q[frame] == int(szp[1])).all())


def test_conv2d_dynamic_w8a8_takes_one_image():
    """An unbatched [C, H, W] image, as F.conv2d takes it: the batched result of the same image."""
    from quantization_amd import _native as Nv
    from quantization_amd.activation import conv2d_dynamic_w8a8
    g = torch.Generator(device="cuda").manual_seed(2)
    w = torch.randn(8, 5, 3, 3, generator=g, device="cuda") * 0.05
    x = torch.randn(5, 11, 9, generator=g, device="cuda") + 0.4
    wq, ws, _ = Nv.quantize_rows_sym_i8(w.view(8, -1))
    wq = wq.view(8, 5, 3, 3)
    y1 = conv2d_dynamic_w8a8(x, wq, ws, None, stride=1, padding=1)
    yb = conv2d_dynamic_w8a8(x.unsqueeze(0), wq, ws, None, stride=1, padding=1)
    assert y1.dim() == 3 and torch.equal(y1, yb[0])


def test_new_activation_entry_points_reject_bad_arguments():
    import ctypes
    from quantization_amd import _native as N
    lib = N.lib()
    x = torch.zeros(1024, device="cuda")
    out = torch.zeros(4, device="cuda")
    ws = torch.zeros(64, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    q = torch.zeros(2048, dtype=torch.int8, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    f = lib.miq_dynamic_quant_params_i8
    assert f(p(x), 0, 0, None, None, None, 0, p(out), None, None, p(ws), 64, p(cnt), None) == -1        # empty tensor
    assert f(p(x), 0, 1024, None, None, None, 0, p(out), None, None, p(ws), 64, None, None) == -1       # no counter
    assert f(p(x), 0, 1024, None, None, None, 4, p(out), None, None, p(ws), 64, p(cnt), None) == -1     # channels without vectors
    assert f(p(x), 9, 1024, None, None, None, 0, p(out), None, None, p(ws), 64, p(cnt), None) == -1     # unknown dtype
    assert f(p(x), 0, 1024, None, None, None, 0, p(out), None, None, None, 0, p(cnt), None) == -3       # no workspace
    g = lib.miq_quantize_tensor_i8
    assert g(p(x), 0, 1, 32, 32, -1, 0, 0, 0, p(out), p(q), None) == -1                                # negative padding
    assert g(p(x), 0, 1, 32, 32, 0, 0, 0, 0, None, p(q), None) == -1                                   # no parameters
    assert g(p(x), 5, 1, 32, 32, 0, 0, 0, 0, p(out), p(q), None) == -1                                 # unknown dtype
    assert g(p(x), 0, 0, 32, 32, 1, 1, 1, 1, p(out), p(q), None) == 0                                  # nothing to do
    assert int(cnt) == 0
