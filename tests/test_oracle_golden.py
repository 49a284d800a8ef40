"""Pin the numpy oracle to the reference: every fixture in tests/golden/ was produced by the
reference's own Python (oracle/gen_golden.py); the oracle must reproduce each bit for bit
(int8 / int64 / scales) or to 1e-6 relative (fp32 layer outputs, where the reference ran a
float GEMM on integer-valued inputs and the oracle an exact integer one)."""
import os

import numpy as np
import pytest

from conftest import Golden, kind_of
from oracle import oracle_np as O

QW = Golden("quantizer_wo.npz")
QA = Golden("quantizer_wa.npz")
FW = Golden("wo_forward.npz")
WA = Golden("wa_chunker.npz")
C1 = Golden("config1_models.npz")


@pytest.mark.parametrize("fix,case", [(QW, c) for c in QW.cases()] + [(QA, c) for c in QA.cases()],
                         ids=lambda v: v if isinstance(v, str) else "")
def test_quantizer_matches_reference(fix, case):
    w = fix.get(case, "w")
    q, s, zp = O.quantize_channel_sym_int8(w, kind=kind_of(case))
    assert q.dtype == np.int8 and zp.dtype == np.int64
    np.testing.assert_array_equal(q, fix.get(case, "q"))
    np.testing.assert_array_equal(s, fix.get(case, "scales"))
    np.testing.assert_array_equal(zp, fix.get(case, "zp"))
    assert s.shape == fix.get(case, "scales").shape


def test_edge_rows_hand_checked():
    """The counter-intuitive rows: signed max (not abs-max), /127.5, eps floor."""
    q, s, _ = O.quantize_channel_sym_int8(QW.get("edge", "w"))
    assert q.tolist() == [[-128] * 4, [0] * 4, [127, 0, 2, -2], [127, -128, 64, 32]]
    eps = np.float32(2.0 ** -23)
    np.testing.assert_array_equal(
        s.reshape(-1), np.array([eps, eps, np.float32(1.275) / np.float32(127.5),
                                 np.float32(2) / np.float32(127.5)], dtype=np.float32))


@pytest.mark.parametrize("case", FW.cases())
def test_wo_forward_matches_reference(case):
    w, x = FW.get(case, "w"), FW.get(case, "x")
    bias = FW.get(case, "bias")
    q, s, _ = O.quantize_channel_sym_int8(w)
    np.testing.assert_array_equal(q, FW.get(case, "q"))
    np.testing.assert_array_equal(s.reshape(-1), FW.get(case, "scales"))
    if case.startswith("lin"):
        y = O.wo_linear_forward(x, q, s, bias)
    else:
        stride, pad = FW.get(case, "stride_pad")
        y = O.wo_conv2d_forward(x, q, s, bias, int(stride), int(pad))
    ref = FW.get(case, "y")
    assert y.shape == ref.shape
    np.testing.assert_allclose(y, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())


FF = Golden("wo_forward_float.npz")


@pytest.mark.parametrize("case", FF.cases())
def test_wo_forward_float_matches_reference(case):
    """Real-valued activations: the oracle's float64-summed product against the reference's own output, at the
    1e-3 relative bar (measured against sum_k |x*w|, the scale a cancelling sum is resolved at); bf16
    activations, which the reference multiplies and rounds in bf16, at bf16 resolution."""
    x, q, s = FF.get(case, "x"), FF.get(case, "q"), FF.get(case, "scales")
    bias = FF.get(case, "bias")
    q2, s2, _ = O.quantize_channel_sym_int8(FF.get(case, "w"))
    np.testing.assert_array_equal(q2, q)
    np.testing.assert_array_equal(s2.reshape(-1), s)
    ref = FF.get(case, "y")
    if "conv" in case:
        stride, pad = (int(v) for v in FF.get(case, "stride_pad"))
        y = O.wo_conv2d_forward_float(x, q, s, bias, stride, pad)
        mag = O.wo_conv2d_forward_float(np.abs(x), np.abs(q.astype(np.int16)), s, None if bias is None else np.abs(bias), stride, pad)
    else:
        y = O.wo_linear_forward_float(x, q, s, bias)
        mag = (np.abs(x).astype(np.float64) @ np.abs(q.astype(np.float64)).T + (0 if bias is None else np.abs(bias))) * s
    rel = 2.0 ** -6 if case.startswith("bf16") else 1e-5
    assert np.all(np.abs(y - ref) <= rel * mag + 1e-30), float(np.max(np.abs(y - ref) / mag))


def test_wa_epilogue_matches_reference():
    for name, conv in (("conv1", True), ("fc2", False)):
        x = WA.get(name, "x")
        wq = WA.get(name, "weight")
        bias, scalers = WA.get(name, "bias"), WA.get(name, "scalers")
        ozp = float(WA.get(name, "output_zero_point")[0])
        if conv:
            acc = O.conv2d_int8(x, wq, 1, 1)
        else:
            acc = O.linear_w8a8(x, np.ascontiguousarray(wq.T))
        y = O.epilogue(acc, bias, scalers, ozp)
        ref = WA.get(name, "y")
        np.testing.assert_allclose(y, ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())


def test_wa_weight_scales_match_wo_quantizer():
    """Weight_Activation stores the same per-channel scales the quantizer produced."""
    for name in ("conv1", "conv2", "fc2"):
        w = WA.get("fp", f"{name}.weight")
        q, s, _ = O.quantize_channel_sym_int8(w)
        np.testing.assert_array_equal(q, WA.get(name, "weight"))
        np.testing.assert_array_equal(s.reshape(-1), WA.get(name, "weight_qscale"))


def test_affine_tensor_params():
    """calculate_params, per-tensor affine branch, against the values the reference's
    calibration derived (input_qscale is Qin's scale)."""
    calib = WA.raw("calib_x")
    s, zp = O.calculate_params_affine_tensor(calib.min(), calib.max())
    np.testing.assert_array_equal(s, WA.get("conv1", "input_qscale"))
    assert zp.dtype == np.int32 and -128 <= int(zp[0]) <= 127


def test_config1_small_layers():
    for model in ("SimpleCNN", "LinearNet", "VGG"):
        for layer in C1.text(model, "layers").split(","):
            key = f"{model}/{layer}"
            if not C1.has(key, "w"):
                continue
            q, s, _ = O.quantize_channel_sym_int8(C1.get(key, "w"))
            np.testing.assert_array_equal(q, C1.get(key, "q"))
            np.testing.assert_array_equal(s.reshape(-1), C1.get(key, "scales"))


def test_linear_w8a8_is_integer_matmul():
    """The reference's acceptance check (linear_int8/test.py:13-15) on a size where a float
    matmul is exact."""
    rng = np.random.default_rng(0)
    a = rng.integers(-128, 128, (37, 96), dtype=np.int8)
    b = rng.integers(-128, 128, (96, 53), dtype=np.int8)
    c = O.linear_w8a8(a, b)
    assert c.dtype == np.int32
    np.testing.assert_array_equal(c, a.astype(np.int64) @ b.astype(np.int64))


def test_conv2d_int8_matches_torch_conv():
    """conv2d_int8/test.py:80-97: compare with F.conv2d on integer-valued inputs."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    for (B, C, H, W, Co, k, s, p) in [(2, 3, 9, 7, 4, 3, 1, 0), (1, 5, 8, 8, 6, 3, 2, 1),
                                       (2, 4, 7, 7, 3, 1, 2, 0), (1, 3, 12, 12, 2, 7, 2, 3)]:
        x = rng.integers(-128, 128, (B, C, H, W), dtype=np.int8)
        w = rng.integers(-128, 128, (Co, C, k, k), dtype=np.int8)
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), stride=s, padding=p)
        np.testing.assert_array_equal(O.conv2d_int8(x, w, s, p), ref.numpy().astype(np.int32))


def test_shard_rows_partition():
    for n in (1, 7, 64, 28672):
        for world in (1, 2, 3, 4, 8):
            spans = [O.shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_torch_baseline_port_matches_reference():
    """oracle/oracle_torch.py (the CPU baseline bench.py times) against the same fixtures."""
    import torch
    from oracle import oracle_torch as T
    for case in ("edge", "f32_10x512", "f32_257x1000", "conv_16x8x3x3", "ties"):
        q, s, zp = T.quantize_channel_sym_int8(torch.from_numpy(QW.get(case, "w")))
        np.testing.assert_array_equal(q.numpy(), QW.get(case, "q"))
        np.testing.assert_array_equal(s.numpy(), QW.get(case, "scales"))
        np.testing.assert_array_equal(zp.numpy(), QW.get(case, "zp"))
    for case in FW.cases():
        q, s, zp = T.quantize_channel_sym_int8(torch.from_numpy(FW.get(case, "w")))
        x = torch.from_numpy(FW.get(case, "x")).float()
        b = FW.get(case, "bias")
        b = None if b is None else torch.from_numpy(b)
        if case.startswith("lin"):
            y = T.wo_linear_forward(x, q, s, zp, b)
        else:
            stride, pad = (int(v) for v in FW.get(case, "stride_pad"))
            y = T.wo_conv2d_forward(x, q, s, zp, b, stride, pad)
        np.testing.assert_array_equal(y.numpy(), FW.get(case, "y"))


def test_pool_oracle_matches_reference_fixture():
    """oracle_np.pool2d_to_int8 against QuantizedMaxPool2d / QuantizedAvgPool2d outputs generated by
    the reference (tests/golden/pools.npz; oracle/gen_golden.py::pool_cases)."""
    from oracle.gen_golden import POOL_CASES
    fix = np.load(os.path.join(os.path.dirname(__file__), "golden", "pools.npz"))
    for name, mode, kind, shape, k, s, p, _ in POOL_CASES:
        y = O.pool2d_to_int8(fix[name + "/x"], k, s, p, mode)
        assert y.dtype == np.int8
        np.testing.assert_array_equal(y, fix[name + "/y"], err_msg=name)


def test_cast_to_int8_wraps_like_the_host_cast():
    v = np.array([1.9, -1.9, 127.9, 128.0, 200.0, 300.5, -129.7, -300.2, 1e10, -1e10, np.nan, np.inf], dtype=np.float32)
    np.testing.assert_array_equal(O.cast_f32_to_int8(v), np.array([1, -1, 127, -128, -56, 44, 127, -44, 0, 0, 0, 0], np.int8))


# ---- the C restatement (oracle/oracle.c), pinned directly: the GPU parity tests use it as their checker ----

from oracle import oracle_c as OC


@pytest.mark.parametrize("case", [c for c in QW.cases() if kind_of(c) == "f32"])
def test_c_oracle_quantizer_matches_reference(case):
    """miq_oracle_quantize_rows_sym_i8 against the reference's own outputs (fp32 cases; the C function takes fp32)."""
    w = QW.get(case, "w")
    q, s, zp = OC.quantize_rows_sym_i8(w)
    np.testing.assert_array_equal(q, QW.get(case, "q"))
    np.testing.assert_array_equal(s, QW.get(case, "scales").reshape(-1))
    np.testing.assert_array_equal(zp, QW.get(case, "zp").reshape(-1))
    assert q.dtype == np.int8 and s.dtype == np.float32 and zp.dtype == np.int64


@pytest.mark.parametrize("case", FW.cases())
def test_c_oracle_forward_matches_reference(case):
    """linear_i8_nt / linear_w8a8 / conv2d_i8 + epilogue_f32 of oracle.c on the reference's weight-only forward
    fixtures: bit-equal to the numpy restatement, and to the reference's float output at 1e-6."""
    x, q, s, bias = FW.get(case, "x"), FW.get(case, "q"), FW.get(case, "scales"), FW.get(case, "bias")
    xi = x.astype(np.int8)
    np.testing.assert_array_equal(xi.astype(x.dtype), x)      # the fixtures' activations are integer-valued
    if case.startswith("lin"):
        acc = OC.linear_i8_nt(xi, q)
        np.testing.assert_array_equal(acc, OC.linear_w8a8(xi, np.ascontiguousarray(q.T)))
        np.testing.assert_array_equal(acc, O.linear_w8a8(xi, np.ascontiguousarray(q.T)))
        y_np = O.wo_linear_forward(x, q, s, bias)
    else:
        stride, pad = (int(v) for v in FW.get(case, "stride_pad"))
        acc = OC.conv2d_i8(xi, q, stride, pad)
        np.testing.assert_array_equal(acc, O.conv2d_int8(xi, q, stride, pad))
        y_np = O.wo_conv2d_forward(x, q, s, bias, stride, pad)
    y = OC.epilogue_f32(acc, bias, s, 0.0)
    np.testing.assert_array_equal(y, O.epilogue(acc, bias, s, 0.0))
    np.testing.assert_array_equal(y, y_np)
    ref = FW.get(case, "y")
    np.testing.assert_allclose(y, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())


def test_c_oracle_wa_epilogue_matches_reference():
    """epilogue_f32 with the weight+activation parameters (bias', scalers, output zero point) of the reference's
    calibrated SimpleCNN: bit-equal to the numpy epilogue, 2e-6 from the reference's float output."""
    for name, conv in (("conv1", True), ("fc2", False)):
        x, wq = WA.get(name, "x"), WA.get(name, "weight")
        bias, scalers = WA.get(name, "bias"), WA.get(name, "scalers")
        ozp = float(WA.get(name, "output_zero_point")[0])
        acc = OC.conv2d_i8(x, wq, 1, 1) if conv else OC.linear_i8_nt(x, wq)
        np.testing.assert_array_equal(acc, O.conv2d_int8(x, wq, 1, 1) if conv else O.linear_w8a8(x, np.ascontiguousarray(wq.T)))
        y = OC.epilogue_f32(acc, bias, scalers, ozp)
        np.testing.assert_array_equal(y, O.epilogue(acc, bias, scalers, ozp))
        ref = WA.get(name, "y")
        np.testing.assert_allclose(y, ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())


def test_c_oracle_against_numpy_on_random_inputs():
    rng = np.random.default_rng(7)
    for shape in ((5, 33), (64, 27), (3, 4, 3, 3)):
        w = (rng.standard_normal(shape) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float32)
        q, s, zp = OC.quantize_rows_sym_i8(w)
        q2, s2, zp2 = O.quantize_channel_sym_int8(w)
        np.testing.assert_array_equal(q, q2)
        np.testing.assert_array_equal(s, s2.reshape(-1))
        np.testing.assert_array_equal(zp, zp2.reshape(-1))
    for (B, C, H, W, Co, k, st, p) in [(2, 3, 9, 7, 4, 3, 1, 0), (1, 5, 8, 8, 6, 3, 2, 1), (1, 3, 12, 12, 2, 7, 2, 3)]:
        x = rng.integers(-128, 128, (B, C, H, W), dtype=np.int8)
        w = rng.integers(-128, 128, (Co, C, k, k), dtype=np.int8)
        np.testing.assert_array_equal(OC.conv2d_i8(x, w, st, p), O.conv2d_int8(x, w, st, p))
    a = rng.integers(-128, 128, (37, 96), dtype=np.int8)
    b = rng.integers(-128, 128, (96, 53), dtype=np.int8)
    np.testing.assert_array_equal(OC.linear_w8a8(a, b), a.astype(np.int64) @ b.astype(np.int64))
    np.testing.assert_array_equal(OC.linear_i8_nt(a, np.ascontiguousarray(b.T)), a.astype(np.int64) @ b.astype(np.int64))
