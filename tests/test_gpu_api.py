"""The reference's Python surface (Qop / Quantizer / Chunker / linear_w8a8 / conv2d_int8) on the
GPU, against the golden fixtures the reference produced and the CPU oracle.  The tests read
like the reference's own usage: build a layer, call from_float, call forward."""
import hashlib

import numpy as np
import pytest
import torch

import sample_models
from conftest import Golden, kind_of
from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

QW = Golden("quantizer_wo.npz")
QA = Golden("quantizer_wa.npz")
FW = Golden("wo_forward.npz")
WA = Golden("wa_chunker.npz")
C1 = Golden("config1_models.npz")
TORCH_KIND = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
Q_PARAMS = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def wo():
    import quantization_amd.quant.Weight_Only as m
    return m


@pytest.fixture(scope="module")
def wa():
    import quantization_amd.quant.Weight_Activation as m
    return m


@pytest.mark.parametrize("pkg,fix,case", [("wo", QW, c) for c in QW.cases()] + [("wa", QA, c) for c in QA.cases()],
                         ids=lambda v: v if isinstance(v, str) else "")
def test_qop_quantize_golden(wo, wa, pkg, fix, case):
    Qop = (wo if pkg == "wo" else wa).Qop
    w = torch.from_numpy(fix.get(case, "w")).to(TORCH_KIND[kind_of(case)]).cuda()
    q = Qop(dtype=torch.int8, symmetric=True, affine="channel", affine_dim=0)
    qt = q.quantize(w)
    assert qt.dtype == torch.int8 and qt.is_cuda and qt.shape == w.shape
    np.testing.assert_array_equal(qt.cpu().numpy(), fix.get(case, "q"))
    assert q.scales.dtype == w.dtype and q.zero_point.dtype == torch.int64
    assert tuple(q.scales.shape) == fix.get(case, "scales").shape == tuple(q.zero_point.shape)
    np.testing.assert_array_equal(q.scales.float().cpu().numpy(), fix.get(case, "scales"))
    np.testing.assert_array_equal(q.zero_point.cpu().numpy(), fix.get(case, "zp"))
    # attributes the reference leaves behind
    assert q.tensor is w and q.tensor_shape == w.shape
    np.testing.assert_array_equal(q.max_val.float().numpy(), O.row_signed_max(fix.get(case, "w")))
    assert torch.equal(q.quantized_tensor, qt.to(w.dtype))


def test_qop_host_tensor_is_staged(wo):
    w = torch.from_numpy(QW.get("f32_10x512", "w"))
    q = wo.Qop(dtype=torch.int8, symmetric=True, affine="channel", affine_dim=0)
    qt = q.quantize(w)
    assert not qt.is_cuda and not q.scales.is_cuda
    np.testing.assert_array_equal(qt.numpy(), QW.get("f32_10x512", "q"))


def test_qop_second_quantize_reuses_scales(wo):
    """The reference computes parameters only once per Qop (Qop.py:144)."""
    g = torch.Generator().manual_seed(2)
    w1, w2 = torch.randn(6, 40, generator=g).cuda(), torch.randn(6, 40, generator=g).cuda() * 3
    q = wo.Qop(dtype=torch.int8, symmetric=True, affine="channel", affine_dim=0)
    q.quantize(w1)
    s1 = q.scales.clone()
    q2 = q.quantize(w2)
    assert torch.equal(q.scales, s1)
    ref = np.clip(np.rint(w2.cpu().numpy() / s1.cpu().numpy()), -128, 127).astype(np.int8)
    np.testing.assert_array_equal(q2.cpu().numpy(), ref)


def test_qop_dequantize_and_error(wo):
    w = torch.from_numpy(QW.get("f32_64x27", "w")).cuda()
    q = wo.Qop(dtype=torch.int8, symmetric=True, affine="channel", affine_dim=0)
    qt = q.quantize(w)
    d = q.dequantize(qt)
    ref = O.dequantize(QW.get("f32_64x27", "q"), QW.get("f32_64x27", "scales"), 0)
    np.testing.assert_array_equal(d.cpu().numpy(), ref)
    err = q.compute_dequantization_error(w.clone(), d.clone())
    assert float(err) >= 0
    # activation=True re-views the parameters [1,N] and scales a [M,N] float tensor per column
    y = torch.randn(5, 64, generator=torch.Generator().manual_seed(1)).cuda()
    dy = q.dequantize(y, activation=True)
    assert tuple(q.scales.shape) == (1, 64)
    np.testing.assert_array_equal(dy.cpu().numpy(), QW.get("f32_64x27", "scales").reshape(1, -1) * y.cpu().numpy())


def test_qop_unsupported_configs_fail_like_the_reference(wo):
    w = torch.randn(4, 8).cuda()
    with pytest.raises(TypeError):  # per-tensor: tensor.shape[None] (Qop.py:110)
        wo.Qop(dtype=torch.int8, symmetric=True, affine="tensor", affine_dim=None).quantize(w)
    with pytest.raises(RuntimeError, match="NoneType"):  # asymmetric per-channel (Qop.py:92)
        wo.Qop(dtype=torch.int8, symmetric=False, affine="channel", affine_dim=0).quantize(w)


def test_qop_calculate_params_affine_tensor(wa):
    q = wa.Qop(dtype=torch.int8, symmetric=False, affine="tensor", affine_dim=None)
    calib = torch.from_numpy(WA.raw("calib_x"))
    q.min_val, q.max_val = calib.min(), calib.max()
    s, zp = q.calculate_params()
    rs, rzp = O.calculate_params_affine_tensor(calib.min().item(), calib.max().item())
    np.testing.assert_array_equal(s.numpy(), rs)
    np.testing.assert_array_equal(zp.numpy(), rzp)
    assert zp.dtype == torch.int32


# ------------------------------------------------------------------------------ layers
def make_module(case):
    w, bias = FW.get(case, "w"), FW.get(case, "bias")
    if case.startswith("lin"):
        m = torch.nn.Linear(w.shape[1], w.shape[0], bias=bias is not None)
    else:
        stride, pad = (int(v) for v in FW.get(case, "stride_pad"))
        m = torch.nn.Conv2d(w.shape[1], w.shape[0], w.shape[2], stride=stride, padding=pad, bias=bias is not None)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w))
        if bias is not None:
            m.bias.copy_(torch.from_numpy(bias))
    return m.cuda()


@pytest.mark.parametrize("case", FW.cases())
def test_wo_quantizer_from_float_and_forward(wo, case):
    ql = wo.Quantizer.from_float(module=make_module(case), data_metry=Q_PARAMS, quantize_output=False)
    assert repr(ql) == FW.text(case, "repr")
    assert ql.weight.dtype == torch.int8
    np.testing.assert_array_equal(ql.weight.cpu().numpy(), FW.get(case, "q"))
    np.testing.assert_array_equal(ql.weight_quant.scales.cpu().numpy().reshape(-1), FW.get(case, "scales"))
    x = torch.from_numpy(FW.get(case, "x")).cuda()
    ref = FW.get(case, "y")
    tol = dict(rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    y_int = ql(x)                       # int8 activations: MFMA path
    assert y_int.dtype == torch.float32 and y_int.shape == ref.shape
    np.testing.assert_allclose(y_int.cpu().numpy(), ref, **tol)
    y_flt = ql(x.float())               # float activations: what the reference is fed
    np.testing.assert_allclose(y_flt.cpu().numpy(), ref, **tol)
    # after a forward the parameters are viewed [1,N,...] as the reference leaves them
    assert ql.weight_quant.scales.shape[0] == 1 and ql.weight_quant.scales.shape[1] == ref.shape[1]


def test_wo_quantizer_to_moves_weight_and_bias(wo):
    ql = wo.Quantizer.from_float(module=make_module("lin_6_4"), data_metry=Q_PARAMS)
    ql.to("cpu")
    assert not ql.weight.is_cuda and not ql.bias.is_cuda
    y = ql(torch.from_numpy(FW.get("lin_6_4", "x")))   # host tensors are staged through the GPU
    assert not y.is_cuda
    np.testing.assert_allclose(y.numpy(), FW.get("lin_6_4", "y"), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("model", ["SimpleCNN", "LinearNet", "VGG"])
def test_wo_chunker_config1_models(wo, model):
    """BASELINE config 1: the sample models, seeded default init, weight-only Chunker.  Every
    layer's int8 weight and scales must hash to what the reference produced."""
    net = sample_models.build(model)
    names = C1.text(model, "layers").split(",")
    mods = dict(net.named_modules())
    for n in names:
        if sha(mods[n].weight.detach().numpy()) != C1.text(f"{model}/{n}", "sha_w"):
            pytest.skip("default initialisation differs from the run that made the fixture")
    net = net.cuda()
    x = sample_models.loader(1, 2, 11)[0][0].cuda()
    fp_out = net(x)
    qnet = wo.Chunker(net, sample_models.loader(1, 2, 11)).model
    qmods = dict(qnet.named_modules())
    for n in names:
        layer = qmods[n]
        assert type(layer).__name__ == "Quantizer"
        assert sha(layer.weight.cpu().numpy()) == C1.text(f"{model}/{n}", "sha_q"), n
        assert sha(layer.weight_quant.scales.cpu().numpy()) == C1.text(f"{model}/{n}", "sha_scales"), n
    # The reference flow end to end (quant/Weight_Only/Main.py:40-42: Chunker(model).model, then a forward): the
    # rewritten model's output on the seeded batch against the reference's own, within 1e-3 of the output's
    # magnitude (the reference sums in fp32 in a different order; here fp32-accurate MFMA products).
    np.testing.assert_allclose(fp_out.detach().cpu().numpy(), C1.raw(f"{model}/y_fp"), rtol=1e-4, atol=1e-5)
    q_out = qnet(x).detach().cpu().numpy()
    ref = C1.raw(f"{model}/y_quant")
    assert q_out.shape == ref.shape
    assert np.abs(q_out - ref).max() <= 1e-3 * np.abs(ref).max(), (np.abs(q_out - ref).max(), np.abs(ref).max())


def test_wa_chunker_simplecnn_golden(wa):
    net = sample_models.build("SimpleCNN")
    for n, p in net.named_parameters():
        if sha(p.detach().numpy()) != WA.text("fp_sha", n):
            pytest.skip("default initialisation differs from the run that made the fixture")
    data = sample_models.loader(2, 4, 5)
    np.testing.assert_array_equal(np.stack([x.numpy() for x, _ in data]), WA.raw("calib_x"))
    qnet = wa.Chunker(net.cuda(), data).model
    for name in ("conv1", "conv2", "fc1", "fc2"):
        layer = getattr(qnet, name)
        assert sha(layer.weight.cpu().numpy()) == WA.text(name, "sha_weight")
        np.testing.assert_array_equal(layer.weight_qscale.cpu().numpy().reshape(-1), WA.get(name, "weight_qscale"))
        for field in ("bias", "scalers", "output_zero_point"):
            got = getattr(layer, field).detach().float().cpu().numpy().reshape(-1)
            np.testing.assert_allclose(got, WA.get(name, field), rtol=1e-5, atol=1e-6, err_msg=f"{name}.{field}")
        assert abs(layer.input_qscale - float(WA.get(name, "input_qscale")[0])) <= 1e-6 * layer.input_qscale
    for name in ("conv1", "fc2"):
        x = torch.from_numpy(WA.get(name, "x")).cuda()
        ref = WA.get(name, "y")
        tol = dict(rtol=1e-3, atol=1e-3 * np.abs(ref).max())
        np.testing.assert_allclose(getattr(qnet, name)(x).cpu().numpy(), ref, **tol)
        np.testing.assert_allclose(getattr(qnet, name)(x.float()).cpu().numpy(), ref, **tol)


# ------------------------------------------------------------------- native entry points
def test_linear_w8a8_reference_script():
    """cuda_kernels/linear_int8/test.py, on its own shapes (randint's upper bound is exclusive there)."""
    from quantization_amd import linear_w8a8
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-128, 127, (1024, 2048), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    b = torch.randint(-128, 127, (2048, 4096), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    out = linear_w8a8.linear_w8a8(a, b)
    assert out.dtype == torch.int32 and out.shape == (1024, 4096)
    ref = torch.nn.functional.linear(a.double(), b.double().T)
    assert torch.allclose(out.double(), ref, atol=1e-3)
    assert torch.equal(out, ref.to(torch.int32))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        linear_w8a8.linear_w8a8(a.cpu(), b)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        linear_w8a8.linear_w8a8(a, b.t().contiguous().t())


def test_conv2d_int8_reference_script():
    """cuda_kernels/conv2d_int8/test.py at a smaller batch: values in [0,7), 3->4 channels, 3x3."""
    from quantization_amd import conv2d_int8
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 7, (4, 3, 512, 512), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    k = torch.randint(0, 7, (4, 3, 3, 3), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert conv2d_int8.conv2d_int8(x, k, out, 1, 0) is None
    assert out.shape == (4, 4, 510, 510)
    ref = torch.nn.functional.conv2d(x.float(), k.float(), stride=1, padding=0)
    assert torch.allclose(out.float(), ref, atol=1e-2)
    out2 = torch.zeros(1, dtype=torch.int32, device="cuda")
    conv2d_int8.conv2d_int8(x, k, out2, 2, 1)            # the stride is honoured (F.conv2d semantics)
    ref2 = torch.nn.functional.conv2d(x.float(), k.float(), stride=2, padding=1)
    assert torch.equal(out2, ref2.to(torch.int32))


# ---- persisted quantization parameters ------------------------------------------------------------
@pytest.mark.gpu
def test_quantized_state_dict_round_trip_weight_only(tmp_path):
    """state_dict of a weight-only quantized model carries the scales (as each layer's extra state), survives
    torch.save / torch.load(weights_only=True), and a freshly built model that loads it reproduces the
    outputs bit for bit; a state_dict without the extra state (what the reference writes) still loads, but the
    layers then refuse to run with scales that belong to other weights until the parameters are supplied."""
    import io
    from quantization_amd.quant.Weight_Only.Chunker import Chunker
    from sample_models import SimpleCNN
    torch.manual_seed(0)
    m1 = SimpleCNN().eval()
    Chunker(m1, None)
    sd = m1.state_dict()
    assert any(k.endswith("_extra_state") for k in sd)
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    sd2 = torch.load(buf, weights_only=True)
    torch.manual_seed(123)                       # different float weights: everything must come from the file
    m2 = SimpleCNN().eval()
    Chunker(m2, None)
    m2.load_state_dict(sd2)
    x = torch.rand(4, 3, 32, 32)
    assert torch.equal(m1(x), m2(x))
    for (n1, a), (n2, b) in zip(m1.named_modules(), m2.named_modules()):
        if hasattr(a, "weight_quant"):
            assert torch.equal(a.weight, b.weight)
            assert torch.equal(a.weight_quant.scales.reshape(-1).cpu(), b.weight_quant.scales.reshape(-1).cpu())
    legacy = {k: v for k, v in sd2.items() if not k.endswith("_extra_state")}
    torch.manual_seed(321)
    m3 = SimpleCNN().eval()
    Chunker(m3, None)
    with pytest.warns(UserWarning, match="no quantization parameters"):
        m3.load_state_dict(legacy)               # strict: the missing extra state is tolerated ...
    with pytest.raises(RuntimeError, match="no quantization parameters"):
        m3(x)                                    # ... but stale scales are never applied to the loaded weights
    for (n1, a), (n3, c) in zip(m1.named_modules(), m3.named_modules()):
        if hasattr(a, "weight_quant"):
            c.set_extra_state(a.get_extra_state())
    assert torch.equal(m1(x), m3(x))


@pytest.mark.gpu
def test_quantized_state_dict_round_trip_weight_activation():
    import io
    from quantization_amd.quant.Weight_Activation.Chunker import Chunker
    from sample_models import SimpleCNN

    def calib(seed):
        g = torch.Generator().manual_seed(seed)
        return [(torch.rand(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(2)]
    torch.manual_seed(0)
    m1 = SimpleCNN().eval()
    Chunker(m1, calib(1))
    buf = io.BytesIO()
    torch.save(m1.state_dict(), buf)
    buf.seek(0)
    torch.manual_seed(5)
    m2 = SimpleCNN().eval()
    Chunker(m2, calib(2))
    m2.load_state_dict(torch.load(buf, weights_only=True))
    x = torch.rand(4, 3, 32, 32)
    y1, y2 = m1(x), m2(x)
    assert y1.dtype == y2.dtype and torch.equal(y1, y2)


@pytest.mark.gpu
def test_layer_forward_enqueues_only_and_captures_into_a_graph(wa, wo):
    """A weight+activation layer whose calibration results live on the GPU, and a weight-only layer kept on the
    host side of a GPU box (module.cpu(), activations on the GPU): after the first call neither forward reads a
    device value back nor re-uploads its weight, so both capture into a HIP graph (a synchronising call fails
    the capture) and replay to the eager results."""
    dm = {"weights": {"dtype": torch.int8, "symmetric": True, "affine": "channel", "affine_dim": 0}}
    torch.manual_seed(3)
    lin = torch.nn.Linear(256, 96)
    qa = wa.Quantizer.from_float(module=lin, data_metry=dm)
    qa.scalers = (torch.rand(1, 96) * 1e-2 + 1e-4).cuda()
    qa.output_zero_point = torch.tensor([3.5]).cuda()
    qa.bias = qa.bias.detach().cuda()
    qa = qa.cuda()
    qo = wo.Quantizer.from_float(module=torch.nn.Linear(256, 64), data_metry=dm)   # weight, bias, scales stay host tensors
    x = torch.randint(-128, 128, (16, 256), dtype=torch.int32).to(torch.int8).cuda()
    xf = torch.randn(16, 256).cuda()
    ya, yo = qa(x), qo(xf)                                   # first calls: the one read of output_zero_point, the uploads
    uploaded = qo._gpu_cache["weight"][2]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        qa(x), qo(xf)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ga, go = qa(x), qo(xf)
    assert qo._gpu_cache["weight"][2] is uploaded            # no second upload
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(ga, ya) and torch.equal(go, yo)
    ref = (x.double() @ qa.weight.double().T + qa.bias.double().reshape(1, -1)) * qa.scalers.double() + 3.5
    assert torch.allclose(ya.double(), ref, rtol=1e-5, atol=1e-5)
    # replacing the zero point is seen (the cache is keyed on the tensor, not on the layer)
    qa.output_zero_point = torch.tensor([-1.0]).cuda()
    assert torch.allclose(qa(x).double(), ref - 4.5, rtol=1e-5, atol=1e-5)


class _GatedMLP(torch.nn.Module):
    def __init__(self, d, h, bias):
        super().__init__()
        self.gate_proj = torch.nn.Linear(d, h, bias=bias)
        self.up_proj = torch.nn.Linear(d, h + 64, bias=False)
        self.down_proj = torch.nn.Linear(h, d, bias=bias)

    def forward(self, x):
        g, u = self.gate_proj(x), self.up_proj(x)
        return self.down_proj(torch.nn.functional.silu(g.float()) * u[..., :g.shape[-1]].float()), g, u


@pytest.mark.parametrize("kind", ["int8", "float32", "bfloat16"])
def test_sibling_layers_fused_into_one_product(kind):
    """Chunker.fuse_siblings(): the gate and up projections of an MLP (called on the same tensor) become one stored weight
    and one product; each layer's output equals the unfused layer's bit for bit, which for int8 activations is the
    oracle's (exact int32 product + the reference's epilogue order).  The layers keep their interface, state_dict keys and
    values; a member moved off the fused weight runs on its own again."""
    from quantization_amd.qlayers import SiblingGroup
    from quantization_amd.quant.Weight_Only.Chunker import Chunker
    torch.manual_seed(11)
    model = _GatedMLP(256, 320, bias=True).cuda()
    ck = Chunker(model, None)
    g = torch.Generator().manual_seed(5)
    if kind == "int8":
        x = torch.randint(-128, 128, (3, 5, 256), generator=g, dtype=torch.int32).to(torch.int8).cuda()
    else:
        x = torch.randn(3, 5, 256, generator=g).to(getattr(torch, kind)).cuda()
    y_ref, g_ref, u_ref = model(x if kind != "int8" else x)
    sd_ref = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    groups = ck.fuse_siblings(x)
    assert len(groups) == 1 and isinstance(groups[0], SiblingGroup)
    assert {id(m) for m in groups[0].layers} == {id(model.gate_proj), id(model.up_proj)}
    assert model.gate_proj.weight.untyped_storage().data_ptr() == model.up_proj.weight.untyped_storage().data_ptr()
    y, gq, uq = model(x)
    assert torch.equal(gq, g_ref) and torch.equal(uq, u_ref) and torch.equal(y, y_ref)
    assert tuple(gq.shape) == (3, 5, 320) and tuple(uq.shape) == (3, 5, 384)
    if kind == "int8":
        from oracle import oracle_np as O
        wq, s = model.gate_proj.weight.cpu().numpy(), model.gate_proj.weight_quant.scales.reshape(-1).cpu().numpy()
        ref = O.wo_linear_forward(x.reshape(-1, 256).cpu().numpy(), wq, s, model.gate_proj.bias.detach().cpu().numpy())
        np.testing.assert_array_equal(gq.reshape(-1, 320).cpu().numpy(), ref)
    # a second activation: nothing stale is served
    x2 = x.clone()
    x2[0, 0, :8] = 1
    g2, u2 = model.gate_proj(x2), model.up_proj(x2)
    assert not torch.equal(g2, gq) and torch.equal(model.gate_proj(x), gq) and torch.equal(model.up_proj(x), uq)
    assert torch.equal(u2, model.up_proj(x2))
    # the same state_dict as before fusing
    sd = model.state_dict()
    for k, v in sd_ref.items():
        assert torch.equal(sd[k], v), k
    # a member that leaves the fused weight (here: re-created buffers) runs on its own, same result
    model.up_proj.weight = model.up_proj.weight.clone()
    assert not groups[0].usable(x)
    assert torch.equal(model.up_proj(x), uq) and torch.equal(model.gate_proj(x), gq)
