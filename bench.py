#!/usr/bin/env python3
"""Headline benchmark: int8 Linear (int8 x int8 -> int32 MFMA GEMM + fused per-channel dequant
epilogue) at M = N = K = 8192 per GPU, the shape BASELINE.json quotes its target on.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one forward of the layer on int8 activations already resident in HBM, with weights
produced by the per-channel quantizer from seeded fp32 data.  With N > 1 each rank owns 8192
output channels of a [8192*N, 8192] layer (weak scaling); a step computes the rank's [8192, 8192]
block straight into its slot of the gather buffer and all-gathers the blocks in place over RCCL, so
every rank ends with all N blocks; the gather of one step overlaps the GEMM of the next (independent
layers of a stack).  What travels is chosen by MIQ_BENCH_GATHER = f32 (default at every N: bit-identical
to one GPU), bf16 (the epilogue rounds once to bf16, half the bytes) or i8 (requantised in the epilogue
with a static activation scale, a quarter); `gather_variants` in bench_extra.json times all three.

Rank 0 prints ONE short JSON line, last.  `value` is whole-job int8 TOPS (2*M*N*K integer operations per
layer).  `roofline` prices the GEMM kernel against the dense int8 MFMA peak with the kernel's own
duration measured by HIP events on the launch stream, and carries the other half of the metric as
`roofline.quantizer` (the quantizer against HBM bandwidth) plus one number per other configuration
(`roofline.others`).  The full tables of the other configurations of BASELINE.json, each with its own
roofline -- the quantizer shapes, the Llama-3-8B MLP at one token (config 4), ResNet-18's convolutions
at batch 256 with a fraction per layer (config 3), and `config5`: the Llama-3-70B shapes [28672,8192],
[8192,28672], [8192,8192] split over the N ranks by output channel (strong scaling) at M = 256 / 2048 /
8192 with the GEMM / all-gather breakdown -- go to the file bench_extra.json (under gpurun_out/ when
that directory exists), not to stdout.  `cpu_baseline` is the reference's CPU path (restated in
oracle/oracle_torch.py, pinned to the reference by fixtures) on a bounded sample of the workload.

The output type of the headline step is fp32 at every --gpus N (MIQ_BENCH_GATHER=bf16|i8 overrides it).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5000.0   # dense int8 MFMA, 2x the ~2.5 PFLOP/s bf16 dense peak of MI355X
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0     # HBM3E spec
EVENT_EVERY = int(os.environ.get("MIQ_BENCH_EVENT_EVERY", "4"))   # the timed region's GEMM launches that carry a HIP event pair

PREWARM_S = 0.3   # seconds of untimed load before the warm-up steps (clock ramp), reported as prewarm_s

GATHER_DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "i8": torch.int8}
ACT_SCALE = 16.0   # static activation scale of the int8 gather (any positive value costs the same)


def event_times(fn, iters):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in evs]


def empty_pair_ms(n=20):
    """What a HIP event pair with nothing between its records measures (median of n): each record is a queue packet of its
    own, and a pair's interval contains it.  Subtracted from the per-launch pairs of the headline GEMM (5 us of 390), whose
    result is also capped by the region's time per step."""
    pairs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in pairs)[n // 2]


def median_ms(fn, iters=30, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = sorted(event_times(fn, iters))
    return ts[len(ts) // 2]


def graphed_ms(step, steps, reps=20, warm=20):
    """Milliseconds per step of `steps` consecutive steps captured once into a HIP graph and replayed: device time of
    launch-bound sequences (a decode step is three 12-microsecond kernels; enqueued call by call from Python, the
    interval between two events measures the host).  Median over `reps` replays."""
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    try:
        graph = torch.cuda.CUDAGraph()
        # thread-local capture: a communicator's watchdog thread (multi-rank runs) may touch the device meanwhile
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(steps):
                step()
        graph.replay()
        torch.cuda.synchronize()
        ts = sorted(event_times(graph.replay, reps))
        return ts[len(ts) // 2] / steps
    except RuntimeError as e:   # no capture on this run: the call-by-call time stands in (it is reported beside this one)
        print(f"[bench] graph capture failed ({str(e).splitlines()[0]}); timing call by call", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        return median_ms(step, 50, 10)


def kernel_source_sha():
    """Digest of the GEMM kernel's sources: counter results under profiles/ are stamped with it."""
    h = hashlib.sha256()
    for f in ("miq_gemm_i8.hip", "miq_gemm_core.h"):
        with open(os.path.join(ROOT, "quantization_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r3_traffic.json,
    written by tools/profile_round.sh), or None when that file was measured on other kernel sources than these."""
    try:
        with open(os.path.join(ROOT, "profiles", "r3_traffic.json")) as f:
            d = json.load(f)
        if d.get("source_sha") != kernel_source_sha() and kernel.startswith("gemm"):
            return None
        return d[kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def hbm_roofline(nbytes, ms):
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes": int(nbytes)}


def mfma_roofline(ops, ms, peak=PEAK_INT8_TOPS):
    t = ops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(t, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(t / peak, 4)}


def weight_only_float(N_, dev):
    """The weight-only Linear on floating-point activations (miq_gemm_w8) on the Llama-3-8B gate/up weight
    [14336,4096]: whole call (the activation-plane pre-pass of fp32 inputs included), HIP events per call.
    Decode sizes are priced against HBM bandwidth on their algorithmic bytes (int8 weight once + activations +
    fp32 output); M = 256 against the dense bf16 MFMA peak on 2*M*N*K."""
    Nn, K = 14336, 4096
    # ten copies (587 MB) taken in turn, as the layers of a model are: no call finds its weight in the 256 MiB
    # Infinity Cache, so the decode sizes are priced on bytes that do come from HBM
    ws = [torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device=dev).to(torch.int8) for _ in range(10)]
    alpha = torch.rand(Nn, device=dev) * 0.01
    res = {"weight": f"int8 [{Nn},{K}], 10 copies in turn", "out_dtype": "float32",
           "timing": "HIP graph of 40 calls, replayed (call_ms); call by call from Python beside it"}
    for M, dt in ((16, torch.bfloat16), (16, torch.float16), (16, torch.float32), (256, torch.bfloat16), (256, torch.float16),
                  (256, torch.float32)):
        x = torch.randn(M, K, device=dev).to(dt)
        out = torch.empty(M, Nn, device=dev)
        it = [0]

        def call():
            it[0] += 1
            N_.gemm_w8(x, ws[it[0] % 10], alpha=alpha, out=out)
        eager = median_ms(call, 50, 50)
        ms = graphed_ms(call, 40)   # forty calls = four passes over the ten weights, as one graph
        nbytes = Nn * K + M * K * x.element_size() + M * Nn * 4
        entry = {"call_ms": round(ms, 4), "call_ms_enqueued_from_python": round(eager, 4),
                 "weight_GBps": round(Nn * K / (ms * 1e-3) / 1e9, 1),
                 "TFLOPs": round(2.0 * M * Nn * K / (ms * 1e-3) / 1e12, 1)}
        if M <= 16:
            entry["roofline"] = hbm_roofline(nbytes, ms)
        res[f"M{M}_{str(dt).split('.')[-1]}"] = entry
    return res


def llama8b_mlp_decode(N_, dev):
    """BASELINE config 4 at one token: the three int8 matmuls of a Llama-3-8B MLP (gate, up [14336,4096]; down
    [4096,14336]) with the fused dequant epilogue, M = 1 and M = 16, fp32 out.  A step is bound by reading the
    three int8 weights once: algorithmic bytes = 3 * 14336 * 4096 + activations + outputs.  `fused_gate_up` is the
    same step with gate and up stored as one [28672,4096] weight (they multiply the same activations): two launches."""
    layers = 4   # four layers' weights (705 MB) taken in turn: every step streams its weights from HBM, not from the Infinity Cache
    ws = []
    for _ in range(layers):
        gu = torch.randint(-128, 128, (28672, 4096), dtype=torch.int32, device=dev).to(torch.int8)
        ws.append({"gate_up": gu, "gate": gu[:14336], "up": gu[14336:], "a_gu": torch.rand(28672, device=dev) * 0.01,
                   "down": torch.randint(-128, 128, (4096, 14336), dtype=torch.int32, device=dev).to(torch.int8),
                   "a_down": torch.rand(4096, device=dev) * 0.01})
    res = {"weights": f"int8 gate/up [14336,4096], down [4096,14336], {layers} layers in turn", "out_dtype": "float32",
           "timing": f"HIP graph of {5 * layers} steps, replayed (step_ms); call by call from Python beside it"}
    for M in (1, 16):
        xa = torch.randint(-128, 128, (M, 4096), dtype=torch.int32, device=dev).to(torch.int8)
        xd = torch.randint(-128, 128, (M, 14336), dtype=torch.int32, device=dev).to(torch.int8)
        o_gu, o_down = torch.empty(M, 28672, device=dev), torch.empty(M, 4096, device=dev)
        o_g, o_u = torch.empty(M, 14336, device=dev), torch.empty(M, 14336, device=dev)
        it = [0]

        def step():
            it[0] += 1
            w = ws[it[0] % layers]
            N_.gemm_i8(xa, w["gate"], alpha=w["a_gu"][:14336], out=o_g)
            N_.gemm_i8(xa, w["up"], alpha=w["a_gu"][14336:], out=o_u)
            N_.gemm_i8(xd, w["down"], alpha=w["a_down"], out=o_down)

        def step_fused():
            it[0] += 1
            w = ws[it[0] % layers]
            N_.gemm_i8(xa, w["gate_up"], alpha=w["a_gu"], out=o_gu)
            N_.gemm_i8(xd, w["down"], alpha=w["a_down"], out=o_down)
        eager = median_ms(step, 50, 50)
        ms = graphed_ms(step, 5 * layers)   # five passes over the layers' weights as one graph
        fused = graphed_ms(step_fused, 5 * layers)
        nbytes = 3 * 14336 * 4096 + M * (2 * 4096 + 14336) + 4 * M * (2 * 14336 + 4096)
        res[f"M{M}"] = {"step_ms": round(ms, 4), "step_ms_enqueued_from_python": round(eager, 4),
                        "weight_GBps": round(3 * 14336 * 4096 / (ms * 1e-3) / 1e9, 1), "roofline": hbm_roofline(nbytes, ms),
                        "fused": {"step_ms": round(fused, 4), "roofline": hbm_roofline(nbytes - M * 4096, fused)}}
    return res


def gemm_other_shapes(N_, dev):
    """The int8 GEMM beside the headline case: BASELINE config 2 (M = N = K = 4096) with fp32 and bf16 output, and the
    headline shape with bf16 output; kernel time by HIP events per launch after a clock ramp, each against the int8 peak."""
    res = {}
    for S, dt in ((4096, torch.float32), (4096, torch.bfloat16), (8192, torch.bfloat16)):
        x = torch.randint(-128, 128, (S, S), dtype=torch.int32, device=dev).to(torch.int8)
        w = torch.randint(-128, 128, (S, S), dtype=torch.int32, device=dev).to(torch.int8)
        alpha = torch.rand(S, device=dev) * 1e-3
        bias = torch.randn(S, device=dev)
        out = torch.empty(S, S, dtype=dt, device=dev)

        def step():
            N_.gemm_i8(x, w, bias=bias, alpha=alpha, out_dtype=dt, out=out)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:   # clock ramp
            for _ in range(10):
                step()
            torch.cuda.synchronize()
        ms = median_ms(step, 50, 10)
        res[f"{S}x{S}x{S}_{str(dt).split('.')[-1]}_out"] = {"kernel_ms": round(ms, 4), "roofline": mfma_roofline(2.0 * S * S * S, ms)}
        del x, w, out
    return res


RESNET18_CONVS = [  # (Cin, H, Cout, k, stride, pad, occurrences in ResNet-18)
    (3, 224, 64, 7, 2, 3, 1), (64, 56, 64, 3, 1, 1, 4), (64, 56, 128, 3, 2, 1, 1), (128, 28, 128, 3, 1, 1, 3),
    (64, 56, 128, 1, 2, 0, 1), (128, 28, 256, 3, 2, 1, 1), (256, 14, 256, 3, 1, 1, 3), (128, 28, 256, 1, 2, 0, 1),
    (256, 14, 512, 3, 2, 1, 1), (512, 7, 512, 3, 1, 1, 3), (256, 14, 512, 1, 2, 0, 1)]


def conv_int8(N_, dev):
    """BASELINE config 3: ResNet-18's twenty convolutions at batch 256 (miq_conv2d_i8, int8 NCHW in, int32 NCHW out):
    whole call (re-layout launch + implicit-GEMM kernel), outputs rotating over three buffers, timed as a replayed HIP graph
    of six calls (the small layers' launches are shorter than their enqueueing from Python; that time is reported beside it).
    Two layers are itemised (the large-image one against HBM bandwidth on int8 input + weight once + int32 output,
    the 14 x 14 one against the int8 MFMA peak); `stack` is the sum over all twenty, against the MFMA peak."""
    B = 256
    res = {"batch": B, "out_dtype": "int32", "timing": "per layer a HIP graph of 6 calls, replayed (call_ms, total_ms); call by call from Python beside it"}
    total_ms, total_ops, eager_ms, roof_ms = 0.0, 0.0, 0.0, 0.0
    per_layer = {}
    for C, H, Co, k, s, p, times in RESNET18_CONVS:
        x = torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
        w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
        Ho = (H + 2 * p - k) // s + 1
        outs = [torch.empty(B, Co, Ho, Ho, dtype=torch.int32, device=dev) for _ in range(3)]
        it = [0]
        pk = N_.conv2d_i8_prepack(w, (B, C, H, H), s, p)   # once per layer, as Quantizer.from_float does

        def step():
            N_.conv2d_i8(x, w, s, p, out_dtype=torch.int32, out=outs[it[0] % 3], packed=pk)
            it[0] += 1
        eager = median_ms(step, 30, 30)
        ms = graphed_ms(step, 6, 10, 6)   # device time: the small layers' two launches are shorter than their enqueueing from Python
        ops = 2.0 * B * Co * Ho * Ho * C * k * k
        total_ms += ms * times
        eager_ms += eager * times
        total_ops += ops * times
        # a layer's own roofline: the longer of its MFMA time and its HBM time (int8 input + weight once + int32 output)
        lay_roof = max(ops / (PEAK_INT8_TOPS * 1e12), (x.numel() + w.numel() + 4 * outs[0].numel()) / (PEAK_HBM_GBS * 1e9)) * 1e3
        roof_ms += lay_roof * times
        per_layer[f"{C}to{Co}_{H}x{H}_k{k}s{s}"] = {"call_ms": round(ms, 4), "roofline_ms": round(lay_roof, 4),
                                                     "frac": round(lay_roof / ms, 3), "x": times}
        if (C, H, Co, k) in ((64, 56, 64, 3), (256, 14, 256, 3)):
            entry = {"call_ms": round(ms, 4), "call_ms_enqueued_from_python": round(eager, 4), "TOPS": round(ops / (ms * 1e-3) / 1e12, 1)}
            if H == 56:
                entry["roofline"] = hbm_roofline(x.numel() + w.numel() + 4 * outs[0].numel(), ms)
            else:
                entry["roofline"] = mfma_roofline(ops, ms)
            res[f"{C}to{Co}_{H}x{H}_k{k}s{s}"] = entry
        del x, w, outs
    res["per_layer"] = per_layer
    res["stack"] = {"layers": 20, "total_ms": round(total_ms, 4), "total_ms_enqueued_from_python": round(eager_ms, 4), "TOPS": round(total_ops / (total_ms * 1e-3) / 1e12, 1),
                    "images_per_s": round(B / (total_ms * 1e-3)), "roofline": mfma_roofline(total_ops, total_ms),
                    "summed_layer_rooflines_ms": round(roof_ms, 4), "frac_of_summed_layer_rooflines": round(roof_ms / total_ms, 4),
                    "filters": "packed once per layer (miq_conv2d_i8_prepack), as Quantizer.from_float does"}
    # The same twenty layers as an int8 pipeline hands them over: channels-last int8 in (border present), int8 out
    # requantised in the epilogue, written with the next layer's border (miq_conv2d_i8_nhwc): one kernel per layer, no
    # re-layout.  The first layer's three input channels are carried as four (a zero channel).
    cl_ms, cl_ops, cl_bytes, cl_eager = 0.0, 0.0, 0.0, 0.0
    for C, H, Co, k, s, p, times in RESNET18_CONVS:
        Cc = 4 if C == 3 else C
        x = torch.zeros(B, H + 2 * p, H + 2 * p, Cc, dtype=torch.int8, device=dev)
        x[:, p:H + p, p:H + p, :C] = torch.randint(-128, 128, (B, H, H, C), dtype=torch.int32, device=dev).to(torch.int8)
        w = torch.zeros(Co, Cc, k, k, dtype=torch.int8, device=dev)
        w[:, :C] = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
        wp = N_.conv2d_i8_pack_weight(w)
        alpha = torch.rand(Co, device=dev) * 1e-4
        Ho = (H + 2 * p - k) // s + 1
        outs = [torch.zeros(B, Ho + 2, Ho + 2, Co, dtype=torch.int8, device=dev) for _ in range(3)]
        it = [0]

        def step_cl():
            N_.conv2d_i8_nhwc(x, wp, w.shape, s, p, alpha=alpha, out_dtype=torch.int8, out_pad=1, out=outs[it[0] % 3])
            it[0] += 1
        cl_eager += median_ms(step_cl, 30, 30) * times
        ms = graphed_ms(step_cl, 6, 10, 6)
        cl_ms += ms * times
        cl_ops += 2.0 * B * Co * Ho * Ho * C * k * k * times
        cl_bytes += (x.numel() + w.numel() + B * Ho * Ho * Co) * times
        del x, w, wp, outs
    res["stack_channels_last_int8"] = {
        "layers": 20, "total_ms": round(cl_ms, 4), "total_ms_enqueued_from_python": round(cl_eager, 4), "TOPS": round(cl_ops / (cl_ms * 1e-3) / 1e12, 1),
        "images_per_s": round(B / (cl_ms * 1e-3)), "roofline": mfma_roofline(cl_ops, cl_ms),
        "hbm_GBps_on_algorithmic_bytes": round(cl_bytes / (cl_ms * 1e-3) / 1e9, 1)}
    return res


def conv_weight_only_float(N_, dev):
    """The same twenty layers as the reference's weight-only Conv2d runs them (quant/Weight_Only/Quantizer.py:72-91): float
    NCHW activations, int8 OIHW weights, fp32 out (miq_conv2d_w8): whole calls (re-layout launch + kernel) as a replayed HIP
    graph of six, outputs rotating over three buffers.  TFLOP/s counts 2 flops per multiply-add of the convolution (the two bf16
    planes of fp32 activations are twice that in MFMA work); roofline against the dense bf16 MFMA peak."""
    B = 256
    res = {"batch": B, "out_dtype": "float32", "timing": "per layer a HIP graph of 6 calls, replayed"}
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        total_ms, total_ops, roof_ms, per_layer = 0.0, 0.0, 0.0, {}
        for C, H, Co, k, s, p, times in RESNET18_CONVS:
            x = torch.randn(B, C, H, H, device=dev).to(dt)
            w = torch.randint(-128, 128, (Co, C, k, k), dtype=torch.int32, device=dev).to(torch.int8)
            alpha = torch.rand(Co, device=dev) * 1e-3
            Ho = (H + 2 * p - k) // s + 1
            outs = [torch.empty(B, Co, Ho, Ho, device=dev) for _ in range(3)]
            it = [0]
            pk = N_.conv2d_w8_prepack(w, x.dtype, (B, C, H, H), s, p)   # once per layer, as Quantizer.from_float does

            def step():
                N_.conv2d_w8(x, w, s, p, alpha=alpha, out=outs[it[0] % 3], packed=pk)
                it[0] += 1
            ms = graphed_ms(step, 6, 8, 4)
            ops = 2.0 * B * Co * Ho * Ho * C * k * k
            total_ms += ms * times
            total_ops += ops * times
            planes = 2 if dt == torch.float32 else 1   # MFMA passes per product
            lay_roof = max(planes * ops / (PEAK_BF16_TFLOPS * 1e12),
                           (x.numel() * x.element_size() + w.numel() + 4 * outs[0].numel()) / (PEAK_HBM_GBS * 1e9)) * 1e3
            roof_ms += lay_roof * times
            per_layer[f"{C}to{Co}_{H}x{H}_k{k}s{s}"] = {"call_ms": round(ms, 4), "roofline_ms": round(lay_roof, 4),
                                                         "frac": round(lay_roof / ms, 3), "x": times}
            del x, w, outs
        res[f"stack_{str(dt).split('.')[-1]}"] = {
            "layers": 20, "total_ms": round(total_ms, 4), "TFLOPs": round(total_ops / (total_ms * 1e-3) / 1e12, 1),
            "images_per_s": round(B / (total_ms * 1e-3)), "roofline": mfma_roofline(total_ops, total_ms, PEAK_BF16_TFLOPS),
            "summed_layer_rooflines_ms": round(roof_ms, 4), "frac_of_summed_layer_rooflines": round(roof_ms / total_ms, 4),
            "per_layer": per_layer}
    return res


def conv_general(N_, dev):
    """Layers the reference hands to F.conv2d with arguments its own int8 kernel lacks (quant/Weight_Only/Quantizer.py:73):
    a dilated dense layer (the implicit-GEMM MFMA kernels with generalised window strides) and a depthwise and a grouped
    one (direct convolution, HBM-bound by construction), through miq_conv2d_general: whole calls, HIP events, fp32 out."""
    layers = [("dilated_3x3_d2_256ch_28x28_b32", 32, 256, 28, 256, 3, 1, 2, 2, 1),
              ("depthwise_3x3_32ch_112x112_b64", 64, 32, 112, 32, 3, 1, 1, 1, 32),
              ("grouped_3x3_g32_128ch_56x56_b32", 32, 128, 56, 128, 3, 1, 1, 1, 32)]
    res = {}
    for name, B, C, H, Co, k, s, p, d, g in layers:
        w = torch.randint(-128, 128, (Co, C // g, k, k), dtype=torch.int32, device=dev).to(torch.int8)
        alpha = torch.rand(Co, device=dev) * 1e-3
        ent = {}
        for dt in (torch.int8, torch.float32):
            x = (torch.randint(-128, 128, (B, C, H, H), dtype=torch.int32, device=dev).to(torch.int8) if dt == torch.int8
                 else torch.randn(B, C, H, H, device=dev))
            y = N_.conv2d_general(x, w, s, p, d, g, alpha=alpha)
            ms = median_ms(lambda: N_.conv2d_general(x, w, s, p, d, g, alpha=alpha, out=y), 10, 5)
            ops = 2.0 * y.numel() * (C // g) * k * k
            nbytes = x.numel() * x.element_size() + y.numel() * 4 + w.numel()
            e = {"call_ms": round(ms, 4), "TOPS": round(ops / (ms * 1e-3) / 1e12, 2)}
            if g > 1:
                e["roofline"] = hbm_roofline(nbytes, ms)
            else:
                e["roofline"] = mfma_roofline(ops, ms, PEAK_INT8_TOPS if dt == torch.int8 else PEAK_BF16_TFLOPS)
            ent[str(dt).split(".")[-1]] = e
        res[name] = ent
    return res


def config5(N_, dev, world, rank, dist, m_list=(256, 2048, 8192)):
    """BASELINE config 5: Llama-3-70B linear shapes split by output channel over the `world` ranks (strong scaling:
    the layer is fixed, each rank owns N / world channels), int8 activations, bf16 epilogue output written straight
    into the gather buffer, in-place all-gather.  Per shape and M: the local GEMM (HIP events, median), the
    all-gather alone (host clock around issue + wait, median) and both in sequence; bytes received per GPU =
    (g-1)/g * M * N * sizeof(out).  Times are the maximum over ranks."""
    from quantization_amd.sharded import ColumnShardedLinear
    rows = []
    for n, k in ((28672, 8192), (8192, 28672), (8192, 8192)):
        n_local = n // world   # every shape divides by 1, 2, 4, 8
        wq = torch.randint(-128, 128, (n_local, k), dtype=torch.int32, device=dev).to(torch.int8)
        scale = torch.rand(n_local, device=dev) * 0.01 + 1e-4
        layer = ColumnShardedLinear(wq, scale, None, n, out_dtype=torch.bfloat16)
        for M in m_list:
            if rank == 0 and world > 1:
                print(f"[bench] config 5: [{n},{k}] over {world} rank(s), M={M}", file=sys.stderr, flush=True)
            x = torch.randint(-128, 128, (M, k), dtype=torch.int32, device=dev).to(torch.int8)
            gemm_ms = median_ms(lambda: layer.compute_block(x), 9, 5)
            got = layer.compute_block(x)
            gather, both = [], []
            for _ in range(7 if world > 1 else 1):
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                layer.start_gather(got).wait()
                torch.cuda.synchronize()
                gather.append((time.perf_counter() - t0) * 1e3)
            for _ in range(7):
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                layer.gather_async(x).wait()
                torch.cuda.synchronize()
                both.append((time.perf_counter() - t0) * 1e3)
            t = torch.tensor([gemm_ms, sorted(gather)[len(gather) // 2], sorted(both)[len(both) // 2]],
                             device=dev if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gemm_ms, gather_ms, layer_ms = (float(v) for v in t.tolist())
            recv = (world - 1) * M * n_local * 2
            rows.append({"shape": f"[{n},{k}]", "M": M, "n_local": n_local, "gemm_ms": round(gemm_ms, 4),
                         "gather_ms": round(gather_ms, 4) if world > 1 else 0.0, "layer_ms": round(layer_ms, 4),
                         "gemm_TOPS_per_gpu": round(2.0 * M * n_local * k / (gemm_ms * 1e-3) / 1e12, 1),
                         "layer_TOPS_all_gpus": round(2.0 * M * n * k / (layer_ms * 1e-3) / 1e12, 1),
                         "bytes_received_per_gpu": recv,
                         "gather_GBps_per_gpu": round(recv / (gather_ms * 1e-3) / 1e9, 1) if world > 1 else None})
            del x, got
        del layer, wq
        torch.cuda.empty_cache()
    return {"workload": "Llama-3-70B linear shapes, output channels split over the ranks, int8 activations, "
                        "bf16 output gathered in place", "gather_dtype": "bfloat16", "ranks": world, "cases": rows}


def activation_quantizer(N_, dev):
    """The dynamic per-tensor activation quantizer in front of the int8 GEMM (quant/Weight_Only/Quantizer.py:38-44
    followed through): quantize_tensor_i8 on a [256,4096] fp32 activation and the whole linear_dynamic_w8a8 at M = 16
    on the Llama-3-8B gate / up weight, whole calls as replayed HIP graphs."""
    from quantization_amd import activation as A
    res = {}
    x = torch.randn(256, 4096, device=dev)
    ms = graphed_ms(lambda: A.quantize_tensor_i8(x), 10, 10, 5)
    res["quantize_tensor_[256,4096]_f32"] = {"call_ms": round(ms, 4), "roofline": hbm_roofline(x.numel() * 5, ms)}
    Nn, K = 14336, 4096
    ws = [torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device=dev).to(torch.int8) for _ in range(6)]
    rs = [A.weight_rowsum(w) for w in ws]
    w_scale = torch.rand(Nn, device=dev) * 0.01 + 1e-4
    for M in (16, 256):
        xm = torch.randn(M, K, device=dev)
        it = [0]

        def call():
            it[0] += 1
            A.linear_dynamic_w8a8(xm, ws[it[0] % 6], w_scale, None, rs[it[0] % 6])
        ms = graphed_ms(call, 12, 10, 5)
        res[f"linear_dynamic_w8a8_M{M}_[14336,4096]"] = {"call_ms": round(ms, 4),
                                                         "roofline": hbm_roofline(Nn * K + M * K * 5 + M * Nn * 4, ms)}
    return res


def quantizer_shapes(N_, dev):
    """The per-channel quantizer on BASELINE config 4's weights and the 4096^2 one of config 2, beside the Llama-3-70B
    shape of the headline record: rotating buffers well past the 256 MiB Infinity Cache, HIP events per launch
    after a clock ramp, mean.  (A pair reads about 2 us more than rocprofv3's kernel trace gives the same launch --
    17.4 us for [4096,4096] there against 19.4 here -- and an empty pair 4.9 us, more than a pair adds to a kernel it
    brackets: the pair's reading stands uncorrected, the pessimistic one of the two.)  Algorithmic bytes: sizeof(w) + 1 per
    element + 12 per row."""
    res = {}
    for rows, cols, dt in ((14336, 4096, torch.float32), (14336, 4096, torch.bfloat16), (4096, 14336, torch.float32),
                           (4096, 4096, torch.float32)):
        esz = 4 if dt == torch.float32 else 2
        nbuf = max(4, int(1.0e9 // (rows * cols * esz)) + 1)
        ws = [(torch.randn(rows, cols, device=dev) * 0.02).to(dt) for _ in range(nbuf)]
        it = [0]

        def step():
            N_.quantize_rows_sym_i8(ws[it[0] % nbuf])
            it[0] += 1
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:
            for _ in range(nbuf):
                step()
            torch.cuda.synchronize()
        ts = event_times(step, 5 * nbuf)
        ms = sum(ts) / len(ts)
        nbytes = rows * cols * (esz + 1) + rows * 12
        res[f"[{rows},{cols}]_{str(dt).split('.')[-1]}"] = {"kernel_ms": round(ms, 4), "roofline": hbm_roofline(nbytes, ms)}
        del ws
    return res


def cpu_baseline(K, N, sample_rows, seed):
    """Reference CPU path on a bounded sample: weight-only forward of `sample_rows` rows against the full weight
    (median of 5), and Qop.quantize with the reference's per-row Python loop on the whole Llama-3-8B gate / up
    weight [14336,4096] (median of 3)."""
    from oracle import oracle_torch as T
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(seed)
    wq_full = torch.randn(14336, 4096, generator=g) * 0.02
    tq = []
    for _ in range(3):
        t0 = time.perf_counter()
        T.quantize_channel_sym_int8(wq_full)
        tq.append(time.perf_counter() - t0)
    tq = sorted(tq)[1]
    del wq_full
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int32).to(torch.int8)
    scale = torch.rand(N, generator=g) * 0.01 + 1e-4
    zp = torch.zeros(N, dtype=torch.int64)
    bias = torch.randn(N, generator=g)
    x = torch.randint(-128, 128, (sample_rows, K), generator=g, dtype=torch.int32).float()
    T.wo_linear_forward(x[:64], wq, scale, zp, bias)   # warm the thread pool
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        T.wo_linear_forward(x, wq, scale, zp, bias)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[2]
    return {
        "value": round(2.0 * sample_rows * N * K / med / 1e12, 4),
        "unit": "TOPS",
        "cores": threads,
        "kind": "port",
        "sample": f"WO forward {sample_rows}x[{N},{K}] (float GEMM + weight cast), median of 5 (+-30% box to box)",
        "spread": [round(2.0 * sample_rows * N * K / t / 1e12, 3) for t in (max(times), min(times))],
        "quantizer_GBps": round((14336 * 4096 * 5 + 14336 * 12) / tq / 1e9, 3),
        "quantizer_sample": "Qop.quantize fp32 [14336,4096], row loop, median of 3",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=8192, help="M = N(per GPU) = K")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline step (no config 3/4/5 extras)")
    ap.add_argument("--config5-m", default="256,2048,8192", help="token counts of the config-5 breakdown (a rehearsal over "
                    "gloo on one GPU moves every gathered byte through the host: give it 256)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # A multi-rank run that stops making progress (a collective that never completes) must end loudly, not hang:
    # after MIQ_BENCH_WATCHDOG seconds (default 900 with more than one rank) every thread's stack is dumped and
    # the process exits.
    watchdog = int(os.environ.get("MIQ_BENCH_WATCHDOG", "900" if world > 1 else "0"))
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # Rehearsal aid: MIQ_BENCH_BACKEND=gloo runs the N-rank code path with every rank on the one GPU of a
    # development box (RCCL needs one GPU per rank); the numbers of such a run mean nothing.
    backend = os.environ.get("MIQ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from quantization_amd import _native as N_
    from quantization_amd.sharded import ColumnShardedLinear

    S = args.size
    M = K = S
    n_total = S * world
    g = torch.Generator(device=dev).manual_seed(1234)          # same activations on every rank
    x = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int32, device=dev).to(torch.int8)
    gw = torch.Generator(device=dev).manual_seed(100 + rank)   # this rank's rows of the weight
    w_local = torch.randn(S, K, generator=gw, device=dev) * 0.02
    bias_local = torch.randn(S, generator=gw, device=dev)

    # --- quantizer (timed on its own: extra.quantizer): the Llama-3-70B gate/up projection ---------
    QR, QC = (28672, 8192) if S == 8192 else (S, K)
    nbuf = 4                                                    # 4 x 940 MB rotates far past the 256 MiB cache
    ws = [torch.randn(QR, QC, generator=gw, device=dev) * 0.02 for _ in range(nbuf)]
    it = [0]

    def quant_step():
        N_.quantize_rows_sym_i8(ws[it[0] % nbuf])
        it[0] += 1
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < PREWARM_S:   # clock ramp, as for the GEMM below
        for _ in range(nbuf):
            quant_step()
        torch.cuda.synchronize()
    qt = event_times(quant_step, 5 * nbuf)
    q_ms = sum(qt) / len(qt)
    q_bytes = QR * QC * 5 + QR * 12
    del ws
    wq, scale, _ = N_.quantize_rows_sym_i8(w_local)

    gather_name = os.environ.get("MIQ_BENCH_GATHER", "f32")   # the same output type at every --gpus N
    if gather_name not in GATHER_DTYPES:
        sys.exit(f"MIQ_BENCH_GATHER must be one of {sorted(GATHER_DTYPES)}")

    def make_layer(name):
        dt = GATHER_DTYPES[name]
        return ColumnShardedLinear(wq, scale, bias_local, n_total, out_dtype=torch.float32 if dt == torch.int8 else dt,
                                   gather_dtype=dt, act_scale=ACT_SCALE if dt == torch.int8 else None)

    def run_steps(layer, n, events=None):
        """n forwards of the layer.  The all-gather of step i runs on RCCL's stream while the GEMM of
        step i+1 runs on the compute stream (the steps are independent layers of a stack); two gather
        buffers alternate, and a buffer is written again only after its previous gather has completed.
        Every gather has completed when this returns.  `events` collects a HIP event pair around every
        EVENT_EVERY-th GEMM launch, on the stream it is launched on (an event record is a packet of its own in the queue:
        around every launch the pairs cost the step 1-2 % of its time)."""
        pending = [None, None]
        for i in range(n):
            slot = i & 1
            if pending[slot] is not None:
                pending[slot].wait()
            sample = events is not None and i % EVENT_EVERY == 0
            if sample:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            got = layer.compute_block(x, slot)
            if sample:
                e1.record()
                events.append((e0, e1))
            pending[slot] = layer.start_gather(got)
        for p in pending:
            if p is not None:
                p.wait()

    region = [None, None]   # HIP events on the launch stream around the whole timed region of the headline run

    def timed(layer, steps, warmup, events=None):
        run_steps(layer, warmup)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if events is not None:
            region[0] = torch.cuda.Event(enable_timing=True)
            region[0].record()
        run_steps(layer, steps, events)
        if events is not None:
            region[1] = torch.cuda.Event(enable_timing=True)
            region[1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    layer = make_layer(gather_name)
    # The chip takes tens of milliseconds of load to leave its idle clocks (the same 30 steps measured
    # 0.456 ms each after 5 warm-up steps and 0.408 ms after 100): bring it to its loaded state with the
    # same step before the W warm-up steps, whatever W the caller asks for.
    # Local GEMMs only: the ranks' loop counts differ, so no collective may be issued here.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < PREWARM_S:
        for _ in range(10):
            layer.compute_block(x)
        torch.cuda.synchronize()
    gemm_events = []
    elapsed = timed(layer, args.steps, args.warmup, gemm_events)

    # --- GEMM kernel duration --------------------------------------------------------------------
    # Two measurements with HIP events on the launch stream: (a) pairs around every EVENT_EVERY-th launch of the timed
    # region, less the interval an EMPTY pair measures (each record is a queue packet of its own, and the pair's
    # interval contains it); (b) one pair around the whole timed region / steps (on one GPU the launches are back to
    # back, so this is a launch's duration plus the gap to the next).  The kernel's duration is the smaller one; it
    # cannot exceed the step that contains it.
    kt = [a.elapsed_time(b) for a, b in gemm_events]
    empty_ms = empty_pair_ms()
    pair_ms = sum(kt) / len(kt) - empty_ms
    region_ms = region[0].elapsed_time(region[1]) / args.steps
    k_ms = min(pair_ms, region_ms) if world == 1 else pair_ms
    k_ms = min(k_ms, elapsed / args.steps * 1e3)
    ops_layer = 2.0 * M * S * K

    extras = {}
    if not args.no_extras:
        if world > 1:
            variants = {}
            for name in ("f32", "bf16", "i8"):
                lay = layer if name == gather_name else make_layer(name)
                el = elapsed if name == gather_name else timed(lay, max(args.steps // 4, 5), 3)
                st = args.steps if name == gather_name else max(args.steps // 4, 5)
                recv = (world - 1) * M * S * torch.empty(0, dtype=GATHER_DTYPES[name]).element_size()
                variants[name] = {"ms_per_step": round(el / st * 1e3, 4), "TOPS": round(ops_layer * world / (el / st) / 1e12, 1),
                                  "bytes_received_per_gpu_per_step": recv,
                                  "GBps_per_gpu_if_step_were_only_the_gather": round(recv / (el / st) / 1e9, 1)}
                if lay is not layer:
                    del lay
                    torch.cuda.empty_cache()
            extras["gather_variants"] = variants
        extras["config5"] = config5(N_, dev, world, rank, dist, tuple(int(v) for v in args.config5_m.split(",")))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = ops_layer * world / (ms_per_step * 1e-3) / 1e12
        achieved = ops_layer / (k_ms * 1e-3) / 1e12
        out_name = {"f32": "float32", "bf16": "bfloat16", "i8": "int8 (requantised in the epilogue)"}[gather_name]
        q_roof = hbm_roofline(q_bytes, q_ms)
        line = {
            "metric": "int8 linear TOPS (M=N=K=8192, fused per-channel dequant epilogue)" if S == 8192 else f"int8 linear TOPS (M=N=K={S})",
            "value": round(value, 1),
            "unit": "TOPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {
                "workload": f"int8 linear M={M} N={S}/GPU K={K}, int8 x int8 -> i32 MFMA -> (acc+bias)*scale -> {out_name}"
                            + ("" if world == 1 else f"; x{world} column shards + all-gather"),
                "parallelism": "single GPU" if world == 1 else f"column-sharded x{world} + all-gather ({gather_name})",
                "pct_of_int8_mfma_peak": round(100.0 * value / (PEAK_INT8_TOPS * world), 2),
            },
            "roofline": {
                "bound": "mfma",
                "achieved": round(achieved, 1),
                "peak": PEAK_INT8_TOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_INT8_TOPS, 4),
                "traffic": profiled_traffic("gemm_i8_256_kernel") if S == 8192 else None,
                "kernel": "gemm_i8_256_kernel",
                "kernel_ms": round(k_ms, 4),
                "quantizer": {"shape": f"fp32 [{QR},{QC}]", "GBps": q_roof["achieved"], "frac": q_roof["frac"],
                              "kernel_ms": round(q_ms, 4), "bytes": q_bytes,
                              "traffic": profiled_traffic("quantize_rows_reg_kernel") if S == 8192 else None},
            },
        }
        extra = {"prewarm_s": PREWARM_S, "kernel_ms_event_pairs": round(pair_ms, 4), "kernel_ms_region": round(region_ms, 4),
                 "quantizer": {"workload": f"per-channel symmetric int8 quantize of fp32 [{QR},{QC}] (5 B/element + 12 B/row)",
                               "kernel_ms": round(q_ms, 4), "empty_pair_ms": round(empty_ms, 4), "roofline": q_roof}}
        extra.update(extras)
        if world == 1 and not args.no_extras:
            extra["quantizer_shapes"] = quantizer_shapes(N_, dev)
            extra["gemm_other_shapes"] = gemm_other_shapes(N_, dev)
            extra["llama3_8b_mlp_decode"] = llama8b_mlp_decode(N_, dev)
            extra["weight_only_float"] = weight_only_float(N_, dev)
            extra["conv_int8"] = conv_int8(N_, dev)
            extra["conv_weight_only_float"] = conv_weight_only_float(N_, dev)
            extra["conv_general"] = conv_general(N_, dev)
            extra["activation_quantizer"] = activation_quantizer(N_, dev)
            # one number per other configuration, inside the record the driver keeps
            qs, go, dec = extra["quantizer_shapes"], extra["gemm_other_shapes"], extra["llama3_8b_mlp_decode"]
            short = {"float32": "f32", "bfloat16": "bf16", "float16": "f16"}
            cw = extra["conv_weight_only_float"]
            line["roofline"]["others"] = {   # fractions of the bound named in bench_extra.json; times in ms
                "quant": {k.replace("_float32", "").replace("_bfloat16", "bf16"): v["roofline"]["frac"] for k, v in qs.items()},
                "gemm": {k.split("x")[0] + "^3_" + short[k.split("_")[1]]: v["roofline"]["frac"] for k, v in go.items()},
                "decode": {m: dec[m]["roofline"]["frac"] for m in ("M1", "M16")},
                "decode_fused": {m: dec[m]["fused"]["roofline"]["frac"] for m in ("M1", "M16")},
                "conv_i8_ms": extra["conv_int8"]["stack"]["total_ms"],
                "conv_i8_frac": extra["conv_int8"]["stack"]["frac_of_summed_layer_rooflines"],
                "conv_i8_nhwc_ms": extra["conv_int8"]["stack_channels_last_int8"]["total_ms"],
                "conv_w8_ms": {short[k.replace("stack_", "")]: v["total_ms"] for k, v in cw.items() if k.startswith("stack_")},
                "conv_w8_f32_frac": cw["stack_float32"]["frac_of_summed_layer_rooflines"],
                "dyn_w8a8_M16_ms": extra["activation_quantizer"]["linear_dynamic_w8a8_M16_[14336,4096]"]["call_ms"],
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, S, 1024, 7)
        # the long tables go to a file; stdout carries the one short record, last
        out_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
        try:
            if world == 1 and not args.no_extras:   # (a run without the tables leaves the last full record alone)
                with open(os.path.join(out_dir, "bench_extra.json"), "w") as f:
                    json.dump({"headline": line, "extra": extra}, f, indent=1)
        except OSError as e:
            print(f"[bench] bench_extra.json not written: {e}", file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
