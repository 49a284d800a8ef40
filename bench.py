#!/usr/bin/env python3
"""Headline benchmark: int8 Linear (int8 x int8 -> int32 MFMA GEMM + fused per-channel dequant
epilogue) at M = N = K = 8192 per GPU, the shape BASELINE.json quotes its target on.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one forward of the layer on int8 activations already resident in HBM, with weights
produced by the per-channel quantizer from seeded fp32 data.  With N > 1 each rank owns 8192
output channels of a [8192*N, 8192] layer (weak scaling) and a step also all-gathers the
[8192, 8192] fp32 output blocks over RCCL, so every rank ends with the full [8192, 8192*N] output;
the gather of one step overlaps the GEMM of the next (independent layers of a stack).

Rank 0 prints ONE JSON line.  `value` is whole-job int8 TOPS (2*M*N*K integer operations per
layer).  `roofline` prices the GEMM kernel against the dense int8 MFMA peak with the kernel's own
duration measured by HIP events on the launch stream; `extra.quantizer` does the same for the
quantizer kernel against HBM bandwidth.  `cpu_baseline` is the reference's CPU path (restated in
oracle/oracle_torch.py, pinned to the reference by fixtures) on a bounded sample of the workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5000.0   # dense int8 MFMA, 2x the ~2.5 PFLOP/s bf16 dense peak of MI355X
PEAK_HBM_GBS = 8000.0     # HBM3E spec


PREWARM_S = 0.3   # seconds of untimed load before the warm-up steps (clock ramp)


def event_times(fn, iters):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return [s.elapsed_time(e) for s, e in evs]


def profiled_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r1_traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
            return json.load(f)[kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def weight_only_float(N_, dev):
    """The weight-only Linear on floating-point activations (miq_gemm_w8) on the Llama-3-8B gate/up weight
    [14336,4096]: whole call (the activation-plane pre-pass of fp32 inputs included), HIP events per call.
    Decode sizes are priced against HBM bandwidth on their algorithmic bytes (int8 weight once + activations +
    fp32 output); M = 256 against the dense bf16 MFMA peak on 2*M*N*K."""
    Nn, K = 14336, 4096
    w = torch.randint(-128, 128, (Nn, K), dtype=torch.int32, device=dev).to(torch.int8)
    alpha = torch.rand(Nn, device=dev) * 0.01
    res = {"weight": f"int8 [{Nn},{K}]", "out_dtype": "float32"}
    for M, dt in ((16, torch.bfloat16), (16, torch.float32), (256, torch.bfloat16), (256, torch.float32)):
        x = torch.randn(M, K, device=dev).to(dt)
        out = torch.empty(M, Nn, device=dev)

        def step():
            N_.gemm_w8(x, w, alpha=alpha, out=out)
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        ts = sorted(event_times(step, 50))
        ms = ts[len(ts) // 2]
        nbytes = Nn * K + M * K * x.element_size() + M * Nn * 4
        entry = {"call_ms": round(ms, 4), "weight_GBps": round(Nn * K / (ms * 1e-3) / 1e9, 1),
                 "TFLOPs": round(2.0 * M * Nn * K / (ms * 1e-3) / 1e12, 1)}
        if M <= 16:
            entry["roofline"] = {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "algorithmic_bytes": nbytes}
        res[f"M{M}_{str(dt).split('.')[-1]}"] = entry
    return res


def conv_int8(N_, dev):
    """The int8 Conv2d (miq_conv2d_i8) on two ResNet-18 layers at batch 256, int32 out: whole call (re-layout
    launch + implicit-GEMM kernel), HIP events per call, outputs rotating over three buffers.  The large-image
    layer is priced against HBM bandwidth on its algorithmic bytes (int8 input and weight once + the int32
    output), the 14 x 14 layer against the dense int8 MFMA peak on 2*B*Cout*Ho*Wo*Cin*9."""
    res = {"batch": 256, "out_dtype": "int32"}
    for C, H, Co, bound in ((64, 56, 64, "hbm"), (256, 14, 256, "mfma")):
        x = torch.randint(-128, 128, (256, C, H, H), dtype=torch.int32, device=dev).to(torch.int8)
        w = torch.randint(-128, 128, (Co, C, 3, 3), dtype=torch.int32, device=dev).to(torch.int8)
        outs = [torch.empty(256, Co, H, H, dtype=torch.int32, device=dev) for _ in range(3)]
        it = [0]

        def step():
            N_.conv2d_i8(x, w, 1, 1, out_dtype=torch.int32, out=outs[it[0] % 3])
            it[0] += 1
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        ts = sorted(event_times(step, 50))
        ms = ts[len(ts) // 2]
        ops = 2.0 * 256 * Co * H * H * C * 9
        nbytes = x.numel() + w.numel() + 4 * outs[0].numel()
        entry = {"call_ms": round(ms, 4), "TOPS": round(ops / (ms * 1e-3) / 1e12, 1)}
        if bound == "hbm":
            entry["roofline"] = {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                 "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "algorithmic_bytes": nbytes}
        else:
            entry["roofline"] = {"bound": "mfma", "achieved": round(ops / (ms * 1e-3) / 1e12, 1), "peak": PEAK_INT8_TOPS,
                                 "unit": "TFLOP/s", "frac": round(ops / (ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4)}
        res[f"{C}to{Co}_{H}x{H}_k3s1"] = entry
        del x, w, outs
    return res


def cpu_baseline(K, N, sample_rows, seed):
    """Reference CPU path on a bounded sample: quantize a [1024, K] slab of the weight with the
    reference's row loop, then weight-only forward of `sample_rows` rows against the full weight."""
    from oracle import oracle_torch as T
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, K, generator=g) * 0.02
    t0 = time.perf_counter()
    T.quantize_channel_sym_int8(w[:1024])
    tq = time.perf_counter() - t0
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int32).to(torch.int8)
    scale = torch.rand(N, generator=g) * 0.01 + 1e-4
    zp = torch.zeros(N, dtype=torch.int64)
    bias = torch.randn(N, generator=g)
    x = torch.randint(-128, 128, (sample_rows, K), generator=g, dtype=torch.int32).float()
    T.wo_linear_forward(x[:64], wq, scale, zp, bias)   # warm the thread pool
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        T.wo_linear_forward(x, wq, scale, zp, bias)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {
        "value": round(2.0 * sample_rows * N * K / best / 1e12, 4),
        "unit": "TOPS",
        "cores": threads,
        "kind": "port",
        "sample": f"weight-only forward of {sample_rows} of the {N} activation rows against the full "
                  f"[{N},{K}] int8 weight (float GEMM + per-call weight cast, as the reference), best of 3",
        "quantizer_GBps": round(1024 * K * 5 / tq / 1e9, 3),
        "quantizer_sample": f"Qop.quantize of a [1024,{K}] fp32 slab (reference's per-row Python loop)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=8192, help="M = N(per GPU) = K")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    layer = ColumnShardedLinear(wq, scale, bias_local, n_total, out_dtype=torch.float32)

    def run_steps(n, events=None):
        """n forwards of the layer.  The all-gather of step i runs on RCCL's stream while the GEMM of
        step i+1 runs on the compute stream (the steps are independent layers of a stack); at most two
        gathers are in flight, and every one has completed when this returns.  `events` collects a HIP
        event pair around every GEMM launch, on the stream it is launched on."""
        pending = None
        for _ in range(n):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            y = layer.local_forward(x)
            if events is not None:
                e1.record()
                events.append((e0, e1))
            _, work = layer.gather_async(y)
            if pending is not None:
                pending.wait()
            pending = work
        if pending is not None:
            pending.wait()

    # The chip takes tens of milliseconds of load to leave its idle clocks (the same 30 steps measured
    # 0.456 ms each after 5 warm-up steps and 0.408 ms after 100): bring it to its loaded state with the
    # same step before the W warm-up steps, whatever W the caller asks for.
    # Local GEMMs only: the ranks' loop counts differ, so no collective may be issued here.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < PREWARM_S:
        for _ in range(10):
            layer.local_forward(x)
        torch.cuda.synchronize()
    run_steps(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gemm_events = []
    run_steps(args.steps, gemm_events)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # --- GEMM kernel duration: the event pairs recorded around every launch of the timed region ----
    kt = [a.elapsed_time(b) for a, b in gemm_events]
    k_ms = sum(kt) / len(kt)
    ops_layer = 2.0 * M * S * K

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = ops_layer * world / (ms_per_step * 1e-3) / 1e12
        achieved = ops_layer / (k_ms * 1e-3) / 1e12
        line = {
            "metric": "int8 linear TOPS (M=N=K=8192, fused per-channel dequant epilogue)",
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
                "workload": f"int8 linear M={M} N={S} per GPU ({n_total} total) K={K}, int8 activations x "
                            "per-channel-quantized int8 weights -> int32 MFMA accumulate -> fused "
                            "(acc+bias)*scale fp32 epilogue"
                            + ("" if world == 1 else f"; column-sharded over {world} GPUs + RCCL all-gather of the fp32 output"),
                "M": M, "N_per_gpu": S, "K": K, "out_dtype": "float32",
                "parallelism": "single GPU" if world == 1 else f"column-sharded x{world} + all-gather",
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
                "note": "integer multiply-adds counted as 2 ops each (TOP/s); algorithmic work 2*M*N*K per launch; "
                        "traffic = bytes per launch through the L2 <-> fabric ports from the rocprofv3 FETCH_SIZE/WRITE_SIZE passes in "
                        "profiles/ (an upper bound of HBM bytes: it counts the fills of the eight per-XCD L2s, 8 XCDs x 4 rounds "
                        "x 12 operand panels x 2 MiB = 768 MiB of first touches per launch, most of which the 256 MiB Infinity "
                        "Cache serves; the operands themselves are 128 MiB, the output 256 MiB)",
            },
            "extra": {
                "quantizer": {
                    "workload": f"per-channel symmetric int8 quantize of fp32 [{QR},{QC}] (5 B/element + 12 B/row)",
                    "GBps": round(q_bytes / (q_ms * 1e-3) / 1e9, 1),
                    "kernel_ms": round(q_ms, 4),
                    "roofline": {"bound": "hbm", "achieved": round(q_bytes / (q_ms * 1e-3) / 1e9, 1),
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(q_bytes / (q_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "traffic": profiled_traffic("quantize_rows_reg_kernel") if S == 8192 else None,
                                 "algorithmic_bytes": q_bytes},
                },
            },
        }
        if world > 1:
            # what the step is bound by at N > 1: every rank receives the other ranks' fp32 blocks over xGMI
            recv = (world - 1) * M * S * 4
            line["extra"]["allgather"] = {
                "bytes_received_per_gpu_per_step": recv,
                "GBps_per_gpu_if_step_were_only_the_gather": round(recv / (ms_per_step * 1e-3) / 1e9, 1),
                "gemm_kernel_ms": round(k_ms, 4),
                "note": "the all-gather of step i overlaps the GEMM of step i+1; a step costs max(gather, GEMM)",
            }
        if world == 1:
            line["extra"]["weight_only_float"] = weight_only_float(N_, dev)
            line["extra"]["conv_int8"] = conv_int8(N_, dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, S, 1024, 7)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
