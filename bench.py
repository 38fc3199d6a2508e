#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X WaveNet inference engine.

Metric (BASELINE.json): samples/sec/GPU and max real-time batch @24 kHz at R=64/S=256/A=256,
20 layers, maxDilation 512, fp16.  One "step" = one run() launch of the hot path generating
SAMPLES_PER_STEP samples for every utterance of the batch (synthetic conditioning / selectors /
random-init weights already resident in HBM).  `value` = utterances x samples / second over all
GPUs, measured at the largest batch whose per-utterance rate stays >= 24 kHz (found by a bounded
sweep before the timed region; override with --batch).  The literal configs[2] point (batch 16)
is reported beside it as `c3_b16`.

Multi-GPU (--gpus N, launched by torch.distributed.run, one process per GPU): utterances shard
with no data-path collective (weak scaling: every GPU runs the same per-GPU batch); each step
ends with ONE RCCL all_gather of the [B/G][N] int32 sample blocks, overlapped with the next step.

Timing: W warm-up steps, then barrier + synchronize, K timed steps, synchronize + barrier, MAX
over ranks.  The dominant kernel's launch duration is measured live with HIP events on the stream
it is launched on, and feeds `roofline`.  `cpu_baseline` times the reference's own CPU
implementation (oracle/_ref when present, else the C restatement oracle/) on one host core over a
bounded sample of the same workload shape (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R, S, A, L, MAXD = 64, 256, 256, 20, 512      # BASELINE.json configs[2] shape (C3)
REALTIME_KHZ = 24.0
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0                 # dense fp16/bf16 MFMA peak
L2_PEAK_GBS = 34500.0                         # aggregate L2 bandwidth (guide, measured)
LDS_PEAK_GBS = 256 * 256 * 2.4                # 256 B/clk/CU x 256 CUs x 2.4 GHz (guide, LDS section)

# algorithmic work per sample per utterance (SURVEY.md 8d / BASELINE.md)
MACS = L * (5 * R * R + S * R) + A * S + A * A
FLOPS = 2 * MACS
WEIGHT_BYTES = 2 * (L * (5 * R * R + S * R) + A * S + A * A)     # fp16 weights streamed per tile pass
HBM_BYTES = 2 * 2 * R * L + 4 + 4                                # cond (fp16) + selector + yOut


def lds_bytes_per_sample(stream_mode, bt=1):
    """Algorithmic LDS bytes moved per generated sample per WORKGROUP (fp16).
    wg kernel (4 waves on bt tiles): per layer every wave reads the x, x[t-d] and h fragment images
    (R/32 KiB each) and its bias quads, and writes its quarter of h, x and the dilated tap; the head
    moves skip / zs images (S/32, A/32 KiB, read by all 4 waves) and the fp32 logits once each way.
    stream kernel (4 consumer waves, one tile each): the whole weight stream is written to LDS once
    by LDS-DMA and read once by every consumer; activations never leave registers."""
    kf = lambda n: n // 32 * 1024
    if stream_mode:
        return 5 * WEIGHT_BYTES + 4 * (16 * A * 4 * 2)
    per_layer = bt * (4 * 3 * kf(R) + 3 * kf(R)) + 4 * 64 * 16 * 3        # exchanges + bias quads
    head = bt * (5 * kf(S) + 5 * kf(A) + 2 * 16 * A * 4 + 2 * kf(R)) + 4 * 64 * 16 * (S // 64 + 2 * A // 64)
    return L * per_layer + head


def make_weights(seed=3):
    """The parity recipe of nv_wavenet_test.cu:36-111: uniform +-0.25/R for embeddings, output head;
    +-0.25/rows for per-layer matrices and biases."""
    rng = np.random.default_rng(seed)
    u = lambda sc, *s: ((rng.random(s, dtype=np.float32) - 0.5) * sc).astype(np.float32)
    w = dict(embP=u(0.5 / R, A, R), embC=u(0.5 / R, A, R),
             Wprev=u(0.25 / R, L, R, 2 * R), Wcur=u(0.25 / R, L, R, 2 * R), Bh=u(0.25 / R, L, 2 * R),
             Wres=u(0.5 / R, L, R, R), Bres=u(0.5 / R, L, R), Wskip=u(0.5 / S, L, R, S), Bskip=u(0.5 / S, L, S),
             Wzs=u(0.5 / R, S, A), Bzs=u(0.5 / R, A), Wza=u(0.5 / R, A, A), Bza=u(0.5 / R, A))
    return w


def build_engine(w, B, N, precision=16):
    from nv_wavenet_amd import WavenetEngine
    e = WavenetEngine(R, S, A, L, MAXD, B, N, impl=3, tanhEmbed=True, precision=precision)
    e.setEmbeddings(w["embP"], w["embC"])
    for l in range(L):
        e.setLayerWeights(l, w["Wprev"][l], w["Wcur"][l], w["Bh"][l], w["Wres"][l], w["Bres"][l], w["Wskip"][l],
                          w["Bskip"][l])
    e.setOutWeights(w["Wzs"], w["Bzs"], w["Wza"], w["Bza"])
    return e


def device_inputs(B, N, seed):
    """Synthetic conditioning [N][L][B][2R] (uniform +-0.25/R) and selectors [N][B] in HBM."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    Lh = torch.empty(N, L, B, 2 * R, dtype=torch.float32, device="cuda")
    Lh.uniform_(-0.25 / R, 0.25 / R, generator=g)
    sel = torch.rand(N, B, dtype=torch.float32, device="cuda", generator=g)
    return Lh, sel


def samples_per_step_for(B):
    # keep the fp32 source of the conditioning under ~24 GB
    per_sample = L * B * 2 * R * 4
    n = int(24e9 // per_sample)
    n = max(64, min(2048, n // 64 * 64))
    return n


def measure_khz(w, B, N, seed=11, mode=None):
    """per-utterance kHz of one launch at batch B (HIP events on the launch stream).
    mode: None = the engine's own choice; "wg" / "stream" force a kernel organisation."""
    import torch
    old = os.environ.get("NVW_MODE")
    if mode:
        os.environ["NVW_MODE"] = mode
    try:
        e = build_engine(w, B, N)
    finally:
        if mode:
            os.environ.pop("NVW_MODE", None)
            if old is not None:
                os.environ["NVW_MODE"] = old
    Lh, sel = device_inputs(B, N, seed)
    e.setInputs(Lh, sel)
    del Lh
    torch.cuda.synchronize()
    e.time_runs(1, min(N, 64), B)
    ms = e.time_runs(1, N, B)
    e.close()
    torch.cuda.empty_cache()
    return N / ms


def cpu_run_once(w, n, B=16, seed=5):
    """n samples of the C3 shape at batch B on the calling core: (seconds, kind)."""
    from oracle import oracle as O
    kind = "reference" if O.have_ref() else "port"
    cls = O.RefOracle if kind == "reference" else O.Oracle
    rng = np.random.default_rng(seed)

    class T:
        pass
    t = T()
    for k, v in w.items():
        setattr(t, k, v)
    Lh = ((rng.random((n, L, B, 2 * R), dtype=np.float32) - 0.5) * (0.5 / R)).astype(np.float32)
    sel = rng.random((n, B), dtype=np.float32) * 0.999
    o = cls(L, B, n, R, S, A, MAXD)
    o.set_model(t)
    o.set_inputs(Lh, sel)
    t0 = time.perf_counter()
    o.run(n)
    dt = time.perf_counter() - t0
    o.close()
    return dt, kind


def cpu_baseline(w, budget_s=20.0):
    """The reference's CPU implementation (nv_wavenet_reference.cpp built into oracle/_ref), C3 shape,
    batch 16, bounded sample: on ONE host core (the reference's own single-threaded path = the reported
    value), then one instance per host core on independent batch slices, as
    sub-object all_cores."""
    import subprocess
    B = 16
    dt, kind = cpu_run_once(w, 4)
    n = int(max(8, min(512, budget_s / (dt / 4))))
    dt, kind = cpu_run_once(w, n)
    out = dict(value=B * n / dt, unit="samples/s", cores=1, kind=kind,
               sample="R%d/S%d/A%d L%d maxD%d fp32, batch %d x %d samples (%.1f s)" % (R, S, A, L, MAXD, B, n, dt),
               khz_per_utterance=n / dt / 1e3)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 64)
    if cores > 1:
        # separate interpreters (no torch, no HIP): a fork of this process would carry the GPU runtime
        n2 = max(8, n // 4)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(n2), "--cpu-seed", str(i)],
                                  stdout=subprocess.PIPE, cwd=ROOT) for i in range(cores)]
        ok = 0
        for pr in procs:
            so, _ = pr.communicate()
            ok += pr.returncode == 0 and b"cpu_worker_seconds" in so
        wall = time.perf_counter() - t0
        if ok == cores:
            out["all_cores"] = dict(value=cores * B * n2 / wall, unit="samples/s", cores=cores,
                                    sample="%d processes x batch %d x %d samples, wall %.1f s incl. start-up" % (cores, B, n2, wall))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (0 = find the max real-time batch)")
    ap.add_argument("--samples", type=int, default=0, help="samples per step (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke runs)")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        dt, _ = cpu_run_once(make_weights(), args.cpu_worker, seed=5 + args.cpu_seed)
        print(json.dumps({"cpu_worker_seconds": dt}))
        return

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from nv_wavenet_amd.sharding import gather_samples
    w = make_weights()
    ncu = torch.cuda.get_device_properties(local_rank).multi_processor_count

    # ---- workload: the largest per-GPU batch that stays real time (bounded sweep, rank 0) ----
    sweep = {}
    c3_b16_khz = None
    if args.batch:
        B = args.batch
    else:
        choice = torch.zeros(1, dtype=torch.int64, device="cuda")
        if rank == 0:
            c3_b16_khz = measure_khz(w, 16, 1024)
            sweep[16] = c3_b16_khz
            best = 16
            for tiles_per_cu in (1, 2, 4):
                cand = 16 * ncu * tiles_per_cu
                khz = measure_khz(w, cand, 128)
                sweep[cand] = khz
                if khz >= REALTIME_KHZ:
                    best = cand
                else:
                    break
            choice[0] = best
        if world > 1:
            if args.backend != "nccl":
                choice = choice.cpu()
            dist.broadcast(choice, 0)
        B = int(choice.item())
    N = args.samples or samples_per_step_for(B)

    # throughput mode (not real time): every SIMD owns a tile, weights streamed once per CU
    thr = None
    if rank == 0 and not args.batch:
        bt = 64 * ncu
        khz_t = sweep.get(bt) or measure_khz(w, bt, 128, mode="stream")
        thr = {"batch_per_gpu": bt, "khz_per_utterance": khz_t, "samples_per_sec_per_gpu": bt * khz_t * 1e3,
               "kernel": "wn::wavenet_stream", "real_time": bool(khz_t >= REALTIME_KHZ)}

    e = build_engine(w, B, N)
    kinfo = e.kernelInfo(B, False)
    Lh, sel = device_inputs(B, N, 100 + rank)
    e.setInputs(Lh, sel)
    del Lh, sel
    torch.cuda.empty_cache()
    ybuf = [torch.zeros(B, N, dtype=torch.int32, device="cuda") for _ in range(2)]
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    pending = []

    def step(i, events=None):
        y = ybuf[i % 2]
        # the gather of step i overlaps the kernel of step i+1 (RCCL runs on its own stream); the gather
        # of step i-2 read the buffer this step overwrites, so it is completed first
        while len(pending) > 1:
            pending.pop(0)()
        if events is not None:
            events[0].record(stream)
        assert e.run(N, B, y, 1, False, sptr)
        if events is not None:
            events[1].record(stream)
        if world > 1:
            _, fin = gather_samples(y, B * world, async_op=True)
            pending.append(fin)

    for i in range(args.warmup):
        step(i)
    while pending:
        pending.pop(0)()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, evs[i])
    while pending:
        pending.pop(0)()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps
    ylast = ybuf[(args.steps - 1) % 2]
    hist = int(torch.unique(ylast).numel())
    e.close()

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = world * B * N / (dt / args.steps)
        khz = N / kern_ms
        units = B * N                                   # utterance-samples per launch
        flops = units * FLOPS
        tiles = (B + 15) // 16
        stream_mode = tiles > 2 * ncu                           # engine's choice (nv_wavenet.hpp)
        bt = 2 if tiles > ncu else 1                            # tiles per workgroup of the latency kernel
        # workgroups (weight-stream passes) per sample
        passes = (tiles + 3) // 4 if stream_mode else (tiles + bt - 1) // bt
        kname = kinfo.split(" ")[0]                             # what the engine reports it launches
        assert ("stream" in kname) == stream_mode and (stream_mode or "BT=%d" % bt in kname), kinfo
        traffic = None
        # HBM bytes per launch from the PMC passes of the latest profiled round (profiles/traffic_rNN.json,
        # written by scripts/make_profiles.sh); only valid for the launch shape it was measured on
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
            try:
                tj = json.load(open(tf))
                if tj.get("batch") == B and tj.get("samples") == N:
                    traffic = tj.get("hbm_bytes_per_launch")
                    break
            except Exception:
                pass
        roofline = dict(bound="mfma", achieved=flops / (kern_ms * 1e-3) / 1e12, peak=MFMA_F16_PEAK_TFLOPS,
                        unit="TFLOP/s", traffic=traffic, kernel=kname, launch=kinfo,
                        kernel_ms=kern_ms,
                        hbm=dict(achieved=units * HBM_BYTES / (kern_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s"),
                        l2_weight_stream=dict(achieved=passes * N * WEIGHT_BYTES / (kern_ms * 1e-3) / 1e9,
                                              peak=L2_PEAK_GBS, unit="GB/s"),
                        lds=dict(achieved=passes * N * lds_bytes_per_sample(stream_mode, 1 if stream_mode else bt) / (kern_ms * 1e-3) / 1e9,
                                 peak=LDS_PEAK_GBS, unit="GB/s"))
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        roofline["hbm"]["frac"] = roofline["hbm"]["achieved"] / HBM_PEAK_GBS
        roofline["l2_weight_stream"]["frac"] = roofline["l2_weight_stream"]["achieved"] / L2_PEAK_GBS
        roofline["lds"]["frac"] = roofline["lds"]["achieved"] / LDS_PEAK_GBS
        out = {
            "metric": "samples/sec (all GPUs) at the max real-time batch @24kHz, R64/S256/A256 20L fp16",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "C3: R=64 S=256 A=256 L=20 maxDilation=512 fp16, autoregressive generation",
                       "batch_per_gpu": B, "global_batch": B * world, "samples_per_step": N,
                       "parallelism": "batch-sharded x%d, final RCCL all_gather" % world},
            "samples_per_sec_per_gpu": value / world,
            "khz_per_utterance": khz, "max_realtime_batch_per_gpu": B if khz >= REALTIME_KHZ else None,
            "realtime_sweep_khz": {str(k): v for k, v in sweep.items()},
            "c3_b16": {"khz_per_utterance": c3_b16_khz, "samples_per_sec": None if c3_b16_khz is None else 16e3 * c3_b16_khz},
            "throughput_mode": thr,
            "distinct_samples_in_last_step": hist,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
