#!/usr/bin/env python3
"""bench.py -- macroblocks/sec of the 1080p IPPP encode hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1920x1080 4:2:0
synthetic input, coded 1920x1072 (the reference crops to multiples of 16), IPPP with
IntraEvery = 30, qp 12, WindowSize 32 (+-16 integer search, +-2 quarter-pel search),
MAXDIFF 3, BasicInterEncoding 0.  One "step" = one closed GOP of 30 pictures for every one of
the S independent streams resident on the GPU (input pictures already in HBM; the RBSP stays in
HBM).  Ranks (one per GPU) encode disjoint stream sets, no data-path collective: weak scaling.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (dominant kernel, live HIP
event timing from the library's launch stream) and `cpu_baseline` (the CPU oracle, test
infrastructure, timed on a bounded sample of the same workload on this host).
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "tests"))

W, H_IN, H = 1920, 1080, 1072
GOP = 30
QP, WINDOW, MAXDIFF = 12, 32, 3
ME_BYTES_PER_MB = 528          # 256 B current luma + 256 B reference luma + 16 B MVs
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(frames_np, nframes):
    """CPU oracle (oracle/fo_cli, a bit-exact port, kind 'port') on the first `nframes` pictures
    of stream 0: one process = one core, like the single-threaded reference."""
    cli = ROOT / "oracle" / "fo_cli"
    if not cli.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "fo_cli"], check=True)
    tmp = Path(os.environ.get("TMPDIR", "/tmp")) / f"ferbench_{os.getpid()}"
    tmp.mkdir(parents=True, exist_ok=True)
    src = tmp / "in.yuv"
    frames_np[:nframes].tofile(src)
    out = subprocess.run([str(cli), "enc", str(W), str(H), str(nframes), str(QP), str(WINDOW), str(MAXDIFF),
                          str(GOP), "0", str(src), str(tmp / "out.264")], check=True, capture_output=True, text=True)
    res = json.loads(out.stdout.strip().splitlines()[-1])
    for f in tmp.iterdir():
        f.unlink()
    tmp.rmdir()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("FER_BENCH_STREAMS", "128")))
    ap.add_argument("--stagger-ms", type=float, default=float(os.environ.get("FER_BENCH_STAGGER_MS", "0")),
                    help="start offset between consecutive contexts (milliseconds)")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("FER_BENCH_CONTEXTS", "2")),
                    help="encoder contexts per GPU, each on its own HIP stream and host thread (streams are split evenly)")
    ap.add_argument("--cpu-frames", type=int, default=3, help="pictures of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from conftest import load_pkg

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    pkg = load_pkg()
    from h264_fer_amd.synth import gen_frames_torch
    S = args.streams
    # one GOP of distinct content per stream (seed differs per stream and per rank), resident in HBM
    frames = gen_frames_torch(W, H_IN, GOP, S, dev, seed=1234 + 1000 * rank, noise=2)
    # centre crop 1080 -> 1072 like ReadFromY4M (F/fileIO.cpp:290-333)
    ys = W * H_IN
    Y = frames[:, :, :ys].view(GOP, S, H_IN, W)[:, :, 4:4 + H, :]
    U = frames[:, :, ys:ys + ys // 4].view(GOP, S, H_IN // 2, W // 2)[:, :, 2:2 + H // 2, :]
    V = frames[:, :, ys + ys // 4:].view(GOP, S, H_IN // 2, W // 2)[:, :, 2:2 + H // 2, :]
    frames = torch.cat([Y.reshape(GOP, S, -1), U.reshape(GOP, S, -1), V.reshape(GOP, S, -1)], dim=2).contiguous()
    torch.cuda.synchronize()

    # The streams are split over `contexts` encoder contexts, each with its own HIP stream and host
    # thread: while one context sits in its latency-bound per-diagonal launches, the other fills the
    # CUs with its throughput-bound kernels (the streams are independent, so this is pure overlap).
    import threading
    NC = max(1, min(args.contexts, S))
    bounds = [S * i // NC for i in range(NC + 1)]
    parts = [frames[:, bounds[i]:bounds[i + 1]].contiguous() for i in range(NC)]
    del frames
    torch.cuda.synchronize()
    encs = [pkg.FerHip(W, H, bounds[i + 1] - bounds[i], qp=QP, window=WINDOW, maxdiff=MAXDIFF, intra_every=GOP)
            for i in range(NC)]
    enc = encs[0]
    nmb = enc.nmb

    def run_ctx(i):
        e, fr = encs[i], parts[i]
        if i and args.stagger_ms > 0:
            # contexts that run in lockstep compete for the same unit (HBM in the feature pass, VALU in the search);
            # a start offset of a fraction of a picture puts one context's memory-bound kernels under the other's
            # compute-bound ones.  The offset is inside the timed region.
            time.sleep(i * args.stagger_ms / 1e3)
        for t in range(GOP):
            e.set_frames_device(fr[t].data_ptr())
            e.encode_picture_device(None)   # AUTO: selectNALUnitType semantics (IDR every GOP)

    def step():
        if NC == 1:
            return run_ctx(0)
        th = [threading.Thread(target=run_ctx, args=(i,)) for i in range(NC)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for e in encs:
        e.get_profile(reset=True)
        e.profile(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof = {}
    status = []
    for e in encs:
        for k, (ms_, n_) in e.get_profile(reset=True).items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + ms_, a[1] + n_)
        e.profile(False)
        status += e.status()
    if any(status):
        raise SystemExit(f"device error flags {status}")
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_mbs = world * S * GOP * nmb * args.steps
    value = total_mbs / dt

    # roofline of the dominant kernel (by accumulated device time; every profiled phase is one kernel, except
    # "sort" = the six launches of the two radix passes and "cavlc" = size + scan + emit)
    p_pictures = (GOP - 1) * args.steps
    KERNELS = {  # phase -> (kernel, algorithmic bytes per macroblock, pictures it runs on)
        "interp": ("k_interp", 256 + 16 * 256, p_pictures),
        "features": ("k_features", 16 * 256 + 256 * (192 + 12), p_pictures),
        "sort_keys": ("k_sort_keys", 256 * (2 + 8), p_pictures),
        "sort": ("k_rs_hist+k_rs_scan+k_rs_scatter (two radix passes)", 2 * 256 * (4 + 8 + 8), p_pictures),
        "sort_finish": ("k_sort_finish", 256 * (8 + 12 + 16), p_pictures),
        "me_pre": ("k_me_pre", ME_BYTES_PER_MB, p_pictures),
        "me_walk": ("k_me_walk", ME_BYTES_PER_MB, p_pictures),
        "me_resolve": ("k_me_resolve", ME_BYTES_PER_MB, p_pictures),
        "p_resid": ("k_p_resid", 1152, p_pictures),
        "intra": ("k_intra_mb", 768, args.steps),
        "cavlc": ("k_cavlc", 800, GOP * args.steps),
    }
    # HBM traffic per macroblock from the committed PMC measurement (profiles/r01_traffic.json); None when absent
    per_mb_traffic = {}
    tj = ROOT / "profiles" / "r01_traffic.json"
    if tj.exists():
        per_mb_traffic = json.loads(tj.read_text()).get("bytes_per_mb", {})

    def line(k):
        name, bpm, pics = KERNELS[k]
        ms_, launches_ = prof[k]
        units = S * nmb * pics  # accumulated over contexts, like ms_ and launches_
        per_launch_s = (ms_ / 1e3) / max(launches_, 1)
        ach = (bpm * units / max(launches_, 1)) / per_launch_s / 1e9 if ms_ > 0 else 0.0
        tr = per_mb_traffic.get(k)
        return {"kernel": name, "achieved": round(ach, 3), "frac": round(ach / HBM_PEAK_GBS, 6),
                "traffic": round(tr * units / max(launches_, 1)) if tr else None,
                "avg_launch_us": round(per_launch_s * 1e6, 2), "launches": launches_, "bytes_per_mb": bpm,
                "total_ms": round(ms_, 2)}

    dom = max(KERNELS, key=lambda k: prof[k][0])
    dl = line(dom)
    roofline = {"bound": "hbm", "kernel": dl["kernel"], "achieved": dl["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dl["frac"], "traffic": dl["traffic"], "avg_launch_us": dl["avg_launch_us"],
                "launches": dl["launches"], "bytes_per_mb": dl["bytes_per_mb"],
                "kernels": {k: line(k) for k in KERNELS}}

    out = None
    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            res = cpu_baseline(parts[0][:, 0].cpu().numpy(), args.cpu_frames)
            cpu = {"value": round(res["mb_per_s"], 1), "unit": "macroblocks/s", "cores": 1, "kind": "port",
                   "sample": f"first {args.cpu_frames} pictures (I+{args.cpu_frames - 1}P) of stream 0, "
                             f"{res['mbs']} MBs in {res['seconds']:.1f} s, oracle/fo_cli single thread"}
        out = {"metric": "macroblocks/sec encode (1080p full-search ME) + bit-exact bitstream vs ref",
               "value": round(value, 1), "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "1080p IPPP encode, full-search +-16 ME (BASELINE configs[2])",
                          "coded_size": f"{W}x{H}", "streams_per_gpu": S, "contexts_per_gpu": NC, "gop": GOP, "qp": QP, "window": WINDOW,
                          "maxdiff": MAXDIFF, "mbs_per_step": world * S * GOP * nmb,
                          "parallelism": f"{world} x {S} independent closed-GOP streams"},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    for e in encs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
