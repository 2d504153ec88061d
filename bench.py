#!/usr/bin/env python3
"""bench.py -- macroblocks/sec of the fer_h264 encode hot path on MI355X.

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1920x1080 4:2:0 synthetic
input, coded 1920x1072 (the reference crops to multiples of 16), IPPP with IntraEvery = 30, qp 12, WindowSize 32
(+-16 integer search, +-2 quarter-pel search), MAXDIFF 3, BasicInterEncoding 0.  One "step" = one closed GOP of 30
pictures for every one of the S independent streams resident on the GPU (input pictures already in HBM; the RBSP
stays in HBM).  Ranks (one per GPU) encode disjoint stream sets, no data-path collective: weak scaling.

`--config 4k` is BASELINE configs[3]: 3840x2160, 64 pictures = 8 closed GOPs of 8 (qp 28, WindowSize 32), the GOPs
sharded over the ranks (gops_of_rank), each rank encoding its GOPs with libferhip, the host merging the NAL units in
GOP order; the SHA-256 of the merged stream is checked against the committed oracle value (strong scaling: the
total work is fixed).

`--gpus N` (N > 1) without a launcher: this process starts N rank processes itself (before any GPU call) and relays
rank 0's line.  Under `python -m torch.distributed.run` the ranks come from the environment.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (dominant kernel, live HIP event timing on the
library's launch stream), `cpu_baseline` (the CPU oracle -- test infrastructure -- timed on a bounded sample of the
same workload on this host) and the extra figures SURVEY.md 8d asks for (PCIe-inclusive rate, integer-VALU fraction,
P_Skip fraction, N-process CPU baseline, output hash check, secondary configurations).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "tests"))

ME_BYTES_PER_MB = 528          # 256 B current luma + 256 B reference luma + 16 B MVs
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_LANE_OPS = 78.6e12        # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
LANE_OPS_PER_MB = 2.1e5        # the reference-exact search at WindowSize 32

CONFIGS = {
    "1080p": dict(W=1920, H_in=1080, H=1072, gop=30, qp=12, window=32, maxdiff=3, scaling="weak",
                  workload="1080p IPPP encode, full-search +-16 ME (BASELINE configs[2])"),
    "4k": dict(W=3840, H_in=2160, H=2160, gop=8, ngops=8, qp=28, window=32, maxdiff=3, scaling="strong",
               workload="4K 3840x2160 IPPP encode, 64 pictures = 8 closed GOPs of 8 sharded over the ranks "
                        "(BASELINE configs[3])"),
}

# phase -> (kernel, algorithmic bytes per macroblock, runs on "P" / "I" / "all" pictures).  What limits a
# kernel is not stated here: it is read from the counters of the last profiling pass (profiles/r03_traffic.json, "limiter":
# HBM bytes per second against the 6.29 TB/s a copy reaches, VALU / scalar instructions against their issue rates).
KERNELS = {
    "interp": ("k_interp", 256 + 16 * 256, "P"),
    "sort_keys": ("k_feat0", 256 + 256 * (12 + 2), "P"),  # luma in; row-major plane-0 records + keys in arrival order out
    # pass 1: keys 2 (histogram) + luma 1 in, records 16 + high digit 1 out; pass 2: digit 1 + 1, records 16 in, walk records 12 +
    # positions 4 + keys 2 out
    "sort": ("k_rs_hist+k_rs_scan+k_rs_scatter (two radix passes)", 256 * (2 + 1 + 16 + 1 + 1 + 1 + 16 + 12 + 4 + 2), "P"),
    "sort_finish": ("k_sort_index+k_bucket_classes+k_sort_quirk", 256 * 6 + 1956, "P"),
    "me_pre": ("k_me_pre", ME_BYTES_PER_MB, "P"),
    "me_walk": ("k_me_walk", ME_BYTES_PER_MB, "P"),
    "me_spec": ("k_me_spec", ME_BYTES_PER_MB, "P"),
    "me_resolve": ("k_me_resolve", ME_BYTES_PER_MB, "P"),
    "p_resid": ("k_p_resid", 1152, "P"),
    "intra": ("k_intra_mb", 768, "I"),
    "cavlc": ("k_cavlc", 800, "all"),
    "frame_sad": ("k_frame_sad", 512, "P"),
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """Parent of an N-rank run: never touches the GPU; children get RANK / WORLD_SIZE / LOCAL_RANK."""
    n = args.gpus
    port = free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), FER_BENCH_CHILD="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0 = procs[0].communicate()[0]
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(rcs) or line is None:
        sys.stderr.write(out0 or "")
        raise SystemExit(f"rank processes failed: return codes {rcs}")
    print(line, flush=True)


def stage_oracle_input(frames_np, nframes, tag):
    """the input file of one fo_cli process, written BEFORE any clock starts"""
    tmp = Path(os.environ.get("TMPDIR", "/tmp")) / f"ferbench_{os.getpid()}_{tag}"
    tmp.mkdir(parents=True, exist_ok=True)
    frames_np[:nframes].tofile(tmp / "in.yuv")
    return tmp


def run_oracle_enc(cfg, tmp, nframes):
    """oracle/fo_cli (a bit-exact port of the reference, kind 'port') on `nframes` pictures: one process = one core.
    fo_cli times the encode itself (input read and output written outside its clock)."""
    cli = ROOT / "oracle" / "fo_cli"
    if not cli.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "fo_cli"], check=True)
    cmd = [str(cli), "enc", str(cfg["W"]), str(cfg["H"]), str(nframes), str(cfg["qp"]), str(cfg["window"]),
           str(cfg["maxdiff"]), str(cfg["gop"]), "0", str(tmp / "in.yuv"), str(tmp / "out.264")]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)


def finish_oracle(proc, tmp):
    out = proc.communicate()[0]
    if proc.returncode != 0:
        raise SystemExit(f"oracle/fo_cli failed ({proc.returncode})")
    res = json.loads(out.strip().splitlines()[-1])
    for f in tmp.iterdir():
        f.unlink()
    tmp.rmdir()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="1080p")
    ap.add_argument("--content", choices=["textured", "letterbox", "flat-half", "still"], default="textured",
                    help="synthetic content variant: letterbox = 128 black-ish rows top and bottom, flat-half = left half one "
                         "value (large flat areas), still = no motion and no noise (P_Skip heavy)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("FER_BENCH_STREAMS", "256")))
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("FER_BENCH_CONTEXTS", "1")),
                    help="encoder contexts per GPU, each on its own HIP stream and host thread (streams are split evenly)")
    ap.add_argument("--cpu-frames", type=int, default=30, help="pictures of the CPU-baseline sample: 30 = one full GOP of one stream, "
                    "about 70 s on one core, run beside the GPU part (0 = skip)")
    ap.add_argument("--cpu-nproc-frames", type=int, default=4, help="pictures each process of the N-process CPU leg encodes")
    ap.add_argument("--secondary", type=int, default=1, help="also time configs[1] (720p I-only) and configs[4] (decode)")
    ap.add_argument("--e2e", type=int, default=1, help="also time the PCIe-inclusive path (pinned host pictures in, host RBSP out)")
    ap.add_argument("--probe-build", type=int, default=0, help="development only: a -DFER_PROBE library with FER_DBG set skips "
                    "stages on purpose, so a wrong output hash is reported (\"ok\": false) instead of ending the run")
    ap.add_argument("--resolve-wgs", type=int, default=0, help="workgroups of the persistent motion-chain launch (0 = library default)")
    ap.add_argument("--overlap-sort", type=int, default=0, help="0 = the sort of the reference picture runs before k_me_pre instead of beside it")
    ap.add_argument("--speculate", type=int, default=1, help="0 = the motion chain searches everything itself (no k_me_spec pre-pass)")
    ap.add_argument("--resolve-group", type=int, default=0, help="streams per ticket group of the motion chain (0 = library default)")
    ap.add_argument("--rehearse-ranks", type=int, default=0, help="1 = rank plumbing only, no GPU and no encoder: spawn the ranks, "
                    "rendezvous (use --dist-backend gloo), barrier, max-reduce of the step time, and the --config 4k gather + merge "
                    "of per-GOP streams with placeholder bytes; prints {\"rehearsal\": ...} instead of a bench line")
    ap.add_argument("--dist-backend", default=os.environ.get("FER_BENCH_BACKEND", "nccl"),
                    help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal: several ranks may share GPU 0)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "FER_BENCH_CHILD" not in os.environ:
        return spawn_ranks(args, sys.argv[1:])

    import numpy as np
    import torch
    import torch.distributed as dist
    from conftest import load_pkg

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_ranks:
        return rehearse_ranks(args, rank, world, dist, torch, load_pkg())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg = CONFIGS[args.config]
    W, H_IN, H, GOP = cfg["W"], cfg["H_in"], cfg["H"], cfg["gop"]
    pkg = load_pkg()
    from h264_fer_amd.synth import gen_frames_torch

    if args.config == "4k":
        # 64 pictures of ONE sequence; GOP g = pictures 8g .. 8g+7, encoded as its own stream by rank g % world
        mine = pkg.gops_of_rank(cfg["ngops"], world, rank)
        S = len(mine)
        if S == 0:
            raise SystemExit("more ranks than GOPs")
        frames = gen_frames_torch(W, H_IN, GOP, S, dev, noise=2, seeds=[1234] * S, t0s=[g * GOP for g in mine])
        total_streams = cfg["ngops"]
    else:
        S = args.streams
        mine = None
        # one GOP of distinct content per stream (seed differs per stream and per rank), resident in HBM
        still = args.content == "still"
        frames = gen_frames_torch(W, H_IN, GOP, S, dev, seed=1234 + 1000 * rank, noise=0 if still else 2)
        if still:
            frames[1:] = frames[0:1]
        total_streams = world * S
    ys = W * H_IN
    Yp = frames[:, :, :ys].view(GOP, S, H_IN, W)
    if args.content == "letterbox":
        Yp[:, :, :132] = 16
        Yp[:, :, -132:] = 16
    elif args.content == "flat-half":
        Yp[:, :, :, : W // 2] = 100
    # centre crop like ReadFromY4M (F/fileIO.cpp:290-333)
    ct = (H_IN - H) // 2
    Y = Yp[:, :, ct:ct + H, :]
    U = frames[:, :, ys:ys + ys // 4].view(GOP, S, H_IN // 2, W // 2)[:, :, ct // 2:ct // 2 + H // 2, :]
    V = frames[:, :, ys + ys // 4:].view(GOP, S, H_IN // 2, W // 2)[:, :, ct // 2:ct // 2 + H // 2, :]
    frames = torch.cat([Y.reshape(GOP, S, -1), U.reshape(GOP, S, -1), V.reshape(GOP, S, -1)], dim=2).contiguous()
    del Y, U, V, Yp
    torch.cuda.synchronize()

    # The streams are split over `contexts` encoder contexts, each with its own HIP stream and host thread: while one
    # context's host thread waits for its 8-byte frame-SAD read-back, the other keeps the GPU fed.
    NC = max(1, min(args.contexts, S))
    bounds = [S * i // NC for i in range(NC + 1)]
    parts = [frames[:, bounds[i]:bounds[i + 1]].contiguous() for i in range(NC)]
    del frames
    torch.cuda.synchronize()
    # the CPU baseline's single-core leg (one full GOP of stream 0 through oracle/fo_cli, about 70 s) starts now and runs
    # beside the GPU part on one of the host's cores
    cpu_job = cpu_sample = None
    if rank == 0 and world == 1 and args.cpu_frames > 0 and args.config == "1080p":
        cpu_sample = parts[0][:, 0].cpu().numpy()
        tmp_ = stage_oracle_input(cpu_sample, min(args.cpu_frames, GOP), "one")
        cpu_job = (run_oracle_enc(cfg, tmp_, min(args.cpu_frames, GOP)), tmp_)
    encs = [pkg.FerHip(W, H, bounds[i + 1] - bounds[i], qp=cfg["qp"], window=cfg["window"], maxdiff=cfg["maxdiff"],
                       intra_every=GOP) for i in range(NC)]
    nmb = encs[0].nmb
    fsz = W * H * 3 // 2
    for e in encs:
        # the persistent motion-chain launch of a context takes its share of the GPU's workgroup slots: with two
        # contexts each leaves room for the other's streaming kernels
        e.tune(1, args.resolve_wgs or (6144 if NC == 1 else max(1536, min(6144, 48 * e.S))))  # two contexts share the workgroup slots; measured: 64 streams -> 1536, 128 and more -> 3072
        if args.resolve_group:
            e.tune(2, args.resolve_group)
        e.tune(3, args.speculate)
        e.tune(4, args.overlap_sort)

    def run_ctx(i, sink=None):
        e, fr = encs[i], parts[i]
        for t in range(GOP):
            e.set_frames_device(fr[t].data_ptr())
            _, _, _, nt = e.encode_picture_device(None)   # AUTO: selectNALUnitType semantics (IDR every GOP)
            if sink is not None:
                sink(i, t, nt)

    def threaded(fn):
        if NC == 1:
            return fn(0)
        err = []

        def wrap(i):
            try:
                fn(i)
            except BaseException as ex:  # noqa: BLE001 -- reported after the join
                err.append(ex)
        th = [threading.Thread(target=wrap, args=(i,)) for i in range(NC)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if err:
            raise err[0]

    def step():
        threaded(run_ctx)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stats0 = [e.stats().copy() for e in encs]
    spec0 = [e.read("SPEC_STAT").copy() for e in encs]
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
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    counts = sum((e.stats() - s0).sum(axis=0) for e, s0 in zip(encs, stats0))   # brojTipova over the timed steps
    spec = sum((e.read("SPEC_STAT") - s0).astype("int64") for e, s0 in zip(encs, spec0))

    total_mbs = total_streams * GOP * nmb * args.steps
    value = total_mbs / dt

    # ---- output check: one more (untimed) GOP through the very path that was timed -- device pictures in, two
    # contexts on two host threads, ferhip_encode_picture_dev -- with the RBSP of every picture read back; stream 0
    # (1080p) or the merged GOP streams (4k) hashed and compared with the committed oracle value
    cap = 2 << 20
    host_rbsp = [torch.empty((GOP, encs[i].S, cap), dtype=torch.uint8).pin_memory() for i in range(NC)]
    host_len = [torch.zeros((GOP, encs[i].S), dtype=torch.int32).pin_memory() for i in range(NC)]
    nal_types = [[None] * GOP for _ in range(NC)]

    def grab(i, t, nt):
        encs[i].copy_rbsp_host(host_rbsp[i][t], host_len[i][t], cap)
        nal_types[i][t] = nt

    threaded(lambda i: run_ctx(i, grab))
    for e in encs:
        e.sync()

    def annexb(i, s):
        sps, pps = encs[i].sps_pps()
        out = bytearray(sps + pps)
        for t in range(GOP):
            n = int(host_len[i][t, s])
            if n > cap:
                raise SystemExit("RBSP larger than the read-back window")
            out += encs[i].write_nal(nal_types[i][t][s], host_rbsp[i][t, s, :n].numpy().tobytes())
        return bytes(out)

    gold = json.loads((ROOT / "tests" / "golden" / "goldens.json").read_text())
    check = {"rbsp_sha256": None, "expected": None, "ok": None}
    if args.config == "4k":
        local_streams = {}
        for i in range(NC):
            for s in range(encs[i].S):
                local_streams[mine[bounds[i] + s]] = annexb(i, s)
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, local_streams)   # host-side concatenation of NAL units, not a data-path collective
            allg = {}
            for g_ in gathered:
                allg.update(g_)
        else:
            allg = local_streams
        if rank == 0:
            merged = pkg.merge_gop_streams([allg[g] for g in range(cfg["ngops"])])
            check["rbsp_sha256"] = hashlib.sha256(merged).hexdigest()
            check["expected"] = gold.get("bench_4k_64f_qp28_w32", {}).get("stream_sha256")
            check["bytes"] = len(merged)
    elif rank == 0 and args.content == "textured":
        s0 = annexb(0, 0)
        check["rbsp_sha256"] = hashlib.sha256(s0).hexdigest()
        check["expected"] = gold.get("bench_1080p_30f_qp12_w32", {}).get("stream_sha256")
        check["bytes"] = len(s0)
        # more streams: one of every ticket queue of the motion chain (local streams 0..7), the last of the first context,
        # the first ones of a second context, the last stream -- whichever of them this run has
        more = gold.get("bench_1080p_30f_qp12_w32_streams", {}).get("stream_sha256", {})
        bad, seen = [], [0]
        for key, want in sorted(more.items(), key=lambda kv: int(kv[0])):
            g_ = int(key)
            if g_ >= S:
                continue
            i = max(k for k in range(NC) if bounds[k] <= g_)
            got = hashlib.sha256(annexb(i, g_ - bounds[i])).hexdigest()
            seen.append(g_)
            if got != want:
                bad.append(g_)
        check["streams_checked"] = seen
        check["streams_bad"] = bad
    if check["expected"] is not None:
        check["ok"] = check["rbsp_sha256"] == check["expected"] and not check.get("streams_bad")
        if not check["ok"] and not args.probe_build:
            raise SystemExit(f"output hash mismatch: {check}")
    del host_rbsp, host_len

    # ---- PCIe-inclusive rate: pinned host pictures in (asynchronous, double-buffered upload), host RBSP out
    value_e2e = None
    if args.e2e and world == 1 and args.config == "1080p":
        pinned = [p.cpu().pin_memory() for p in parts]
        out_h = [torch.empty((2, encs[i].S, cap), dtype=torch.uint8).pin_memory() for i in range(NC)]
        len_h = [torch.zeros((2, encs[i].S), dtype=torch.int32).pin_memory() for i in range(NC)]

        def e2e_ctx(i):
            e, pin = encs[i], pinned[i]
            e.upload_frames(pin[0].data_ptr())
            for t in range(GOP):
                if t + 1 < GOP:
                    e.upload_frames(pin[t + 1].data_ptr())
                e.set_frames_uploaded()
                e.encode_picture_device(None)
                e.copy_rbsp_host(out_h[i][t & 1], len_h[i][t & 1], cap)
            e.sync()

        threaded(e2e_ctx)   # warm-up (allocates the staging buffers)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        threaded(e2e_ctx)
        torch.cuda.synchronize()
        value_e2e = S * GOP * nmb / (time.perf_counter() - t1)
        del pinned, out_h, len_h

    # ---- roofline of the dominant kernel (by accumulated device time; every profiled phase is one kernel, except
    # "sort" = the six launches of the two radix passes and "cavlc" = size + scan + emit)
    pics = {"P": (GOP - 1) * args.steps, "I": args.steps, "all": GOP * args.steps}
    per_mb_traffic, limiters = {}, {}
    for name in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        tj = ROOT / "profiles" / name
        if tj.exists():
            pj = json.loads(tj.read_text())
            per_mb_traffic = pj.get("bytes_per_mb", {})
            limiters = pj.get("limiter", {})
            break

    def line(k):
        name, bpm, on = KERNELS[k]
        limiter = limiters.get(k, "not measured")
        ms_, launches_ = prof.get(k, (0.0, 0))
        units = S * nmb * pics[on]  # accumulated over contexts, like ms_ and launches_
        per_launch_s = (ms_ / 1e3) / max(launches_, 1)
        ach = (bpm * units / max(launches_, 1)) / per_launch_s / 1e9 if ms_ > 0 else 0.0
        tr = per_mb_traffic.get(k)
        return {"kernel": name, "limiter": limiter, "achieved": round(ach, 3), "frac": round(ach / HBM_PEAK_GBS, 6),
                "traffic": round(tr * units / max(launches_, 1)) if tr else None,
                "avg_launch_us": round(per_launch_s * 1e6, 2), "launches": launches_, "bytes_per_mb": bpm,
                "total_ms": round(ms_, 2)}

    live = [k for k in KERNELS if prof.get(k, (0.0, 0))[0] > 0]
    dom = max(live, key=lambda k: prof[k][0])
    dl = line(dom)
    valu_frac = value * LANE_OPS_PER_MB / VALU_LANE_OPS / max(world, 1) if args.config == "1080p" else None
    roofline = {"bound": "hbm", "kernel": dl["kernel"], "limiter": dl["limiter"], "achieved": dl["achieved"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dl["frac"], "traffic": dl["traffic"],
                "avg_launch_us": dl["avg_launch_us"], "launches": dl["launches"], "bytes_per_mb": dl["bytes_per_mb"],
                "valu_frac": round(valu_frac, 5) if valu_frac is not None else None,
                "valu_note": "per-GPU macroblocks/s x 2.1e5 lane-ops per macroblock (reference-exact search, SURVEY 8d) / 78.6e12 lane-ops/s",
                "kernels": {k: line(k) for k in live}}

    out = None
    if rank == 0:
        cpu = cpu_n = None
        if cpu_job is not None:
            # the single-core leg ran beside the GPU part (one of the host's cores; fo_cli clocks the encode itself)
            res = finish_oracle(*cpu_job)
            cpu = {"value": round(res["mb_per_s"], 1), "unit": "macroblocks/s", "cores": 1, "kind": "port",
                   "sample": f"first {args.cpu_frames} pictures (I+{args.cpu_frames - 1}P{', one full GOP' if args.cpu_frames == GOP else ''}) "
                             f"of stream 0, {res['mbs']} MBs in {res['seconds']:.1f} s, oracle/fo_cli single thread, input staged before its clock starts"}
            # N independent processes (the reference's only route to several cores): N = host cores, a bounded sample each,
            # inputs staged before the clock starts
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ncpu = max(1, min(ncpu, 64))
            nf = max(1, min(args.cpu_nproc_frames, GOP))
            stage = [stage_oracle_input(cpu_sample, nf, f"n{k}") for k in range(ncpu)]
            t2 = time.perf_counter()
            jobs = [run_oracle_enc(cfg, t_, nf) for t_ in stage]
            mbs = sum(finish_oracle(p_, t_)["mbs"] for p_, t_ in zip(jobs, stage))
            wall = time.perf_counter() - t2
            cpu_n = {"value": round(mbs / wall, 1), "unit": "macroblocks/s", "cores": ncpu, "kind": "port",
                     "sample": f"{ncpu} fo_cli processes, the first {nf} pictures (I+{nf - 1}P) of stream 0 each, wall {wall:.1f} s, "
                               "inputs staged before the clock starts"}
        secondary = None
        if world == 1 and args.secondary and args.config == "1080p":
            for e in encs:
                e.close()
            encs.clear()
            secondary = run_secondary(pkg, torch, dev)
        coded = int(counts.sum())
        out = {"metric": "macroblocks/sec encode (1080p full-search ME) + bit-exact bitstream vs ref",
               "value": round(value, 1), "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": cfg["workload"], "content": args.content,
                          "coded_size": f"{W}x{H}", "streams_per_gpu": S, "contexts_per_gpu": NC, "gop": GOP, "qp": cfg["qp"],
                          "window": cfg["window"], "maxdiff": cfg["maxdiff"], "mbs_per_step": total_streams * GOP * nmb,
                          "parallelism": f"{world} x {S} independent closed-GOP streams"},
               "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_nproc": cpu_n,
               "value_e2e": round(value_e2e, 1) if value_e2e else None,
               "value_e2e_note": "pinned host pictures in (double-buffered asynchronous H2D), host RBSP out (D2H), same streams and GOP",
               "skip_fraction": round(float(counts[0]) / coded, 6) if coded else None,
               "mb_types": {"p_skip": int(counts[0]), "p16x16": int(counts[1]), "p16x8": int(counts[2]),
                            "p8x16": int(counts[3]), "p8x8": int(counts[4])},
               "speculation": {"partitions_decided_by_chain": int(spec[0]), "predictor_guess_hits": int(spec[1]),
                               "hit_rate": round(float(spec[1]) / max(int(spec[0]), 1), 5),
                               "pskip_verdicts": int(spec[2]), "pskip_guess_hits": int(spec[3])},
               "output_check": check, "secondary": secondary}
        print(json.dumps(out), flush=True)
    for e in encs:
        e.close()
    if world > 1:
        dist.destroy_process_group()


def rehearse_ranks(args, rank, world, dist, torch, pkg):
    """The multi-rank plumbing of a --gpus N run without a GPU and without an encoder (nothing is encoded: the per-GOP
    "streams" are placeholder bytes): process-group rendezvous, the barrier + max-over-ranks reduction of the step time,
    the --config 4k host-side gather of per-GOP streams and their merge in GOP order."""
    cfg = CONFIGS[args.config]
    if world > 1:
        dist.init_process_group("gloo" if args.dist_backend != "nccl" or not torch.cuda.is_available() else "nccl")
        dist.barrier()
    dt = 1.0 + rank
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ngops = cfg.get("ngops", world)
    mine = pkg.gops_of_rank(ngops, world, rank)
    local_streams = {g: b"\x00\x00\x00\x01" + bytes([0x65, g, rank]) for g in mine}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, local_streams)
        allg = {}
        for g_ in gathered:
            allg.update(g_)
    else:
        allg = local_streams
    if rank == 0:
        order = [allg[g][5] for g in range(ngops)]
        owners = [allg[g][6] for g in range(ngops)]
        print(json.dumps({"rehearsal": True, "world": world, "max_dt": dt, "gops_in_order": order == list(range(ngops)),
                          "gop_owner": owners}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_secondary(pkg, torch, dev):
    """BASELINE configs[1] (720p I-only, many pictures in flight) and configs[4] (1080p decode), short runs."""
    import numpy as np
    from h264_fer_amd.synth import gen_frames_torch
    res = {}
    # ---- 720p I-frame encode: S streams x T pictures, IntraEvery 1
    W, H, S, T = 1280, 720, 256, 4
    fr = gen_frames_torch(W, H, 1, S, dev, seed=77, noise=2)[0].contiguous()
    e = pkg.FerHip(W, H, S, qp=12, window=16, maxdiff=3, intra_every=1)
    for _ in range(1):
        e.set_frames_device(fr.data_ptr())
        e.encode_picture_device(None)
    e.sync()
    t0 = time.perf_counter()
    for _ in range(T):
        e.set_frames_device(fr.data_ptr())
        e.encode_picture_device(None)
    e.sync()
    dt = time.perf_counter() - t0
    ok = not any(e.status())
    res["720p_intra"] = {"workload": "720p I-frame encode (BASELINE configs[1])", "value": round(S * T * e.nmb / dt, 1),
                         "unit": "macroblocks/s", "streams": S, "pictures": T, "qp": 12, "status_ok": ok}
    e.close()
    del fr
    # ---- 1080p decode of the encoder's own IPPP output (host Annex-B in, pictures left in HBM)
    W, H, S, T = 1920, 1072, 16, 30  # one full GOP: the I picture's 252 anti-diagonal launches are amortised as in the encoder's workload
    fr = gen_frames_torch(W, H, T, S, dev, seed=1234, noise=2).cpu().numpy()
    e = pkg.FerHip(W, H, S, qp=12, window=32, maxdiff=3, intra_every=30)
    streams, _ = e.encode_streams(fr)
    nmb = e.nmb
    e.close()
    reps = 8
    batch = streams * reps
    pkg.decode_streams(batch, T, want_pictures=False)   # first call allocates the window arena
    t0 = time.perf_counter()
    _, pics, _, _ = pkg.decode_streams(batch, T, want_pictures=False)
    dt = time.perf_counter() - t0
    res["1080p_decode"] = {"workload": "1080p decode (CAVLC parse + dequant + inverse 4x4), BASELINE configs[4]",
                           "value": round(len(batch) * T * nmb / dt, 1), "unit": "macroblocks/s", "streams": len(batch),
                           "pictures": T, "all_decoded": pics == [T] * len(batch)}
    pkg.load_library().ferhip_decode_release()
    return res


if __name__ == "__main__":
    main()
