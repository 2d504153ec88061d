"""Turns the raw outputs of tools/prof_r03.sh (gpurun_out/r03/) into the committed summaries under profiles/:
r03_kernel_stats_bench.md, r03_hbm_traffic_pmc.md, r03_sq_counters.md, r03_traffic.json (HBM bytes per macroblock and the
limiter of every phase, derived from the counters), r03_bench_prof.json (the bench line of the profiled run)."""
import collections
import csv
import glob
import json
import os
from pathlib import Path


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


root = Path(__file__).resolve().parent.parent
go = root / "gpurun_out" / "r03"
prof = root / "profiles"
NMB = 8040
HBM_COPY = 6.29e12          # bytes/s a copy reaches (MI355X_MICROARCH.md)
CLK = 2.37e9                # measured under this load (tools/clock_watch.sh)
SIMDS, CUS = 1024, 256


def first_json(path):
    for line in open(path).read().splitlines():
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line[:line.rfind("}") + 1])
    raise SystemExit(f"no JSON line in {path}")


b1 = first_json(go / "bench_prof1.json")
(prof / "r03_bench_prof.json").write_text(json.dumps({"one_context_256_streams": b1}) + "\n")
S_BENCH = b1["config"]["streams_per_gpu"]
rows = list(csv.DictReader(open(newest(str(go / "stats1/*/*_kernel_stats.csv")))))
md = ["# Round 3 — rocprofv3 kernel summary of the benchmark command", "",
      "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0`", "",
      "1080p IPPP, ONE context of %d streams (the default: every kernel alone on the GPU), one warm-up GOP + one timed GOP + the "
      "output-check GOP; MI355X.  Bench line of the same run: %.2f M MB/s, roofline kernel `%s` (average launch %.1f us by HIP events "
      "inside bench.py).  The `at::native` kernels belong to the synthetic-input generator." % (S_BENCH, b1["value"] / 1e6,
                                                                                             b1["roofline"]["kernel"], b1["roofline"]["avg_launch_us"]), "",
      "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---|---|---|---|"]
avg_ns = {}
for r in rows:
    short = r["Name"].split("(")[0].replace("void ", "")
    avg_ns[short] = float(r["AverageNs"])
for r in rows[:28]:
    md.append("| `%s` | %s | %.3f | %.2f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
md += ["", "HIP-event averages of the same run (bench.py `roofline.kernels`, timed GOP only; `sort` = six launches of the two radix "
       "passes, `sort_finish` = index + classes + mis-filed layout, `cavlc` = size + scan + emit):", "",
       "| phase | kernel | avg launch us (HIP events) | launches | algorithmic GB/s | limiter (from the counters below) |", "|---|---|---|---|---|---|"]


def load(path, counter):
    agg = collections.defaultdict(float)
    cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            cnt[k] += 1
    return agg, cnt


fa, fc = load(newest(str(go / "pmc_fetch/*/*_counter_collection.csv")), "FETCH_SIZE")
wa, wc = load(newest(str(go / "pmc_write/*/*_counter_collection.csv")), "WRITE_SIZE")
S = 16
alg = {"k_interp": 4352, "k_feat0": 256 + 256 * 14, "k_me_pre": 528, "k_me_walk": 528, "k_me_spec": 528, "k_me_resolve": 528, "k_p_resid": 1152,
       "k_intra_mb": 768, "k_cavlc": 800, "k_frame_sad": 512, "k_rs_scatter<0>": 256 * 18, "k_rs_scatter<1>": 256 * 35}
tmd = ["# Round 3 — HBM traffic counters (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)", "",
       "`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/quick_hd.py 16 2` (and `WRITE_SIZE`): 1080p, 16 streams, I+P, two encodes.",
       "FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, the "
       "`x2` column applies that correction (uncalibrated for narrow accesses).  Bytes per macroblock = (2 x FETCH + WRITE) per dispatch "
       "x dispatches of the kernel per picture / (16 streams x 8040 macroblocks).", "",
       "| kernel | dispatches | per picture | FETCH MB/dispatch | x2 | WRITE MB/dispatch | (2F+W) bytes per MB and picture | algorithmic bytes per MB |",
       "|---|---|---|---|---|---|---|---|"]
per_mb = {}
for k in sorted(fa, key=lambda k: -fa[k]):
    n = fc[k]
    f = fa[k] * 1024 / n / 1e6
    w = wa.get(k, 0) * 1024 / max(wc.get(k, 1), 1) / 1e6
    per_pic = 254 if k.startswith("k_intra") else (2 if k.startswith(("k_rs_hist", "k_rs_scan")) else 1)
    bpm = (2 * f + w) * 1e6 * per_pic / (S * NMB)
    a = [v for kk, v in alg.items() if k.startswith(kk)]
    tmd.append("| `%s` | %d | %d | %.2f | %.2f | %.2f | %.0f | %s |" % (k[:40], n, per_pic, f, 2 * f, w, bpm, a[0] if a else ""))
    per_mb[k] = bpm
(prof / "r03_hbm_traffic_pmc.md").write_text("\n".join(tmd) + "\n")


def g(*prefixes):
    return sum(v for k, v in per_mb.items() if any(k.startswith(p) for p in prefixes))


bytes_per_mb = {"interp": round(g("k_interp")), "sort_keys": round(g("k_feat0")), "sort": round(g("k_rs_")),
                "sort_finish": round(g("k_sort_index", "k_bucket_classes", "k_sort_quirk")),
                "me_pre": round(g("k_me_pre")), "me_walk": round(g("k_me_walk")), "me_spec": round(g("k_me_spec")),
                "me_resolve": round(g("k_me_resolve")), "p_resid": round(g("k_p_resid")), "intra": round(g("k_intra")),
                "frame_sad": round(g("k_frame_sad")), "cavlc": round(g("k_cavlc", "k_bits"))}

# SQ counters
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(newest(str(go / "pmc_sq/*/*_counter_collection.csv")))):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"]))
        cnt[k] += 1
SQS = 32
parts = SQS * 32160
smd = ["# Round 3 — SQ counters per kernel (where the wave cycles go)", "",
       "`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU "
       "SQ_INSTS_SALU SQ_BUSY_CYCLES -- python3 tools/quick_hd.py 32 2` (1080p, 32 streams, one I + one P picture, encoded twice; "
       "counters in their own pass).  Instructions per 8x8 partition = per dispatch / (32 streams x 32 160 partitions).  Round 2: "
       "`k_me_pre` 1 891 VALU / 647 SALU, `k_me_walk` 1 607 / 1 521, `k_me_resolve` 2 149 / 1 200 per partition.", "",
       "| kernel | dispatches | VALU / partition | SALU / partition | VALU active % | any inst active % | parked (s_waitcnt/barrier) % | issue-stalled % |",
       "|---|---|---|---|---|---|---|---|"]
inst = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not k.startswith("k_"):
        continue
    n, wcy = cnt[k], v["SQ_WAVE_CYCLES"]
    inst[k] = (v["SQ_INSTS_VALU"] / n / parts, v["SQ_INSTS_SALU"] / n / parts)
    smd.append("| `%s` | %d | %.0f | %.0f | %.1f | %.1f | %.1f | %.1f |" % (
        k[:30], n, inst[k][0], inst[k][1], 100 * v["SQ_ACTIVE_INST_VALU"] / wcy,
        100 * v["SQ_ACTIVE_INST_ANY"] / wcy, 100 * v["SQ_WAIT_ANY"] / wcy, 100 * v["SQ_WAIT_INST_ANY"] / wcy))
smd += ["", "For a per-macroblock kernel multiply by 4 (k_p_resid: the line above x 4 wave instructions per macroblock); `k_intra_mb` is "
        "one dispatch per anti-diagonal: its per-macroblock count is the line x 4 x 252 dispatches of a picture."]

# limiter of every phase: the largest of (HBM bytes/s : 6.29 TB/s), (VALU wave instructions x 4 cycles : SIMD cycles), (scalar
# instructions : one per CU cycle), each over the kernel's duration in the benchmark run; below 0.5 everywhere = latency
phase_kernels = {"interp": ["k_interp", "k_interp_pad"], "sort_keys": ["k_feat0"], "sort": ["k_rs_hist", "k_rs_scan", "k_rs_scatter<0>", "k_rs_scatter<1>"],
                 "sort_finish": ["k_sort_index", "k_bucket_classes", "k_sort_quirk"], "me_pre": ["k_me_pre<32>"], "me_walk": ["k_me_walk"],
                 "me_spec": ["k_me_spec<32>"], "me_resolve": ["k_me_resolve<32>"], "p_resid": ["k_p_resid"], "intra": ["k_intra_mb"],
                 "cavlc": ["k_cavlc<false>", "k_bits_scan", "k_cavlc<true>"], "frame_sad": ["k_frame_sad"]}
limiter, fracs = {}, {}
for ph, ks in phase_kernels.items():
    t = 0.0
    valu = salu = 0.0
    for k in ks:
        per_pic = 254 if k.startswith("k_intra") else (2 if k.startswith(("k_rs_hist", "k_rs_scan")) else 1)
        t += avg_ns.get(k, 0.0) * 1e-9 * per_pic / (S_BENCH * NMB)          # seconds per macroblock
        if k in inst:
            valu += inst[k][0] * 4 * per_pic
            salu += inst[k][1] * 4 * per_pic
    if t <= 0:
        continue
    f_hbm = bytes_per_mb.get(ph, 0) / t / HBM_COPY
    f_valu = valu * 4 / (SIMDS * CLK) / t
    f_salu = salu / (CUS * CLK) / t
    best = max((f_hbm, "hbm"), (f_valu, "valu"), (f_salu, "salu"))
    limiter[ph] = best[1] if best[0] >= 0.5 else "latency"
    fracs[ph] = {"hbm": round(f_hbm, 3), "valu": round(f_valu, 3), "salu": round(f_salu, 3)}
smd += ["", "## What limits each phase", "",
        "Over the kernel's duration in the benchmark run (kernel summary, %d streams): HBM = (2F+W) bytes per second against the "
        "6.29 TB/s a copy reaches; VALU = wave instructions x 4 cycles against the SIMD cycles (4 cycles per instruction is what "
        "these kernels' mixes sustain, `r02_valu_throughput.md`); scalar = scalar instructions against one per CU cycle at %.2f GHz. "
        "The largest fraction names the limiter when it reaches 0.5, `latency` otherwise (`r03_traffic.json`, read by bench.py)." % (S_BENCH, CLK / 1e9), "",
        "| phase | HBM | VALU issue | scalar issue | limiter |", "|---|---|---|---|---|"]
for ph in phase_kernels:
    if ph in fracs:
        smd.append("| %s | %.2f | %.2f | %.2f | %s |" % (ph, fracs[ph]["hbm"], fracs[ph]["valu"], fracs[ph]["salu"], limiter[ph]))
(prof / "r03_sq_counters.md").write_text("\n".join(smd) + "\n")

tj = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 tools/quick_hd.py 16 2`, see "
                 "r03_hbm_traffic_pmc.md; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch x dispatches per picture divided by the "
                 "macroblocks of a picture; the x2 FETCH correction of MI355X_MICROARCH.md is calibrated for wide coalesced reads only. "
                 "limiter: see r03_sq_counters.md",
      "bytes_per_mb": bytes_per_mb, "limiter": limiter, "fractions": fracs}
(prof / "r03_traffic.json").write_text(json.dumps(tj, indent=1) + "\n")
for k, v in b1["roofline"]["kernels"].items():
    md.append("| %s | `%s` | %.1f | %d | %.1f | %s |" % (k, v["kernel"], v["avg_launch_us"], v["launches"], v["achieved"], limiter.get(k, "")))
(prof / "r03_kernel_stats_bench.md").write_text("\n".join(md) + "\n")
print("bench", b1["value"])
print("traffic", bytes_per_mb)
print("limiter", limiter)
print("fractions", fracs)

# decode leg
try:
    drows = list(csv.DictReader(open(newest(str(go / "dec/*/*_kernel_stats.csv")))))
    rates = [l.strip() for l in open(go / "dec.log") if l.startswith("streams")]
    dmd = ["# Round 3 — rocprofv3 kernel summary of the 1080p decode leg", "",
           "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/dec_rate.py 8`: 128 streams x one 30-picture GOP of the "
           "encoder's own 1080p IPPP output (bench.py's secondary configuration), decoded three times (one warm-up).", "",
           "Rates printed by the same run: " + "; ".join(rates), "",
           "| kernel | calls | total ms | avg us |", "|---|---|---|---|"]
    for r in drows:
        if "k_dec" in r["Name"]:
            dmd.append("| `%s` | %s | %.1f | %.1f |" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    (prof / "r03_kernel_stats_decode.md").write_text("\n".join(dmd) + "\n")
except Exception as e:  # the decode pass is optional
    print("decode profile skipped:", e)
