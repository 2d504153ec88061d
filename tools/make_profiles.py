"""Turns the raw outputs of tools/prof_r02.sh (gpurun_out/r02/) into the committed summaries under profiles/:
r02_kernel_stats_bench.md (+ one-context table), r02_hbm_traffic_pmc.md, r02_traffic.json, r02_sq_counters.md,
r02_bench_prof.json (the bench lines of the profiled runs)."""
import collections
import csv
import glob
import json
from pathlib import Path

import os


def newest(pattern):
    """gpurun merges new files into gpurun_out/ without removing those of earlier runs: take the latest"""
    return max(glob.glob(pattern), key=os.path.getmtime)


root = Path(__file__).resolve().parent.parent
go = root / "gpurun_out" / "r02"
prof = root / "profiles"


def first_json(path):
    for line in open(path).read().splitlines():
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line[:line.rfind("}") + 1])
    raise SystemExit(f"no JSON line in {path}")


def stats_table(d, title, cmd, bl):
    rows = list(csv.DictReader(open(newest(str(go / d / "*/*_kernel_stats.csv")))))
    out = [f"## {title}", "", f"`{cmd}`", "",
           "(bench line of the same run: %.2f M MB/s, roofline kernel `%s`: average launch %.1f us by HIP events inside bench.py)"
           % (bl["value"] / 1e6, bl["roofline"]["kernel"], bl["roofline"]["avg_launch_us"]), "",
           "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---|---|---|---|"]
    for r in rows[:26]:
        out.append("| `%s` | %s | %.3f | %.2f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, r["Percentage"]))
    out += ["", "HIP-event averages of the same run (bench.py `roofline.kernels`, timed GOP only; rocprofv3 above also counts the "
            "warm-up GOP and the output-check GOP; `sort` = six launches of the two radix passes, `sort_finish` = index + ranges + "
            "mis-filed layout, `cavlc` = size + scan + emit):", "",
            "| phase | kernel | avg launch us (HIP events) | launches | algorithmic GB/s | limiter |", "|---|---|---|---|---|---|"]
    for k, v in bl["roofline"]["kernels"].items():
        out.append("| %s | `%s` | %.1f | %d | %.1f | %s |" % (k, v["kernel"], v["avg_launch_us"], v["launches"], v["achieved"], v["limiter"]))
    return out


b2 = first_json(go / "bench_prof2.json")
b1 = first_json(go / "bench_prof1.json")
(prof / "r02_bench_prof.json").write_text(json.dumps({"two_contexts": b2, "one_context": b1}) + "\n")
md = ["# Round 2 — rocprofv3 kernel summaries of the benchmark command", "",
      "1080p IPPP, one warm-up GOP + one timed GOP (+ the output-check GOP); MI355X.  The `at::native` kernels belong "
      "to the synthetic-input generator.", ""]
md += stats_table("stats2", "Default: 256 streams in two contexts of 128",
                  "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0", b2)
md += ["", "With two contexts the kernels of one context share the CUs with the other's: a kernel's duration includes that "
       "(the streaming kernels of the reference preparation stretch most), and the per-kernel totals add up to more than the wall "
       "time.  The isolated table follows.", ""]
md += stats_table("stats1", "One context of 128 streams (every kernel alone on the GPU)",
                  "... bench.py --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0 --contexts 1 --streams 128", b1)
(prof / "r02_kernel_stats_bench.md").write_text("\n".join(md) + "\n")


def load(path, counter):
    agg = collections.defaultdict(float)
    cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        agg[k] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            cnt[k] += 1
    return agg, cnt


fa, fc = load(newest(str(go / "pmc_fetch/*/*_counter_collection.csv")), "FETCH_SIZE")
wa, wc = load(newest(str(go / "pmc_write/*/*_counter_collection.csv")), "WRITE_SIZE")
S, nmb = 16, 8040
alg = {"k_interp": 4352, "k_feat0": 256 + 256 * 30, "k_me_pre": 528, "k_me_walk": 528, "k_me_resolve": 528, "k_p_resid": 1152,
       "k_intra_mb": 768, "k_cavlc": 800, "k_frame_sad": 512, "k_rs_scatter": 256 * 33}
md = ["# Round 2 — HBM traffic counters (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)", "",
      "`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/quick_hd.py 16 2` (and `WRITE_SIZE`): 1080p, 16 streams, I+P, two encodes.",
      "FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, the "
      "`x2` column applies that correction (uncalibrated for narrow accesses).  Per-MB = (2 x FETCH + WRITE) / (16 streams x 8040 "
      "macroblocks) per dispatch (per launch group for the per-diagonal intra kernel).  Round 1 for comparison: `k_me_resolve` "
      "134 568, `k_me_pre` 81 941, `k_me_walk` 66 978, `k_features` 56 510 (gone), `k_sort_finish` 26 604, `k_sort_keys` 18 051.", "",
      "| kernel | dispatches | FETCH MB/dispatch | x2 | WRITE MB/dispatch | (2F+W) bytes per MB | algorithmic bytes per MB |",
      "|---|---|---|---|---|---|---|"]
per_mb = {}
for k in sorted(fa, key=lambda k: -fa[k]):
    n = fc[k]
    f = fa[k] * 1024 / n / 1e6
    w = wa.get(k, 0) * 1024 / max(wc.get(k, 1), 1) / 1e6
    short = k.split("(")[0].replace("void ", "")
    disp_per_pic = 254 if "intra" in short else 1
    bpm = (2 * f + w) * 1e6 * disp_per_pic / (S * nmb)
    a = [v for kk, v in alg.items() if short.startswith(kk)]
    md.append("| `%s` | %d | %.2f | %.2f | %.2f | %.0f | %s |" % (short[:40], n, f, 2 * f, w, bpm, a[0] if a else ""))
    per_mb[short] = bpm
(prof / "r02_hbm_traffic_pmc.md").write_text("\n".join(md) + "\n")


def g(prefix):
    return sum(v for k, v in per_mb.items() if k.startswith(prefix))


tj = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 tools/quick_hd.py 16 2`, see "
                 "r02_hbm_traffic_pmc.md; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch divided by the macroblocks one "
                 "dispatch processes; the x2 FETCH correction of MI355X_MICROARCH.md is calibrated for wide coalesced reads only",
      "bytes_per_mb": {"interp": round(g("k_interp")), "sort_keys": round(g("k_feat0")), "sort": round(g("k_rs")),
                       "sort_finish": round(g("k_sort_index") + g("k_bucket_ranges") + g("k_sort_quirk")),
                       "me_pre": round(g("k_me_pre")), "me_walk": round(g("k_me_walk")), "me_resolve": round(g("k_me_resolve")),
                       "p_resid": round(g("k_p_resid")), "intra": round(g("k_intra")), "frame_sad": round(g("k_frame_sad")),
                       "cavlc": round(g("k_cavlc") + g("k_bits"))}}
(prof / "r02_traffic.json").write_text(json.dumps(tj, indent=1) + "\n")

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
parts = 32 * 32160
md = ["# Round 2 — SQ counters per kernel (where the wave cycles go)", "",
      "`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU "
      "SQ_INSTS_SALU SQ_BUSY_CYCLES -- python3 tools/quick_hd.py 32 2` (1080p, 32 streams, one I + one P picture, encoded twice; "
      "counters in their own pass).  Instructions per 8x8 partition = per dispatch / (32 streams x 32 160 partitions).  Round 1: "
      "`k_me_pre` 2 200 VALU, `k_me_walk` 2 100 VALU / 1 830 SALU, `k_me_resolve` 2 150 VALU per partition.", "",
      "| kernel | dispatches | VALU / partition | SALU / partition | VALU active % | any inst active % | parked (s_waitcnt/barrier) % | issue-stalled % |",
      "|---|---|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not k.startswith("k_"):
        continue
    n, wcy = cnt[k], v["SQ_WAVE_CYCLES"]
    md.append("| `%s` | %d | %.0f | %.0f | %.1f | %.1f | %.1f | %.1f |" % (
        k[:30], n, v["SQ_INSTS_VALU"] / n / parts, v["SQ_INSTS_SALU"] / n / parts, 100 * v["SQ_ACTIVE_INST_VALU"] / wcy,
        100 * v["SQ_ACTIVE_INST_ANY"] / wcy, 100 * v["SQ_WAIT_ANY"] / wcy, 100 * v["SQ_WAIT_INST_ANY"] / wcy))
md += ["", "For a per-macroblock kernel multiply by 4 (k_p_resid: the line above x 4 wave instructions per macroblock); `k_intra_mb` is "
       "one dispatch per anti-diagonal: its per-macroblock count is the line x 4 x 252 dispatches of a picture.",
       "", "The scalar unit is one per CU: `k_me_walk` at 1 830 scalar instructions per partition issued 0.9 of them per CU cycle -- "
       "it was scalar-issue bound, which is why its control flow moved into the lanes (batch tables) this round.  At the counts above "
       "the three motion kernels are VALU-issue bound: instructions x partitions x 4 cycles / (1024 SIMDs x clock) is 85-100 % of "
       "the measured duration of `k_me_pre` and `k_me_walk`, and about 50 % for `k_me_resolve` (whose row chain keeps a fifth of its "
       "workgroups waiting at the start and end of every stream group)."]
(prof / "r02_sq_counters.md").write_text("\n".join(md) + "\n")
print("bench 2ctx", b2["value"], "1ctx", b1["value"])
print("traffic", tj["bytes_per_mb"])
