"""Turn the raw outputs of a profiling run under gpurun_out/ into the committed summaries in profiles/:
bench_r01.json + bench_prof2.json (bench lines), prof_bench2 (rocprofv3 --stats of the bench command),
pmc_fetch2 / pmc_write2 (FETCH_SIZE / WRITE_SIZE passes of tools/quick_hd.py 16 2)."""
import collections
import csv
import glob
import json
from pathlib import Path

root = Path(__file__).resolve().parent.parent
go = root / "gpurun_out"


def first_json(path):
    for line in open(path).read().splitlines():
        line = line.strip()
        if line.startswith("{"):
            end = line.rfind("}")
            return json.loads(line[:end + 1])
    raise SystemExit(f"no JSON line in {path}")


b = first_json(go / "bench_r01.json")
bp = first_json(go / "bench_prof2.json")
(root / "profiles" / "r01_bench.json").write_text(json.dumps(b) + "\n")

rows = list(csv.DictReader(open(glob.glob(str(go / "prof_bench2/*/*_kernel_stats.csv"))[0])))
out = ["# Round 1 — rocprofv3 kernel summary of the benchmark command", "",
       "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0`",
       "(defaults: 1080p IPPP, 128 streams in two contexts of 64, one warm-up GOP + one timed GOP; MI355X; bench line of the "
       "same run: %.2f M MB/s, roofline kernel `%s`: average launch %.1f us by HIP events inside bench.py)"
       % (bp["value"] / 1e6, bp["roofline"]["kernel"], bp["roofline"]["avg_launch_us"]), "",
       "| kernel | calls | total ms | avg us | % of kernel time |", "|---|---|---|---|---|"]
for r in rows[:24]:
    out.append("| `%s` | %s | %.3f | %.2f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
out += ["", "HIP-event averages of the same run (bench.py `roofline.kernels`, timed GOP only; rocprofv3 above also counts "
        "the warm-up GOP; `sort` and `cavlc` are groups of launches):", "",
        "| phase | avg launch us (HIP events) | launches |", "|---|---|---|"]
for k, v in bp["roofline"]["kernels"].items():
    out.append("| %s (`%s`) | %.1f | %d |" % (k, v["kernel"], v["avg_launch_us"], v["launches"]))
out += ["", "Notes: the two contexts overlap on the GPU, so per-kernel totals add up to more than the wall time and a kernel's "
        "duration includes the time it shares the CUs with the other context's kernels (`k_features` alone on the GPU takes "
        "7.7 ms per 64-stream launch, 3.8 TB/s of algorithmic bytes; next to the other context two to three times as long). "
        "Each P picture of a context is one launch of every `k_me_*` / `k_features` / `k_p_resid` kernel (64 streams per "
        "launch); `k_intra_mb` is launched once per macroblock anti-diagonal of an I picture; the `at::native` kernels belong "
        "to the synthetic-input generator.", ""]
(root / "profiles" / "r01_kernel_stats_bench.md").write_text("\n".join(out))


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


fa, fc = load(glob.glob(str(go / "pmc_fetch2/*/*_counter_collection.csv"))[0], "FETCH_SIZE")
wa, wc = load(glob.glob(str(go / "pmc_write2/*/*_counter_collection.csv"))[0], "WRITE_SIZE")
S, nmb = 16, 8040
alg = {"k_features": 56320, "k_interp": 4352, "k_me_pre": 528, "k_me_walk": 528, "k_me_resolve": 528, "k_p_resid": 1152,
       "k_intra_mb": 768, "k_cavlc": 800, "k_frame_sad": 512}
md = ["# Round 1 — HBM traffic counters (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes)", "",
      "`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/quick_hd.py 16 2` (and `WRITE_SIZE`): 1080p, 16 streams, "
      "I+P, two encodes.",
      "FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on "
      "gfx950, the `x2` column applies that correction (uncalibrated for narrow accesses). Per-MB = (2 x FETCH + WRITE) / "
      "(16 streams x 8040 macroblocks) per dispatch (per launch group for the per-diagonal intra kernel).", "",
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
(root / "profiles" / "r01_hbm_traffic_pmc.md").write_text("\n".join(md) + "\n")


def g(prefix):
    return sum(v for k, v in per_mb.items() if k.startswith(prefix))


tj = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 tools/quick_hd.py 16 2`, see "
                 "r01_hbm_traffic_pmc.md; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch divided by the macroblocks one "
                 "dispatch processes; the x2 FETCH correction of MI355X_MICROARCH.md is calibrated for wide coalesced reads only",
      "bytes_per_mb": {"interp": round(g("k_interp")), "features": round(g("k_features")), "sort_keys": round(g("k_sort_keys")), "sort": round(g("k_rs")), "sort_finish": round(g("k_sort_finish")),
                       "me_pre": round(g("k_me_pre")), "me_walk": round(g("k_me_walk")), "me_resolve": round(g("k_me_resolve")),
                       "p_resid": round(g("k_p_resid")), "intra": round(g("k_intra")),
                       "cavlc": round(g("k_cavlc") + g("k_bits"))}}
(root / "profiles" / "r01_traffic.json").write_text(json.dumps(tj, indent=1) + "\n")
print("bench", b["value"], b["roofline"]["kernel"], b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline"]["traffic"])
print("traffic", tj["bytes_per_mb"])
