# Dev helper (GPU box): rocprofv3 kernel summary of the decode leg.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/psd
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -- python3 tools/dec_rate.py $1 > $O/a.log 2>&1
f=$(find $O/a -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("k_dec") or "k_dec" in r["Name"]:
        print(f'{r["Name"][:50]:50s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"])/1e6:10.1f} avg_us {float(r["AverageNs"])/1e3:10.1f}')
PY
find $O -name '*.csv' -delete; tail -2 $O/a.log
