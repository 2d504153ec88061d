# Dev helper (GPU box): rocprofv3 kernel summary of a short 1080p run.  usage: bash tools/prof_stats.sh [streams]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ps
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -- python3 tools/quick_hd.py ${1:-128} 2 > $O/a.log 2>&1
f=$(find $O/a -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    if r["Name"].startswith("void at::"): continue
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} min {float(r["MinNs"])/1e3:10.1f} max {float(r["MaxNs"])/1e3:10.1f}')
PY
find $O -name '*.csv' -delete
