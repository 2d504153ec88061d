# Dev helper (GPU box): SQ counters of the decode kernels.  usage: bash tools/prof_pmc_dec.sh "SQ_WAVE_CYCLES SQ_INSTS_SALU ..."
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pd
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $O/a -- python3 tools/dec_rate.py $2 > $O/a.log 2>&1
f=$(find $O/a -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not k.startswith('k_dec'): continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); cnt[k] += 1
for k, v in agg.items():
    print(k, cnt[k], {c: "%.4g" % (x / cnt[k]) for c, x in v.items()})
PY
find $O -name '*.csv' -delete
