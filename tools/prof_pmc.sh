# Dev helper (GPU box): one --pmc pass over a short 1080p run, per-kernel summary.  usage: bash tools/prof_pmc.sh "SQ_WAVE_CYCLES SQ_INSTS_VALU ..." [streams]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pp
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $O/a -- python3 tools/quick_hd.py ${2:-32} 2 > $O/a.log 2>&1
f=$(find $O/a -name '*counter_collection.csv' | head -1); python3 tools/pmc_summary.py $f ${2:-32} > $O/sum.txt
find $O -name '*.csv' -delete
cat $O/sum.txt
