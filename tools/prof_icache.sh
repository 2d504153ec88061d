# instruction-cache counters of the motion kernels (run from the repo root through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ic
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU --output-format csv -d $O/a -- python3 tools/quick_hd.py 32 2 > $O/a.log 2>&1
f=$(find $O/a -name '*counter_collection.csv' | head -1); python3 tools/pmc_summary.py $f 32 > $O/sum_a.txt
find $O -name '*.csv' -size +20M -delete
cat $O/sum_a.txt | head -14
