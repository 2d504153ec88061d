# VALU / SALU instruction counts per partition of the motion kernels (run through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pq
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 tools/quick_hd.py 32 2 > $O/a.log 2>&1
f=$(find $O/a -name '*counter_collection.csv' | head -1); python3 tools/pmc_summary.py $f 32 > $O/sum.txt
find $O -name '*.csv' -delete
head -8 $O/sum.txt
