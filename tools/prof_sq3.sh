# SQ counter passes for the motion kernels (round 3): run from the repo root through gpurun
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03sq
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $O/a -- python3 tools/quick_hd.py 32 2 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $O/b -- python3 tools/quick_hd.py 32 2 > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/c -- python3 tools/quick_hd.py 32 2 > $O/c.log 2>&1
for p in a b c; do f=$(find $O/$p -name '*counter_collection.csv' | head -1); python3 tools/pmc_summary.py $f 32 > $O/sum_$p.txt; done
find $O -name '*.csv' -size +20M -delete
find $O -name '*_agent_info.csv' -delete
du -sh $O
