# Collects the raw material of profiles/r03_* on the GPU box (run from the repo root through gpurun).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --secondary 0 --e2e 0 > $O/bench_prof1.json 2> $O/bench_prof1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/quick_hd.py 16 2 > $O/q_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/quick_hd.py 16 2 > $O/q_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 tools/quick_hd.py 32 2 > $O/q_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python3 tools/dec_rate.py 8 > $O/dec.log 2> $O/dec.err
find $O -name '*_kernel_trace.csv' -delete
find $O -name '*_agent_info.csv' -delete
du -sh $O; echo done; ls $O
