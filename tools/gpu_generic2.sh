mkdir -p gpurun_out
timeout -k 10 300 python tools/dbg/generic_rate.py 2>/dev/null > gpurun_out/generic_rate.log; cat gpurun_out/generic_rate.log
timeout -k 10 300 python tools/dbg/generic_rate.py 100000000 2>/dev/null > gpurun_out/generic_rate100.log; cat gpurun_out/generic_rate100.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gen -- python3 $GRAFT_REPO_ROOT/tools/dbg/generic_rate.py > /dev/null 2>&1; cp $(find /tmp/prof_gen -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/generic_kernel_stats.csv; grep grhip $GRAFT_REPO_ROOT/gpurun_out/generic_kernel_stats.csv | cut -c1-60,120-300
