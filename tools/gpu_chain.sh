mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in 8 64 256; do python3 $R/tools/bench_chain.py $S 10000000 2>/dev/null | tail -1; done | tee $R/gpurun_out/chain_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_chain -- python3 $R/tools/bench_chain.py 64 10000000 > $R/gpurun_out/prof_chain.log 2>&1
find $R/gpurun_out/prof_chain -name "*kernel_stats.csv" | head -1 | xargs cut -c1-200 | head -12
