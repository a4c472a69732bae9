#!/bin/bash
# rocprofv3 kernel statistics of the full chain for one batch size / clock-recovery shape (run on the GPU box)
# usage: bash tools/gpu_chain_prof.sh TAG S [bench_chain.py options]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
tag=$1; shift
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/tools/bench_chain.py "$@" > $O/${tag}_bench.log 2>&1 || echo "(profiler exit code $?)"
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
grep samples_per_stream $O/${tag}_bench.log | cut -c60-300
head -4 $O/${tag}_kernel_stats.csv | cut -c1-60,100-400
