#!/bin/bash
# rocprofv3 kernel statistics of the channeliser at 32 / 64 / 128 channels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_pfb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pfb -- python3 $R/tools/dbg/pfb_sizes.py > $R/gpurun_out/pfb_sizes_prof.log 2>&1 || echo "(profiler exit code $?)"
cp $(find /tmp/prof_pfb -name "*kernel_stats.csv" | head -1) $R/gpurun_out/pfb_sizes_kernel_stats.csv
grep "M=" $R/gpurun_out/pfb_sizes_prof.log
