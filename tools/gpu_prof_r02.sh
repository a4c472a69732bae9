# Round-2 evidence: rocprofv3 --kernel-trace --stats of every benchmark whose numbers DESIGN.md quotes, the program
# directly behind `--` (no env / bash -c hop); one summary CSV + the script's own log per benchmark in
# gpurun_out/prof_r02/, copied to profiles/r02_* afterwards.   usage: bash tools/gpu_prof_r02.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
run() {    # name, script, args...
    name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 "$@" > $O/$name.log 2>&1 || echo "$name: profiler exit code $?" >> $O/progress.log
    python3 - "$name" "$O" <<'PY'
import csv, glob, sys
s, o = sys.argv[1], sys.argv[2]
for f in glob.glob("/tmp/prof_%s/**/*kernel_stats.csv" % s, recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("%s/%s_kernel_stats.csv" % (o, s), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
    rm -rf /tmp/prof_$name
    echo "$name done" >> $O/progress.log
}
run chain1024 $R/tools/bench_chain.py 1024
run chain1536 $R/tools/bench_chain.py 1536
run chain2048 $R/tools/bench_chain.py 2048
run chain2432 $R/tools/bench_chain.py 2432
run chain64 $R/tools/bench_chain.py 64
run blocks $R/tools/bench_blocks.py
run cfg3 $R/tools/bench_cfg3.py
