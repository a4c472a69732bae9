# Kernel-trace statistics of the SURVEY 8f blocks' benchmarks (tools/bench_decim.py, bench_pfbdec.py,
# bench_framer.py) on the GPU box; leaves one summary CSV per script in gpurun_out/prof_widened/.
# usage: bash tools/gpu_prof_widened.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_widened
mkdir -p $O
for s in bench_decim bench_pfbdec bench_framer; do
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$s -- python3 $R/tools/$s.py > $O/$s.log 2>&1 || exit 1
    python3 - "$s" "$O" <<'PY'
import csv, glob, sys
s, o = sys.argv[1], sys.argv[2]
for f in glob.glob("/tmp/prof_%s/**/*kernel_stats.csv" % s, recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("%s/%s_kernel_stats.csv" % (o, s), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
    rm -rf /tmp/prof_$s
    echo "$s done" >> $O/progress.log
done
