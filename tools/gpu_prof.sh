# Profiles bench.py on the GPU box.  usage: bash tools/gpu_prof.sh <tag>
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python3 $R/bench.py --steps 10 --warmup 3 --captures 8 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc1 -- $BENCH > $O/bench_pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc2 -- $BENCH > $O/bench_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc3 -- $BENCH > $O/bench_pmc3.log 2>&1
cd $O
find . -name "*.csv" | head -30
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read()[:1500])
for d in ("pmc1","pmc2","pmc3"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            cnt[(k,r["Counter_Name"])]+=1
        for k,v in acc.items():
            if "fir_tiled" in k:
                print("==",d,k)
                for c,val in v.items():
                    print("   %-24s per-dispatch %.4g  (n=%d)"%(c,val/cnt[(k,c)],cnt[(k,c)]))
PY
