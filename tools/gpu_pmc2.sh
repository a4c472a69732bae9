cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_pmc2
mkdir -p $O
BENCH="python3 $R/bench.py --steps 6 --warmup 2 --captures 8 --no-cpu-baseline --chain-captures 0"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $BENCH > $O/bench_p$i.log 2>&1
done
cd $O
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if "fir_tiled" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    for c,val in acc.items():
        print("%-28s per-dispatch %.5g"%(c,val/cnt[c]))
PY
