# PMC passes for the tiled kernel.  usage: bash tools/gpu_pmc.sh <tag>
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python3 $R/bench.py --steps 6 --warmup 2 --captures 8 --no-cpu-baseline --chain-captures 0"
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM" \
           "SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
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
        print("%-26s per-dispatch %.5g  (n=%d)"%(c,val/cnt[c],cnt[c]))
PY
