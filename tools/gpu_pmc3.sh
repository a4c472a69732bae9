# Memory-pipeline counters of the matrix-core FIR kernel beside the read-bandwidth micro-benchmark of the same
# access shape (tools/dbg/readbw.hip): where the two differ is where the kernel loses its bandwidth.
# One --pmc pass per counter set, nothing else traced.   usage: bash tools/gpu_pmc3.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_pmc3
mkdir -p $O
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_BUFFER_READ_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES" \
           "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY TCC_CYCLE TCC_HIT TCC_MISS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/k$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --chain-captures 0 > $O/bench_k$i.log 2>&1 || echo "set $i bench failed" >> $O/progress.log
  timeout -k 10 100 rocprofv3 --pmc $set --output-format csv -d $O/m$i -- $R/tools/dbg/readbw > $O/readbw_m$i.log 2>&1 || echo "set $i readbw failed" >> $O/progress.log
  echo "set $i done" >> $O/progress.log
done
cd $O
python3 - > $O/summary.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("[km]*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fir_mfma" in k or "fir_like" in k:
            k = k[:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-36s per-dispatch %.6g  (%d dispatches)" % (c, v / cnt[k][c], cnt[k][c]))
PY
cat $O/summary.txt
rm -rf $O/k*/ $O/m*/
