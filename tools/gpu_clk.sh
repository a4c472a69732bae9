# shader clock and VALU utilisation of the tiled kernel at 1 and 2 workgroups per CU
# (GRHIP_WGPCU is honoured by diagnostic builds only: make variant NAME=diag EXTRA=-DGRHIP_DIAG, GRHIP_LIB=.../libgrhip_diag.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_clk
rm -rf $O; mkdir -p $O
for w in 2; do
  export GRHIP_WGPCU=$w
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/w$w -- python3 $R/bench.py --steps 6 --warmup 2 --captures 64 --ramp-ms 300 --no-cpu-baseline --chain-captures 0 > $O/bench_w$w.log 2>&1
done
cd $O
python3 - <<'PY'
import csv, glob, collections
for w in (2,):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob("w%d/**/*counter_collection.csv"%w, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fir_tiled" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    dur=[]
    for f in glob.glob("w%d/**/*kernel_trace.csv"%w, recursive=True):
        for r in csv.DictReader(open(f)):
            if "fir_tiled" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    d=sum(dur)/max(1,len(dur))
    print("WG/CU",w,"kernel ns",d,"n",len(dur))
    for c,v in acc.items():
        print("   %-22s %.5g"%(c,v/cnt[c]))
    if d and "GRBM_GUI_ACTIVE" in acc:
        g=acc["GRBM_GUI_ACTIVE"]/cnt["GRBM_GUI_ACTIVE"]
        print("   clock if GUI_ACTIVE is summed over 8 XCDs: %.0f MHz"%(g/8/d*1e3))
PY
