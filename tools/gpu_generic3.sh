# PMC of the bit-exact window kernel (each counter set in its own pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_generic; rm -rf $O; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/p0 -- python3 $R/tools/dbg/generic_rate.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/p1 -- python3 $R/tools/dbg/generic_rate.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $O/p2 -- python3 $R/tools/dbg/generic_rate.py > /dev/null 2>&1
cd $O
python3 - <<'PY'
import csv, glob, collections, json
pm = {}
for d in ("p0", "p1", "p2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "fir_generic_win" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for c, v in acc.items():
            pm[c] = v / cnt[c]
json.dump(pm, open("summary.json", "w"), indent=1)
print(json.dumps(pm, indent=1))
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
