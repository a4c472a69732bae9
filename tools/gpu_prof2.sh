# Profiles `python3 bench.py <args>` on the GPU box: rocprofv3 kernel stats + PMC passes (each in its own run),
# summarised for the kernels whose name contains <match>.  usage: bash tools/gpu_prof2.sh <tag> <match> [bench args]
TAG=${1:-r02}; MATCH=${2:-fir_mfma}; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
python3 $R/bench.py "$@" > $O/bench_line.json 2> $O/bench_stderr.log
BENCH="python3 $R/bench.py --no-cpu-baseline --chain-captures 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/bench_trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > $O/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- $BENCH > $O/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- $BENCH > $O/bench_pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds -- $BENCH > $O/bench_pmc_lds.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/pmc_mfma -- $BENCH > $O/bench_pmc_mfma.log 2>&1
cd $O
MATCH="$MATCH" python3 - <<'PY'
import csv, glob, collections, json, os
M = os.environ["MATCH"]
EXCL = os.environ.get("EXCLUDE", ", true>(")      # e.g. ", true>(": the GRHIP_MODE_FAST_REFTAPS instantiation that the bench line also times
out = {}
for f in glob.glob("trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("kernel_stats_summary.csv", "w") as g:
        w = csv.writer(g); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    for r in rows:
        if M in r["Name"] and EXCL not in r["Name"]:
            out["kernel"] = r["Name"][:120]; out["avg_ns"] = float(r["AverageNs"]); out["min_ns"] = float(r["MinNs"]); out["calls"] = int(r["Calls"])
pm = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds", "pmc_mfma"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if M in r["Kernel_Name"] and EXCL not in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for c, v in acc.items():
            pm[c] = v / cnt[c]
out["pmc_per_dispatch"] = pm
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # rocprofv3 reports KiB; FETCH_SIZE counts 1/2 of a 16-B/lane coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM)
    out["hbm_bytes_per_launch"] = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
    out["fetch_bytes_corrected"] = 2.0 * pm["FETCH_SIZE"] * 1024.0
    out["write_bytes"] = pm["WRITE_SIZE"] * 1024.0
try:
    line = json.loads(open("bench_line.json").read().strip().splitlines()[-1])
    out["captures"] = line["config"]["captures_per_gpu_per_step"]; out["samples"] = line["config"]["samples_per_capture"]
    out["bench_kernel_ms"] = line["roofline"]["kernel_ms"]
    if "GRBM_GUI_ACTIVE" in pm and "avg_ns" in out:
        out["effective_clock_ghz"] = pm["GRBM_GUI_ACTIVE"] / 8.0 / out["avg_ns"]
except Exception as e:
    out["bench_line_error"] = str(e)
json.dump(out, open("summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/trace/*/*.db 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
