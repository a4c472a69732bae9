# round 3, call E: the judged profile of the default bench (kernel stats + PMC passes, FAST instantiation only),
# then A/B of the spread-issue variant of the legacy kernel against the plain diagnostic build (same box, interleaved)
mkdir -p gpurun_out
bash tools/gpu_prof2.sh r03 fir_mfma > gpurun_out/r03_prof.log 2>&1; tail -5 gpurun_out/r03_prof.log
L=$PWD/gnuradio-3.5.0-dmr_amd
VARIANTS="GRHIP_LIB=$L/libgrhip_diag.so GRHIP_LIB=$L/libgrhip_spread.so" bash tools/gpu_ab.sh > gpurun_out/r03_spread_ab.log 2>&1; cat gpurun_out/r03_spread_ab.log
