# Round 3, call A: the three attribution probes of DESIGN 2 (diagnostic builds), then the profile of the default bench command
L=$PWD/gnuradio-3.5.0-dmr_amd
mkdir -p gpurun_out
( echo "== shipped library, with the complex-tap paths"; COMPLEX_TAPS=1 timeout -k 10 300 python tools/dbg/demod_attrib.py 2>&1 | grep -v amdgpu.ids
  echo "== probe (i): IEEE divide in the FAST epilogue (libgrhip_probe1.so)"; GRHIP_LIB=$L/libgrhip_probe1.so timeout -k 10 300 python tools/dbg/demod_attrib.py 2>&1 | grep "GPU FAST"
  echo "== probe (ii): every pre-mix phasor from double precision (libgrhip_probe2.so)"; GRHIP_LIB=$L/libgrhip_probe2.so timeout -k 10 300 python tools/dbg/demod_attrib.py 2>&1 | grep "GPU FAST (matrix" ) > gpurun_out/r03_demod_attribution.log 2>&1
cat gpurun_out/r03_demod_attribution.log
bash tools/gpu_prof2.sh r03 fir_mfma > gpurun_out/r03_prof.log 2>&1; tail -40 gpurun_out/r03_prof.log | head -60
