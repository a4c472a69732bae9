# Round 3, final evidence: the profile of the default bench command (line, rocprofv3 stats, PMC passes), the attribution table with the
# REFTAPS row, the two chain profiles that ran out of memory in call B
mkdir -p gpurun_out
bash tools/gpu_prof2.sh r03 fir_mfma > gpurun_out/r03_prof.log 2>&1; tail -12 gpurun_out/r03_prof.log
( echo "== shipped library, with the complex-tap paths"; COMPLEX_TAPS=1 timeout -k 10 300 python tools/dbg/demod_attrib.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03_demod_attribution_final.log 2>&1
grep "REFTAPS" gpurun_out/r03_demod_attribution_final.log
bash tools/gpu_chain_prof.sh r03_chain2048_four_level 2048 --four
bash tools/gpu_chain_prof.sh r03_chain2432 2432
