mkdir -p gpurun_out
bash tools/gpu_prof2.sh r03 fir_mfma > gpurun_out/r03_prof.log 2>&1; tail -8 gpurun_out/r03_prof.log
bash tools/gpu_chain_prof.sh r03_chain2048_four_level 2048 --four
bash tools/gpu_chain_prof.sh r03_chain2432 2432
timeout -k 10 400 python tools/dbg/chain_flips.py > gpurun_out/r03_chain_flips.log 2>&1; grep -v amdgpu gpurun_out/r03_chain_flips.log
