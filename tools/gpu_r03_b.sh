# Round 3, call B: channeliser tests (oversampled on the fast kernel), its rates, the chain profiles, the blocks table
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fft_pfb.py -x -q -m gpu -k "pfb" > gpurun_out/pfb_tests.log 2>&1; tail -4 gpurun_out/pfb_tests.log
timeout -k 10 300 python tools/dbg/pfb_oversampled.py > gpurun_out/r03_pfb_oversampled.log 2>&1; cat gpurun_out/r03_pfb_oversampled.log | grep -v amdgpu
for s in 2048 64 1024; do bash tools/gpu_chain_prof.sh r03_chain$s $s; done
bash tools/gpu_chain_prof.sh r03_chain2048_four_level 2048 --four
bash tools/gpu_chain_prof.sh r03_chain2432 2432
