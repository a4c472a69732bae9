# round 3, last call: the judged profile of the default bench (its line now carries bit_exact_engine), then the whole GPU suite
mkdir -p gpurun_out
bash tools/gpu_prof2.sh r03 fir_mfma > gpurun_out/r03_prof.log 2>&1; tail -4 gpurun_out/r03_prof.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_full_gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r03_full_gpu_tests.log
exit $rc
