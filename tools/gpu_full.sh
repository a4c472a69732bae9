# the whole GPU suite, then the default bench line and the smoke entry
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_full_gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/r03_full_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r03_final_bench_line.json 2> gpurun_out/r03_final_bench.err; rc=$?; cut -c1-300 gpurun_out/r03_final_bench_line.json
exit $rc
