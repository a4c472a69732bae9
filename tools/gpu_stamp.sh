mkdir -p gpurun_out
timeout -k 10 300 python tools/stamp_report_mfma.py > gpurun_out/stamp_lg.log 2>&1; grep -v amdgpu gpurun_out/stamp_lg.log | head -30
