mkdir -p gpurun_out
timeout -k 10 300 python tools/dbg/mm_forms.py 20 400 70000 2.0 11 fast > gpurun_out/mm_forms.log 2>&1; grep -v amdgpu gpurun_out/mm_forms.log | tail -30
timeout -k 10 300 python tools/dbg/mm_forms.py 20 400 70000 2.0 11 generic > gpurun_out/mm_forms2.log 2>&1; grep -v amdgpu gpurun_out/mm_forms2.log | tail -30
