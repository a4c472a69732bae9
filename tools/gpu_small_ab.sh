L=$PWD/gnuradio-3.5.0-dmr_amd
for rep in 1 2; do for v in libgrhip.so libgrhip_m8388608.so; do echo "== $v"; GRHIP_LIB=$L/$v timeout -k 10 120 python tools/bench_small_calls.py 2>/dev/null | grep -v amdgpu | cut -c1-90; done; done
