# ablations of the role-split kernel (diagnostic builds, wrong results): which unit pins the period?
mkdir -p gpurun_out; rm -f gpurun_out/rs_abl.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for v in diag:1 abl1:1 abl2:1 abl4:1 abl8:1 abl16:1 abl6:1 abl7:1 abl22:1 abl23:1 diag:0; do
  lib=${v%%:*}; rs=${v##*:}
  env GRHIP_LIB=$L/libgrhip_$lib.so GRHIP_MF_RS=$rs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures 64 --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5))" >> gpurun_out/rs_abl.log || exit 1
done
cat gpurun_out/rs_abl.log
