"""Secondary measurements (not the headline line): every other block of SURVEY 8(a)
at its BASELINE size on one GPU, device-resident, as algorithmic GB/s against the
8 TB/s HBM peak.  usage: python tools/bench_blocks.py"""
import json
import time
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import grhip_loader

g = grhip_loader.import_grhip()
wl = g.workload
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
PEAK = 8000.0


def timeit(fn, reps=50, warm=3, ramp_s=0.3):
    # an idle MI355X needs tens of ms of load before its shader clock reaches steady state
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(10):
            fn()
        st.synchronize()
    for _ in range(warm):
        fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, items, bytes_per_item, unit="Msamples/s"):
    gbs = items * bytes_per_item / (ms * 1e-3) / 1e9
    print(json.dumps({"block": name, "ms": round(ms, 4), "rate": round(items / ms / 1e3, 1), "unit": unit,
                      "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK, 4)}), flush=True)


rng = np.random.default_rng(0)

# cfg1: fir_filter_ccf 64 taps, D=1, 1 M samples (and 16 M to get out of launch-latency land)
for n in (1_000_000, 16_000_000, 128_000_000):
    x = torch.randn((n + 64, 2), device=dev)
    y = torch.empty((n, 2), device=dev)
    blk = g.fir_filter_ccf(1, wl.lowpass_taps(64, 0.1, 1.0))
    report("fir_filter_ccf 64t D=1 n=%d" % n, timeit(lambda: blk.work_device(n, x, y, st)), n, 16)

# published-baseline shape (mp-sched, fir_filter_fff 256 taps, decimation 1)
n = 128_000_000
xf = torch.randn(n + 256, device=dev)
yf = torch.empty(n, device=dev)
blk = g.fir_filter_fff(1, wl.lowpass_taps(256, 0.1, 1.0))
report("fir_filter_fff 256t D=1 FAST (real-data overlap-save engine since round 2; round 1: tiled kernel, float-pair mode)", timeit(lambda: blk.work_device(n, xf, yf, st), reps=20), n, 8)
blk.set_mode(g.MODE_GENERIC)
report("fir_filter_fff 256t D=1 GENERIC (bit-exact order)", timeit(lambda: blk.work_device(n, xf, yf, st), reps=20), n, 8)

# unfused xlating (10 B / input sample) and quad_demod (12 B / item)
n = 160_000_000
c = wl.CFG2
x = torch.randn((n + 256, 2), device=dev)
y = torch.empty((n // 4, 2), device=dev)
blk = g.freq_xlating_fir_filter_ccc(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"])
def run_xl():
    blk.reset(); blk.work_device(n // 4, x, y, st)
report("freq_xlating_fir_filter_ccc 256t D=4", timeit(run_xl), n, 10)
cp = (wl.cfg2_proto_taps() * np.exp(1j * 0.01 * np.arange(256))).astype(np.complex64)
blk2 = g.freq_xlating_fir_filter_ccc(4, cp, c["center_freq"], c["fs"])
def run_xl2():
    blk2.reset(); blk2.work_device(n // 4, x, y, st)
report("freq_xlating_fir_filter_ccc 256 COMPLEX taps D=4", timeit(run_xl2), n, 10)
d = torch.empty(n, device=dev)
xx = torch.randn((n + 1, 2), device=dev)
qd = g.quadrature_demod_cf(1.0)
report("quadrature_demod_cf", timeit(lambda: qd.work_device(n, xx, d, st)), n, 12)

# correlator (2 B / bit)
n = 64_000_000
bits = torch.randint(0, 2, (n,), dtype=torch.uint8, device=dev)
ob = torch.empty(n, dtype=torch.uint8, device=dev)
ca = g.correlate_access_code_bb(wl.access_code_string(), 4)
report("correlate_access_code_bb", timeit(lambda: ca.work_device(n, bits, ob, st)), n, 2, "Mbits/s")

# binary_slicer_fb on its own (the chain fuses it into the correlator): 5 B per item
nb_ = 128_000_000
bx = torch.randn(nb_, device=dev)
bo = torch.empty(nb_, dtype=torch.uint8, device=dev)
bs = g.binary_slicer_fb()
report("binary_slicer_fb", timeit(lambda: bs.work_device(nb_, bx, bo, st), reps=20), nb_, 5)

# pager_slicer_fb: a serial DC tracker, one wavefront per stream: the rate of ONE stream
ns_ = 4_000_000
sx = torch.randn(ns_, device=dev)
so = torch.empty(ns_, dtype=torch.uint8, device=dev)
ps = g.pager_slicer_fb(0.002)
report("pager_slicer_fb one stream of 4 M symbols (one wavefront, latency-bound)",
       timeit(lambda: ps.work_device(ns_, sx, so, st), reps=5, ramp_s=0.05), ns_, 5, "Msymbols/s")

# cfg3: fft_vcc 4096-pt over 2^24 samples; pfb_channelizer M=8, 256-tap prototype, 2^24 samples
N, nvec = 4096, 4096
xv = torch.randn((N * nvec, 2), device=dev)
yv = torch.empty((N * nvec, 2), device=dev)
ff = g.fft_vcc(N, True, [], False)
report("fft_vcc 4096-pt x 4096", timeit(lambda: ff.work_device(nvec, xv, yv, st)), N * nvec, 16)
M, nout = 8, (1 << 24) // 8
taps = wl.lowpass_taps(256, 0.5 / M, 1.0)
pf = g.pfb_channelizer_ccf(M, taps, 1.0)
per = nout + 64
xs = torch.randn((M * per, 2), device=dev)
yo = torch.empty((nout * M, 2), device=dev)
pf.general_work_device(nout, xs, per, yo, st)       # first call returns 0 (d_updated)
report("pfb_channelizer_ccf M=8 256t", timeit(lambda: pf.general_work_device(nout, xs, per, yo, st)), nout * M, 16)

# gr_fft_filter_ccc: overlap-add, 256 complex taps -> 512-point transforms, 257 samples per block
tp = (wl.cfg2_proto_taps() * np.exp(1j * 0.01 * np.arange(256))).astype(np.complex64)
blk = g.fft_filter_ccc(1, tp)
ns = blk.nsamples()
n = ns * 65536
x = torch.randn((n, 2), device=dev)
y = torch.empty((n, 2), device=dev)
report("fft_filter_ccc 256t D=1 (fused overlap-save, 4096-pt blocks)", timeit(lambda: blk.work_device(n, x, y, st), reps=20), n, 16)
# long filters: where fast convolution beats the direct form
tl = (wl.lowpass_taps(2000, 0.05, 1.0) * np.exp(1j * 0.01 * np.arange(2000))).astype(np.complex64)
blk = g.fft_filter_ccc(1, tl)
ns = blk.nsamples()
n = ns * (16_000_000 // ns)
x = torch.randn((n + 2048, 2), device=dev)
y = torch.empty((n, 2), device=dev)
report("fft_filter_ccc 2000t D=1 (fused overlap-save)", timeit(lambda: blk.work_device(n, x, y, st), reps=20), n, 16)
blk2 = g.fir_filter_ccc(1, tl)
report("fir_filter_ccc 2000t D=1 FAST (dispatches to the overlap-save engine)",
       timeit(lambda: blk2.work_device(n, x, y, st), reps=5), n, 16)
tm = tl[:1000]
blk3 = g.fir_filter_ccc(1, tm)
report("fir_filter_ccc 1000t D=1 FAST (dispatches to the overlap-save engine)", timeit(lambda: blk3.work_device(n, x, y, st), reps=10), n, 16)
blk4 = g.fir_filter_ccf(1, wl.lowpass_taps(256, 0.1, 1.0))
report("fir_filter_ccf 256t D=1 FAST (overlap-save engine: 256 taps per phase)", timeit(lambda: blk4.work_device(n, x, y, st), reps=10), n, 16)
blk5 = g.fir_filter_ccf(1, wl.lowpass_taps(128, 0.1, 1.0))
report("fir_filter_ccf 128t D=1 FAST (overlap-save engine since round 2; round 1: tiled kernel)", timeit(lambda: blk5.work_device(n, x, y, st), reps=10), n, 16)

# freq_xlating_fir_filter_ccc at decimations the tiled kernel does not take: high-decimation direct kernel + rotator
# (tools/bench_decim.py compares it with the overlap-save engine shape by shape)
n = 160_000_000
x = torch.randn((n + 512, 2), device=dev)
y = torch.empty((n // 20, 2), device=dev)
blk = g.freq_xlating_fir_filter_ccc(20, wl.lowpass_taps(400, 0.02, 1.0).astype(np.complex64), c["center_freq"], c["fs"])
def run_xl20():
    blk.reset(); blk.work_device(n // 20, x, y, st)
report("freq_xlating_fir_filter_ccc 400t D=20 real prototype (high-decimation direct kernel, pre-mix form)", timeit(run_xl20, reps=10), n, 8.4)

y16 = torch.empty((n // 16, 2), device=dev)
blk16 = g.freq_xlating_fir_filter_ccc(16, wl.lowpass_taps(400, 0.02, 1.0).astype(np.complex64), c["center_freq"], c["fs"])
def run_xl16():
    blk16.reset(); blk16.work_device(n // 16, x, y16, st)
report("freq_xlating_fir_filter_ccc 400t D=16 real prototype (overlap-save engine, folded inverse; round 1: direct kernel)", timeit(run_xl16, reps=10), n, 8.5)
