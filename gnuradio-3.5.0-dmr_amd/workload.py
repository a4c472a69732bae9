"""Synthetic workloads for the DMR demodulation hot path.

Pure numpy; no GPU, no oracle.  Used by tests/ and bench.py to build the same
seeded inputs everywhere.

cfg1  fir_filter_ccf, 64-tap Hamming low-pass, 1 M uniform complex samples
cfg2  freq_xlating_fir_filter_ccc(256 taps, decim 4) -> quadrature_demod_cf on a
      10 MS/s 4FSK capture (250 kBd, deviation +-1/+-3 * 33.75 kHz = DMR's
      648/1944 Hz at 4800 Bd scaled to 250 kBd, so that the outer symbols sit
      inside the 200 kHz low-pass; SURVEY 8(d)'s 81 kHz would put them in the
      stop band), carrier at -1.25 MHz because the 3.5.0 block translates
      -center_freq to DC
cfg4  ... -> clock_recovery_mm_ff(omega=10) -> binary_slicer -> correlate_access_code
"""
import math

import numpy as np

SEED_BASE = 0x444D5200

CFG2 = dict(
    fs=10e6, sym_rate=250e3, deviation=33.75e3, carrier=-1.25e6, center_freq=+1.25e6,
    ntaps=256, decim=4, cutoff=200e3, esn0_db=20.0, n_samples=10_000_000,
)
CFG2["demod_gain"] = (CFG2["fs"] / CFG2["decim"]) / (2 * math.pi * CFG2["deviation"])

CFG4 = dict(
    omega=10.0, gain_mu=0.175, mu=0.5, omega_relative_limit=0.005,
    sync_period_syms=2640, threshold=4,
    # 48-bit DMR-style sync word (BS-sourced voice), MSB first
    sync_hex="755FD7DF75F7",
)
CFG4["gain_omega"] = 0.25 * CFG4["gain_mu"] * CFG4["gain_mu"]


def sync_bits():
    v = int(CFG4["sync_hex"], 16)
    return np.array([(v >> (47 - i)) & 1 for i in range(48)], dtype=np.uint8)


def access_code_string():
    return "".join(str(int(b)) for b in sync_bits())


def lowpass_taps(ntaps, cutoff, fs, gain=1.0):
    """Hamming-windowed sinc, unity DC gain, any length.  Same formula as
    gr_firdes::low_pass / gr_firdes::window(WIN_HAMMING)
    (gnuradio-core/src/lib/general/gr_firdes.cc:105-148,731-734), which only
    produces odd lengths; for even lengths the sinc is centred on (ntaps-1)/2."""
    n = np.arange(ntaps, dtype=np.float64)
    w = 0.54 - 0.46 * np.cos(2 * math.pi * n / (ntaps - 1)) if ntaps > 1 else np.ones(1)
    c = (ntaps - 1) / 2.0
    fwT0 = 2 * math.pi * cutoff / fs
    t = n - c
    with np.errstate(invalid="ignore", divide="ignore"):
        h = np.where(t == 0, fwT0 / math.pi, np.sin(t * fwT0) / (t * math.pi))
    h = (h * w).astype(np.float32).astype(np.float64)
    h = h * (gain / h.sum())
    return h.astype(np.float32)


def uniform_complex(n, seed=SEED_BASE):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, size=(n, 2)).astype(np.float32)
    return x.view(np.complex64).reshape(n)


def fsk4_symbols(n_syms, seed):
    """random dibits -> 4FSK levels with the sync word planted every
    sync_period_syms symbols.  A 48-bit sync is sent as 48 two-level (+3/-3)
    symbols on the sign bit so that the binary slicer after M&M recovers it
    (the configured chain only decides the sign of each symbol)."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    levels = np.array([-3.0, -1.0, 1.0, 3.0])
    sym = levels[rng.integers(0, 4, size=n_syms)]
    sb = sync_bits().astype(np.float64) * 2 - 1
    per = CFG4["sync_period_syms"]
    for s in range(100, n_syms - 48, per):
        sym[s:s + 48] = 3.0 * sb
    return sym


def fsk4_capture(n_samples, stream_id=0, cfg=CFG2, return_symbols=False):
    """rectangular-pulse 4FSK at complex baseband offset `carrier`, AWGN at
    Es/N0.  Phase accumulates in float64 (the modulator is not on the measured
    path); output complex64."""
    seed = SEED_BASE + stream_id
    sps = int(round(cfg["fs"] / cfg["sym_rate"]))
    n_syms = (n_samples + sps - 1) // sps + 1
    sym = fsk4_symbols(n_syms, seed)
    f_inst = cfg["carrier"] + cfg["deviation"] * np.repeat(sym, sps)[:n_samples]
    ph = 2 * math.pi * np.cumsum(f_inst / cfg["fs"])
    ph = np.fmod(ph, 2 * math.pi)
    rng = np.random.default_rng(seed)
    # Es = sps * |x|^2 ; noise variance per complex sample N0 (unit-power signal)
    n0 = sps / (10.0 ** (cfg["esn0_db"] / 10.0))
    sigma = math.sqrt(n0 / 2.0)
    x = np.empty(n_samples, dtype=np.complex64)
    CH = 1 << 20
    for s in range(0, n_samples, CH):
        e = min(s + CH, n_samples)
        noise = rng.normal(0.0, sigma, size=(e - s, 2))
        x[s:e] = (np.cos(ph[s:e]) + noise[:, 0]) + 1j * (np.sin(ph[s:e]) + noise[:, 1])
    if return_symbols:
        return x, sym
    return x


def cfg2_proto_taps(cfg=CFG2):
    """256-tap real low-pass prototype handed to freq_xlating_fir_filter_ccc as
    complex taps (imaginary part zero)."""
    return lowpass_taps(cfg["ntaps"], cfg["cutoff"], cfg["fs"]).astype(np.complex64)


def with_history(x, history_items):
    """what the scheduler shows a block on its first call: history-1 zeros in
    front (gnuradio-core/src/lib/runtime/gr_flat_flowgraph.cc:150)."""
    return np.concatenate([np.zeros(history_items, dtype=x.dtype), x])
