"""Test infrastructure shared by tests/ and __graft_entry__.smoke(): the tolerance rule for
quadrature-demodulator outputs of the FAST path.  Not part of the product."""
import numpy as np

# gr_fast_atan2f (general/gr_fast_atan2f.cc:147-157) is DISCONTINUOUS where it switches from
# `base_angle = z` to the table interpolation: at z = TAN_MAP_RES = 1/255 the first gives
# 0.0039216 rad and the second 0.0019766 rad (the table is indexed with z*256 - 0.5 but holds
# atan(i/255)).  An input within rounding of that z lands on either side; the reference's own
# SSE and generic builds disagree there too.  Size of the step, in radians:
ATAN_STEP = 0.0039215689 - 0.0019765894
CFG2_GAIN = 2.5e6 / (2.0 * np.pi * 33750.0)


def demod_report(got, ref, skip=64, tol=1e-5, gain=CFG2_GAIN):
    """The numbers behind demod_close, for reporting (smoke(), tests): a dict with
      steady_rel_inf    max |got - ref| / max|ref| after the start-up transient, step-exempted samples removed
      per_element_rel   max |got - ref| / |ref| over the steady samples with |ref| > 0.1 max|ref| (same removal)
      step_exempted     samples that differ by the reference's own arctangent step at z = 1/255
      transient_rel_inf max |got - ref| / max|ref| inside the first `skip` outputs
      ok                the demod_close verdict"""
    got = np.asarray(got); ref = np.asarray(ref)
    ok, worst = demod_close(got, ref, skip, tol, gain)
    rep = {"ok": bool(ok), "steady_rel_inf": float(worst), "per_element_rel": 0.0, "step_exempted": 0,
           "transient_rel_inf": 0.0}
    if got.shape != ref.shape or len(ref) <= skip:
        return rep
    full = float(np.abs(ref).max())
    g = got[skip:].astype(np.float64); r = ref[skip:].astype(np.float64)
    err = np.abs(g - r)
    step = abs(gain) * ATAN_STEP
    at_step = (err > tol * float(np.abs(r).max())) & (np.abs(err - step) <= 0.02 * step)
    rep["step_exempted"] = int(at_step.sum())
    keep = ~at_step
    big = keep & (np.abs(r) > 0.1 * float(np.abs(r).max()))
    if big.any():
        rep["per_element_rel"] = float((err[big] / np.abs(r[big])).max())
    rep["transient_rel_inf"] = float(np.abs(got[:skip].astype(np.float64) - ref[:skip]).max() / max(full, 1e-30))
    return rep


def demod_close(got, ref, skip=64, tol=1e-5, gain=CFG2_GAIN):
    """Tolerance check for quadrature-demod outputs of the FAST path.

    Steady state (after the FIR's start-up transient of ntaps/decim = 64 outputs):
    |got - ref| <= tol * max|ref| there -- the north-star 1e-5, relative to the
    demodulator's output range.  Samples that differ by exactly the reference's own step at
    z = 1/255 (see above) are tolerated if they are rare (<= 2e-5 of the samples, at least 2).
    Inside the transient the FIR output climbs from 0, the demodulator takes the angle of
    numbers that are ~1e-3 of full scale, and an angle is only as accurate as |dy|/|y|; there
    the check is 1e-2 of the output range (it still catches a wrong sample, not rounding).
    Returns (ok, worst_steady_relative_error)."""
    got = np.asarray(got); ref = np.asarray(ref)
    if got.shape != ref.shape:
        return False, float("inf")
    if len(ref) <= skip:
        return bool(np.abs(got - ref).max() <= 1e-2 * max(np.abs(ref).max(), 1e-30)), 0.0
    s = float(np.abs(ref[skip:]).max())
    err = np.abs(got[skip:].astype(np.float64) - ref[skip:].astype(np.float64))
    bad = err > tol * s
    if bad.any():
        step = abs(gain) * ATAN_STEP
        at_step = np.abs(err[bad] - step) <= 0.02 * step
        if at_step.all() and int(bad.sum()) <= max(2, int(2e-5 * len(err))):
            err = err[~bad]
    e_steady = float(err.max())
    e_trans = float(np.abs(got[:skip] - ref[:skip]).max())
    full = float(np.abs(ref).max())
    return (e_steady <= tol * s) and (e_trans <= 1e-2 * full), e_steady / s
