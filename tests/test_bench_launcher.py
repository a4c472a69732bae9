"""`python bench.py --gpus N` must itself start N ranks.  CPU test of that launch
path with N = 2: gloo, a stub step (no GPU), everything else -- child processes, rendezvous on
127.0.0.1, taps broadcast, capture sharding, max-over-ranks, rank 0 printing ONE line -- as in
the real run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--captures", "4", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3
    ranks = sorted(res["ranks"])
    assert [x[0] for x in ranks] == [0, 1]               # two distinct ranks reported ...
    assert [x[1] for x in ranks] == [0, 1]               # ... on two distinct local devices
    assert ranks[0][3] != ranks[1][3]                    # ... in two processes
    assert ranks[0][2] == [0, 1, 2, 3] and ranks[1][2] == [4, 5, 6, 7]    # weak scaling: 4 captures per rank
    assert ranks[0][4] == ranks[1][4] == 256             # rank 1 got rank 0's taps
    assert res["ms_per_step"] >= 2.0                     # MAX over ranks (rank 1 sleeps 2 ms per step)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--launcher-selftest"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_failing_rank_fails_the_run():
    # the real (GPU) path on a box without a GPU: every rank exits non-zero, so must the launcher
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    import torch
    if torch.cuda.is_available():
        return
    assert r.returncode != 0


def test_one_failing_rank_ends_the_others_promptly():
    """rank 1 dies before the rendezvous while rank 0 sits in init_process_group -- the launcher must notice,
    end rank 0 and exit non-zero within seconds, not after the distributed timeout"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--captures", "2", "--launcher-selftest"],
             {"GRHIP_SELFTEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0
    assert "rank(s) failed" in (r.stderr + r.stdout)
    assert time.time() - t0 < 90


def test_cpu_chain_baseline_legs_agree():
    """bench.py's cpu_baseline "chain" legs (SURVEY 8d i-iii): the stage-after-stage run and the thread-per-block pipeline over
    64 k-item chunks are the same computation -- same number of access-code flags -- and every leg reports a rate"""
    import importlib.util
    import grhip_loader
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    wl = grhip_loader.import_grhip().workload
    po = grhip_loader.import_oracle()
    x = wl.fsk4_capture(600_000)
    r = b.cpu_chain_baseline(wl, x, wl.cfg2_proto_taps(), "ref" if po.have_ref() else "oracle")
    assert r["serial_1_thread"]["sync_flags"] == r["thread_per_block"]["sync_flags"] >= 4
    for leg in ("serial_1_thread", "thread_per_block", "all_cores"):
        assert r[leg]["value"] > 0 and r[leg]["cores"] >= 1


def test_bench_generator_is_the_parity_workload():
    """bench.py synthesises its captures with torch on the device; the parity tests use workload.fsk4_capture (numpy).
    Same waveform: the two are phase-coherent sample by sample (noise apart, which comes from different generators at the
    same level), so the benchmark runs on the signal the parity tests check."""
    import importlib.util
    import numpy as np
    import torch
    import grhip_loader
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    wl = grhip_loader.import_grhip().workload
    n = 400_000
    for sid in (0, 5):
        xb = b.synth_captures(torch, wl, 1, n, sid, torch.device("cpu"))[0].numpy()
        xb = xb[:, 0] + 1j * xb[:, 1]
        xw = wl.fsk4_capture(n, stream_id=sid)
        c = wl.CFG2
        sps = int(round(c["fs"] / c["sym_rate"]))
        n0 = sps / (10.0 ** (c["esn0_db"] / 10.0))            # noise variance per complex sample, unit-power signal
        # E[x_b conj(x_w)] = |s|^2 = 1 when the noiseless parts agree in phase at every sample
        coh = np.mean(xb * np.conj(xw))
        tol = 6 * np.sqrt((2 * n0 + n0 * n0) / n)
        assert abs(coh.real - 1.0) < tol + 1e-3 and abs(coh.imag) < tol + 1e-3, coh
        # same noise level: E|x|^2 = 1 + n0 for both
        for x in (xb, xw):
            assert abs(np.mean(np.abs(x) ** 2) - (1 + n0)) < 0.02 * (1 + n0)
        # and the difference of the two is noise only: E|x_b - x_w|^2 = 2 n0
        assert abs(np.mean(np.abs(xb - xw) ** 2) - 2 * n0) < 0.03 * 2 * n0 + 1e-3
