"""Robustness of the two-stream factorisation's cross-stream edges (stream memory operations, in-kernel polls) and of
the option combinations ADVICE r4 named.  The reference returns errors from inside the optimiser loop instead of throwing
(gpmcmc.py:331-339): a poll that gives up must come back as an error code with the handle still usable."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch

    assert torch.cuda.is_available()
    from andvaranaut_amd import MiGP
    from oracle import gp_oracle as orc

    return MiGP, orc


def test_poll_that_sees_nothing_demotes_the_handle_and_the_call_still_returns_the_result():
    """a poll that runs into its limit does not fail the evaluation.  The handle switches its cross-stream
    edges to events, evaluates the same theta again inside the same call and says so once (gpmcmc.py:331-339: a failed
    evaluation is survived inside the optimiser loop -- here it is not even seen)."""
    MiGP, orc = _mods()
    N, d = 4096, 8  # 32 tile columns: two streams, the (a2) edge is a poll at the end of a leaf
    X, y = orc.synth_problem(N, d, seed=3)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF", need_grad=False)
    ref = gp.lml(theta)
    assert gp.info == 0 and gp.get_option(26) == 2 and gp.get_option(40) == 0
    gp.set_option(27, 12)  # give up after 2^12 sleeps (well under a millisecond) instead of seconds
    gp.set_option(28, 1)   # the next evaluation leaves one main-stream signal unwritten
    t0 = time.perf_counter()
    assert gp.lml(theta) == ref and gp.info == 0  # (event edges return the polls' bits)
    assert time.perf_counter() - t0 < 1.0
    assert gp.get_option(26) == 0 and gp.get_option(40) == 1 and gp.get_option(28) == 0
    assert "events from now on" in gp.last_error()
    assert gp.lml(theta) == ref
    # re-armed by the caller: polls again, same bits; a second lost signal demotes again
    gp.set_option(26, 2)
    assert gp.get_option(40) == 0 and gp.lml(theta) == ref
    gp.set_option(28, 1)
    assert gp.lml(theta) == ref and gp.get_option(26) == 0
    # the hook needs a bounded poll: refused with runtime waits or events
    with pytest.raises(RuntimeError, match="option 28 needs option 26 = 2"):
        gp.set_option(28, 1)
    gp.set_option(26, 1)
    with pytest.raises(RuntimeError, match="option 28 needs option 26 = 2"):
        gp.set_option(28, 1)
    gp.close()


def test_hook_that_meets_no_edge_is_cleared_by_the_evaluation():
    """option 28 used to survive evaluations whose schedule never reaches the edge it drops (column mode from the
    start) and hit a later one."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(1536, 4, seed=5)  # 12 tile columns: column mode on two streams, no (a2) edge
    theta = orc.synth_theta(4)
    gp = MiGP(X, y, "RBF", need_grad=False)
    ref = gp.lml(theta)
    gp.set_option(28, 1)
    assert gp.lml(theta) == ref and gp.get_option(28) == 0 and gp.get_option(40) == 0
    gp.close()


def test_batch_with_a_poll_timeout_is_evaluated_again_as_a_whole():
    MiGP, orc = _mods()
    N, d = 4096, 8
    X, y = orc.synth_problem(N, d, seed=4)
    th = np.stack([orc.synth_theta(d, kv=1.5 + 0.1 * i) for i in range(3)])
    gp = MiGP(X, y, "RBF", need_grad=False)
    ref = gp.lml_batch(th)
    gp.set_option(27, 12)
    gp.set_option(28, 1)
    assert np.array_equal(gp.lml_batch(th), ref)
    assert gp.get_option(40) == 1 and gp.get_option(26) == 0
    assert np.array_equal(gp.lml_batch(th), ref)
    gp.close()


_SERIAL_CHILD = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from andvaranaut_amd import MiGP
from oracle import gp_oracle as orc
N, d = 3072, 6   # 24 tile columns: column mode on two streams, a poll at the end of every leaf
X, y = orc.synth_problem(N, d, seed=13)
theta = orc.synth_theta(d)
t0 = time.perf_counter()
gp = MiGP(X, y, "Matern52", need_grad=False)
t_create = time.perf_counter() - t0
demoted_at_create = gp.get_option(40)
t0 = time.perf_counter()
v = gp.lml(theta)
t_first = time.perf_counter() - t0
t0 = time.perf_counter()
v2 = gp.lml(theta)
t_second = time.perf_counter() - t0
ref = orc.lml(X, y, ["Matern52"], [], theta)
print("RESULT", repr(float(v)), repr(float(v2)), repr(float(ref)), gp.info, demoted_at_create, gp.get_option(40), gp.get_option(26), t_create, t_first, t_second)
gp.close()
"""


@pytest.mark.parametrize("env", [{"AMD_SERIALIZE_KERNEL": "3"}, {"HIP_LAUNCH_BLOCKING": "1"}, {}])
def test_default_schedule_terminates_under_serialised_dispatch(env):
    """a FRESH child process with kernel dispatch serialised in its environment before any GPU call evaluates a
    two-stream size and gets the oracle's LML, without waiting for the multi-second poll limit even once: mi_gp_create's probe
    (a few ms) finds the condition and the handle starts with event edges.  The unserialised control keeps its polls."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    child_env = dict(os.environ)
    for k in ("AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_COPY"):
        child_env.pop(k, None)
    child_env.update(env)
    out = subprocess.run([sys.executable, "-c", _SERIAL_CHILD, ROOT], capture_output=True, text=True, env=child_env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    v, v2, ref = float(line[1]), float(line[2]), float(line[3])
    info, at_create, demoted, mode = int(line[4]), int(line[5]), int(line[6]), int(line[7])
    t_first, t_second = float(line[9]), float(line[10])
    assert info == 0 and v == v2 and abs(v - ref) <= 1e-10 * abs(ref), line
    if env:
        assert at_create == 1 and demoted == 1 and mode == 0, line
        assert t_first < 2.0 and t_second < 1.0, line  # (the default poll limit is seconds PER POLL: 24 leaves would be a minute)
    else:
        assert at_create == 0 and demoted == 0 and mode == 2, line


@pytest.mark.parametrize("N,nh", [(3072, 6), (640, 6), (640, 10)])
def test_handles_evaluating_concurrently_on_two_streams_each(N, nh):
    """Six handles x two streams with in-kernel polls (option 26 = 2): no poll-limit error, every value equal to the serial
    schedule's (include/mi_gp.h documents six as the tested limit for mode 2).  N = 640 is five tile columns -- on two streams
    since round 6.  TEN handles are beyond the documented limit: a handle whose poll runs into its limit there demotes itself to
    event edges and evaluates again inside the call (option 40), so the values are still the serial schedule's and no call fails."""
    MiGP, orc = _mods()
    d = 6  # (N = 3072: 24 tile columns, the two-stream driver in column mode from the start)
    X, y = orc.synth_problem(N, d, seed=9)
    thetas = [orc.synth_theta(d, kv=1.2 + 0.05 * i) for i in range(12)]
    gps = [MiGP(X, y, "Matern52", need_grad=False) for _ in range(nh)]
    assert all(g.get_option(0, 1) == 1 for g in gps)
    serial = [gps[0].lml(t) for t in thetas]
    out = [[None] * len(thetas) for _ in gps]
    errs = []

    def work(i):
        try:
            for rep in range(3):
                for j, t in enumerate(thetas):
                    out[i][j] = gps[i].lml(t)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(gps))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(len(gps)):
        assert out[i] == serial
    if nh <= 6:
        assert [g.get_option(40, 0) for g in gps] == [0] * nh  # nobody had to give up the polls
    for g in gps:
        g.close()


def test_early_inverse_levels_leave_the_gradient_bit_identical():
    """Option 30: the first block-doubling levels of U = L^-T run inside the factorisation's tail on the main stream
    (LML + gradient from 64 tile columns on) -- same launches per tile, regrouped over node batches: same bits."""
    MiGP, orc = _mods()
    N, d = 8320, 8  # 65 tile columns: a ragged last tile and a trailing partial node on every level
    X, y = orc.synth_problem(N, d, seed=21)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "RBF")
    gp.set_option(30, 0)
    v0, g0 = gp.lml_grad(theta)
    for ms, cols in ((16, 48), (64, 48), (4, 16)):
        gp.set_option(30, ms)
        gp.set_option(31, cols)
        v, g = gp.lml_grad(theta)
        assert v == v0 and np.array_equal(g, g0), (ms, cols)
    # ... and the conditional through U (mi_gp_predict_u forms U behind mi_gp_factor: nothing was done early there)
    mu, var = gp.predict(theta, X[:300], via_inverse=True)
    mu2, var2 = gp.predict(theta, X[:300], via_inverse=False)
    assert np.allclose(mu, mu2, rtol=1e-7, atol=1e-8) and np.allclose(var, var2, rtol=1e-6, atol=1e-9)
    gp.close()


@pytest.mark.parametrize("N,d", [(1500, 4), (3000, 6), (3700, 5), (4300, 5), (8320, 8)])
def test_column_mode_and_extended_panels_return_the_same_bits_on_every_schedule(N, d):
    """Round 5: the last 24 tile columns are factored column by column (option 37; problems of up to 31 tile columns as a whole
    since round 6, option 46: N = 3700 is 29), the super-panels in front of them also
    update the next panel's first column (option 35), short in-panel updates run on the thin kernel (option 32).  All three are
    rules of the SHAPE: one stream (option 0 = 0), forced two streams (0 = 2), event edges instead of stream memory
    operations (26 = 0 / 1), the other scheduling knobs -- every schedule returns the default's bits, LML and gradient."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(N, d, seed=N)
    theta = orc.synth_theta(d)
    gp = MiGP(X, y, "Matern52")
    ntc = (N + 127) // 128
    if ntc >= 20:
        gp.set_option(2, 4 if ntc <= 60 else 8)  # pin the width the two-stream default picks (one stream would take 8)
    v0, g0 = gp.lml_grad(theta)
    ref = orc.lml(X, y, ["Matern52"], [], theta)
    assert abs(v0 - ref) <= 1e-10 * abs(ref)
    for opts in ([(0, 0)], [(0, 2)], [(26, 0)], [(26, 1)], [(26, 0), (0, 2)], [(21, 16)], [(30, 0)], [(45, 0)], [(45, 0), (26, 0)]):
        for k, v in opts:
            gp.set_option(k, v)
        v1, g1 = gp.lml_grad(theta)
        assert v1 == v0 and np.array_equal(g1, g0), opts
        assert gp.lml(theta) == v0, opts
        for k, v in {0: 1, 26: 2, 21: 8, 30: 16, 45: 1}.items():
            gp.set_option(k, v)
    gp.close()


def test_round5_arithmetic_options_agree_to_rounding_and_off_is_the_round4_chain():
    """Options 32 (thin kernel), 35 (extended panels), 37 (column mode) regroup sums: every combination agrees with the
    oracle to the contract's 1e-10 and with each other to 1e-11 on a well-conditioned problem; a batch returns the single
    evaluation's bits under each of them."""
    MiGP, orc = _mods()
    N, d = 4300, 5
    X, y = orc.synth_problem(N, d, seed=8)
    theta = orc.synth_theta(d)
    th = np.stack([orc.synth_theta(d, kv=1.2 + 0.2 * i) for i in range(3)])
    ref = orc.lml(X, y, ["RBF"], [], theta)
    gp = MiGP(X, y, "RBF", need_grad=False)
    vals = []
    for o32, o35, o37, o46 in [(2048, 32, 24, 31), (0, 32, 24, 31), (2048, 0, 24, 31), (2048, 32, 0, 31), (0, 0, 0, 31), (2048, 64, 34, 31), (64, 8, 12, 31),
                               (2048, 32, 24, 34), (2048, 32, 12, 40)]:  # (46: the whole problem, 34 tile columns, in column mode)
        gp.set_option(32, o32)
        gp.set_option(35, o35)
        gp.set_option(37, o37)
        gp.set_option(46, o46)
        v = gp.lml(theta)
        assert abs(v - ref) <= 1e-10 * abs(ref), (o32, o35, o37)
        vals.append(v)
        single = np.array([gp.lml(t) for t in th])
        for group in (8, 1, 3):  # option 38: a batch's k-segmented main-stream updates in column mode -- scheduling only
            gp.set_option(38, group)
            assert np.array_equal(single, gp.lml_batch(th)), (o32, o35, o37, group)
    assert max(vals) - min(vals) <= 1e-11 * abs(ref), vals
    gp.close()


@pytest.mark.parametrize("ntc,ragged", [(3, 0), (4, -20), (5, 0), (7, 0), (8, -37), (9, 0), (19, -1), (20, 0), (23, -100), (24, 0), (25, -3), (28, 0), (29, -64), (31, 0), (32, -7), (33, 0)])
def test_tile_column_counts_around_the_round5_thresholds(ntc, ragged):
    """One stream below 4 tile columns (8 until round 6), column mode from the start up to 31 (24 until round 6), an entry into it behind one or more super-panels
    above, extended panels from 20: every boundary of those rules (and ragged last tiles) against the oracle, LML and gradient,
    and a batch of three against the single entry point."""
    MiGP, orc = _mods()
    N, d = 128 * ntc + ragged, 3
    X, y = orc.synth_problem(N, d, seed=ntc)
    theta = orc.synth_theta(d)
    ref, gref = orc.lml_grad(X, y, ["Matern32"], [], theta)
    gp = MiGP(X, y, "Matern32")
    v, g = gp.lml_grad(theta)
    assert abs(v - ref) <= 1e-10 * abs(ref), (N, v, ref)
    scale = np.maximum(np.abs(gref), 1e-3 * np.max(np.abs(gref)))
    assert np.max(np.abs(g - gref) / scale) <= 1e-7, (N, g, gref)
    assert gp.lml(theta) == v
    th = np.stack([orc.synth_theta(d, kv=1.0 + 0.3 * i) for i in range(3)])
    assert np.array_equal(gp.lml_batch(th), np.array([gp.lml(t) for t in th]))
    gp.close()


@pytest.mark.parametrize("N,d", [(200, 3), (1500, 4), (3000, 6)])
def test_spin_wait_on_the_sequence_word_returns_what_a_stream_synchronisation_returns(N, d):
    """Round 6, option 47: the host spins on the sequence number the evaluation's last kernel publishes in the pinned result
    buffer (behind the scalars / the gradient, released at system scope) instead of synchronising the stream.  Same values,
    bit for bit, with the spin off (0), on (default 2000 us) and with a budget every evaluation runs into (1 us: the call
    falls back to the stream synchronisation and the next 15 calls do not spin) -- LML, LML + gradient, batches, the
    conditional factor behind predict; thetas alternate so that a stale buffer would show."""
    MiGP, orc = _mods()
    X, y = orc.synth_problem(N, d, seed=N)
    ths = [orc.synth_theta(d, kv=1.0 + 0.1 * i) for i in range(6)]
    Xs = X[:16] + 0.01
    gp = MiGP(X, y, "Matern52")
    assert gp.get_option(47, -1) == 2000
    ref = None
    for budget in (0, 2000, 1, 2000):
        gp.set_option(47, budget)
        got = []
        for rep in range(3):
            for t in ths:
                v = gp.lml(t)
                v2, g = gp.lml_grad(t)
                assert v2 == v
                got.append((v, g.tobytes()))
        vb, gb = gp.lml_grad_batch(np.stack(ths))
        got.append((vb.tobytes(), gb.tobytes()))
        got.append(gp.lml_batch(np.stack(ths[::-1])).tobytes())
        m, var = gp.predict(ths[2], Xs)
        got.append((m.tobytes(), var.tobytes()))
        if ref is None:
            ref = got
            o = orc.lml(X, y, ["Matern52"], [], ths[0])
            assert abs(got[0][0] - o) <= 1e-10 * abs(o)
        assert got == ref, budget
    gp.close()
