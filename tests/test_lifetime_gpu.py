"""Library lifetime on the GPU box: deferred context release, xrsfm_ba_quiesce, dlclose right after a large solve.

The reference builds and frees a ceres::Problem per call (/root/reference/src/optimization/ba_solver.cc:596,645,536); the
replacement keeps device blocks cached between calls and releases the host side of a large context on ONE library-owned
thread that is joined when the library is unloaded (xrsfm_ba.hip: Reaper).  These tests run the hazards the round-2 review
named: a process that unloads the library while such a release may still be running, and many large contexts destroyed back
to back."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_DLCLOSE = textwrap.dedent("""
    import sys, _ctypes, numpy as np
    sys.path.insert(0, %r)
    import torch  # noqa: F401
    from xrsfm_amd import capi, synth
    d = synth.make_problem(1000, 500000, 4, seed=4)          # config L: 2M observations, far above the deferred-release threshold
    prob = capi.ProblemArrays(**{k: np.array(d[k], copy=True) for k in capi.ProblemArrays.FIELDS})
    s = capi.solve(prob)                                     # create + run + download + destroy (hands the context to the reaper)
    assert s.n_successful > 3, s.n_successful
    lib = capi.load()
    h = lib._handle
    capi._lib = None
    _ctypes.dlclose(h)                                       # right away: the release of ~1 GB of host vectors may still be running
    maps = open('/proc/self/maps').read()
    print('STILL_MAPPED' if 'libxrsfm_ba.so' in maps else 'UNLOADED')
    print('DONE')
""")


def test_dlclose_right_after_a_large_solve(lib):
    r = subprocess.run([sys.executable, "-c", _DLCLOSE % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DONE" in r.stdout, r.stdout + r.stderr
    assert "UNLOADED" in r.stdout, "dlclose did not unload the library: " + r.stdout


def test_many_large_contexts_destroyed_back_to_back(lib):
    """Eight contexts of 400k observations each, created and destroyed in a row: the reaper's backlog is bounded (further
    contexts are released in place), xrsfm_ba_quiesce() waits for all of it and empties the device cache, and a solve after
    that still works and gives the result of the first one bit for bit."""
    import numpy as np
    from xrsfm_amd import capi, synth
    d = synth.make_problem(100, 100000, 4, seed=11)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    first = None
    for _ in range(8):
        prob = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
        s = capi.solve(prob)
        state = (prob.cam_q.copy(), prob.cam_t.copy(), s.final_cost)
        if first is None:
            first = state
        assert np.array_equal(state[0], first[0]) and np.array_equal(state[1], first[1]) and state[2] == first[2]
    cached = capi.quiesce()
    assert cached > 0                      # the destroyed contexts' device blocks were in the cache ...
    assert capi.quiesce() == 0             # ... and are gone now
    prob = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
    s = capi.solve(prob)
    assert np.array_equal(prob.cam_q, first[0]) and s.final_cost == first[2]


_WATCHDOG = textwrap.dedent("""
    import os, sys, time, numpy as np
    sys.path.insert(0, %r)
    import torch  # noqa: F401
    from xrsfm_amd import capi
    from tests import helpers as H
    arr = H.make(20, 1200, 4, seed=141)
    ctx = capi.Context(H.to_product(arr))
    ctx.comm_init(1, 0, capi.comm_unique_id())          # XRSFM_BA_FORCE_COMM=1: a real 1-rank RCCL communicator, multi() is true
    t0 = time.time()
    try:
        ctx.run()
        print('RUN_RETURNED_OK')
    except RuntimeError as e:
        print('RUN', str(e))
    print('WAITED %%.2f' %% (time.time() - t0))
    for name, call in (('RUN2', ctx.run), ('DOWNLOAD', ctx.download), ('RESET', ctx.reset)):
        try:
            call()
            print(name, 'RETURNED_OK')
        except RuntimeError as e:
            print(name, str(e))
    t0 = time.time()
    ctx.close()
    print('DESTROY %%.2f' %% (time.time() - t0))
    del os.environ['XRSFM_BA_DEBUG_STALL_S']; del os.environ['XRSFM_BA_FORCE_COMM']
    s = capi.solve(H.to_product(arr))                   # the library is still usable: a fresh context on a fresh stream
    print('AFTER', s.n_successful)
    print('DONE')
""")


def test_watchdog_trips_poisons_the_context_and_destroy_returns(lib):
    """VERDICT round 5, 8(c).  XRSFM_BA_DEBUG_STALL_S holds the context's stream for 6 s in front of the first scalar hand-over of the run (a stand-in for an
    all-reduce no peer joins; the kernel ends by itself), XRSFM_BA_WATCHDOG_S=1: xrsfm_ba_run must come back with XRSFM_BA_ECOMM
    (-4) after about a second instead of spinning, every later entry point must refuse the poisoned context with XRSFM_BA_ESTATE
    (-5), xrsfm_ba_destroy must return (ncclCommAbort; the library itself waits for nothing), and the library must go on working."""
    env = dict(os.environ, XRSFM_BA_FORCE_COMM="1", XRSFM_BA_WATCHDOG_S="1", XRSFM_BA_DEBUG_STALL_S="6")
    r = subprocess.run([sys.executable, "-c", _WATCHDOG % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    out = r.stdout
    assert r.returncode == 0 and "DONE" in out, out + r.stderr
    line = {ln.split()[0]: ln for ln in out.splitlines() if ln.strip()}
    assert "-4" in line["RUN"] and "RUN_RETURNED_OK" not in out, out
    assert 0.9 < float(line["WAITED"].split()[1]) < 5.0, out              # the watchdog, not the end of the 6 s stall
    for name in ("RUN2", "DOWNLOAD", "RESET"):
        assert "-5" in line[name], out
    # destroy RETURNS: ncclCommAbort makes RCCL's own kernels leave and then waits for the communicator's stream work — here the stand-in
    # stall kernel, which no abort flag reaches and which ends by itself after 6 s; nothing of the poisoned context goes back to the caches
    assert float(line["DESTROY"].split()[1]) < 8.0, out
    assert int(line["AFTER"].split()[1]) > 2, out
    assert "no progress" in r.stderr


def test_backward_substitution_timeout_stops_the_run_at_the_next_hand_over(lib, monkeypatch, capfd):
    """VERDICT round 5, 8(b) / ADVICE round 4 #5.  XRSFM_BA_DEBUG_BWD_TIMEOUT=1 makes every hand-off of the one-launch backward
    substitution (k_lv_bwd_all) wait for a tag that never comes, with a short spin bound: the kernel raises scalar slot S_BWD_ERR,
    which reaches the host with the very next hand-over of the scalar block — xrsfm_ba_run must stop THERE with
    XRSFM_BA_EINTERNAL (-7), not iterate on a garbage step until the end of the solve.  The context stays usable."""
    import numpy as np
    from xrsfm_amd import capi
    from tests import helpers as H
    arr = H.make(60, 3000, 4, seed=77)                    # 6 tile columns: a level schedule with >= 2 levels
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["level_schedule"] == 1 and plan["levels"] >= 2
    ref = H.to_product(arr)
    s_ref = capi.solve(ref, capi.default_options(linear_solver=1, max_iterations=8))
    monkeypatch.setenv("XRSFM_BA_DEBUG_BWD_TIMEOUT", "1")
    ctx = capi.Context(H.to_product(arr))
    capfd.readouterr()
    with pytest.raises(RuntimeError, match="-7"):
        ctx.run(capi.default_options(linear_solver=1, max_iterations=8, verbose=1))
    ctx.close()
    cap = capfd.readouterr()
    # the run ended with the FIRST step's scalars: only iteration 0 was ever reported
    it_lines = [ln for ln in cap.out.splitlines() if ln[:4].strip().isdigit()]
    assert len(it_lines) <= 1, cap.out
    assert "timed out" in cap.err
    monkeypatch.delenv("XRSFM_BA_DEBUG_BWD_TIMEOUT")
    again = H.to_product(arr)
    s = capi.solve(again, capi.default_options(linear_solver=1, max_iterations=8))
    assert s.final_cost == s_ref.final_cost and np.array_equal(again.cam_q, ref.cam_q)
