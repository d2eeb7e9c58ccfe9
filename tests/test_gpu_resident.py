"""GPU: the resident one-theta evaluator (mcalf_set_resident) -- one workgroup that stays on the chip between the solvers'
one-theta calls (lnlhood_pc / _dy / _mn, hires_fitter.py:250-285) and answers them from a page-locked mailbox without a
launch.  Same code as the launched form, so the same bits; it leaves by itself after its idle limit, and nothing else of
the context notices it."""
import os
import sys
import time

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import workloads
from cases import oracle_synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("cfg,conv", [("A", "numpy"), ("C", "numpy"), ("B", "jax")])
def test_resident_evaluator_gives_the_bits_of_the_launched_call(cfg, conv):
    kw, _, seed = workloads.config(cfg, oracle_synth)
    P = workloads.draw_P(kw, 96, np.random.default_rng(seed + 21))
    with mcalf_amd.als_fitter(None, conv_mode=conv, **kw) as fit:
        P[5, fit.startind + 1:] = np.nan                     # every term dropped by the nansum (hires_fitter.py:294)
        want = np.array([fit.lnlhood_dy(p) for p in P])
        assert fit.last_launch().inline_setup == 1
        want_batch = fit.loglike_batch(P)
        want_chi2 = fit.chi2(P[0])
        want_model = fit.reconstruct_spec(P[1])
        fit.set_resident(300)
        got = np.array([fit.lnlhood_dy(p) for p in P])
        assert fit.last_launch().inline_setup == 3          # answered without a launch
        assert np.array_equal(got, want, equal_nan=True)
        assert [fit.lnlhood_pc(p)[0] for p in P[:4]] == list(want[:4]) and fit.lnlhood_mn(list(P[2]), fit.ndim, fit.ndim) == want[2]
        # the context's other entries while the evaluator is alive
        assert np.array_equal(fit.loglike_batch(P), want_batch, equal_nan=True)
        assert fit.chi2(P[0]) == want_chi2 and np.array_equal(fit.reconstruct_spec(P[1]), want_model)
        assert fit.lnlhood_dy(P[3]) == want[3]
        # it leaves by itself: a device-wide synchronisation long after the last call returns at once ...
        time.sleep(0.02)
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.05
        # ... and the next call starts another one
        again = np.array([fit.lnlhood_dy(p) for p in P[:16]])
        assert np.array_equal(again, want[:16], equal_nan=True) and fit.last_launch().inline_setup == 3
        fit.set_resident(0)                                  # told to leave now; back to launches
        assert fit.lnlhood_dy(P[7]) == want[7] and fit.last_launch().inline_setup == 1
    # (the context closes with nothing left behind)


def test_a_context_closed_while_its_evaluator_is_alive_and_two_contexts_side_by_side():
    kw, _, seed = workloads.config("B", oracle_synth)
    P = workloads.draw_P(kw, 40, np.random.default_rng(seed + 22))
    with mcalf_amd.als_fitter(None, **kw) as ref:
        want = np.array([ref.lnlhood_dy(p) for p in P])
    a = mcalf_amd.als_fitter(None, **kw)
    a.set_resident(100000)                                   # a long idle limit: closing tells the kernel to leave, it does not wait
    assert [a.lnlhood_dy(p) for p in P[:4]] == list(want[:4])
    t0 = time.perf_counter()
    a.close()
    assert time.perf_counter() - t0 < 0.05
    a = mcalf_amd.als_fitter(None, **kw)
    b = mcalf_amd.als_fitter(None, **kw)
    a.set_resident(2000)
    b.set_resident(2000)
    got_a, got_b = [], []
    for p in P:                                              # interleaved: two mailboxes, two resident workgroups
        got_a.append(a.lnlhood_dy(p))
        got_b.append(b.lnlhood_dy(p))
    assert np.array_equal(got_a, want) and np.array_equal(got_b, want)
    t0 = time.perf_counter()
    a.close()                                                # (freeing device memory synchronises the device: that waits for
    b.close()                                                # b's evaluator at most its idle limit, 2 ms)
    assert time.perf_counter() - t0 < 0.1


def test_a_tiled_spectrum_and_batches_keep_the_launched_path():
    kw, _, seed = workloads.config("E", oracle_synth)
    P = workloads.draw_P(kw, 6, np.random.default_rng(seed + 23), damped=2)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = [fit.lnlhood_dy(p) for p in P]
        fit.set_resident(300)
        assert [fit.lnlhood_dy(p) for p in P] == want and fit.last_launch().inline_setup == 1     # five tiles: launched
        with pytest.raises(RuntimeError, match="idle limit"):
            fit.set_resident(-1)


def test_solver_ranks_with_resident_evaluators_get_the_bits_of_one(monkeypatch):
    """Two processes, a context each, MCALF_RESIDENT_US in their environment (how an MPI launcher would turn it on)."""
    import dropin_ranks
    one = dropin_ranks.run("B", 1, 120)
    monkeypatch.setenv("MCALF_RESIDENT_US", "400")
    two = dropin_ranks.run("B", 2, 120)
    assert one["bit_equal_across_ranks"] and two["bit_equal_across_ranks"]
    assert one["shared_logL_rank0"] == two["shared_logL_rank0"]
    assert max(one["max_abs_dlogL_vs_oracle"], two["max_abs_dlogL_vs_oracle"]) < 1e-4
