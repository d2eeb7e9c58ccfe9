"""GPU: the drop-in mode the solvers use -- several processes (PolyChord's MPI ranks, cli.py:110), each with its own
context on the one GPU, each calling lnlhood_pc one theta at a time -- gives every process the same bits a single
process gets, within the parity bar of the oracle."""
import os
import sys
import time

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import workloads
from cases import oracle_synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _broker_rank(name, rank, rows, q):
    """A solver rank of the broker test: no device context, one theta per call."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from mcalf_amd import broker
    cl = broker.BrokerClient(name, rank)
    q.put((rank, [cl.lnlhood_pc(p)[0] for p in rows]))
    cl.close()


def test_two_concurrent_processes_get_the_bits_of_one():
    import dropin_ranks
    one = dropin_ranks.run("B", 1, 150)
    two = dropin_ranks.run("B", 2, 150)
    assert one["bit_equal_across_ranks"] and two["bit_equal_across_ranks"]
    assert one["shared_logL_rank0"] == two["shared_logL_rank0"]           # 32 thetas, bit for bit
    assert max(one["max_abs_dlogL_vs_oracle"], two["max_abs_dlogL_vs_oracle"]) < 1e-4
    assert two["aggregate_logL_per_s"] > 0


def test_ranks_behind_the_likelihood_broker_get_the_bits_of_their_own_context():
    """mc-alf_amd/broker.py: ONE process owns the device context and serves solver ranks (processes WITHOUT a GPU context)
    through shared memory, evaluating their thetas in batches.  Every rank's logL equals what `lnlhood_pc` of an own
    context gives, to the bit (a live point's value does not depend on the batch it arrives in)."""
    import multiprocessing as mp
    from mcalf_amd import broker
    kw, _, seed = workloads.config("B", oracle_synth)
    P = workloads.draw_P(kw, 24, np.random.default_rng(seed + 5))
    name = f"mcalf_gputest_{os.getpid()}"
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = [fit.lnlhood_pc(p)[0] for p in P]
        with broker.LikelihoodBroker(fit, name, slots=4) as b:
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            procs = [ctx.Process(target=_broker_rank, args=(name, r, P[r::3], q)) for r in range(3)]
            for p in procs:
                p.start()
            got = {}
            t0 = time.time()
            while len(got) < 3 and time.time() - t0 < 240:
                b.poll()
                while not q.empty():
                    r, vals = q.get()
                    got[r] = vals
            for p in procs:
                p.join(timeout=30)
            assert len(got) == 3
            for r in range(3):
                assert got[r] == want[r::3]
            assert b.stats["thetas"] == 24


def test_the_librarys_broker_loop_over_two_contexts_gives_the_same_bits():
    """mcalf_broker_serve: the serving loop inside the library, one C thread over TWO contexts of the same problem (requests
    that arrive while a launch is in flight leave on the other context).  Three ranks without a device context; every logL
    equals lnlhood_pc of an own context to the bit; the loop returns once the stop flag is raised."""
    import multiprocessing as mp
    import threading
    from mcalf_amd import broker
    kw, _, seed = workloads.config("B", oracle_synth)
    P = workloads.draw_P(kw, 30, np.random.default_rng(seed + 6))
    name = f"mcalf_gputest_native_{os.getpid()}"
    with mcalf_amd.als_fitter(None, **kw) as fit, mcalf_amd.als_fitter(None, **kw) as fit2:
        want = [fit.lnlhood_pc(p)[0] for p in P]
        with broker.LikelihoodBroker([fit, fit2], name, slots=4) as b:
            assert b.native
            server = threading.Thread(target=b.serve_native, kwargs={"max_seconds": 240.0})
            server.start()
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            procs = [ctx.Process(target=_broker_rank, args=(name, r, P[r::3], q)) for r in range(3)]
            for p in procs:
                p.start()
            got = {}
            t0 = time.time()
            while len(got) < 3 and time.time() - t0 < 240 and server.is_alive():
                try:
                    r, vals = q.get(timeout=0.2)
                    got[r] = vals
                except Exception:  # noqa: BLE001 - queue.Empty
                    pass
            b.stop()
            server.join(timeout=60)
            for p in procs:
                p.join(timeout=30)
            assert not server.is_alive() and len(got) == 3
            for r in range(3):
                assert got[r] == want[r::3]
            assert b.stats["thetas"] == 30 and 1 <= b.stats["batches"] <= 30
        # a context listed twice is refused before anything is served
        with broker.LikelihoodBroker([fit, fit], name + "x", slots=2) as b2:
            with pytest.raises(RuntimeError, match="listed twice"):
                b2.serve_native(max_seconds=1.0)


@pytest.mark.parametrize("cfg", ["A", "B", "E"])
def test_small_calls_read_their_completion_off_the_results(cfg, monkeypatch):
    """One theta per call (and small batches) through the host-pointer entry: the call returns when every result slot of the
    page-locked block has been written (no stream wait: `stream_polled`), with the bits the stream-wait form gives
    (MCALF_STREAM_POLL=0) -- for a row of NaN parameters too (every term dropped by the nansum, hires_fitter.py:294)."""
    kw, _, seed = workloads.config(cfg, oracle_synth)
    P = workloads.draw_P(kw, 40, np.random.default_rng(seed + 11), damped=2 if cfg == "E" else 0)
    P[7, 1:] = np.nan
    got = {}
    for poll in ("1", "0"):
        monkeypatch.setenv("MCALF_STREAM_POLL", poll)
        with mcalf_amd.als_fitter(None, **kw) as fit:
            one = np.array([fit.lnlhood_dy(p) for p in P[:12]])
            ll1 = fit.last_launch()
            many = fit.loglike_batch(P)
            ll = fit.last_launch()
            assert ll.path == mcalf_amd._lib.MCALF_PATH_HOST_ZEROCOPY and ll.stream_polled == int(poll) == ll1.stream_polled
            got[poll] = (one, many)
    assert np.isfinite(np.delete(got["1"][1], 7)).all()
    assert np.array_equal(got["1"][0], got["1"][1][:12], equal_nan=True)
    for k in range(2):
        assert np.array_equal(got["1"][k], got["0"][k], equal_nan=True)


def test_ranks_behind_the_broker_with_resident_workgroups_get_the_same_bits():
    """mcalf_broker_serve_resident: every rank's mailbox in the shared block is polled by a workgroup that stays on the GPU; the
    serving thread only (re)starts the launch of those workgroups (at start-up and after they have all left -- the idle limit
    here is short enough for both to happen).  Ranks hold no device context; same client class; bits of an own context."""
    import multiprocessing as mp
    import threading
    from mcalf_amd import broker
    kw, _, seed = workloads.config("B", oracle_synth)
    P = workloads.draw_P(kw, 30, np.random.default_rng(seed + 8))
    name = f"mcalf_gputest_resident_{os.getpid()}"
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = [fit.lnlhood_pc(p)[0] for p in P]
        with broker.LikelihoodBroker(fit, name, slots=4, resident_us=150) as b:
            server = threading.Thread(target=b.serve_native, kwargs={"max_seconds": 240.0})
            server.start()
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            procs = [ctx.Process(target=_broker_rank, args=(name, r, P[r::3], q)) for r in range(3)]
            for p in procs:
                p.start()
            got = {}
            t0 = time.time()
            while len(got) < 3 and time.time() - t0 < 240 and server.is_alive():
                try:
                    r, vals = q.get(timeout=0.2)
                    got[r] = vals
                except Exception:  # noqa: BLE001 - queue.Empty
                    pass
            stats = b.stats
            b.stop()
            server.join(timeout=60)
            for p in procs:
                p.join(timeout=30)
            assert not server.is_alive() and len(got) == 3
            for r in range(3):
                assert got[r] == want[r::3]
            assert stats["thetas"] == 30 and stats["batches"] >= 1      # (launches of the resident grid: it leaves when every rank is quiet)
        # the context is as good as before
        assert fit.lnlhood_pc(P[0])[0] == want[0]
