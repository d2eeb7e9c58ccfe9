#!/usr/bin/env python3
"""Soak (GPU box): the broker with resident workgroups under ranks whose gaps straddle the idle limit -- the workgroups leave
and are restarted again and again while other ranks keep asking; every answer compared with an own context's bits.
    python tools/soak_resident_broker.py [ranks] [calls per rank] [idle_us]"""
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def rank(name, r, calls, idle, rows_file, q):
    from mcalf_amd import broker
    d = np.load(rows_file)
    P, want = d["P"], d["want"]
    cl = broker.BrokerClient(name, r)
    rng = np.random.default_rng(1000 + r)
    gaps = np.array([0, 0, 0, 0.3, 0.8, 1.0, 1.2, 3.0, 10.0]) * idle * 1e-6
    bad = 0
    for _ in range(calls):
        k = int(rng.integers(len(P)))
        g = float(gaps[int(rng.integers(gaps.size))])
        if g > 0:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < g:
                pass
        if cl.lnlhood_dy(P[k]) != want[k]:
            bad += 1
    q.put((r, bad))
    cl.close()


def main():
    import mcalf_amd
    from mcalf_amd import broker, workloads
    from cases import oracle_synth
    nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    idle = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    kw, _, seed = workloads.config("A", oracle_synth)
    P = workloads.draw_P(kw, 64, np.random.default_rng(seed + 33))
    name = "mcalf_soak_%d" % os.getpid()
    rows_file = "/tmp/mcalf_soak_%d.npz" % os.getpid()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = np.array([fit.lnlhood_dy(p) for p in P])
        np.savez(rows_file, P=P, want=want)
        with broker.LikelihoodBroker(fit, name, slots=nr, resident_us=idle) as b:
            th = threading.Thread(target=b.serve_native, kwargs={"max_seconds": 600.0})
            th.start()
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            procs = [ctx.Process(target=rank, args=(name, r, calls, idle, rows_file, q)) for r in range(nr)]
            t0 = time.time()
            for p in procs:
                p.start()
            res = [q.get(timeout=550) for _ in procs]
            for p in procs:
                p.join(timeout=30)
            st = b.stats
            b.stop()
            th.join(timeout=60)
    os.remove(rows_file)
    bad = sum(b_ for _, b_ in res)
    print("DONE: %d ranks x %d calls in %.1f s, %d bad, %d launches of the resident grid (idle limit %d us), %d thetas acknowledged"
          % (nr, calls, time.time() - t0, bad, st["batches"], idle, st["thetas"]))
    sys.exit(1 if bad or st["thetas"] != nr * calls else 0)


if __name__ == "__main__":
    main()
