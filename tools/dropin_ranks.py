#!/usr/bin/env python3
"""Drop-in mode under several solver ranks (what BASELINE's north-star calls "drops into the existing solvers
unchanged"): R processes -- PolyChord's MPI workers, cli.py:37-41,110 -- each with its own als_fitter context on the
ONE GPU of a box, each calling `lnlhood_pc(theta)` serially, one theta per call.  Reports per-call latency and the
aggregate logL/s for every R, with a parity check (oracle) and a cross-process bit-equality check in every run.

    python tools/dropin_ranks.py [--config B] [--ranks 1,2,4,6,6x2,6x4] [--calls 2000] [--out profiles/r04_dropin.json]

The parent never touches the GPU (a GPU box admits at most 6 processes on its card at once).  `bR` = R solver ranks WITHOUT device
contexts behind one likelihood broker (mc-alf_amd/broker.py: one process owns the GPU and evaluates the ranks' thetas in
batches); `PxT` = P processes with T solver threads each, every thread with its OWN context (ctypes releases the GIL for the duration of the library call):
the way to put more than six contexts on the card of a box -- the GPU sees P x T independent streams of one-theta
calls, the host side of a process serialises its threads' few microseconds of Python per call."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(cfg, rank, nranks, calls, work):
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import mcalf_amd
    from mcalf_amd import workloads
    from cases import oracle_synth, problem_from_kwargs
    from oracle import numpy_oracle as orc

    kw, _, seed = workloads.config(cfg, oracle_synth)
    common = workloads.draw_P(kw, 32, np.random.default_rng(seed + 1000), damped=2 if cfg == "E" else 0)   # same in every rank
    own = workloads.draw_P(kw, calls, np.random.default_rng(seed + 2000 + rank), damped=2 if cfg == "E" else 0)
    fit = mcalf_amd.als_fitter(None, **kw)
    for p in own[:50]:
        fit.lnlhood_pc(p)
    open(os.path.join(work, f"ready{rank}"), "w").close()
    t0 = time.time()
    while not os.path.exists(os.path.join(work, "go")):
        assert time.time() - t0 < 120
        time.sleep(0.0005)
    lat = np.empty(calls)
    tb = time.perf_counter()
    for i, p in enumerate(own):
        t1 = time.perf_counter()
        fit.lnlhood_pc(p)
        lat[i] = time.perf_counter() - t1
    wall = time.perf_counter() - tb
    shared = [fit.lnlhood_pc(p)[0] for p in common]                      # while the other ranks may still be running
    prob = problem_from_kwargs(kw)
    want = np.array([orc.lnlhood_worker(prob, p) for p in common[:8]])
    res = {"rank": rank, "calls": calls, "wall_s": wall, "us_mean": float(lat.mean() * 1e6), "us_median": float(np.median(lat) * 1e6),
           "us_p99": float(np.percentile(lat, 99) * 1e6), "shared_logL": shared,
           "max_abs_dlogL_vs_oracle": float(np.abs(np.array(shared[:8]) - want).max())}
    fit.close()
    with open(os.path.join(work, f"res{rank}.json"), "w") as fh:
        json.dump(res, fh)


def worker_threads(cfg, rank, nranks, calls, work, nthreads):
    """`nthreads` solver threads in this process, one context each; the process reports their sum."""
    import threading
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import mcalf_amd
    from mcalf_amd import workloads
    from cases import oracle_synth, problem_from_kwargs
    from oracle import numpy_oracle as orc

    kw, _, seed = workloads.config(cfg, oracle_synth)
    common = workloads.draw_P(kw, 32, np.random.default_rng(seed + 1000), damped=2 if cfg == "E" else 0)
    fits = [mcalf_amd.als_fitter(None, **kw) for _ in range(nthreads)]
    owns = [workloads.draw_P(kw, calls, np.random.default_rng(seed + 2000 + rank * 64 + t), damped=2 if cfg == "E" else 0)
            for t in range(nthreads)]
    for f, own in zip(fits, owns):
        for p in own[:50]:
            f.lnlhood_pc(p)
    open(os.path.join(work, f"ready{rank}"), "w").close()
    t0 = time.time()
    while not os.path.exists(os.path.join(work, "go")):
        assert time.time() - t0 < 120
        time.sleep(0.0005)
    lats = [np.empty(calls) for _ in range(nthreads)]
    walls = [0.0] * nthreads

    def loop(t):
        f, own, lat = fits[t], owns[t], lats[t]
        tb = time.perf_counter()
        for i, p in enumerate(own):
            t1 = time.perf_counter()
            f.lnlhood_pc(p)
            lat[i] = time.perf_counter() - t1
        walls[t] = time.perf_counter() - tb
    th = [threading.Thread(target=loop, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    shared = [fits[0].lnlhood_pc(p)[0] for p in common]
    same_in_proc = all([f.lnlhood_pc(p)[0] for p in common] == shared for f in fits[1:])
    prob = problem_from_kwargs(kw)
    want = np.array([orc.lnlhood_worker(prob, p) for p in common[:8]])
    lat = np.concatenate(lats)
    res = {"rank": rank, "calls": calls * nthreads, "wall_s": max(walls), "us_mean": float(lat.mean() * 1e6),
           "us_median": float(np.median(lat) * 1e6), "us_p99": float(np.percentile(lat, 99) * 1e6), "shared_logL": shared,
           "threads_bit_equal": bool(same_in_proc),
           "max_abs_dlogL_vs_oracle": float(np.abs(np.array(shared[:8]) - want).max())}
    for f in fits:
        f.close()
    with open(os.path.join(work, f"res{rank}.json"), "w") as fh:
        json.dump(res, fh)


def broker_server(cfg, name, slots, work, lanes=0, resident_us=0):
    """The one process with device contexts: serves the clients' thetas in batches (mc-alf_amd/broker.py).  lanes = 0: the
    Python loop, one context; lanes >= 1: the loop inside the library (mcalf_broker_serve) over that many contexts."""
    import threading
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import mcalf_amd
    from mcalf_amd import broker, workloads
    from cases import oracle_synth
    kw, _, _ = workloads.config(cfg, oracle_synth)
    fits = [mcalf_amd.als_fitter(None, **kw) for _ in range(max(lanes, 1))]
    stop_file = os.path.join(work, "stop")
    with broker.LikelihoodBroker(fits if not resident_us else fits[0], name, slots=slots, resident_us=resident_us) as b:
        open(os.path.join(work, "server_ready"), "w").close()
        if lanes == 0 and not resident_us:
            b.serve(stop_when=lambda: os.path.exists(stop_file), native=False)
        else:
            def watch():
                while not os.path.exists(stop_file):
                    time.sleep(0.01)
                b.stop()
            threading.Thread(target=watch, daemon=True).start()
            b.serve_native()
        with open(os.path.join(work, "server.json"), "w") as fh:
            json.dump(b.stats, fh)
    for f in fits:
        f.close()


def broker_client(cfg, rank, nranks, calls, work, name):
    """A solver rank without a device context: one theta per call through the broker."""
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from mcalf_amd import broker, workloads
    from cases import oracle_synth, problem_from_kwargs
    from oracle import numpy_oracle as orc
    kw, _, seed = workloads.config(cfg, oracle_synth)
    common = workloads.draw_P(kw, 32, np.random.default_rng(seed + 1000), damped=2 if cfg == "E" else 0)
    own = workloads.draw_P(kw, calls, np.random.default_rng(seed + 2000 + rank), damped=2 if cfg == "E" else 0)
    cl = broker.BrokerClient(name, rank)
    for p in own[:50]:
        cl.lnlhood_pc(p)
    open(os.path.join(work, f"ready{rank}"), "w").close()
    t0 = time.time()
    while not os.path.exists(os.path.join(work, "go")):
        assert time.time() - t0 < 120
        time.sleep(0.0005)
    lat = np.empty(calls)
    tb = time.perf_counter()
    for i, p in enumerate(own):
        t1 = time.perf_counter()
        cl.lnlhood_pc(p)
        lat[i] = time.perf_counter() - t1
    wall = time.perf_counter() - tb
    shared = [cl.lnlhood_pc(p)[0] for p in common]
    want = np.array([orc.lnlhood_worker(problem_from_kwargs(kw), p) for p in common[:8]]) if rank == 0 else np.array(shared[:8])
    res = {"rank": rank, "calls": calls, "wall_s": wall, "us_mean": float(lat.mean() * 1e6), "us_median": float(np.median(lat) * 1e6),
           "us_p99": float(np.percentile(lat, 99) * 1e6), "shared_logL": shared,
           "max_abs_dlogL_vs_oracle": float(np.abs(np.array(shared[:8]) - want).max())}
    cl.close()
    with open(os.path.join(work, f"res{rank}.json"), "w") as fh:
        json.dump(res, fh)


def run_broker(cfg, nranks, calls, lanes=0, resident_us=0):
    work = tempfile.mkdtemp(prefix="mcalf_broker_")
    name = "mcalf_dropin_%d" % os.getpid()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    me = os.path.abspath(__file__)
    server = subprocess.Popen([sys.executable, me, "--broker-server", cfg, name, str(max(nranks, 1)), work, str(lanes), str(resident_us)], env=env)
    t0 = time.time()
    while not os.path.exists(os.path.join(work, "server_ready")):
        if time.time() - t0 > 300 or server.poll() is not None:
            raise SystemExit("the broker did not come up")
        time.sleep(0.01)
    procs = [subprocess.Popen([sys.executable, me, "--broker-client", cfg, str(r), str(nranks), str(calls), work, name], env=env)
             for r in range(nranks)]
    while not all(os.path.exists(os.path.join(work, f"ready{r}")) for r in range(nranks)):
        if time.time() - t0 > 300 or any(p.poll() not in (None, 0) for p in procs):
            for p in procs + [server]:
                p.kill()
            raise SystemExit("a client did not come up")
        time.sleep(0.01)
    open(os.path.join(work, "go"), "w").close()
    for p in procs:
        if p.wait(timeout=600) != 0:
            server.kill()
            raise SystemExit("a client failed")
    open(os.path.join(work, "stop"), "w").close()
    server.wait(timeout=60)
    res = [json.load(open(os.path.join(work, f"res{r}.json"))) for r in range(nranks)]
    st = json.load(open(os.path.join(work, "server.json")))
    same = all(r["shared_logL"] == res[0]["shared_logL"] for r in res)
    return {"ranks": nranks, "processes": nranks, "threads_per_process": 1, "broker": True, "calls_per_rank": calls,
            "server_loop": ("resident workgroups (mcalf_broker_serve_resident), idle limit %d us" % resident_us) if resident_us else
                           ("python, 1 context" if lanes == 0 else "library (mcalf_broker_serve), %d context(s)" % lanes),
            "thetas_per_launch": st["thetas"] / max(st["batches"], 1),
            "aggregate_logL_per_s": sum(r["calls"] / r["wall_s"] for r in res),
            "us_per_call_mean": sum(r["us_mean"] for r in res) / nranks,
            "us_per_call_median": sorted(r["us_median"] for r in res)[nranks // 2],
            "us_per_call_p99_max": max(r["us_p99"] for r in res),
            "bit_equal_across_ranks": same, "shared_logL_rank0": res[0]["shared_logL"],
            "max_abs_dlogL_vs_oracle": max(r["max_abs_dlogL_vs_oracle"] for r in res)}


def run(cfg, nranks, calls, nthreads=1):
    work = tempfile.mkdtemp(prefix="mcalf_dropin_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", cfg, str(r), str(nranks), str(calls), work,
                               str(nthreads)], env=env) for r in range(nranks)]
    t0 = time.time()
    while not all(os.path.exists(os.path.join(work, f"ready{r}")) for r in range(nranks)):
        if time.time() - t0 > 300 or any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                p.kill()
            raise SystemExit("a worker did not come up")
        time.sleep(0.01)
    open(os.path.join(work, "go"), "w").close()
    for p in procs:
        if p.wait(timeout=600) != 0:
            raise SystemExit("a worker failed")
    res = [json.load(open(os.path.join(work, f"res{r}.json"))) for r in range(nranks)]
    same = all(r["shared_logL"] == res[0]["shared_logL"] and r.get("threads_bit_equal", True) for r in res)
    return {"ranks": nranks * nthreads, "processes": nranks, "threads_per_process": nthreads, "calls_per_rank": calls,
            "aggregate_logL_per_s": sum(r["calls"] / r["wall_s"] for r in res),
            "us_per_call_mean": sum(r["us_mean"] for r in res) / nranks,
            "us_per_call_median": sorted(r["us_median"] for r in res)[nranks // 2],
            "us_per_call_p99_max": max(r["us_p99"] for r in res),
            "bit_equal_across_ranks": same, "shared_logL_rank0": res[0]["shared_logL"],
            "max_abs_dlogL_vs_oracle": max(r["max_abs_dlogL_vs_oracle"] for r in res)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--broker-server":
        broker_server(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 0,
                      int(sys.argv[7]) if len(sys.argv) > 7 else 0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--broker-client":
        broker_client(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], sys.argv[7])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        nthr = int(sys.argv[7]) if len(sys.argv) > 7 else 1
        if nthr > 1:
            worker_threads(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], nthr)
        else:
            worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--ranks", default="1,2,4,6")
    ap.add_argument("--calls", type=int, default=2000)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rows = []
    for spec in args.ranks.split(","):
        if spec.startswith("b"):                            # bR: R ranks WITHOUT device contexts behind one likelihood broker
            if "r" in spec:                                 # bRrU: resident workgroups, idle limit U microseconds
                nr, _, us = spec[1:].partition("r")
                rows.append(run_broker(args.config, int(nr), args.calls, 0, int(us)))
                continue
            nr, _, ln = spec[1:].partition("l")             # (Python loop); bRlL: the library's loop over L contexts
            rows.append(run_broker(args.config, int(nr), args.calls, int(ln) if ln else 0))
            continue
        pr, _, thr = spec.partition("x")
        rows.append(run(args.config, int(pr), args.calls, int(thr) if thr else 1))
    ref = rows[0]["shared_logL_rank0"]
    out = {"what": "R solver ranks, one als_fitter context each on ONE MI355X, lnlhood_pc(theta) one theta per call "
                   "(PolyChord's MPI workers, cli.py:37-41,110)", "config": args.config, "runs": rows,
           "bit_equal_across_runs": all(r["shared_logL_rank0"] == ref for r in rows)}
    for r in rows:
        del r["shared_logL_rank0"]
        if r.get("broker"):
            print("R = %d ranks behind ONE broker [%s] (%.1f thetas per launch):" % (r["ranks"], r["server_loop"], r["thetas_per_launch"]), end=" ")
        print("R = %d (%d processes x %d threads): %.1f us per call (median %.1f, worst p99 %.1f), %.0f logL/s aggregate, bit-equal %s, |dlogL| vs oracle %.1e"
              % (r["ranks"], r["processes"], r["threads_per_process"], r["us_per_call_mean"], r["us_per_call_median"], r["us_per_call_p99_max"], r["aggregate_logL_per_s"],
                 r["bit_equal_across_ranks"], r["max_abs_dlogL_vs_oracle"]))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
