#!/usr/bin/env python3
"""Benchmark of the MC-ALF likelihood hot path on MI355X.

A "step" is one `loglike_batch` pass over one batch of synthetic live points (BASELINE.json
config B at N=1: batch=1024, ncomp=8, CIV doublet, npix=4000, fixed 8 km/s LSF) with the
parameter matrix already resident in HBM and logL left in HBM.  With N GPUs every rank
runs the same per-GPU batch (weak scaling; config B -> 8 x B, the C -> D pattern of
BASELINE.json) and the per-sample logL shards are gathered to rank 0 over RCCL each step.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument`
# otherwise); the launch environments export it already, this only covers a bare shell
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import mcalf_amd  # noqa: E402
from mcalf_amd import _lib, workloads  # noqa: E402
from mcalf_amd import dist as mdist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 peak


def hip_synth(kw, p):
    """Synthesise a truth spectrum with the HIP path itself (no oracle in the product path)."""
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, dtype=float))


def cpu_baseline(kw, P, budget_s):
    """Time the numpy/scipy oracle (the closest runnable stand-in for the reference's numpy
    path) on a bounded sample of the same workload.  Checker/baseline only."""
    from oracle import numpy_oracle as oracle
    wl, flux, err = kw["spectrum"]
    prob = oracle.Problem(wl, flux, err, kw["linepars"], tuple(kw["ncomp"]), nfill=kw.get("nfill", 0),
                          specres=kw["specres"], Nrange=kw["Nrange"], brange=kw["brange"], zrange=kw["zrange"],
                          Nrangefill=kw.get("Nrangefill", [11.5, 16]), brangefill=kw.get("brangefill", [1, 30]),
                          fitrange=kw["fitrange"])
    oracle.lnlhood_worker(prob, P[0])          # warm
    vals, t0, done = [], time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s or done < (len(P) if budget_s <= 0 else 0):   # cycle over the batch until the time budget is spent (budget 0: every row once)
        row = P[done % len(P)]
        v = oracle.lnlhood_worker(prob, row)
        if done < len(P):
            vals.append(v)
        done += 1
    dt = time.perf_counter() - t0
    return np.array(vals), dt, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="B", choices=["A", "B", "C", "D", "E"])
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline time budget (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real multi-GPU path) or gloo: a rehearsal of the N>1 control flow on a "
                         "one-GPU box (all ranks share cuda:0, logL shards gathered through host memory)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the second CPU baseline (plain-C/OpenMP oracle); 0 = skip it")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = the config's batch per GPU (default, BASELINE's C -> D pattern); strong = the "
                         "config's batch split over the GPUs")
    ap.add_argument("--pinned", action="store_true", help="with --host-api: P and logL in page-locked host memory")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent batches kept in flight (contexts + streams); 1 = the headline configuration")
    ap.add_argument("--no-launch-events", action="store_true",
                    help="diagnostic: do not bracket each fused launch with HIP events (measures their overhead)")
    ap.add_argument("--host-api", action="store_true",
                    help="diagnostic: time the host-pointer entry (H2D of P and D2H of logL inside the step); "
                         "never the headline value")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    # MCALF_BENCH_FORCE_DIST=1 (with torch.distributed.run --nproc-per-node 1) runs the N>1 code path --
    # process group, RCCL gather ring, barrier, all_reduce -- on a single rank: a one-GPU check of the
    # collective plumbing the driver's multi-GPU runs use.
    use_dist = world > 1 or os.environ.get("MCALF_BENCH_FORCE_DIST") == "1"
    rehearsal = world > 1 and args.backend == "gloo"
    if rehearsal:
        local_rank = 0                                   # every rank on the one GPU of the box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    kw, batch, seed = workloads.config(args.config, hip_synth)
    if args.batch:
        batch = args.batch
    if args.config == "D":                               # BASELINE config D: 32768 rows job-wide = 4096 per GPU on 8 GPUs
        if batch % world:
            raise SystemExit(f"config D: {batch} rows do not split over {world} ranks")
        batch //= world
    if args.scaling == "strong" and world > 1:
        if batch % world:
            raise SystemExit(f"--scaling strong: batch {batch} is not a multiple of {world} ranks")
        batch //= world                                  # per-rank rows; the job-wide batch stays the config's
    # every rank draws the job-wide matrix from one seed and keeps its contiguous row block
    P_all = workloads.draw_P(kw, batch * world, np.random.default_rng(seed), damped=2 if args.config == "E" else 0)
    P_host = np.ascontiguousarray(P_all[rank * batch:(rank + 1) * batch])
    fit = mcalf_amd.als_fitter(None, device=local_rank, **kw)
    npix, ndim, nlines = fit.obj_wl.size, fit.ndim, fit.numlines
    nc = P_host[:, fit.startind].astype(int)
    comp_pix = float(nc.sum()) * npix                       # component x pixel evals per step (this rank)
    line_pix = float((nc * nlines + fit.nfill).sum()) * npix

    dP = torch.from_numpy(P_host).to(dev)
    # two logL buffers in flight: the (latency-bound) gather of step k overlaps the kernel of step k+1
    plan = mdist.LogLGather(batch * world, "cpu" if rehearsal else dev, depth=2, always_collective=use_dist)
    assert (plan.lo, plan.hi) == (rank * batch, (rank + 1) * batch)
    dlogL = torch.empty(batch, dtype=torch.float64, device=dev) if (rehearsal or not use_dist) else None
    _lib.check(fit._lib.mcalf_reserve(fit._ctx, batch), fit._ctx)
    stream = torch.cuda.current_stream()
    st = C.c_void_p(stream.cuda_stream)
    # optional: further independent batches in flight, each with its own context, stream and output
    extra = []
    if args.inflight > 1 and world == 1 and not args.host_api:
        for _ in range(args.inflight - 1):
            f2 = mcalf_amd.als_fitter(None, device=local_rank, **kw)
            _lib.check(f2._lib.mcalf_reserve(f2._ctx, batch), f2._ctx)
            s2 = torch.cuda.Stream()
            extra.append((f2, s2, torch.empty(batch, dtype=torch.float64, device=dev)))
    turn = [0]
    P_host_api, out_host_api = P_host, None
    if args.host_api and args.pinned:                    # page-locked host buffers: the copies become plain DMA
        P_host_api = torch.from_numpy(P_host).pin_memory().numpy()
        out_host_api = torch.empty(batch, dtype=torch.float64).pin_memory().numpy()
    launch = fit._lib.mcalf_loglike_batch_device
    ctx, pP = fit._ctx, dP.data_ptr()
    last_out = [dlogL]

    def step():
        if args.host_api:
            fit.loglike_batch(P_host_api, out=out_host_api)
            return
        if extra:
            k = turn[0] % (len(extra) + 1)
            turn[0] += 1
            if k > 0:
                f2, s2, o2 = extra[k - 1]
                rc = launch(f2._ctx, pP, batch, o2.data_ptr(), C.c_void_p(s2.cuda_stream))
                if rc:
                    _lib.check(rc, f2._ctx)
                return
        out = dlogL if dlogL is not None else plan.local          # plan.local waits for the slot's old gather
        rc = launch(ctx, pP, batch, out.data_ptr(), st)
        if rc:
            _lib.check(rc, ctx)
        last_out[0] = out
        if use_dist:
            if rehearsal:
                plan.local.copy_(dlogL)                  # through host memory (gloo)
            plan.gather_async()                          # RCCL gather of the logL shards to rank 0

    gathered = [None]

    def fence():
        if use_dist:
            gathered[0] = plan.finish()                  # every outstanding gather has landed on rank 0
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    red_dev = "cpu" if rehearsal else dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([comp_pix, line_pix], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    comp_pix_job, line_pix_job = (float(v) for v in tot.tolist())

    # dominant kernel: average launch duration from HIP events on the launch stream (N=1: the
    # stream carries nothing but the fused kernel; N>1: the gather is on RCCL's own stream)
    stream_ms = ev0.elapsed_time(ev1) / args.steps        # everything on the launch stream per step
    # Dominant kernel alone: a second pass of K launches, each bracketed by HIP events on the launch
    # stream inside the library.  Kept out of the timed region because the brackets themselves cost
    # ~5 us per step (measured), which would otherwise be charged to `value`.
    kern_ms, nl = C.c_double(0.0), C.c_int32(0)
    if not args.host_api and not args.no_launch_events:
        _lib.check(fit._lib.mcalf_profile_begin(fit._ctx, args.steps), fit._ctx)
        for _ in range(args.steps):
            step()
        fence()
        _lib.check(fit._lib.mcalf_profile_end(fit._ctx, C.byref(kern_ms), C.byref(nl)), fit._ctx)
    kern_ms = kern_ms.value if nl.value else stream_ms      # mean duration of mcalf_fused_kernel alone
    logL_dev = last_out[0].cpu().numpy()
    gather_check = None
    if use_dist and rank == 0 and gathered[0] is not None:
        # the vector rank 0 holds after the last gather: its own block must be what it computed, and every
        # other block a finite logL of that rank's rows
        g = gathered[0].cpu().numpy()
        gather_check = {"rows": int(g.size), "own_block_equal": bool(np.array_equal(g[:batch], logL_dev)),
                        "all_finite": bool(np.isfinite(g).all())}

    out = None
    if rank == 0:
        n_half = fit.info.n_cap
        alg_bytes = (8 * ndim + 8) * batch + 24 * npix
        alg_flops = 40.0 * line_pix + (31 + 4 * n_half) * npix * batch
        ach_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None                       # HBM bytes per launch from rocprofv3 PMC passes (offline, profiles/)
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                tr = json.load(fh).get(args.config)
            if tr and not args.batch:
                traffic = tr["traffic_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        out = {
            "metric": "component-pixel Voigt evals/s (sum_s ncomp_s * npix / t), logL batch on MI355X",
            "value": comp_pix_job * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "batches_in_flight": max(1, args.inflight if (world == 1 and not args.host_api) else 1), "entry": ("host pointers (PCIe inclusive%s)" % (", page-locked buffers" if args.pinned else "")) if args.host_api else "device pointers",
            "config": {"workload": ("BASELINE config A: CIV 1548/1550, the reference's multicomponent mock spectrum (tests/golden)"
                                    if args.config == "A" else "BASELINE config E: HI 1215 damped" if args.config == "E"
                                    else f"BASELINE config {args.config}: CIV 1548/1550 synthetic spectrum"), "batch_per_gpu": batch, "global_batch": batch * world,
                       "npix": npix, "ncomp": list(kw["ncomp"]), "nlines": nlines, "nfill": fit.nfill, "ndim": ndim,
                       "specres": list(kw["specres"]), "lsf_taps": 2 * n_half + 1, "tiles_per_sample": fit.info.ntiles,
                       "parallelism": (f"dp{world} rows sharded, {'gloo REHEARSAL on one GPU' if rehearsal else 'RCCL'} gather of logL to rank 0"
                                       if world > 1 else "single GPU")},
            "logL_per_s": batch * world * args.steps / elapsed,
            "line_pixel_evals_per_s": line_pix_job * args.steps / elapsed,
            "kernel_ms": kern_ms, "stream_ms_per_step": stream_ms,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mcalf_fused_kernel", "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "fused path is FP64-VALU bound, not HBM bound (SURVEY.md 8d); see roofline_valu"},
            "roofline_valu": {"bound": "fp64_valu", "achieved": alg_flops / (kern_ms * 1e-3) / 1e12,
                              "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": alg_flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                              "algorithmic_flops_per_launch": alg_flops},
        }
        if args.cpu_seconds > 0 and world == 1:          # the CPU baseline is an N=1 figure (rank 0 only)
            vals, dt, done = cpu_baseline(kw, P_host, args.cpu_seconds)
            k = len(vals)
            evals = sum(float(nc[i % batch]) for i in range(done)) * npix
            out["cpu_baseline"] = {
                "value": evals / dt, "unit": "evals/s", "cores": 1, "kind": "port",
                "sample": f"{done} logL evaluations cycling over the {batch} rows of the same parameter matrix, "
                          f"numpy/scipy float64 oracle (oracle/numpy_oracle.py), {dt:.1f} s, {dt / done * 1e3:.2f} ms per logL",
                "host_cpus": os.cpu_count()}
            out["parity"] = {"max_abs_dlogL_vs_oracle": float(np.abs(vals - logL_dev[:k]).max()), "rows": k}
            if args.cpu_threads > 0:
                # second CPU figure: the plain-C/OpenMP restatement (oracle/c), several host threads
                from oracle import c_oracle, numpy_oracle
                wl_, flux_, err_ = kw["spectrum"]
                prob = numpy_oracle.Problem(wl_, flux_, err_, kw["linepars"], tuple(kw["ncomp"]), nfill=kw.get("nfill", 0),
                                            specres=kw["specres"], Nrange=kw["Nrange"], brange=kw["brange"],
                                            zrange=kw["zrange"], Nrangefill=kw.get("Nrangefill", [11.5, 16]),
                                            brangefill=kw.get("brangefill", [1, 30]), fitrange=kw["fitrange"])
                nthr = max(1, min(args.cpu_threads, os.cpu_count() or 1))
                co = c_oracle.COracle(prob, threads=nthr)
                rows = P_host[: min(batch, 64 * nthr)]
                co.loglike_batch(rows[:nthr])
                tc, reps = time.perf_counter(), 0
                while time.perf_counter() - tc < 5.0:
                    cvals = co.loglike_batch(rows)
                    reps += 1
                dtc = time.perf_counter() - tc
                out["cpu_baseline_c_openmp"] = {
                    "value": reps * float(nc[: len(rows)].sum()) * npix / dtc, "unit": "evals/s", "cores": nthr, "kind": "port",
                    "sample": f"{reps} x {len(rows)} rows, oracle/c/mcalf_oracle.c (gcc -O2 -fopenmp), {dtc:.1f} s",
                    "max_abs_dlogL_vs_gpu": float(np.abs(cvals - logL_dev[: len(rows)]).max())}
    if out is not None and world > 1 and args.cpu_seconds > 0:
        # N>1: no CPU timing, only a parity spot check of rank 0's first rows against the oracle
        vals, _, _ = cpu_baseline(kw, P_host[:8], 0.0)
        out["parity"] = {"max_abs_dlogL_vs_oracle": float(np.abs(vals - logL_dev[:len(vals)]).max()), "rows": len(vals)}
    if out is not None and gather_check is not None:
        out["gather_check"] = gather_check
    fit.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
