#!/usr/bin/env python3
"""Benchmark of the MC-ALF likelihood hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic live points.

  N = 1 (default)  BASELINE.json config C -- the largest single-GPU configuration: batch 4096, ncomp 8-11,
                   4 fillers, LSF resolution floated in [8, 9] km/s, CIV doublet, 4000 pixels.
  N > 1 (default)  BASELINE.json config D -- the same problem with 32768 live points job-wide, cut into
                   contiguous row blocks over the N ranks (strong scaling: the job-wide batch is fixed), the
                   per-sample logL gathered to rank 0 over RCCL every step.  The gather is timed THREE ways in the
                   one process group, in this order: torch.distributed.gather (two buffers in flight), the
                   library's own exchange on the launch stream, the library's exchange on its side stream
                   (overlap); each leg has its `ms_per_step`, `gather_check` and the ranks' min / max kernel time
                   under `gathers`, the best one is `value`.  The two library legs run under a watchdog: a leg
                   that has not finished after 30 s is recorded as "timed_out", the line is printed from what has
                   been measured and the process exits non-zero.  The N = 1 line carries the one-GPU time of the
                   same 32768 rows (`strong_scaling_reference`).

The N = 1 line also carries `other_configs` (B, E's 2048-row shard, E in full, D through host pointers: each both ways with the
path taken and a parity spot check), `library_config` (the knobs the library ran under) and -- where this process sees several
GPUs -- `multi_device_one_process`: config D through ONE context over 1 / 2 / 4 / all visible devices (mcalf_create_multi).

`value` is timed through the device-pointer entry, parameters resident in HBM when the timed region starts and
logL left in HBM (the bench contract).  `value_host_api` is the same K steps through the host-pointer entry
`mcalf_loglike_batch` -- H2D of P and D2H of logL inside every step, SURVEY.md section 8(d)'s definition --
measured in the same run and printed next to it.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import threading
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
MIN_PASS_MS = 50.0             # a timed pass shorter than this is repeated and the median pass reported

WORKLOAD_LABEL = {
    "A": "BASELINE config A: CIV 1548/1550, the reference's multicomponent mock spectrum (tests/golden)",
    "B": "BASELINE config B: CIV 1548/1550 synthetic spectrum, ncomp 8, fixed 8 km/s LSF",
    "C": "BASELINE config C: CIV 1548/1550 synthetic spectrum, ncomp 8-11 + 4 fillers, LSF floated in [8, 9] km/s",
    "D": "BASELINE config D: config C's problem with 32768 live points job-wide",
    "E": "BASELINE config E: HI 1215 damped, ncomp 16, 20000 pixels",
}


def hip_synth(kw, p):
    """Synthesise a truth spectrum with the HIP path itself (no oracle in the product path)."""
    with mcalf_amd.als_fitter(None, **kw) as fit:
        return fit.reconstruct_spec(np.asarray(p, dtype=float))


def oracle_problem(kw):
    from oracle import numpy_oracle as oracle
    wl, flux, err = kw["spectrum"]
    return oracle.Problem(wl, flux, err, kw["linepars"], tuple(kw["ncomp"]), nfill=kw.get("nfill", 0),
                          specres=kw["specres"], Nrange=kw["Nrange"], brange=kw["brange"], zrange=kw["zrange"],
                          Nrangefill=kw.get("Nrangefill", [11.5, 16]), brangefill=kw.get("brangefill", [1, 30]),
                          fitrange=kw["fitrange"])


def cpu_baseline(kw, P, budget_s):
    """Time the numpy/scipy oracle (the closest runnable stand-in for the reference's numpy
    path) on a bounded sample of the same workload.  Checker/baseline only."""
    from oracle import numpy_oracle as oracle
    prob = oracle_problem(kw)
    oracle.lnlhood_worker(prob, P[0])          # warm
    vals, t0, done = [], time.perf_counter(), 0
    # cycle over the batch until the time budget is spent (budget 0: every row once)
    while time.perf_counter() - t0 < budget_s or done < (len(P) if budget_s <= 0 else 0):
        row = P[done % len(P)]
        v = oracle.lnlhood_worker(prob, row)
        if done < len(P):
            vals.append(v)
        done += 1
    dt = time.perf_counter() - t0
    return np.array(vals), dt, done


def usable_cpus():
    """Host cores this process may actually use: the scheduler affinity mask and the cgroup CPU quota (a GPU
    box gives a one-GPU job a share of its cores), not just os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for quota, period in (("/sys/fs/cgroup/cpu.max", None),
                          ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period is None:
                q, per = open(quota).read().split()
            else:
                q, per = open(quota).read().strip(), open(period).read().strip()
            if q not in ("max", "-1"):
                n = min(n, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    return n


def pick_workload(world, config=None, scaling=None):
    """Default workload of a run: one GPU -> BASELINE config C (the largest single-GPU configuration); several ->
    config D (32768 live points job-wide) as a strong-scaling job, the rows split over the ranks."""
    return config or ("C" if world == 1 else "D"), scaling or "strong"


def rows_per_rank(job_batch, world, scaling):
    """Rows each rank evaluates: the configuration's batch split over the ranks (strong) or repeated on every rank
    (weak)."""
    if scaling == "strong":
        if job_batch % world:
            raise SystemExit(f"--scaling strong: batch {job_batch} is not a multiple of {world} ranks")
        return job_batch // world
    return job_batch


def load_profile_json(name, config):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh).get(config)
    except (OSError, ValueError):
        return None


def library_source_hash(lib):
    """The sha256 prefix of the kernel sources the loaded library was built from (mcalf_version(): '... src <hash>')."""
    ver = lib.mcalf_version().decode()
    return ver.rsplit(" src ", 1)[1] if " src " in ver else "unstamped"


def stamped(entry, lib_hash):
    """A PMC-derived entry of profiles/*.json is only quoted when it was measured on the kernel that is running:
    (entry, None) when its `source_hash` equals the library's, else (None, why)."""
    if not entry:
        return None, "no PMC record for this configuration under profiles/"
    have = entry.get("source_hash")
    if have != lib_hash or lib_hash == "unstamped":
        return None, (f"profiles/ record was measured on kernel sources {have}, the loaded library is {lib_hash}: "
                      "stale counters are not quoted (re-run tools/profiles.sh)")
    return entry, None


def issue_model(iss, kern_ms):
    """Instruction-issue bound of the fused kernel from the PMC instruction counts (profiles/pmc.json `issue`).

    A wave issues at most one instruction at a time and the SIMD's arbiter one per class per cycle, so with W
    waves per SIMD a launch cannot finish before  max(per-class pipe time, serial issue time of a wave / overlap):
      t_pipe  = wave-instructions of the busiest pipe on a SIMD x its measured cycles per instruction
      t_issue = all wave-instructions of a SIMD x the measured issue interval of a single wave / W
    cycles per instruction come from tools/micro/issue_rate.hip on this pool (`cycles`); `frac` = bound / measured."""
    simds = iss["simds"]
    clk = iss["clock_mhz"] * 1e6
    cyc = iss["cycles"]
    per_simd = {k: iss["insts"][k] / simds for k in iss["insts"]}
    t_class = {k: per_simd[k] * cyc["pipe"][k] / clk * 1e3 for k in per_simd if k in cyc["pipe"]}
    # classes that share a pipe add up: both VALU classes issue on the SIMD's vector pipe, SALU and branches on the
    # scalar pipe; the LDS pipe is priced by its own busy counter (SQ_LDS_IDX_ACTIVE, per CU) when there is one
    t_pipe = {"valu": t_class.get("valu_f64", 0.0) + t_class.get("valu_other", 0.0),
              "scalar": t_class.get("salu", 0.0) + t_class.get("branch", 0.0),
              "lds": t_class.get("lds", 0.0)}
    if iss.get("lds_pipe_cycles_per_cu"):
        t_pipe["lds"] = iss["lds_pipe_cycles_per_cu"] / clk * 1e3
    t_issue = sum(per_simd[k] * cyc["wave"][k] for k in per_simd if k in cyc["wave"]) / iss["waves_per_simd"] / clk * 1e3
    bound = max(max(t_pipe.values()), t_issue)
    return {"bound": "instruction issue", "unit": "ms", "achieved": kern_ms, "peak": bound, "frac": bound / kern_ms,
            "pipe_ms": t_pipe, "class_ms": t_class, "busiest_pipe": max(t_pipe, key=t_pipe.get),
            "serial_issue_ms": t_issue, "waves_per_simd": iss["waves_per_simd"],
            "wave_instructions_per_launch": iss["insts"], "cycles_per_instruction": cyc,
            "note": "frac = the time the busiest pipe (or one wave's serial issue, shared by the SIMD's waves) needs for the "
                    "MEASURED instruction stream at the MEASURED issue rates, over the kernel's duration; 1 - frac is time no "
                    "pipe limit explains: dependency and LDS-latency stalls that four waves per SIMD do not cover"}


PATH_NAME = {_lib.MCALF_PATH_HOST_STREAM: "one streaming launch (MCALF_PATH_HOST_STREAM)",
             _lib.MCALF_PATH_HOST_ZEROCOPY: "zero-copy small call, completion read off the results (MCALF_PATH_HOST_ZEROCOPY)",
             _lib.MCALF_PATH_HOST_PIPELINED: "row-block pipeline (MCALF_PATH_HOST_PIPELINED)",
             _lib.MCALF_PATH_HOST_STAGED: "staged copies (MCALF_PATH_HOST_STAGED)"}


def median_pass_ms(fn, k, sync, min_passes=3):
    """Milliseconds per step of `fn`: passes of exactly k steps bracketed by `sync()` -- at least `min_passes` of them, more
    while a pass is too short to time (MIN_PASS_MS) -- the median pass.  (Never ONE pass: the first calls on a fresh
    page-locked buffer include a one-time ~7 ms stall inside hipMemcpyAsync -- profiles/r06_host_entry_outlier.txt -- which
    a single pass of 20 steps reports as +0.35 ms per step.)"""
    def one():
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        sync()
        return time.perf_counter() - t0
    first = one()
    npass = max(min_passes, 1 if first * 1e3 >= MIN_PASS_MS else min(15, 2 * int(math.ceil(MIN_PASS_MS / max(first * 1e3, 1e-3))) + 1))
    times = sorted([first] + [one() for _ in range(npass - 1)])
    return times[len(times) // 2] / k * 1e3


def config_leg(config, rows=None, steps=20, device=0, parity_rows=64):
    """One further BASELINE configuration on this GPU, both ways: the device-resident step (P in HBM, logL left in HBM: the
    bench contract's `value`) and SURVEY.md 8(d)'s step through the host-pointer entry (P from host memory and logL back
    inside every step; pageable numpy arrays, then page-locked ones), the path the library took, and a parity spot check
    of `parity_rows` rows spread over the batch against the numpy / scipy oracle (the checker; not timed)."""
    kw, batch, seed = workloads.config(config, hip_synth)
    rows = rows or batch
    P = np.ascontiguousarray(workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2 if config == "E" else 0)[:rows])
    dev = torch.device("cuda", device)
    with mcalf_amd.als_fitter(None, device=device, **kw) as fit:
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, rows), fit._ctx)
        dP = torch.from_numpy(P).to(dev)
        out = torch.empty(rows, dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream()
        st = C.c_void_p(stream.cuda_stream)
        launch, ctx, pP, pO = fit._lib.mcalf_loglike_batch_device, fit._ctx, dP.data_ptr(), out.data_ptr()

        def dev_step():
            rc = launch(ctx, pP, rows, pO, st)
            if rc:
                _lib.check(rc, ctx)
        sync = torch.cuda.synchronize
        for _ in range(3):
            dev_step()
        ms_dev = median_pass_ms(dev_step, steps, sync)
        logl_dev = out.cpu().numpy().copy()
        ll_dev = fit.last_launch()

        def dev_sync_step():                              # the floor of ANY synchronous step: resident inputs, nothing copied
            dev_step()
            stream.synchronize()
        ms_dev_sync = median_pass_ms(dev_sync_step, steps, sync)
        out_h = np.empty(rows)
        fit.loglike_batch(P, out=out_h)
        fit.loglike_batch(P, out=out_h)
        ms_host = median_pass_ms(lambda: fit.loglike_batch(P, out=out_h), steps, sync)
        llh = fit.last_launch()
        P_pin = torch.from_numpy(P).pin_memory().numpy()
        out_pin = torch.full((rows,), float("nan"), dtype=torch.float64).pin_memory().numpy()
        fit.loglike_batch(P_pin, out=out_pin)
        ms_pin = median_pass_ms(lambda: fit.loglike_batch(P_pin, out=out_pin), steps, sync)
        nc = P[:, fit.startind].astype(int)
        comp_pix = float(nc.sum()) * fit.obj_wl.size
        idx = np.unique(np.linspace(0, rows - 1, min(parity_rows, rows)).astype(int))
        from oracle import numpy_oracle as oracle
        prob = oracle_problem(kw)
        want = np.array([oracle.lnlhood_worker(prob, P[i]) for i in idx])
        return {"workload": WORKLOAD_LABEL[config], "rows": rows, "npix": int(fit.obj_wl.size), "ndim": fit.ndim,
                "tiles_per_sample": fit.info.ntiles, "steps": steps,
                "ms_per_step_device_resident": ms_dev, "value_device_resident": comp_pix / (ms_dev * 1e-3),
                "ms_per_step_host_api": ms_host, "value_host_api": comp_pix / (ms_host * 1e-3),
                "ms_per_step_host_api_pinned": ms_pin,
                "ms_per_step_device_entry_synchronised_every_step": ms_dev_sync,
                "host_over_device": ms_host / ms_dev, "host_over_device_pinned": ms_pin / ms_dev,
                "host_over_synchronous_floor": ms_host / ms_dev_sync,
                "path_host_api": PATH_NAME.get(llh.path, str(llh.path)), "row_blocks_host_api": llh.row_blocks,
                "device_launch": {"persistent": bool(ll_dev.persistent), "grid": ll_dev.grid, "items": ll_dev.items},
                "bit_equal_host_vs_device_entry": bool(np.array_equal(out_h, logl_dev) and np.array_equal(out_pin, logl_dev)),
                "parity": {"rows": int(idx.size), "max_abs_dlogL_vs_oracle": float(np.abs(logl_dev[idx] - want).max())}}


def multi_device_leg(devices, steps=10):
    """ONE process, several GPUs (mcalf_create_multi): BASELINE config D's 32768 rows through the host-pointer entry of a
    context over `devices` (contiguous row blocks, one per device, no collective) against the same call on the first device
    alone.  Runs in a child process of `bench.py` (see `run_multi_device_leg`), on nodes where this process sees several GPUs."""
    kw, batch, seed = workloads.config("D", hip_synth)
    P = np.ascontiguousarray(workloads.draw_P(kw, batch, np.random.default_rng(seed)))
    sync = torch.cuda.synchronize
    res = {"workload": WORKLOAD_LABEL["D"] + ", all rows in ONE process", "rows": batch, "devices": list(devices), "steps": steps,
           "entry": "mcalf_loglike_batch (pageable numpy arrays in, logL out inside every step)"}
    ref = None
    for n in sorted({1, 2, 4, len(devices)}):
        if n > len(devices):
            continue
        use = list(devices[:n])
        with mcalf_amd.als_fitter(None, device=use if n > 1 else use[0], **kw) as fit:
            out = np.empty(batch)
            for _ in range(3):
                fit.loglike_batch(P, out=out)
            ms = median_pass_ms(lambda: fit.loglike_batch(P, out=out), steps, sync)
            if ref is None:
                ref = (ms, out.copy())
            nc = P[:, fit.startind].astype(int)
            res["n%d" % n] = {"devices": use, "ms_per_step": ms, "value": float(nc.sum()) * fit.obj_wl.size / (ms * 1e-3),
                              "speedup_vs_one_device": ref[0] / ms, "devices_used": int(fit.last_launch().devices_used),
                              "bit_equal_to_one_device": bool(np.array_equal(out, ref[1]))}
    return res


def run_multi_device_leg(timeout_s=240.0):
    """The leg above in a CHILD process with a time limit (a fresh process: a hang or a fault on hardware this path has never
    met must not take the bench line with it).  None when this process sees one GPU (MCALF_BENCH_MULTI_DEVICES=0,0 forces a
    device list -- entries may repeat -- for a plumbing check on a one-GPU box)."""
    import subprocess
    forced = os.environ.get("MCALF_BENCH_MULTI_DEVICES")
    devices = [int(x) for x in forced.split(",")] if forced else list(range(torch.cuda.device_count()))
    if len(devices) < 2:
        return None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--multi-device-leg", ",".join(map(str, devices))],
                           capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "child exited with %d" % r.returncode, "stderr_tail": r.stderr[-400:]}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": "the multi-device leg did not finish within %.0f s" % timeout_s}
    except Exception as exc:                                 # noqa: BLE001 -- never lose the bench line to this leg
        return {"error": repr(exc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=["A", "B", "C", "D", "E"],
                    help="default: C on one GPU (largest single-GPU BASELINE configuration), D on several")
    ap.add_argument("--batch", type=int, default=0, help="override the job-wide batch of the configuration")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N>1: strong = the configuration's batch split over the ranks (default; BASELINE's "
                         "north-star asks for strong scaling), weak = the configuration's batch on every rank")
    ap.add_argument("--chunks", type=int, default=-1,
                    help="row blocks per batch inside the library (mcalf_set_chunks): -1 library default, 0 automatic, "
                         "1 = one fused launch per step (what the rocprofv3 summaries in profiles/ are taken with)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline time budget (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the plain-C/OpenMP CPU baseline; 0 = every host core (os.cpu_count()), -1 = skip")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real multi-GPU path) or gloo: a rehearsal of the N>1 control flow on a "
                         "one-GPU box (all ranks share cuda:0, logL shards gathered through host memory)")
    ap.add_argument("--gather", default="all", choices=["all", "torch", "inlib", "inlib_overlap"],
                    help="N>1: which gather of the logL shards is timed.  all (default) = the three of them, one after the "
                         "other in the one process group, best one reported as `value`: torch = torch.distributed.gather "
                         "(RCCL through torch, two buffers in flight); inlib = the library's own communicator "
                         "(mcalf_comm_init + mcalf_loglike_gatherv_device: kernels and one grouped ncclSend/ncclRecv exchange "
                         "on the launch stream); inlib_overlap = the same with the exchange on the context's side stream")
    ap.add_argument("--leg-timeout", type=float, default=30.0,
                    help="N>1: seconds a library-gather leg may take before the watchdog prints what has been measured and exits")
    ap.add_argument("--init-timeout", type=float, default=120.0,
                    help="N>1: seconds the library's communicator may take to come up (ncclCommInitRank over all ranks; its "
                         "own allowance, so that a slow first initialisation is not mistaken for a hung exchange)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-pointer (PCIe-inclusive) passes")
    ap.add_argument("--no-strong-ref", action="store_true", help="N=1: skip the config-D-on-one-GPU reference")
    ap.add_argument("--no-model-leg", action="store_true", help="N=1: skip the model-output (reconstruct_spec) passes")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N=1: skip the legs over the other BASELINE configurations (B, E's 2048-row shard, E in full)")
    ap.add_argument("--multi-device-leg", default=None,
                    help="internal: run ONLY the one-process multi-device leg over these device entries (comma-separated) and "
                         "print it as one JSON line (bench.py starts itself this way as a child process)")
    ap.add_argument("--no-multi-device", action="store_true",
                    help="N=1: skip the one-process multi-device leg (it only runs where this process sees several GPUs)")
    ap.add_argument("--only-other-configs", default=None,
                    help="diagnostic: time ONLY these legs (comma-separated: B, C, E, E2048, D) and print them as one JSON line")
    ap.add_argument("--inflight", type=int, default=1,
                    help="diagnostic: independent batches kept in flight (contexts + streams); 1 = the headline")
    args = ap.parse_args()

    if args.multi_device_leg:
        print(json.dumps(multi_device_leg([int(x) for x in args.multi_device_leg.split(",")])))
        return
    if args.only_other_configs:
        legs = {}
        for name in args.only_other_configs.split(","):
            cfg, rows = (name[0], int(name[1:])) if len(name) > 1 else (name, None)
            legs[name] = config_leg(cfg, rows, steps=args.steps)
        lib = _lib.load()
        print(json.dumps({"other_configs": legs, "library": lib.mcalf_version().decode()}))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    config, scaling = pick_workload(world, args.config, args.scaling)
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
    rccl_ranks = dist.get_world_size() if use_dist else 1

    kw, job_batch, seed = workloads.config(config, hip_synth)
    if args.batch:
        job_batch = args.batch
    batch = rows_per_rank(job_batch, world, scaling)     # per-rank rows; strong: the job-wide batch stays the config's
    damped = 2 if config == "E" else 0
    # every rank draws the job-wide matrix from one seed and keeps its contiguous row block
    P_all = workloads.draw_P(kw, batch * world, np.random.default_rng(seed), damped=damped)
    P_host = np.ascontiguousarray(P_all[rank * batch:(rank + 1) * batch])
    del P_all
    if os.environ.get("MCALF_BENCH_SORT") in ("asc", "desc"):
        # diagnostic only (never set by the driver): rows ordered by their ncomp slot, to measure what an ordered
        # hand-out of the work items would buy
        start = int(len(np.atleast_1d(kw["specres"])) > 1) + int(len(np.atleast_1d(kw.get("contval", [1.0]))) > 1)
        key = P_host[:, start] * (1.0 if os.environ["MCALF_BENCH_SORT"] == "asc" else -1.0)
        P_host = np.ascontiguousarray(P_host[np.argsort(key, kind="stable")])
    fit = mcalf_amd.als_fitter(None, device=local_rank, **kw)
    if args.chunks >= 0:
        fit.set_chunks(args.chunks)
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
    if args.inflight > 1 and world == 1:
        for _ in range(args.inflight - 1):
            f2 = mcalf_amd.als_fitter(None, device=local_rank, **kw)
            _lib.check(f2._lib.mcalf_reserve(f2._ctx, batch), f2._ctx)
            s2 = torch.cuda.Stream()
            extra.append((f2, s2, torch.empty(batch, dtype=torch.float64, device=dev)))
    turn = [0]
    launch = fit._lib.mcalf_loglike_batch_device
    ctx, pP = fit._ctx, dP.data_ptr()
    last_out = [dlogL]
    inlib_box = [None]                                   # the library-gather object of the leg that is running
    gathered = [None]
    red_dev = "cpu" if rehearsal else dev

    def step():
        if extra:
            k = turn[0] % (len(extra) + 1)
            turn[0] += 1
            if k > 0:
                f2, s2, o2 = extra[k - 1]
                rc = launch(f2._ctx, pP, batch, o2.data_ptr(), C.c_void_p(s2.cuda_stream))
                if rc:
                    _lib.check(rc, f2._ctx)
                return
        inlib = inlib_box[0]
        if inlib is not None:
            inlib.step(dP, stream)
            last_out[0] = inlib.local
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

    def fence():
        if use_dist:
            if inlib_box[0] is not None:
                gathered[0] = inlib_box[0].finish(stream)    # joins the exchanges of the library's side stream
            else:
                gathered[0] = plan.finish()              # every outstanding gather has landed on rank 0
            dist.barrier()
        torch.cuda.synchronize()

    def timed_pass(fn, k):
        """Exactly k steps bracketed by barrier + synchronize on both sides; seconds."""
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        fence()
        return time.perf_counter() - t0

    def measure(fn, k, red_dev, min_passes=1):
        """Median over repeated timed passes (each pass = exactly k steps); every pass time is the MAX over ranks."""
        first = timed_pass(fn, k)
        ref = first
        if use_dist:                                     # every rank must run the SAME number of passes (collectives inside)
            t1 = torch.tensor([first], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            ref = float(t1.item())
        npass = max(min_passes, 1 if ref * 1e3 >= MIN_PASS_MS else min(15, 2 * int(math.ceil(MIN_PASS_MS / max(ref * 1e3, 1e-3))) + 1))
        times = [first] + [timed_pass(fn, k) for _ in range(npass - 1)]
        t = torch.tensor(times, dtype=torch.float64, device=red_dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = sorted(float(v) for v in t.tolist())
        return times[len(times) // 2], times

    def kernel_pass():
        """Dominant kernel alone: a further pass of K launches, each bracketed by HIP events on the launch stream
        inside the library (one fused launch per step: the library issues the batch as ONE row block while it is
        being profiled).  Kept out of the timed region because the brackets themselves cost ~5 us per step."""
        km, nl = C.c_double(0.0), C.c_int32(0)
        _lib.check(fit._lib.mcalf_profile_begin(fit._ctx, args.steps), fit._ctx)
        for _ in range(args.steps):
            step()
        fence()
        _lib.check(fit._lib.mcalf_profile_end(fit._ctx, C.byref(km), C.byref(nl)), fit._ctx)
        return km.value if nl.value else None

    def split_pass(kernels_ms_ref):
        """Where rank 0's step goes, for the gather that is currently selected (N > 1): a further pass of a few steps, every
        step started behind a barrier and run SYNCHRONOUSLY -- nothing overlaps, so the three segments add up to a step of
        this pass (an upper bound of the pipelined `ms_per_step`):
          kernels   wall time from the launch call to the launch stream being idle (set-up + fused kernel, launch latency);
          exchange  from there until the gathered blocks have landed on rank 0 (the slowest peer's kernels are inside);
          join      what the caller then still waits for before it may read the vector (assembling / joining the side stream).
        The library's entry enqueues kernels and exchange in ONE call: its `kernels` is the torch leg's figure for the same
        kernels on the same rank (`kernels_ms_ref`), its `exchange` the rest of that call's segment."""
        if not use_dist:
            return None
        k = max(2, min(args.steps, 8))
        seg = np.zeros(3)
        inlib = inlib_box[0]
        for _ in range(k):
            fence()
            t0 = time.perf_counter()
            if inlib is None:
                out = dlogL if dlogL is not None else plan.local
                rc = launch(ctx, pP, batch, out.data_ptr(), st)
                if rc:
                    _lib.check(rc, ctx)
                stream.synchronize()
                t1 = time.perf_counter()
                if rehearsal:
                    plan.local.copy_(dlogL)
                plan.gather_async()
                w = plan._work[plan._last]
                if w is not None:
                    w.wait()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                plan.finish()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
            else:
                inlib.step(dP, stream)
                stream.synchronize()                      # (without overlap: kernels AND exchange; with it: the kernels only)
                t2 = time.perf_counter()
                t1 = min(t2, t0 + kernels_ms_ref * 1e-3) if kernels_ms_ref is not None else t2
                inlib.finish(stream)
                t3 = time.perf_counter()
            seg += (t1 - t0, t2 - t1, t3 - t2)
        fence()
        seg = seg / k * 1e3
        how = ("wall clock on rank 0, every step behind a barrier and synchronous (no overlap between steps or with the exchange); " +
               ("kernels: launch call -> launch stream idle; exchange: -> blocks landed on rank 0; join: -> gathered vector assembled"
                if inlib is None else
                "kernels: the torch leg's figure for the same kernels; exchange: rest of the time until the launch stream is idle "
                "(with overlap the exchange runs on the side stream and is waited for in join); join: mcalf_comm_join + synchronise"))
        return {"steps": k, "kernels_ms": float(seg[0]), "exchange_ms": float(seg[1]), "join_ms": float(seg[2]),
                "step_ms_synchronous": float(seg.sum()), "how": how}

    def run_leg(kernels_ms_ref=None):
        """Warm-up, timed passes, kernel pass, rank-0 split and gather check of the gather that is currently selected."""
        for _ in range(args.warmup):
            step()
        el, passes = measure(step, args.steps, red_dev)
        own = last_out[0].cpu().numpy().copy()
        check = None
        if use_dist and rank == 0 and gathered[0] is not None:
            # the vector rank 0 holds after the last gather: its own block must be what it computed, and every
            # other block a finite logL of that rank's rows
            g = gathered[0].cpu().numpy()
            check = {"rows": int(g.size), "own_block_equal": bool(np.array_equal(g[:batch], own)),
                     "all_finite": bool(np.isfinite(g).all())}
        km = kernel_pass()
        km = km if km is not None else el / args.steps * 1e3
        kmin = kmax = km
        if use_dist:                                     # load imbalance: the ranks' own kernel times
            kt = torch.tensor([km, -km], dtype=torch.float64, device=red_dev)
            dist.all_reduce(kt, op=dist.ReduceOp.MAX)
            kmax, kmin = float(kt[0].item()), -float(kt[1].item())
        comm_ranks = None
        if inlib_box[0] is not None:                     # the LIBRARY's communicator says how many ranks it spans, not torch
            nr, rk = C.c_int32(0), C.c_int32(-1)
            _lib.check(fit._lib.mcalf_comm_info(fit._ctx, C.byref(nr), C.byref(rk)), fit._ctx)
            comm_ranks = int(nr.value)
        return {"elapsed": el, "pass_times": passes, "logL": own, "gather_check": check, "kernel_ms": km,
                "kernel_ms_min_over_ranks": kmin, "kernel_ms_max_over_ranks": kmax, "comm_ranks": comm_ranks,
                "rank0_split": split_pass(kernels_ms_ref)}

    GATHER_WHAT = {
        "torch": "torch.distributed.gather, two buffers in flight",
        "inlib": "library (mcalf_loglike_gatherv_device: kernels and one grouped ncclSend/ncclRecv exchange on the launch stream)",
        "inlib_overlap": "library (mcalf_loglike_gatherv_device: kernels on the launch stream, grouped ncclSend/ncclRecv on the "
                         "context's exchange stream, two buffer pairs in flight)",
    }
    legs, leg_notes = {}, {}
    names = ["torch"]
    if use_dist:
        names = ["torch", "inlib", "inlib_overlap"] if args.gather == "all" else [args.gather]
    # the library's exchange needs one RCCL rank per device; a gloo rehearsal on ONE GPU can only drive it over the
    # tests' stand-in transport (MCALF_RCCL_LIB)
    can_inlib = use_dist and (not rehearsal or bool(os.environ.get("MCALF_RCCL_LIB")))

    def emit_partial_and_exit(name, what=None):
        """Watchdog of a library-gather leg: print what has been measured, exit non-zero (a fresh exit, never a re-exec)."""
        leg_notes[name] = "timed_out"
        if rank == 0:
            done = {k: {"ms_per_step": v["elapsed"] / args.steps * 1e3, "gather_check": v["gather_check"]} for k, v in legs.items()}
            best = min(legs, key=lambda k: legs[k]["elapsed"]) if legs else None
            line = {"metric": "component-pixel Voigt evals/s", "unit": "evals/s", "n_gpus": world, "steps": args.steps,
                    "warmup": args.warmup, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
                    "data": "synthetic", "value": (comp_pix * world * args.steps / legs[best]["elapsed"]) if best else None,
                    "ms_per_step": (legs[best]["elapsed"] / args.steps * 1e3) if best else None,
                    "gather_reported": best, "gathers": {**done, name: "timed_out"},
                    "config": {"workload": WORKLOAD_LABEL[config], "batch_per_gpu": batch, "global_batch": batch * world},
                    "error": what or f"gather leg {name!r} did not finish within {args.leg_timeout:.0f} s"}
            print(json.dumps(line), flush=True)
        os._exit(3)

    for name in names:
        if name != "torch" and not can_inlib:
            leg_notes[name] = "skipped: a gloo rehearsal on one GPU has no RCCL transport for the library's exchange"
            continue
        watchdog = None
        if name != "torch":
            init_dog = threading.Timer(args.init_timeout, emit_partial_and_exit, args=(
                name, f"the communicator of gather leg {name!r} did not come up within {args.init_timeout:.0f} s"))
            init_dog.daemon = True
            init_dog.start()
            inlib_box[0] = mdist.InLibGather(fit, batch * world, dev, depth=2 if name == "inlib_overlap" else 1)
            init_dog.cancel()
            watchdog = threading.Timer(args.leg_timeout, emit_partial_and_exit, args=(name,))
            watchdog.daemon = True
            watchdog.start()
        ref = legs["torch"]["rank0_split"]["kernels_ms"] if ("torch" in legs and legs["torch"]["rank0_split"]) else None
        legs[name] = run_leg(ref)
        if watchdog is not None:
            watchdog.cancel()
    reported = min(legs, key=lambda k: legs[k]["elapsed"])
    inlib_box[0] = None                                  # (what follows fences through the torch plan again)
    elapsed, pass_times = legs[reported]["elapsed"], legs[reported]["pass_times"]
    logL_dev, gather_check, kern_ms = legs[reported]["logL"], legs[reported]["gather_check"], legs[reported]["kernel_ms"]
    tot = torch.tensor([comp_pix, line_pix], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    comp_pix_job, line_pix_job = (float(v) for v in tot.tolist())

    # PCIe-inclusive passes through the host-pointer entry (N = 1): pageable numpy arrays as a sampler holds
    # them, then page-locked ones
    host_api = None
    if world == 1 and not args.no_host_api:
        out_host = np.empty(batch)
        fit.loglike_batch(P_host, out=out_host)
        t_host, _ = measure(lambda: fit.loglike_batch(P_host, out=out_host), args.steps, red_dev, min_passes=3)
        same = bool(np.array_equal(out_host, logL_dev))
        llh = fit.last_launch()
        P_pin = torch.from_numpy(P_host).pin_memory().numpy()
        out_pin = torch.full((batch,), float("nan"), dtype=torch.float64).pin_memory().numpy()
        fit.loglike_batch(P_pin, out=out_pin)
        t_pin, _ = measure(lambda: fit.loglike_batch(P_pin, out=out_pin), args.steps, red_dev, min_passes=3)
        # the same entry with a synchronisation after every step: what a synchronous step costs when nothing moves
        def dev_sync_step():
            rc = launch(ctx, pP, batch, last_out[0].data_ptr(), st)
            if rc:
                _lib.check(rc, ctx)
            stream.synchronize()
        t_dsync, _ = measure(dev_sync_step, args.steps, red_dev)
        dev_ms = elapsed / args.steps * 1e3
        host_api = {"ms_per_step": t_host / args.steps * 1e3, "value": comp_pix * args.steps / t_host,
                    "ms_per_step_pinned": t_pin / args.steps * 1e3, "value_pinned": comp_pix * args.steps / t_pin,
                    "host_over_device": t_host / args.steps * 1e3 / dev_ms,
                    "host_over_device_pinned": t_pin / args.steps * 1e3 / dev_ms,
                    "ms_per_step_device_entry_synchronised_every_step": t_dsync / args.steps * 1e3,
                    "bit_equal_to_device_entry": same,
                    "bit_equal_to_device_entry_pinned": bool(np.array_equal(out_pin, logL_dev)),
                    "path": PATH_NAME.get(llh.path, str(llh.path)),
                    "stream_setup_workgroups": llh.stream_setup_wgs, "completion_polled": bool(llh.stream_polled),
                    "what": "mcalf_loglike_batch: P [batch][ndim] f64 from host memory and logL [batch] f64 back to host memory "
                            "inside every step, one synchronous call per step (SURVEY.md 8(d)); pageable numpy arrays / "
                            "page-locked arrays.  host_over_device = this step over the device-resident step (`ms_per_step`)"}
        # latency of ONE theta through the reference's callable (lnlhood_dy), the one-launch variant of small calls
        th = P_host[0].copy()
        for _ in range(50):
            fit.lnlhood_dy(th)
        t1 = time.perf_counter()
        for _ in range(500):
            fit.lnlhood_dy(th)
        host_api["single_call_us"] = (time.perf_counter() - t1) / 500 * 1e6
        # the same through the resident evaluator (mcalf_set_resident: no launch per call; opt-in)
        if fit.info.ntiles == 1 and fit.ndim <= 64:
            fit.set_resident(500)
            for _ in range(50):
                fit.lnlhood_dy(th)
            t1 = time.perf_counter()
            for _ in range(500):
                fit.lnlhood_dy(th)
            host_api["single_call_us_resident"] = (time.perf_counter() - t1) / 500 * 1e6
            fit.set_resident(0)

    # N = 1 leg of the strong-scaling job: config D's 32768 rows on this one GPU
    strong_ref = None
    if world == 1 and config == "C" and not args.no_strong_ref and not args.batch:
        kwD, batchD, seedD = workloads.config("D", hip_synth)
        PD = workloads.draw_P(kwD, batchD, np.random.default_rng(seedD))
        dPD = torch.from_numpy(PD).to(dev)
        outD = torch.empty(batchD, dtype=torch.float64, device=dev)
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, batchD), fit._ctx)

        def stepD():
            rc = launch(ctx, dPD.data_ptr(), batchD, outD.data_ptr(), st)
            if rc:
                _lib.check(rc, ctx)
        for _ in range(2):
            stepD()
        kD = max(3, args.steps // 8)
        tD, _ = measure(stepD, kD, red_dev)
        ncD = PD[:, fit.startind].astype(int)
        strong_ref = {"workload": WORKLOAD_LABEL["D"] + ", all rows on ONE GPU", "global_batch": batchD, "steps": kD,
                      "ms_per_step": tD / kD * 1e3, "value": float(ncD.sum()) * npix * kD / tD}
        if not args.no_host_api:
            # the same rows through the host-pointer entry (pageable arrays): the fixed cost of a synchronous call and of
            # the streaming launch's start, spread over eight times the rows
            outDh = np.empty(batchD)
            fit.loglike_batch(PD, out=outDh)
            tDh, _ = measure(lambda: fit.loglike_batch(PD, out=outDh), kD, red_dev, min_passes=3)
            strong_ref["host_api"] = {"ms_per_step": tDh / kD * 1e3, "host_over_device": tDh / tD,
                                      "path": PATH_NAME.get(fit.last_launch().path, str(fit.last_launch().path)),
                                      "bit_equal_to_device_entry": bool(np.array_equal(outDh, outD.cpu().numpy()))}
        del dPD, outD

    # Model-output entry (reconstruct_spec for the whole batch, hires_fitter.py:409-449; consumer cli.py:414-418):
    # the only mode in which HBM bytes matter -- 8 * npix B per live point are written
    model_leg = None
    if world == 1 and not args.no_model_leg and not extra:
        dflux = torch.empty((batch, npix), dtype=torch.float64, device=dev)
        model_fn = fit._lib.mcalf_model_batch_device

        def stepM():
            rc = model_fn(ctx, pP, batch, 0, dflux.data_ptr(), st)
            if rc:
                _lib.check(rc, ctx)
        for _ in range(3):
            stepM()
        kM = max(5, args.steps // 4)
        tM, _ = measure(stepM, kM, red_dev)
        kmM, nlM = C.c_double(0.0), C.c_int32(0)
        _lib.check(fit._lib.mcalf_profile_begin(fit._ctx, kM), fit._ctx)
        for _ in range(kM):
            stepM()
        fence()
        _lib.check(fit._lib.mcalf_profile_end(fit._ctx, C.byref(kmM), C.byref(nlM)), fit._ctx)
        # one row checked against the log-likelihood entry: logL recomputed on the host from the model spectrum
        row = dflux[0].cpu().numpy()
        ispec2 = 1.0 / fit.obj_noise ** 2
        ll0 = -0.5 * np.nansum(ispec2 * (fit.obj - row) ** 2 - np.log(ispec2) + np.log(2.0 * np.pi))
        bytes_alg = (8 * ndim + 8 * npix) * batch + 8 * npix          # parameters in, model out, frequencies once
        model_leg = {"entry": "mcalf_model_batch_device (targonly = 0), model spectra left in HBM", "steps": kM,
                     "ms_per_step": tM / kM * 1e3, "kernel_ms": kmM.value if nlM.value else None,
                     "value": comp_pix * kM / tM, "unit": "evals/s",
                     "algorithmic_bytes_per_launch": bytes_alg,
                     "hbm_gbs": bytes_alg / ((kmM.value if nlM.value else tM / kM * 1e3) * 1e-3) / 1e9,
                     "hbm_frac": bytes_alg / ((kmM.value if nlM.value else tM / kM * 1e3) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernel_ratio_to_logL_mode": (kmM.value / kern_ms) if nlM.value else None,
                     "logL_from_model_row0_minus_logL_entry": float(ll0 - logL_dev[0])}
        del dflux

    # The other single-GPU BASELINE configurations, each both ways (device-resident / host-pointer step), on the driver's
    # clock: B, E's per-GPU shard at N = 8 (2048 rows) and E in full (16384 rows).  (D through host pointers: the
    # strong-scaling reference above carries it.)
    other = None
    if world == 1 and config == "C" and not args.no_other_configs and not args.batch and not extra:
        other = {"B": config_leg("B", None, steps=max(20, args.steps), device=local_rank),
                 "E_shard_2048": config_leg("E", 2048, steps=max(10, args.steps // 2), device=local_rank),
                 "E_full_16384": config_leg("E", None, steps=max(5, args.steps // 5), device=local_rank)}

    out = None
    if rank == 0:
        n_half = fit.info.n_cap
        alg_bytes = (8 * ndim + 8) * batch + 24 * npix
        alg_flops = 40.0 * line_pix + (31 + 4 * n_half) * npix * batch
        ach_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        std_batch = not args.batch and (world == 1 or scaling == "strong")
        lib_hash = library_source_hash(fit._lib)
        tr, tr_note = stamped(load_profile_json("traffic.json", config) if std_batch and world == 1 else None, lib_hash)
        pmc, pmc_note = stamped(load_profile_json("pmc.json", config) if std_batch and world == 1 else None, lib_hash)
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "component-pixel Voigt evals/s (sum_s ncomp_s * npix / t), logL batch on MI355X: `value` with the inputs "
                      "resident in HBM when the timed region starts, `value_host_api` through the host-pointer entry with P in "
                      "and logL out inside every step",
            "value_definition": "the bench contract of this build: `value` is whole-job throughput with inputs already resident in "
                                "HBM when the timed region starts; a PCIe-inclusive rate is reported beside it and is never "
                                "`value`.  SURVEY.md 8(d) defines the step with H2D of P and D2H of logL inside: that figure is "
                                "`value_host_api` / `ms_per_step_host_api` (pageable numpy arrays, one synchronous call per "
                                "step), timed in the same run; `host_api.host_over_device` is their ratio",
            "value": comp_pix_job * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "entry": "device pointers (P resident in HBM, logL left in HBM)",
            "timing": {"passes": len(pass_times), "steps_per_pass": args.steps, "reported": "median pass",
                       "pass_ms": [round(t * 1e3, 4) for t in pass_times]},
            "batches_in_flight": max(1, args.inflight if world == 1 else 1),
            "config": {"workload": WORKLOAD_LABEL[config], "batch_per_gpu": batch, "global_batch": batch * world,
                       "npix": npix, "ncomp": list(kw["ncomp"]), "nlines": nlines, "nfill": fit.nfill, "ndim": ndim,
                       "specres": list(kw["specres"]), "lsf_taps": 2 * n_half + 1, "tiles_per_sample": fit.info.ntiles,
                       "row_blocks_per_batch": fit.chunks_for(batch),
                       "parallelism": (f"dp{world}: rows sharded in contiguous blocks, "
                                       f"{'gloo REHEARSAL on one GPU' if rehearsal else 'RCCL'} gather of logL to rank 0 every step"
                                       if world > 1 else "single GPU")},
            "logL_per_s": batch * world * args.steps / elapsed,
            "line_pixel_evals_per_s": line_pix_job * args.steps / elapsed,
            "kernel_ms": kern_ms,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": tr["traffic_bytes_per_launch"] if tr else None,
                         "traffic_note": tr_note,
                         "kernel": "mcalf_fused_kernel (one launch over the whole batch)",
                         "kernel_source_hash": lib_hash,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the fused path is FP64-VALU bound, not HBM bound (arithmetic intensity > 1e3 FLOP/B, "
                                 "SURVEY.md 8d): the HBM fraction is ~1e-3 by construction; see roofline_valu"},
            "roofline_valu": {"bound": "fp64_valu", "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "throughput_equivalent": alg_flops / (kern_ms * 1e-3) / 1e12,
                              "throughput_equivalent_frac": alg_flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                              "algorithmic_flops_per_launch": alg_flops,
                              "note": "throughput_equivalent prices SURVEY 8(d)'s nominal 40 FLOP per line-pixel; the "
                                      "far-wing interpolation legitimately skips most of those evaluations, so it is "
                                      "NOT a bound (it exceeds 1 at config E) and not pipe utilisation: `achieved` / "
                                      "`frac` are the FP64 operations the pipes executed (PMC instruction mix), "
                                      "valu_busy the share of cycles the vector pipe was issuing",
                              "pmc_note": pmc_note},
        }
        if pmc:
            out["roofline_valu"].update({k: pmc[k] for k in ("valu_busy", "executed_flops_per_launch", "executed_flops_note",
                                                             "valu_active_lane_fraction", "valu_active_lane_fraction_note", "executed_flops_lane_weighted_estimate",
                                                             "source") if k in pmc})
            if "executed_flops_per_launch" in pmc:
                ex = pmc["executed_flops_per_launch"] / (kern_ms * 1e-3) / 1e12
                out["roofline_valu"].update({"achieved": ex, "frac": ex / FP64_VALU_PEAK_TFLOPS})
            if "issue" in pmc:
                out["roofline_issue"] = issue_model(pmc["issue"], kern_ms)
        # (the same two figures under explicit names, whichever convention a reader expects for `value`)
        out["value_device_resident"], out["ms_per_step_device_resident"] = out["value"], ms_per_step
        if host_api:
            out["value_host_api"] = host_api["value"]
            out["ms_per_step_host_api"] = host_api["ms_per_step"]
            out["host_api"] = host_api
        if strong_ref:
            # what 8 GPUs can reach at best when each gets this batch and the job is config D: the one-GPU time of
            # the 32768 rows over 8 x this step (no collective yet)
            strong_ref["ceiling_8gpu_speedup"] = strong_ref["ms_per_step"] / ms_per_step
            strong_ref["ceiling_8gpu_efficiency"] = strong_ref["ms_per_step"] / (8.0 * ms_per_step)
            out["strong_scaling_reference"] = strong_ref
        if model_leg:
            out["model_output"] = model_leg
        if other:
            if strong_ref and "host_api" in strong_ref:
                other["D_host"] = {"workload": strong_ref["workload"], "rows": strong_ref["global_batch"],
                                   "ms_per_step_device_resident": strong_ref["ms_per_step"],
                                   "ms_per_step_host_api": strong_ref["host_api"]["ms_per_step"],
                                   "host_over_device": strong_ref["host_api"]["host_over_device"],
                                   "path_host_api": strong_ref["host_api"].get("path"),
                                   "bit_equal_host_vs_device_entry": strong_ref["host_api"]["bit_equal_to_device_entry"]}
            out["other_configs"] = other
        out["library_config"] = fit.get_config()
        if world == 1 and config == "C" and not args.no_multi_device and not args.batch:
            md = run_multi_device_leg()
            # (None on a one-GPU lease; on a node whose GPUs this process all sees: ONE process driving them)
            out["multi_device_one_process"] = md if md is not None else "skipped: this process sees one GPU"
        ll = fit.last_launch()
        out["launch"] = {"persistent": bool(ll.persistent), "grid": ll.grid, "items": ll.items,
                         "lines_per_sync": ll.lines_per_sync, "ordered_handout": bool(ll.ordered)}
        if use_dist:
            out["rccl_ranks"] = rccl_ranks
            out["gather_reported"] = reported
            out["gather"] = GATHER_WHAT[reported] + ("; ONE rank: the library's exchange is a device-to-device copy, no RCCL call "
                                                      "is made" if (rccl_ranks == 1 and reported != "torch") else "")
            # every gather that was timed in this process group, in the order they ran; `value` is the best one
            out["gathers"] = {}
            for name in names:
                if name in legs:
                    lg = legs[name]
                    out["gathers"][name] = {
                        "what": GATHER_WHAT[name], "ms_per_step": lg["elapsed"] / args.steps * 1e3,
                        "value": comp_pix_job * args.steps / lg["elapsed"], "gather_check": lg["gather_check"],
                        "kernel_ms_min_over_ranks": lg["kernel_ms_min_over_ranks"],
                        "kernel_ms_max_over_ranks": lg["kernel_ms_max_over_ranks"],
                        "rank0_split": lg["rank0_split"],
                        # who says how many ranks took part: torch's process group for its gather, the library's own
                        # communicator (mcalf_comm_info) for the library's legs
                        "rccl_ranks": lg["comm_ranks"] if lg["comm_ranks"] is not None else rccl_ranks,
                        "rccl_ranks_source": "mcalf_comm_info" if lg["comm_ranks"] is not None else "torch.distributed.get_world_size",
                        "passes": len(lg["pass_times"])}
                else:
                    out["gathers"][name] = leg_notes.get(name, "not run")
            if rehearsal:
                out["gathers_note"] = ("gloo REHEARSAL on one GPU: every rank shares cuda:0; the library legs run over the tests' "
                                       "stand-in transport when MCALF_RCCL_LIB names it, else they are skipped")
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
            if args.cpu_threads >= 0:
                # second CPU figure: the plain-C/OpenMP restatement (oracle/c) on every host core this process may
                # use; when that is more than 32 threads a 16-thread run is timed as well (a box may expose 256
                # logical CPUs to a job that is scheduled on a fraction of them) and the faster one is reported
                from oracle import c_oracle
                share = usable_cpus()
                counts = [args.cpu_threads] if args.cpu_threads > 0 else ([share, 16] if share > 32 else [share])
                runs = []
                for nthr in counts:
                    co = c_oracle.COracle(oracle_problem(kw), threads=nthr)
                    rows = P_host[: min(batch, 16 * nthr)]
                    co.loglike_batch(rows[:nthr])
                    tc, reps = time.perf_counter(), 0
                    while time.perf_counter() - tc < 4.0:
                        cvals = co.loglike_batch(rows)
                        reps += 1
                    dtc = time.perf_counter() - tc
                    runs.append({
                        "value": reps * float(nc[: len(rows)].sum()) * npix / dtc, "unit": "evals/s", "cores": nthr,
                        "kind": "port",
                        "sample": f"{reps} x {len(rows)} rows, oracle/c/mcalf_oracle.c (gcc -O2 -fopenmp, {nthr} threads), {dtc:.1f} s",
                        "max_abs_dlogL_vs_gpu": float(np.abs(cvals - logL_dev[: len(rows)]).max())})
                best = max(runs, key=lambda r: r["value"])
                best["host_cpus"], best["usable_cpus"] = os.cpu_count(), share
                if len(runs) > 1:
                    best["other_thread_counts"] = [{"cores": r["cores"], "value": r["value"]} for r in runs if r is not best]
                out["cpu_baseline_c_openmp"] = best
    if out is not None and world > 1 and args.cpu_seconds > 0:
        # N>1: no CPU timing, only a parity spot check of rank 0's first rows against the oracle
        vals, _, _ = cpu_baseline(kw, P_host[:8], 0.0)
        out["parity"] = {"max_abs_dlogL_vs_oracle": float(np.abs(vals - logL_dev[:len(vals)]).max()), "rows": len(vals)}
    if out is not None and gather_check is not None:
        out["gather_check"] = gather_check
    fit.close()
    for f2, _, _ in extra:
        f2.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
