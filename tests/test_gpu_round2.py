"""GPU: round-2 rows -- the file-path constructor with several fit ranges, BASELINE config D's
first shard, and the jaxns-facing closure of row J."""
import os

import numpy as np
import pytest

import mcalf_amd
from mcalf_amd import workloads
from cases import oracle_synth, problem_from_kwargs
from oracle import c_oracle
from oracle import numpy_oracle as o

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGL_ATOL = 1e-4          # BASELINE.json north_star: logL within 1e-4 absolute
FLUX_RTOL = 1e-6          # and flux within 1e-6 relative


def test_constructed_from_a_file_path_with_two_fit_ranges_and_computed_velstep():
    """als_fitter(specfile, fitrange=[two windows], ...) exactly as cli.py:73-76 builds it: the ASCII table is
    read from disk, the two windows are masked (hires_fitter.py:75-82), `velstep` is the clipped median of the
    per-pixel steps INCLUDING the jump across the gap (:84-87), and the periodic LSF wraps across the masked
    spectrum (:463).  The oracle is built from the same arrays; its velstep is its own restatement."""
    path = os.path.join(GOLD, "civ_mock_spec_multicomp.txt")
    ranges = [[6184.0, 6197.5], [6199.0, 6212.0]]
    args = dict(nfill=2, specres=[7.5, 9.0], contval=[0.95, 1.05], Nrange=[12.0, 14.5], brange=[8.0, 35.0],
                zrange=[2.995, 3.005])
    d = np.loadtxt(path)
    keep = ((d[:, 0] > 6184.0) & (d[:, 0] < 6197.5)) | ((d[:, 0] > 6199.0) & (d[:, 0] < 6212.0))      # :75-82
    wl, fl, er = d[keep, 0], d[keep, 1], d[keep, 2]
    steps = (wl[1:] - wl[:-1]) / wl[1:] * 2.9979245e5                                                 # :84
    assert (steps > 10).sum() == 1                               # the jump across the masked gap ...
    velstep = float(np.median(steps[steps < 10]))                # ... is what 3-sigma clipping removes (:85-87)
    prob = o.Problem(wl, fl, er, o.CIV_LINES, (2, 5), fitrange=ranges, velstep=velstep, **args)
    with mcalf_amd.als_fitter(path, ranges, ["CIV 1548", "CIV 1550"], [2, 5], **args) as fit:
        assert fit.numfitranges == 2 and fit.obj_wl.size == wl.size < 1998
        assert np.array_equal(fit.obj_wl, wl) and np.array_equal(fit.obj, fl) and np.array_equal(fit.obj_noise, er)
        assert fit.velstep == velstep                            # computed by the constructor, not passed in
        assert np.diff(fit.obj_wl).max() > 1.0                   # the mask really leaves a hole
        assert (fit.ndim, fit.startind, fit.endind) == (prob.ndim, prob.startind, prob.endind)
        for a, b in zip(fit.bounds, prob.bounds):
            assert np.array_equal(np.asarray(a, float), np.asarray(b, float))
        rng = np.random.default_rng(77)
        cubes = rng.random((96, fit.ndim))
        P = np.array([fit._scale_cube_pc(c) for c in cubes])
        assert np.array_equal(P, np.array([o.scale_cube_pc(prob, c) for c in cubes]))
        got = fit.loglike_batch(P)
        want = o.loglike_batch(prob, P)
        assert np.abs(got - want).max() < LOGL_ATOL
        assert (np.abs(got - want) / np.abs(want)).max() < 1e-10
        m = fit.model_batch(P[:6])
        for i in range(6):
            ref = o.reconstruct_spec(prob, P[i])
            assert np.abs(m[i] - ref).max() < 1e-11 and (np.abs(m[i] / ref - 1)).max() < FLUX_RTOL
        # the reference's single-point callables on the same object
        assert abs(fit.lnlhood_pc(P[0])[0] - want[0]) < LOGL_ATOL and fit.lnlhood_pc(P[0])[1] == []
        assert abs(fit.chi2(P[1]) - o.chi2(prob, P[1])) < 1e-6 * abs(o.chi2(prob, P[1]))


def test_config_D_first_shard_against_both_oracles():
    """BASELINE config D: 32768 live points drawn with default_rng(3), contiguous 4096-row shards per GPU.
    Rank 0's shard -- rows 0..4095 of that draw -- against the C oracle (every row) and the numpy / scipy oracle
    (a spread of rows); the other ranks' shards are the same code on other rows."""
    kw, batch, seed = workloads.config("D", oracle_synth)
    assert (batch, seed) == (32768, 3)
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed))[:4096]
    prob = problem_from_kwargs(kw)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(P)
    assert np.isfinite(got).all()
    co = c_oracle.COracle(prob, threads=min(16, os.cpu_count() or 1))
    want_c = co.loglike_batch(P)
    assert np.abs(got - want_c).max() < LOGL_ATOL
    assert (np.abs(got - want_c) / np.abs(want_c)).max() < 1e-10
    idx = np.arange(0, 4096, 128)
    want = o.loglike_batch(prob, P[idx])
    assert np.abs(got[idx] - want).max() < LOGL_ATOL
    nc = P[:, 1].astype(int)
    assert set(np.unique(nc)) == {8, 9, 10}                      # int() on a uniform [8, 11] draw (:207-208)


@pytest.mark.parametrize("free", [True, False])
def test_jaxns_facing_closure_row_J(free):
    """get_jax_likelihood() (hires_fitter.py:521, 685-693; cli.py:237, 256): float32 theta in, float32 logL out,
    JAX-path semantics; equals the float64 restatement of that path on the float32-rounded theta."""
    kw, _, seed = workloads.config("C", oracle_synth)
    if not free:
        kw = dict(kw, specres=[8.0])
    P = workloads.draw_P(kw, 48, np.random.default_rng(seed + 17))
    P32 = P.astype(np.float32)
    prob = problem_from_kwargs(kw)
    want = np.array([o.jax_loglike_f64(prob, p.astype(np.float64)) for p in P32])
    with mcalf_amd.als_fitter(None, **kw) as fit:                # a numpy-path object, as cli.py builds it
        ll = fit.get_jax_likelihood(use_jax=False)
        one = ll(P32[0])
        assert np.ndim(one) == 0 and np.asarray(one).dtype == np.float32
        many = ll(P32)
        assert many.dtype == np.float32 and many.shape == (48,)
        assert many[0] == one
        assert np.array_equal(ll(P32.reshape(6, 8, -1)), many.reshape(6, 8))
        # float64 result of the same context, before the float32 rounding of the return value
        exact = ll.fitter.loglike_batch(P32.astype(np.float64))
        assert np.abs(exact - want).max() < LOGL_ATOL
        assert np.array_equal(many, exact.astype(np.float32))
        assert np.abs(many.astype(np.float64) - want).max() <= 1e-4 + np.abs(want).max() * 2.0 ** -23
        with pytest.raises(ValueError):
            ll(P32[:, :-1])
        try:
            import jax  # noqa: F401
        except ImportError:
            with pytest.raises(ImportError):
                fit.get_jax_likelihood(use_jax=True)
        # a context that already has the JAX semantics hands out a closure over itself
    with mcalf_amd.als_fitter(None, conv_mode="jax", **kw) as fj:
        assert fj.get_jax_likelihood(use_jax=False).fitter is fj


def test_jax_kernel_grid_comes_from_the_second_specres_entry():
    """hires_fitter.py:549-550: with a free resolution the fixed kernel grid is sized from res_lims[1], the SECOND
    entry of specres -- not the maximum.  specres=[9, 8] therefore gives a shorter grid than [8, 9]."""
    kw, _, seed = workloads.config("C", oracle_synth)
    kw = dict(kw, specres=[9.0, 8.0])
    P = workloads.draw_P(kw, 6, np.random.default_rng(seed + 3))
    prob = problem_from_kwargs(kw)
    assert o.jax_half_size(prob) == int(np.ceil(np.float32(3.0348 * (8.0 / 2.354820) / prob.velstep)))
    with mcalf_amd.als_fitter(None, conv_mode="jax", **kw) as fit:
        assert fit.info.n_cap == o.jax_half_size(prob)
        got = fit.loglike_batch(P)
    want = np.array([o.jax_loglike_f64(prob, p) for p in P])
    assert np.abs(got - want).max() < LOGL_ATOL


def test_in_library_gather_on_a_one_rank_communicator():
    """mcalf_comm_* + mcalf_loglike_gather_device: RCCL is loaded at run time, the context owns the
    communicator, the gather lands the rank's block in the root's vector.  One rank here (a GPU box has one GPU);
    the N > 1 exchange is the same grouped ncclSend / ncclRecv with more peers."""
    import ctypes as C
    import torch
    from mcalf_amd import _lib
    from mcalf_amd import dist as mdist
    kw, _, seed = workloads.config("C", oracle_synth)
    P = workloads.draw_P(kw, 700, np.random.default_rng(seed + 41))
    with mcalf_amd.als_fitter(None, **kw) as fit:
        want = fit.loglike_batch(P)
        n, r = C.c_int32(-1), C.c_int32(-1)
        _lib.check(fit._lib.mcalf_comm_info(fit._ctx, C.byref(n), C.byref(r)), fit._ctx)
        assert (n.value, r.value) == (0, -1)                      # no communicator yet
        dP = torch.from_numpy(P).cuda()
        with pytest.raises(RuntimeError, match="mcalf_comm_init"):
            _lib.check(fit._lib.mcalf_loglike_gather_device(fit._ctx, dP.data_ptr(), 700, dP.data_ptr(), dP.data_ptr(), 0, None),
                       fit._ctx)
        g = mdist.InLibGather(fit, 700, "cuda")
        _lib.check(fit._lib.mcalf_comm_info(fit._ctx, C.byref(n), C.byref(r)), fit._ctx)
        assert (n.value, r.value) == (1, 0)
        for _ in range(3):
            g.step(dP)
        torch.cuda.synchronize()
        assert np.array_equal(g.local.cpu().numpy(), want) and np.array_equal(g.all.cpu().numpy(), want)
        g.close()
        _lib.check(fit._lib.mcalf_comm_info(fit._ctx, C.byref(n), C.byref(r)), fit._ctx)
        assert n.value == 0


def test_two_rank_job_through_the_product_path_on_one_gpu():
    """bench.py --gpus 2 --backend gloo: two processes, each with its own context on the one GPU of the box, the
    real kernels, the row sharding of mcalf_amd.dist and a (gloo) gather of the logL shards to rank 0 -- the N > 1
    control flow of the driver's multi-GPU runs with everything but the RCCL transport."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "D",
           "--batch", "2048", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["rccl_ranks"] == 2
    assert out["config"]["batch_per_gpu"] == 1024 and out["config"]["global_batch"] == 2048
    assert out["gather_check"] == {"rows": 2048, "own_block_equal": True, "all_finite": True}
    assert out["parity"]["max_abs_dlogL_vs_oracle"] < LOGL_ATOL
    # the three gathers of the N > 1 line: torch's is timed; without a transport for the library's exchange (one GPU, no
    # stand-in named) the two library legs say so instead of running
    assert out["gather_reported"] == "torch" and list(out["gathers"]) == ["torch", "inlib", "inlib_overlap"]
    g = out["gathers"]["torch"]
    assert g["gather_check"] == out["gather_check"] and g["ms_per_step"] == pytest.approx(out["ms_per_step"])
    assert 0 < g["kernel_ms_min_over_ranks"] <= g["kernel_ms_max_over_ranks"]
    # what the first real multi-GPU run will be read by: per-leg rank count and its source, the ranks' kernel times, and
    # where rank 0's (synchronous) step goes
    assert (g["rccl_ranks"], g["rccl_ranks_source"]) == (2, "torch.distributed.get_world_size")
    sp = g["rank0_split"]
    assert set(sp) >= {"steps", "kernels_ms", "exchange_ms", "join_ms", "step_ms_synchronous", "how"}
    assert sp["kernels_ms"] > 0 and sp["exchange_ms"] >= 0 and sp["join_ms"] >= 0
    assert sp["step_ms_synchronous"] == pytest.approx(sp["kernels_ms"] + sp["exchange_ms"] + sp["join_ms"])
    assert all(isinstance(out["gathers"][k], str) and out["gathers"][k].startswith("skipped") for k in ("inlib", "inlib_overlap"))


def test_from_the_reference_s_own_ini_and_ascii_inputs_to_the_golden_logl(tmp_path, monkeypatch):
    """What a user of the reference does (cli.py:55-76): `mcalf fit.cfg` -> `readconfig` -> `als_fitter(specfile, wavefit,
    linelist, ncomp, ...)` read from disk.  Here with the reference's OWN example configuration and data file
    (tests/golden/fit.cfg, civ_mock_spec_multicomp.txt: byte copies) laid out as the configuration expects them
    (`./testdata/...`), through `als_fitter.from_config`: 1998 pixels in 6180-6220 A, ncomp 8-11 -> ndim 34, velstep the
    clipped median of the file's own grid; logL and chi2 at the 10-component truth equal the golden values (the 11th
    component slot is inactive: int(p[0]) = 10), and the reference's solver-facing callables agree with each other."""
    import json
    import shutil
    from mcalf_amd.routines import hires_fitter as h
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    os.makedirs(tmp_path / "testdata")
    shutil.copy(os.path.join(gold, "civ_mock_spec_multicomp.txt"), tmp_path / "testdata" / "civ_mock_spec_multicomp.txt")
    shutil.copy(os.path.join(gold, "fit.cfg"), tmp_path / "fit.cfg")
    monkeypatch.chdir(tmp_path)
    pars = h.readconfig("fit.cfg")
    g = json.load(open(os.path.join(gold, "derived_goldens.json")))
    with h.als_fitter.from_config(pars) as fit:
        assert (fit.ndim, fit.startind, fit.endind, fit.ncompmin, fit.ncompmax, fit.nfill) == (34, 0, 34, 8, 11, 0)
        assert fit.obj_wl.size == 1998 and abs(fit.velstep - g["velstep"]) < 1e-12
        assert not fit.freespecres and not fit.freecont and fit.fitlines == ["CIV 1548", "CIV 1550"]
        p = np.concatenate([workloads.truth_vector(10), [13.0, 3.0, 20.0]])      # the 11th slot: never evaluated
        pc = fit.lnlhood_pc(p)
        assert pc[1] == [] and abs(pc[0] - g["G3_logL_truth"]) < 1e-7
        assert fit.lnlhood_dy(p) == pc[0] and fit.lnlhood_mn(list(p), 34, 34) == pc[0]
        assert abs(fit.chi2(p) - g["G3_chi2_truth"]) < 1e-7
        # the prior transform of the solver branch (cli.py:110: _scale_cube_pc) stays inside the configuration's box
        th = fit._scale_cube_pc(np.full(34, 0.5))
        assert th[0] == 9.0 and abs(th[1] - 13.25) < 1e-12 and abs(th[2] - 3.0) < 1e-12 and abs(th[3] - 25.0) < 1e-12
