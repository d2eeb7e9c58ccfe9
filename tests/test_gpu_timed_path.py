"""GPU: the path that bench.py times -- `mcalf_loglike_batch_device`, ONE persistent launch over the whole batch --
against the oracles on every row of BASELINE configs C (4096 rows) and D (32768 rows), with the launch mode read
back from the library (`mcalf_last_launch`) instead of inferred from the batch size; and the pipelined host-pointer
entry (pageable / page-locked, every block plan) asserted to be the path taken."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth, problem_from_kwargs, require_streaming_shape
from oracle import c_oracle
from oracle import numpy_oracle as o

pytestmark = pytest.mark.gpu

LOGL_ATOL = 1e-4          # BASELINE.json north_star: logL within 1e-4 absolute


def _device_logl(fit, dP, n):
    out = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("cfg", ["C", "D"])
def test_one_persistent_launch_over_the_full_batch_against_both_oracles(cfg):
    """Exactly what bench.py times (config C: its N = 1 headline; config D: the strong-scaling N = 1 leg and, in
    4096-row shards, every GPU's share at N = 8): device entry, one launch, persistent grid, ordered hand-out."""
    kw, batch, seed = workloads.config(cfg, oracle_synth)
    assert (batch, seed) == {"C": (4096, 2), "D": (32768, 3)}[cfg]
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed))
    prob = problem_from_kwargs(kw)
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, batch), fit._ctx)
        got = _device_logl(fit, dP, batch)
        ll = fit.last_launch()
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert ll.path == _lib.MCALF_PATH_DEVICE and ll.row_blocks == 1
        assert ll.persistent == 1 and ll.grid == 2 * cus and ll.items == batch      # ONE persistent launch
        assert ll.selfhalo == 1 and ll.lines_per_sync == 5
        assert ll.ordered == (1 if cfg == "C" else 0)           # ordered hand-out up to 16 items per workgroup slot
        again = _device_logl(fit, dP, batch)                     # the queue hands items out in another order:
        assert np.array_equal(got, again)                        # values do not depend on who evaluates what when
    assert np.isfinite(got).all()
    co = c_oracle.COracle(prob, threads=min(16, os.cpu_count() or 1))
    want_c = co.loglike_batch(P)                                 # EVERY row
    assert np.abs(got - want_c).max() < LOGL_ATOL
    assert (np.abs(got - want_c) / np.abs(want_c)).max() < 1e-10
    idx = np.arange(0, batch, batch // 32)                       # a spread of rows against the numpy / scipy oracle
    want = o.loglike_batch(prob, P[idx])
    assert np.abs(got[idx] - want).max() < LOGL_ATOL


def test_config_e_per_gpu_shard_through_the_device_entry_against_both_oracles():
    """BASELINE config E's share of ONE GPU at N = 8 -- rows 0 .. 2047 of the 16384 of `default_rng(4)` (contiguous row
    blocks, rank 0's) -- through `mcalf_loglike_batch_device`: one persistent launch over 2048 x 5 pixel tiles (multi-tile
    instantiation, no ordered hand-out), the finalize kernel behind it; every row against the C oracle, a spread against
    the numpy / scipy oracle (damped Ly-alpha wings: the wide-wing Faddeeva regime)."""
    kw, batch, seed = workloads.config("E", oracle_synth)
    assert (batch, seed) == (16384, 4)
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2)[:2048].copy()
    n = P.shape[0]
    prob = problem_from_kwargs(kw)
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = _device_logl(fit, dP, n)
        ll = fit.last_launch()
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert fit.info.ntiles == 5
        assert (ll.path, ll.row_blocks, ll.persistent, ll.grid, ll.items) == (_lib.MCALF_PATH_DEVICE, 1, 1, 2 * cus, 5 * n)
        assert (ll.selfhalo, ll.ordered, ll.inline_setup) == (0, 0, 0)
    assert np.isfinite(got).all()
    want_c = c_oracle.COracle(prob, threads=min(16, os.cpu_count() or 1)).loglike_batch(P)      # EVERY row
    assert np.abs(got - want_c).max() < LOGL_ATOL
    assert (np.abs(got - want_c) / np.abs(want_c)).max() < 1e-10
    idx = np.arange(0, n, n // 16)
    assert np.abs(got[idx] - o.loglike_batch(prob, P[idx])).max() < LOGL_ATOL


def test_config_e_full_batch_every_row_against_the_c_oracle():
    """BASELINE config E at FULL size: all 16384 rows of `default_rng(4)` through the device entry (ONE persistent launch
    over 16384 x 5 pixel tiles) and through the host-pointer entry, every row against the C oracle (16 threads, ~20 s):
    |dlogL| < 1e-4 absolute (north_star) and 1e-10 relative; the two entries agree bit for bit."""
    kw, batch, seed = workloads.config("E", oracle_synth)
    P = workloads.draw_P(kw, batch, np.random.default_rng(seed), damped=2)
    prob = problem_from_kwargs(kw)
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = _device_logl(fit, dP, batch)
        ll = fit.last_launch()
        assert (ll.path, ll.row_blocks, ll.persistent, ll.items) == (_lib.MCALF_PATH_DEVICE, 1, 1, 5 * batch)
        host = fit.loglike_batch(P)
        assert fit.last_launch().path in (_lib.MCALF_PATH_HOST_PIPELINED, _lib.MCALF_PATH_HOST_STREAM)
    assert np.isfinite(got).all() and np.array_equal(host, got)
    want = c_oracle.COracle(prob, threads=min(16, os.cpu_count() or 1)).loglike_batch(P)        # EVERY row
    assert np.abs(got - want).max() < LOGL_ATOL
    assert (np.abs(got - want) / np.abs(want)).max() < 1e-10


def test_jax_semantics_persistent_launch_against_the_f64_restatement():
    """conv_mode='jax' (hires_fitter.py:521-695: fixed kernel grid, zero padding, edge reset, floor on the ncomp slot) on
    config C with 2600 rows through the device entry -- the persistent `<true, ...>` instantiation a vectorised jaxns
    batch would run -- against `jax_loglike_f64` on a spread of 64 rows; and bit-equal to the same rows evaluated as
    small calls (the one-launch variant)."""
    kw, _, seed = workloads.config("C", oracle_synth)
    kw = dict(kw, conv_mode="jax")
    n = 2600
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 41))
    prob = problem_from_kwargs(kw)
    dP = torch.from_numpy(P).cuda()
    idx = np.arange(0, n, n // 64)[:64]
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = _device_logl(fit, dP, n)
        ll = fit.last_launch()
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert (ll.path, ll.persistent, ll.grid, ll.items, ll.inline_setup) == (_lib.MCALF_PATH_DEVICE, 1, 2 * cus, n, 0)
        small = fit.loglike_batch(P[idx])
        assert fit.last_launch().inline_setup == 1
    want = np.array([o.jax_loglike_f64(prob, p) for p in P[idx]])
    assert np.abs(got[idx] - want).max() < LOGL_ATOL
    assert np.array_equal(got[idx], small)


def test_persistent_switch_is_really_taken_and_changes_nothing(monkeypatch):
    """MCALF_PERSIST / MCALF_ORDER / MCALF_LINES_PER_SYNC are scheduling choices.  The mode is read back from the
    library for every context (device entry, >= 4 items per workgroup slot so that the persistent grid applies)."""
    for cfg, n in (("C", 2600), ("E", 450)):
        kw, _, seed = workloads.config(cfg, oracle_synth)
        P = workloads.draw_P(kw, n, np.random.default_rng(seed + 77), damped=2 if cfg == "E" else 0)
        dP = torch.from_numpy(P).cuda()
        results = {}
        for persist, order, lps in (("1", "1", "4"), ("1", "1", "5"), ("1", "0", "4"), ("0", "1", "4"), ("0", "0", "5")):
            monkeypatch.setenv("MCALF_PERSIST", persist)
            monkeypatch.setenv("MCALF_ORDER", order)
            monkeypatch.setenv("MCALF_LINES_PER_SYNC", lps)
            with mcalf_amd.als_fitter(None, **kw) as fit:
                results[(persist, order, lps)] = _device_logl(fit, dP, n)
                ll = fit.last_launch()
                items = n * fit.info.ntiles
                assert ll.items == items and items >= 4 * 512
                assert ll.persistent == int(persist), (cfg, persist)
                assert ll.grid == (ll.grid if persist == "1" else items) and (persist == "0" or ll.grid < items)
                assert ll.lines_per_sync in (4, 5)
        ref = results[("1", "1", "4")]
        assert np.isfinite(ref).all()
        for key, val in results.items():
            assert np.array_equal(val, ref), (cfg, key)


def test_pipelined_host_entry_is_the_path_taken_and_is_bit_equal(monkeypatch):
    """Large scalar-output calls through host pointers (batch * ndim > 65536 doubles).  Default plan: ONE streaming
    launch (`run_host_stream`, MCALF_PATH_HOST_STREAM) once the launch reaches the persistent-grid threshold; an
    explicit block count, MCALF_HOST_PLAN or MCALF_STREAM=0 select the row-block pipeline (`run_host_pipelined`):
    pageable arrays are staged, page-locked ones are used by the copy engines directly and the kernels write logL
    into the caller's page-locked array.  Every plan gives the device entry's bits."""
    kw, _, seed = workloads.config("C", oracle_synth)
    n = 2600
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 13))
    assert P.size > 65536
    Ppin = torch.from_numpy(P).pin_memory().numpy()
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        require_streaming_shape(fit)
        whole = _device_logl(fit, dP, n)
        assert np.isfinite(whole).all()
        for k, blocks_pageable, blocks_pinned in ((0, 1, 1), (1, 1, 1), (2, 2, 2), (5, 5, 5)):
            fit.set_chunks(k)
            path = _lib.MCALF_PATH_HOST_STREAM if k == 0 else _lib.MCALF_PATH_HOST_PIPELINED
            got = fit.loglike_batch(P)
            ll = fit.last_launch()
            assert (ll.path, ll.row_blocks, ll.pinned_in, ll.pinned_out) == (path, blocks_pageable, 0, 0)
            assert np.array_equal(got, whole), k
            opin = torch.full((n,), float("nan"), dtype=torch.float64).pin_memory().numpy()
            fit.loglike_batch(Ppin, out=opin)
            ll = fit.last_launch()
            assert (ll.path, ll.row_blocks, ll.pinned_in, ll.pinned_out) == (path, blocks_pinned, 1, 1)
            assert np.array_equal(opin, whole), k
            # mixed: page-locked rows, pageable results
            mixed = np.full(n, np.nan)
            fit.loglike_batch(Ppin, out=mixed)
            assert fit.last_launch().pinned_in == 1 and fit.last_launch().pinned_out == 0
            assert np.array_equal(mixed, whole)
            assert np.array_equal(fit.chi2_batch(P), fit.chi2_batch(P[::-1].copy())[::-1])
        fit.set_chunks(0)
        # small calls take the zero-copy block instead
        assert np.array_equal(fit.loglike_batch(P[:100]), whole[:100])
        assert fit.last_launch().path == _lib.MCALF_PATH_HOST_ZEROCOPY
        # below the persistent-grid threshold (4 items per workgroup slot) a large call is pipelined in row blocks
        nsmall = 1500
        assert nsmall * P.shape[1] > 65536
        assert np.array_equal(fit.loglike_batch(P[:nsmall]), whole[:nsmall])
        ll = fit.last_launch()
        assert (ll.path, ll.row_blocks) == (_lib.MCALF_PATH_HOST_PIPELINED, 3)
    # the row-block pipeline as the default plan (MCALF_STREAM=0): a first block of 128 KiB of rows (348 of these), then
    # doubling, the last block takes the rest -- 348 + 696 + 1556 pageable; twice the first block page-locked: 696 + 1904
    monkeypatch.setenv("MCALF_STREAM", "0")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        assert np.array_equal(fit.loglike_batch(P), whole)
        ll = fit.last_launch()
        assert (ll.path, ll.row_blocks, ll.pinned_in) == (_lib.MCALF_PATH_HOST_PIPELINED, 3, 0)
        opin = torch.full((n,), float("nan"), dtype=torch.float64).pin_memory().numpy()
        fit.loglike_batch(Ppin, out=opin)
        ll = fit.last_launch()
        assert (ll.path, ll.row_blocks, ll.pinned_in, ll.pinned_out) == (_lib.MCALF_PATH_HOST_PIPELINED, 2, 1, 1)
        assert np.array_equal(opin, whole)
    monkeypatch.delenv("MCALF_STREAM")
    # explicit relative plans; 1 : 1 : 5000 of 2600 rows leaves block 0 EMPTY and one row in block 1
    for plan in ("1,3,4", "1,1,5000"):
        monkeypatch.setenv("MCALF_HOST_PLAN", plan)
        with mcalf_amd.als_fitter(None, **kw) as fit:
            assert np.array_equal(fit.loglike_batch(P), whole), plan
            assert fit.last_launch().row_blocks == 3
    monkeypatch.setenv("MCALF_HOST_PLAN", "1,30,30")
    kwB, _, seedB = workloads.config("E", oracle_synth)        # E: ndim 49; 1400 rows > 65536 doubles
    PE = workloads.draw_P(kwB, 1400, np.random.default_rng(seedB + 5), damped=2)
    with mcalf_amd.als_fitter(None, **kwB) as fit:
        a = fit.loglike_batch(PE)
        assert fit.last_launch().path == _lib.MCALF_PATH_HOST_PIPELINED
        fit.set_chunks(1)
        assert np.array_equal(fit.loglike_batch(PE), a)


@pytest.mark.parametrize("cfg,conv,n", [("C", "numpy", 4096), ("C", "jax", 2300), ("E", "numpy", 1400), ("B", "numpy", 2700)])
def test_streaming_host_entry_one_launch_no_copy_commands(cfg, conv, n, monkeypatch):
    """`run_host_stream`: the whole call is ONE launch of `mcalf_fused_kernel<..., kStream = true>` -- the grid reads the
    parameter rows from page-locked memory while the host is still staging them, sets the live points up itself (every
    XCD its own rows: a few dedicated workgroups per XCD run ahead of the XCD's item queue) and writes logL into
    page-locked memory.  Asserted to be the path
    taken, for pageable / page-locked / mixed arrays, single- and multi-tile spectra, both boundary modes, logL and chi2;
    bit-equal to the device entry (set-up kernel + fused kernel) and within tolerance of the C oracle on every row; the
    set-up geometry (dedicated workgroups, rows per claim, completion by polling or by the stream) changes nothing; a
    wait that runs out inside the kernel fails over to the row-block pipeline with the same bits."""
    kw, _, seed = workloads.config(cfg, oracle_synth)
    kw = dict(kw, conv_mode=conv)
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 77), damped=2 if cfg == "E" else 0)
    assert P.size > 65536
    if cfg == "E":
        # a tiled spectrum streams up to 65536 work items (1400 x 5 here) and takes the row-block pipeline beyond (measured:
        # profiles/r06_tiled_stream_crossover.txt); MCALF_STREAM=2 streams every size
        with mcalf_amd.als_fitter(None, **kw) as fit:
            require_streaming_shape(fit)
            small_tiled = fit.loglike_batch(P)
            assert fit.last_launch().path == _lib.MCALF_PATH_HOST_STREAM
            big = np.ascontiguousarray(np.tile(P, (10, 1)))                    # 14000 x 5 = 70000 work items
            big_ll = fit.loglike_batch(big)
            assert fit.last_launch().path == _lib.MCALF_PATH_HOST_PIPELINED
            assert np.array_equal(big_ll[:n], small_tiled) and np.array_equal(big_ll[-n:], small_tiled)
        monkeypatch.setenv("MCALF_STREAM", "2")
    Ppin = torch.from_numpy(P).pin_memory().numpy()
    dP = torch.from_numpy(P).cuda()
    ref = {}
    for env in ({}, {"MCALF_STREAM_WGS": "40", "MCALF_STREAM_CHUNK": "8"}, {"MCALF_STREAM_WGS": "5", "MCALF_STREAM_CHUNK": "104"},
                {"MCALF_STREAM_POLL": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with mcalf_amd.als_fitter(None, **kw) as fit:
            require_streaming_shape(fit)
            if not ref:
                ref["logl"] = _device_logl(fit, dP, n)
                assert fit.last_launch().path == _lib.MCALF_PATH_DEVICE
                fit.set_chunks(1)
                ref["chi2"] = fit.chi2_batch(P)                   # (an explicit block count: the pipeline, one block)
                assert fit.last_launch().path == _lib.MCALF_PATH_HOST_PIPELINED
                fit.set_chunks(0)
            for rep in range(3):
                if rep == 1:
                    # a DIFFERENT matrix in between (the rows in reverse order): the workspaces the waves of a launch
                    # hand each other are reused by every call, and a stale read of the previous call's records must
                    # not hide behind identical values
                    assert np.array_equal(fit.loglike_batch(P[::-1].copy()), ref["logl"][::-1]), (env, "reversed")
                    assert fit.last_launch().path == _lib.MCALF_PATH_HOST_STREAM
                    continue
                got = fit.loglike_batch(P)
                ll = fit.last_launch()
                assert (ll.path, ll.row_blocks, ll.persistent, ll.ordered, ll.pinned_in, ll.pinned_out) == \
                    (_lib.MCALF_PATH_HOST_STREAM, 1, 1, 0, 0, 0)
                assert ll.grid == 2 * torch.cuda.get_device_properties(0).multi_processor_count
                assert ll.items == n * fit.info.ntiles
                assert ll.stream_setup_wgs == -(-int(env.get("MCALF_STREAM_WGS", 16)) // 8) * 8    # (so many per XCD)
                # (completion read off the kernel's page-locked word; tiled spectra: the finalize kernel's word, round 6)
                assert ll.stream_polled == (0 if env.get("MCALF_STREAM_POLL") == "0" else 1)
                assert np.array_equal(got, ref["logl"]), (env, rep)
            opin = torch.full((n,), float("nan"), dtype=torch.float64).pin_memory().numpy()
            fit.loglike_batch(Ppin, out=opin)
            ll = fit.last_launch()
            assert (ll.path, ll.pinned_in, ll.pinned_out) == (_lib.MCALF_PATH_HOST_STREAM, 1, 1)
            assert np.array_equal(opin, ref["logl"]), env
            mixed = np.full(n, np.nan)
            fit.loglike_batch(Ppin, out=mixed)
            assert (fit.last_launch().pinned_in, fit.last_launch().pinned_out) == (1, 0)
            assert np.array_equal(mixed, ref["logl"])
            assert np.array_equal(fit.chi2_batch(P), ref["chi2"], equal_nan=True)
            assert fit.last_launch().path == _lib.MCALF_PATH_HOST_STREAM
        for k in env:
            monkeypatch.delenv(k)
    # against the C oracle: every row (config E's 20000-pixel rows: every fourth)
    if conv == "numpy":
        rows = np.arange(0, n, 4 if cfg == "E" else 1)
        want = c_oracle.COracle(problem_from_kwargs(kw), threads=min(16, os.cpu_count() or 1)).loglike_batch(P[rows])
        assert np.abs(ref["logl"][rows] - want).max() < LOGL_ATOL
    # a wait that runs out (here: a limit of 10 ns, shorter than any PCIe round trip) raises the kernel's status word;
    # the grid drains, the call fails over to the pipeline and still returns the right bits
    monkeypatch.setenv("MCALF_STREAM_TIMEOUT", "1e-8")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(P)
        assert fit.last_launch().path == _lib.MCALF_PATH_HOST_PIPELINED
        assert fit.last_launch().stream_fallback == _lib.MCALF_STREAM_FALLBACK_TIMEOUT
        assert np.array_equal(got, ref["logl"])
        # ... and the queues were re-armed: with the limit restored the next context streams again
    monkeypatch.delenv("MCALF_STREAM_TIMEOUT")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        assert np.array_equal(fit.loglike_batch(P), ref["logl"])
        assert fit.last_launch().path == _lib.MCALF_PATH_HOST_STREAM


@pytest.mark.parametrize("cfg,conv", [("A", "numpy"), ("C", "numpy"), ("E", "numpy"), ("C", "jax"), ("E", "jax")])
def test_one_launch_variant_of_small_calls_gives_the_two_kernel_bits(cfg, conv, monkeypatch):
    """Calls of up to 2 x CUs work items (the one-theta-at-a-time solvers) run ONE kernel whose workgroups set their
    live point up themselves; larger ones run set-up kernel + fused kernel.  Same set-up code: a live point's value
    must not depend on the size of the batch it arrives in -- logL, chi2, model, single components, cube input."""
    kw, _, seed = workloads.config(cfg, oracle_synth)
    kw = dict(kw, conv_mode=conv)
    rng = np.random.default_rng(seed + 321)
    P = workloads.draw_P(kw, 9, rng, damped=2 if cfg == "E" else 0)
    cubes = rng.random((5, P.shape[1]))
    out = {}
    for inline_max in ("0", None):
        if inline_max is None:
            monkeypatch.delenv("MCALF_INLINE_MAX", raising=False)
        else:
            monkeypatch.setenv("MCALF_INLINE_MAX", inline_max)
        with mcalf_amd.als_fitter(None, **kw) as fit:
            res = []
            for n in (1, 2, 9):
                res.append(fit.loglike_batch(P[:n]))
                assert fit.last_launch().inline_setup == (0 if inline_max == "0" else 1)
            res.append(fit.chi2_batch(P[:3]))
            res.append(fit.model_batch(P[:2]))
            assert fit.last_launch().inline_setup == (0 if inline_max == "0" else 1)
            res.append(fit.model_batch(P[:2], targonly=True))
            res.append(fit.onecomp_batch(np.array([[8.0, 1.0, 13.5, P[0][fit.startind + 2], 20.0]]), line=0))
            res.append(fit.onecomp_batch(np.array([[7.5, 0.97, 13.1, P[0][fit.startind + 2], 12.0]]), fill=True))
            th, ll = fit.loglike_cube_batch(cubes)
            res += [th, ll]
            res.append(np.array([fit.lnlhood_pc(P[3])[0], fit.lnlhood_dy(P[4])]))
            out[inline_max] = res
    for a, b in zip(out["0"], out[None]):
        assert np.array_equal(a, b, equal_nan=True)
    # ... and a large batch (two-kernel path, persistent grid for C) gives these rows the same bits
    big = workloads.draw_P(kw, 2600 if cfg != "E" else 600, rng, damped=2 if cfg == "E" else 0)
    big[:9] = P
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = fit.loglike_batch(big)
        assert fit.last_launch().inline_setup == 0
    assert np.array_equal(got[:9], out[None][2])


def test_model_output_entry_on_the_persistent_grid_against_the_oracle():
    """`mcalf_model_batch_device` as bench.py's model-output leg drives it -- one persistent launch, the plain
    model-output epilogue -- against the oracle's reconstruct_spec on a spread of rows, and bit-equal to the same
    rows evaluated one at a time (the one-launch variant): hires_fitter.py:409-449, consumer cli.py:414-418."""
    kw, _, seed = workloads.config("C", oracle_synth)
    n = 2600
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 55))
    prob = problem_from_kwargs(kw)
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        npix = fit.obj_wl.size
        flux = torch.full((n, npix), float("nan"), dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for targonly in (0, 1):
            _lib.check(fit._lib.mcalf_model_batch_device(fit._ctx, dP.data_ptr(), n, targonly, flux.data_ptr(), st), fit._ctx)
            torch.cuda.synchronize()
            ll = fit.last_launch()
            assert ll.path == _lib.MCALF_PATH_DEVICE and ll.persistent == 1 and ll.items == n and ll.inline_setup == 0
            got = flux.cpu().numpy()
            assert np.isfinite(got).all()
            for i in range(0, n, 325):
                ref = o.reconstruct_spec(prob, P[i], targonly=bool(targonly))
                assert np.abs(got[i] - ref).max() < 1e-11 and np.abs(got[i] / ref - 1).max() < 1e-6
                one = fit.model_batch(P[i:i + 1], targonly=bool(targonly))[0]
                assert fit.last_launch().inline_setup == 1
                assert np.array_equal(one, got[i])
        # logL recomputed on the host from the device's model spectra = the log-likelihood entry (hires_fitter.py:292-294)
        _lib.check(fit._lib.mcalf_model_batch_device(fit._ctx, dP.data_ptr(), n, 0, flux.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        m = flux.cpu().numpy()[:64]
        ispec2 = 1.0 / fit.obj_noise ** 2
        ll_host = -0.5 * np.nansum(ispec2 * (fit.obj - m) ** 2 - np.log(ispec2) + np.log(2 * np.pi), axis=1)
        assert np.abs(ll_host - _device_logl(fit, dP, n)[:64]).max() < 1e-6


def test_ordered_persistent_row_blocks_and_graph_replay():
    """Row blocks that are each large enough for the persistent grid get their own queue and their own hand-out
    order (block-local indices); a captured persistent launch replays correctly (the set-up kernel of every replay
    resets the queue and rebuilds the order)."""
    kw, _, seed = workloads.config("C", oracle_synth)
    n = 4600
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 91))
    dP = torch.from_numpy(P).cuda()
    with mcalf_amd.als_fitter(None, **kw) as fit:
        _lib.check(fit._lib.mcalf_reserve(fit._ctx, n), fit._ctx)
        fit.set_chunks(1)
        whole = _device_logl(fit, dP, n)
        ll = fit.last_launch()
        assert ll.persistent == 1 and ll.ordered == 1 and ll.row_blocks == 1
        fit.set_chunks(2)                                        # 2 x 2300 rows: both blocks persistent and ordered
        for _ in range(2):
            assert np.array_equal(_device_logl(fit, dP, n), whole)
        ll = fit.last_launch()
        assert ll.row_blocks == 2 and ll.persistent == 1 and ll.ordered == 1 and ll.items == 2300
        fit.set_chunks(1)
        out = torch.zeros(n, dtype=torch.float64, device="cuda")
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), whole)


@pytest.mark.parametrize("block", ["64", "192", "512"])
def test_set_up_geometry_and_chunked_ordering_do_not_change_results(block, monkeypatch):
    """MCALF_SETUP_BLOCK: the set-up kernel's workgroup size (one wave per live point) and with it the number of
    keys the ordering workgroup holds at a time -- with 64 threads it sorts 4096 live points in four chunks and
    reloads its keys for the second pass.  Scheduling only."""
    kw, _, seed = workloads.config("C", oracle_synth)
    n = 4096
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 7))
    dP = torch.from_numpy(P).cuda()
    monkeypatch.delenv("MCALF_SETUP_BLOCK", raising=False)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        ref = _device_logl(fit, dP, n)
    monkeypatch.setenv("MCALF_SETUP_BLOCK", block)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        got = _device_logl(fit, dP, n)
        assert fit.last_launch().ordered == 1 and fit.last_launch().persistent == 1
        small = fit.loglike_batch(P[:700])                       # two-kernel path, not persistent
        assert fit.last_launch().persistent == 0 and fit.last_launch().inline_setup == 0
    assert np.array_equal(got, ref) and np.array_equal(small, ref[:700])

