"""GPU: ONE context over several devices (mcalf_create_multi; SURVEY.md 8(b) `ndevices`, section 5).  The reference's
large batches arise inside one process (jaxns vmaps the likelihood over the live points: cli.py:274-280): the
host-pointer entries of a multi-device context cut their rows into contiguous blocks, one per device entry, and every
entry writes its block straight into the caller's array.  A test box has ONE GPU, so the entries here all name device 0
(two / three independent sub-contexts, a helper thread each) -- which exercises the sharding arithmetic, the concurrent
calls (two streaming launches on one device at once), the ragged split and the error conventions; the speed-up over
real devices is unmeasured on hardware.  Everything must equal the single-device context bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth

pytestmark = pytest.mark.gpu


def _shards(batch, n):
    from mcalf_amd import dist as mdist
    return [mdist.shard_bounds(batch, n, k) for k in range(n)]


@pytest.mark.parametrize("cfg,rows,devices", [("C", 4096, [0, 0]), ("C", 4096, [0, 0, 0]), ("E", 2048, [0, 0]), ("E", 1000, [0, 0, 0])])
def test_multi_device_context_equals_the_single_context_bit_for_bit(cfg, rows, devices):
    kw, _, seed = workloads.config(cfg, oracle_synth)
    P = workloads.draw_P(kw, rows, np.random.default_rng(seed), damped=2 if cfg == "E" else 0)
    with mcalf_amd.als_fitter(None, **kw) as one:
        want = one.loglike_batch(P)
        want_chi2 = one.chi2_batch(P[:700])
        want_model = one.model_batch(P[:600])
        cubes = np.random.default_rng(5).random((rows, one.ndim))
        want_theta, want_cube = one.loglike_cube_batch(cubes)
    n = len(devices)
    with mcalf_amd.als_fitter(None, device=devices, **kw) as fit:
        assert fit.info.ndevices == n and list(fit.info.devices[:n]) == devices
        assert fit.get_config().startswith("devices=%d " % n)
        got = fit.loglike_batch(P)
        assert np.array_equal(got, want)
        ll = fit.last_launch()
        assert ll.devices_used == n
        # every device entry ran ITS contiguous block through the plan its size selects
        for k, (lo, hi) in enumerate(_shards(rows, n)):
            sub = fit.last_launch(sub=k)
            assert sub.items == (hi - lo) * fit.info.ntiles or sub.path == _lib.MCALF_PATH_HOST_PIPELINED, (k, sub.items)
            assert sub.path in (_lib.MCALF_PATH_HOST_STREAM, _lib.MCALF_PATH_HOST_PIPELINED, _lib.MCALF_PATH_HOST_ZEROCOPY)
        # page-locked caller arrays: every entry reads / writes its slice of them directly
        P_pin = torch.from_numpy(P).pin_memory().numpy()
        out_pin = torch.full((rows,), float("nan"), dtype=torch.float64).pin_memory().numpy()
        assert np.array_equal(fit.loglike_batch(P_pin, out=out_pin), want)
        # the other batched entries shard alike (700 / 600 rows: two entries of >= 256 rows)
        assert np.array_equal(fit.chi2_batch(P[:700]), want_chi2) and fit.last_launch().devices_used == min(n, 2)
        assert np.array_equal(fit.model_batch(P[:600]), want_model)
        theta, ll_cube = fit.loglike_cube_batch(cubes)
        assert np.array_equal(theta, want_theta) and np.array_equal(ll_cube, want_cube)
        # small calls and the reference's one-theta callables stay on entry 0
        assert np.array_equal(fit.loglike_batch(P[:100]), want[:100]) and fit.last_launch().devices_used == 1
        assert fit.lnlhood_dy(P[3]) == want[3] and fit.lnlhood_pc(P[4]) == (want[4], [])
        fit.set_resident(300)
        assert fit.lnlhood_dy(P[5]) == want[5] and fit.last_launch(sub=0).inline_setup == (3 if fit.info.ntiles == 1 else fit.last_launch(sub=0).inline_setup)
        fit.set_resident(0)
        # repeated calls: the helper threads take job after job
        for _ in range(3):
            assert np.array_equal(fit.loglike_batch(P), want)


def test_multi_device_context_refuses_what_belongs_to_one_device():
    kw, _, seed = workloads.config("B", oracle_synth)
    P = workloads.draw_P(kw, 16, np.random.default_rng(seed))
    with mcalf_amd.als_fitter(None, device=[0, 0], **kw) as fit:
        lib, ctx = fit._lib, fit._ctx
        dP = torch.from_numpy(P).cuda()
        out = torch.empty(16, dtype=torch.float64, device="cuda")
        rc = lib.mcalf_loglike_batch_device(ctx, dP.data_ptr(), 16, out.data_ptr(), None)
        assert rc == _lib.MCALF_ERR_INVALID and b"multi-device" in lib.mcalf_last_error(ctx)
        assert lib.mcalf_profile_begin(ctx, 4) == _lib.MCALF_ERR_INVALID
        ident = (C.c_char * 128)()
        assert lib.mcalf_comm_init(ctx, ident, 1, 0) == _lib.MCALF_ERR_INVALID
        # still usable afterwards, and mcalf_last_launch_sub knows its range
        assert np.isfinite(fit.loglike_batch(P)).all()
        info = _lib.mcalf_launch_info_t()
        assert lib.mcalf_last_launch_sub(ctx, 2, C.byref(info)) == _lib.MCALF_ERR_INVALID
    with pytest.raises(RuntimeError, match="device entry 1"):
        mcalf_amd.als_fitter(None, device=[0, 99], **kw)


def test_get_config_names_the_knobs_and_the_environment(monkeypatch):
    kw, _, _ = workloads.config("B", oracle_synth)
    monkeypatch.setenv("MCALF_STREAM", "0")
    monkeypatch.setenv("MCALF_HOST_FIRST_KB", "64")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        cfg = fit.get_config()
    assert "devices=1 " in cfg and " stream=0 " in cfg and " host_first_kb=64 " in cfg
    assert cfg.endswith("[env: MCALF_HOST_FIRST_KB MCALF_STREAM]")
    monkeypatch.delenv("MCALF_STREAM")
    monkeypatch.delenv("MCALF_HOST_FIRST_KB")
    with mcalf_amd.als_fitter(None, **kw) as fit:
        cfg = fit.get_config()
    assert " stream=1 " in cfg and " host_first_kb=128 " in cfg and cfg.endswith("[env: none]")


def test_bench_one_process_multi_device_leg():
    """`bench.py`'s one-process multi-device leg (run by the N = 1 bench in a child process wherever the process sees
    several GPUs; here forced onto two entries of the one GPU): config D's 32768 rows through a context over the entries
    against the first entry alone -- bit-equal, every entry used."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--multi-device-leg", "0,0"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["rows"] == 32768 and d["devices"] == [0, 0]
    assert d["n1"]["devices_used"] == 1 and d["n2"]["devices_used"] == 2
    assert d["n2"]["bit_equal_to_one_device"] and d["n2"]["ms_per_step"] > 0 and d["n2"]["speedup_vs_one_device"] > 0.5
