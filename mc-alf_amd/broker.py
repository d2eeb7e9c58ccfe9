"""Drop-in mode for MANY solver ranks on one GPU: a likelihood broker.

The reference's solvers call the likelihood one theta at a time -- PolyChord runs one MPI rank per core, each calling
`lnlhood_pc(theta)` serially (cli.py:37-41, 110; hires_fitter.py:250-262).  With one device context per rank every call
is its own kernel launch, and launches of DIFFERENT processes do not overlap beyond a few: measured on one MI355X, four
ranks reach 1.0e5 logL/s, six 1.2e5, and more ranks add nothing (profiles/r04_dropin.json) -- while ONE launch evaluates
a whole batch of live points at the cost of one.

The broker puts that to use without changing the solver: ONE process owns the device context and serves the ranks
through a block of POSIX shared memory.  A rank writes its theta into its slot and bumps the slot's request counter; the
server collects every slot with an open request, evaluates them as ONE batch (`mcalf_loglike_batch`: one launch, the
one-launch variant of small calls), writes the values back and acknowledges.  A live point's value does not depend on
the batch it arrives in (tests/test_gpu_timed_path.py), so every rank gets the bits it would get from its own context.

    server (once per node / GPU):   with LikelihoodBroker(fit, "mcalf0", slots=64) as b: b.serve()
    rank r (no GPU context at all): cl = BrokerClient("mcalf0", slot=r); logL, derived = cl.lnlhood_pc(theta)

`serve()` runs the loop inside the library when the evaluator is an `als_fitter` (`mcalf_broker_serve`: one C thread, no
interpreter in the round), and with SEVERAL contexts of the same problem (`LikelihoodBroker([fit_a, fit_b, ...], ...)`)
requests that arrive while a launch is in flight leave at once on the next free context instead of waiting for it to
end.  `poll()` / `serve(native=False)` are the same protocol in Python, for any object with `loglike_batch`.

`LikelihoodBroker(fit, name, slots, resident_us=500)` goes one step further for spectra of one pixel tile: every rank gets
a WORKGROUP that stays on the GPU and takes the rank's requests straight from its mailbox in the shared block
(`mcalf_broker_serve_resident`; the block is page-locked by the server).  No server thread and no kernel launch is on a
call's path any more -- the server only restarts the launch of those workgroups when a request finds none (they leave,
together, after `resident_us` microseconds without a request from any rank).  Same client class, same callables, same bits.

The client mirrors the solver-facing callables of `als_fitter` (lnlhood_pc / _dy / _mn / lnlhood_worker, _scale_cube_pc
/ _mn) with the same return conventions.  Shared-memory ordering: plain stores and loads of CPython on x86-64 (total
store order): theta before the request counter, logL before the acknowledgement.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from multiprocessing import shared_memory

import numpy as np

_MAGIC = 0x4D43414C46425231          # "MCALFBR1"
_MAGIC_RESIDENT = 0x4D43414C46425232  # "MCALFBR2": mailboxes polled by resident workgroups
_HDR = 8                             # uint64 words: magic, ndim, slots, startind, stop, served batches, served thetas, server pid
_BOX = 576                           # bytes of a mailbox (MCALF_MAILBOX_BYTES): u32 req, quit, ack, state; f64 result; 5 reserved; f64 row[64]
_PENDING = 0x7FF8C0DEC0DE0001        # MCALF_RESULT_PENDING


def _layout(ndim: int, slots: int):
    """Byte offsets of the arrays inside the block: header, request / acknowledge counters (one cache line per slot, so
    that ranks do not share lines), bounds, results, parameter rows."""
    off = {"hdr": 0}
    pos = _HDR * 8
    for name, n in (("req", slots * 8), ("ack", slots * 8), ("lo", ndim), ("hi", ndim), ("logl", slots * 8), ("theta", slots * ndim)):
        pos = (pos + 63) & ~63
        off[name] = pos
        pos += n * 8
    return off, pos


def _layout_resident(ndim: int, slots: int):
    off = {"hdr": 0}
    pos = _HDR * 8
    for name, n in (("lo", ndim * 8), ("hi", ndim * 8), ("box", slots * _BOX)):
        pos = (pos + 63) & ~63
        off[name] = pos
        pos += n
    return off, (pos + 4095) & ~4095


class _ResidentViews:
    def __init__(self, buf, ndim, slots):
        off, _ = _layout_resident(ndim, slots)
        self.off = off
        self.hdr = np.ndarray((_HDR,), dtype=np.uint64, buffer=buf, offset=off["hdr"])
        self.lo = np.ndarray((ndim,), dtype=np.float64, buffer=buf, offset=off["lo"])
        self.hi = np.ndarray((ndim,), dtype=np.float64, buffer=buf, offset=off["hi"])
        self.words = np.ndarray((slots, _BOX // 4), dtype=np.uint32, buffer=buf, offset=off["box"])     # [:, 0] req, 1 quit, 2 ack, 3 state
        self.res_bits = np.ndarray((slots, _BOX // 8), dtype=np.uint64, buffer=buf, offset=off["box"])[:, 2]
        self.res = np.ndarray((slots, _BOX // 8), dtype=np.float64, buffer=buf, offset=off["box"])[:, 2]
        self.rows = np.ndarray((slots, _BOX // 8), dtype=np.float64, buffer=buf, offset=off["box"])[:, 8:]


class _Views:
    def __init__(self, buf, ndim, slots):
        off, _ = _layout(ndim, slots)
        self.hdr = np.ndarray((_HDR,), dtype=np.uint64, buffer=buf, offset=off["hdr"])
        self.req = np.ndarray((slots, 8), dtype=np.uint64, buffer=buf, offset=off["req"])[:, 0]
        self.ack = np.ndarray((slots, 8), dtype=np.uint64, buffer=buf, offset=off["ack"])[:, 0]
        self.lo = np.ndarray((ndim,), dtype=np.float64, buffer=buf, offset=off["lo"])
        self.hi = np.ndarray((ndim,), dtype=np.float64, buffer=buf, offset=off["hi"])
        self.logl = np.ndarray((slots, 8), dtype=np.float64, buffer=buf, offset=off["logl"])[:, 0]
        self.theta = np.ndarray((slots, ndim), dtype=np.float64, buffer=buf, offset=off["theta"])


def _attach(name):
    """Attach to an existing block WITHOUT telling the process's resource tracker: Python < 3.13 registers an attached block
    as if this process had created it, and the tracker then unlinks it when the rank exits -- under the other ranks' feet (the
    block is the server's).  (Un-registering afterwards is no way out either: ranks spawned by the server share ITS tracker.)"""
    from multiprocessing import resource_tracker
    keep = resource_tracker.register
    resource_tracker.register = lambda *a, **k: None
    try:
        return shared_memory.SharedMemory(name=name)
    finally:
        resource_tracker.register = keep


def _process_alive(pid: int) -> bool:
    """Whether process `pid` exists and still runs (a zombie -- dead, not yet reaped by its parent -- does not)."""
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:                                 # (another user's process: it exists)
        return True
    try:
        with open(f"/proc/{pid}/stat") as fh:
            return fh.read().rsplit(")", 1)[1].split()[0] != "Z"
    except (OSError, IndexError):
        return True


class LikelihoodBroker:
    """The serving side.  `fit` is anything with `ndim`, `startind`, `bounds` (as `als_fitter` holds them) and
    `loglike_batch(P) -> logL` -- or a sequence of such evaluators of the SAME problem, one launch in flight on each
    (native loop only); the device contexts live here and nowhere else."""

    def __init__(self, fit, name: str, slots: int = 64, resident_us: int = 0):
        self.resident_us = int(resident_us)
        self.fits = list(fit) if isinstance(fit, (list, tuple)) else [fit]
        fit = self.fits[0]
        if any(int(f.ndim) != int(fit.ndim) for f in self.fits):
            raise ValueError("the broker's evaluators must describe the same problem")
        self.fit, self.name, self.slots = fit, name, int(slots)
        self.ndim = int(fit.ndim)
        if self.resident_us > 0:
            if not self.native or self.ndim > 64:
                raise ValueError("resident evaluators need a library context and at most 64 parameters")
            _, size = _layout_resident(self.ndim, self.slots)
        else:
            _, size = _layout(self.ndim, self.slots)
        self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
        self.shm.buf[:size] = bytes(size)
        self.v = (_ResidentViews if self.resident_us > 0 else _Views)(self.shm.buf, self.ndim, self.slots)
        # min / max of every bounds entry, as _scale_cube_pc takes them (hires_fitter.py:205-206)
        self.v.lo[:] = [np.min(b) for b in fit.bounds]
        self.v.hi[:] = [np.max(b) for b in fit.bounds]
        self.v.hdr[1], self.v.hdr[2], self.v.hdr[3] = self.ndim, self.slots, int(fit.startind)
        self.v.hdr[7] = os.getpid()                             # (the ranks watch the SERVING process -- serve() / serve_native() write their
                                                                # own pid again: the loop may run in a forked child of this one)
        self.v.hdr[0] = _MAGIC_RESIDENT if self.resident_us > 0 else _MAGIC   # last: a client that sees the magic sees a complete header
        self._batch = np.empty((self.slots, self.ndim))

    def poll(self) -> int:
        """Serve every request that is open right now as ONE batch; the number of thetas served."""
        if self.resident_us > 0:
            raise RuntimeError("a broker with resident evaluators is served by serve() / serve_native() only")
        v = self.v
        req = v.req.copy()                                      # (a request that arrives after this copy waits one round)
        open_ = np.nonzero(req != v.ack)[0]
        n = open_.size
        if n == 0:
            return 0
        batch = self._batch[:n]
        np.take(v.theta, open_, axis=0, out=batch)
        out = self.fit.loglike_batch(batch)
        v.logl[open_] = out
        v.ack[open_] = req[open_]                               # acknowledge AFTER the values are in place
        v.hdr[5] += 1
        v.hdr[6] += n
        return n

    @property
    def native(self) -> bool:
        """Whether every evaluator is a device context of the library (the loop can then run inside it)."""
        return all(getattr(f, "_ctx", None) is not None and hasattr(getattr(f, "_lib", None), "mcalf_broker_serve") for f in self.fits)

    def serve_native(self, idle_sleep_after: float = 0.05, max_seconds: float = 0.0) -> None:
        """The loop inside the library (mcalf_broker_serve): returns when the stop flag is raised, or after `max_seconds`."""
        from . import _lib
        self.v.hdr[7] = os.getpid()                             # the process that serves (a server that dies raises no stop flag)
        base = C.addressof(C.c_char.from_buffer(self.shm.buf))
        ok = False
        try:
            if self.resident_us > 0:
                off = self.v.off
                rc = self.fit._lib.mcalf_broker_serve_resident(self.fit._ctx, base + off["box"], self.slots, base + off["hdr"] + 4 * 8,
                                                               self.resident_us, base + off["hdr"] + 5 * 8, float(max_seconds))
            else:
                off, _ = _layout(self.ndim, self.slots)
                d = _lib.mcalf_broker_t(slots=self.slots, ndim=self.ndim, req=base + off["req"], ack=base + off["ack"], counter_stride=8,
                                        theta=base + off["theta"], theta_stride=self.ndim, logl=base + off["logl"], logl_stride=8,
                                        stop=base + off["hdr"] + 4 * 8, stats=base + off["hdr"] + 5 * 8, idle_sleep_after_s=idle_sleep_after)
                ctxs = (C.c_void_p * len(self.fits))(*[f._ctx for f in self.fits])
                rc = self.fit._lib.mcalf_broker_serve(ctxs, len(self.fits), C.byref(d), float(max_seconds))
            _lib.check(rc, self.fit._ctx)
            ok = True
        finally:
            # a loop that ended with an ERROR serves nobody any more: the ranks that are waiting must get an exception, not
            # spin for ever (a loop that ended because of max_seconds may be entered again: no flag)
            if not ok and self.v is not None:
                self.v.hdr[4] = 1

    def serve(self, idle_sleep_after: float = 0.05, stop_when=None, native=None) -> None:
        """Serve until a client (or `stop()`) raises the stop flag, or `stop_when()` says so.  Spins while requests keep
        coming; after `idle_sleep_after` seconds without one it yields the core between polls.  `native` (default: when
        every evaluator is a library context and no `stop_when` is given) runs the loop inside the library."""
        if native is None:
            native = self.resident_us > 0 or (self.native and stop_when is None)
        if native:
            return self.serve_native(idle_sleep_after)
        self.v.hdr[7] = os.getpid()
        last = time.perf_counter()
        ok = False
        try:
            while not self.v.hdr[4]:
                if self.poll():
                    last = time.perf_counter()
                elif time.perf_counter() - last > idle_sleep_after:
                    if stop_when is not None and stop_when():
                        break
                    time.sleep(0.0002)
            ok = True
        finally:
            if not ok and self.v is not None:                   # (the evaluator raised: see serve_native)
                self.v.hdr[4] = 1

    @property
    def stats(self):
        if self.resident_us > 0:                                # (nobody counts on a call's path: the acknowledged request numbers do)
            return {"batches": int(self.v.hdr[5]), "thetas": int(self.v.words[:, 2].astype(np.int64).sum())}
        return {"batches": int(self.v.hdr[5]), "thetas": int(self.v.hdr[6])}

    def stop(self):
        self.v.hdr[4] = 1

    def close(self):
        if self.v is not None:
            self.v.hdr[4] = 1                                   # a rank that is still waiting gets an exception, not a hang
        self.v = None
        try:
            self.shm.close()
            self.shm.unlink()
        except (FileNotFoundError, BufferError):
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BrokerClient:
    """A solver rank's side: the likelihood callables of `als_fitter`, served by the broker `name` through slot `slot`
    (one slot per rank; MPI rank numbers do).  Holds no device context."""

    def __init__(self, name: str, slot: int, timeout: float = 60.0, call_timeout: float | None = 120.0, server_pid: int | None = None):
        """`timeout`: seconds to wait for the broker's block to appear.  `call_timeout`: seconds a single likelihood call may
        wait for its answer before it raises (None: for ever); independently of it a call raises as soon as the server
        PROCESS is gone (its pid is in the header, read afresh at every check) -- a server that dies raises no stop flag.
        `server_pid=0` turns the pid check off and leaves `call_timeout` alone to decide: for ranks in another PID namespace
        than the server (containers), where the header's pid names nothing or somebody else."""
        self.call_timeout = call_timeout
        self._watch_pid = server_pid
        t0 = time.time()
        while True:
            try:
                self.shm = _attach(name)
                hdr = np.ndarray((_HDR,), dtype=np.uint64, buffer=self.shm.buf)
                if hdr[0] in (_MAGIC, _MAGIC_RESIDENT):
                    break
                del hdr                                     # (the view exports the buffer: close() would raise BufferError)
                self.shm.close()
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout:
                raise RuntimeError(f"no likelihood broker named {name!r}")
            time.sleep(0.01)
        self.ndim, self.slots, self.startind = int(hdr[1]), int(hdr[2]), int(hdr[3])
        self.server_pid = int(hdr[7]) if server_pid is None else int(server_pid)
        if not (0 <= slot < self.slots):
            raise ValueError(f"slot {slot} outside the broker's {self.slots} slots")
        self.slot = int(slot)
        self.resident = int(hdr[0]) == _MAGIC_RESIDENT
        if self.resident:
            self.v = _ResidentViews(self.shm.buf, self.ndim, self.slots)
            self._row = self.v.rows[self.slot][: self.ndim]
            self._req = self.v.words[self.slot]                  # [0] is the request number
            self._bits = self.v.res_bits[self.slot: self.slot + 1]
            self._pending = np.uint64(_PENDING)
        else:
            self.v = _Views(self.shm.buf, self.ndim, self.slots)
            self._row = self.v.theta[self.slot]
        self.bounds = np.stack([self.v.lo, self.v.hi], axis=1).copy()
        self._ptp = self.bounds[:, 1] - self.bounds[:, 0]

    # -- the likelihood (hires_fitter.py:250-328) ---------------------------------------------------------------
    def lnlhood_worker(self, p):
        int(p[self.startind])                              # raises where the reference's int() does (:428)
        v, s = self.v, self.slot
        if self.resident:
            # the rank's mailbox, polled by its workgroup on the GPU: result <- pending, row, then the request number
            bits, pend = self._bits, self._pending
            bits[0] = pend
            self._row[:] = p                                # (raises for a wrong length)
            words = self._req
            seq = words[0] + np.uint32(1)
            words[0] = seq
            n = 0
            # answered: the slot no longer holds the pending pattern, or -- should the answer itself be that pattern -- the
            # workgroup's acknowledgement (written behind the result) carries this request's number
            while bits[0] == pend and words[2] != seq:
                n += 1
                if not (n & 0x3FFF):
                    self._still_served(n)
            return float(v.res[s])
        self._row[:] = p                                    # (raises for a wrong length)
        seq = v.req[s] + np.uint64(1)
        v.req[s] = seq                                      # theta first, then the request
        ack = v.ack
        n = 0
        while ack[s] != seq:
            n += 1
            if not (n & 0x3FFF):
                self._still_served(n)
        return float(v.logl[s])

    def _still_served(self, n):
        """Called every 16384 looks of a waiting call: raise when nobody is going to answer -- the stop flag is up, the
        server process is gone (no flag is raised by a process that dies), or the call has waited `call_timeout` seconds."""
        if self.v.hdr[4]:
            raise RuntimeError("the likelihood broker has stopped")
        if n == 0x4000:
            self._t_wait = time.monotonic()
            return
        if self._watch_pid is None:
            self.server_pid = int(self.v.hdr[7])                # (the serving loop writes its own pid when it starts: read it afresh)
        if self.server_pid and not _process_alive(self.server_pid):
            raise RuntimeError(f"the likelihood broker's process ({self.server_pid}) is gone")
        if self.call_timeout is not None and time.monotonic() - self._t_wait > self.call_timeout:
            raise RuntimeError(f"the likelihood broker did not answer within {self.call_timeout} s")

    def lnlhood_pc(self, p):
        return self.lnlhood_worker(p), []

    def lnlhood_dy(self, p):
        return self.lnlhood_worker(p)

    def lnlhood_mn(self, p, ndim, nparam):
        return self.lnlhood_worker(np.array([p[x] for x in range(self.ndim)]))

    # -- the prior transforms (hires_fitter.py:202-216) ---------------------------------------------------------
    def _scale_cube_pc(self, cube):
        theta = np.array(cube, dtype=float, copy=True)     # separately rounded multiply and add, as numpy evaluates :206
        theta *= self._ptp
        theta += self.bounds[:, 0]
        theta[self.startind] = int(theta[self.startind])   # :207-208
        return theta

    def _scale_cube_mn(self, cube, ndim, nparam):
        for k in range(ndim):                               # :211-216, in place (possibly a C double pointer)
            cube[k] = cube[k] * self._ptp[k] + self.bounds[k, 0]
        return cube

    def stop_broker(self):
        self.v.hdr[4] = 1

    def close(self):
        self.v = self._row = None
        try:
            self.shm.close()
        except BufferError:
            pass
