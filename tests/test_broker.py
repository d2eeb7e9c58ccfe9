"""CPU: the likelihood broker's protocol (mc-alf_amd/broker.py) with a stand-in evaluator -- several client PROCESSES,
one server; every client gets the value of ITS theta, requests are served in batches, the callables keep the
reference's return conventions, a stopped broker does not leave a client spinning."""
import multiprocessing as mp
import os
import time

import numpy as np
import pytest

from mcalf_amd import broker


class FakeFit:
    """Stands in for als_fitter: ndim 5, ncomp slot 1, a likelihood that is a plain function of the row."""
    ndim, startind = 5, 1
    bounds = [[8.0, 9.0], [1, 3], [12.0, 14.5], [2.99, 3.01], [10.0, 40.0]]

    def __init__(self):
        self.batches = []

    def loglike_batch(self, P):
        self.batches.append(len(P))
        time.sleep(0.002)                                   # (a launch takes a while: requests pile up meanwhile)
        return -0.5 * (P ** 2).sum(axis=1) + P[:, 0]


def _client(name, slot, n, q):
    cl = broker.BrokerClient(name, slot)
    rng = np.random.default_rng(100 + slot)
    ok = True
    for _ in range(n):
        p = rng.random(5) * 10 + 1
        want = -0.5 * (p ** 2).sum() + p[0]
        got, derived = cl.lnlhood_pc(p)
        ok = ok and got == want and derived == []
        ok = ok and cl.lnlhood_dy(p) == want and cl.lnlhood_mn(list(p), 5, 5) == want
    cube = rng.random(5)
    th = cl._scale_cube_pc(cube)
    q.put((slot, ok, th.tolist(), cube.tolist()))
    cl.close()


def test_broker_serves_many_client_processes_in_batches():
    name = f"mcalf_test_{os.getpid()}"
    fit = FakeFit()
    with broker.LikelihoodBroker(fit, name, slots=8) as b:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_client, args=(name, s, 40, q)) for s in range(6)]
        for p in procs:
            p.start()
        t0 = time.time()
        done = []
        while len(done) < 6 and time.time() - t0 < 120:
            b.poll()
            while not q.empty():
                done.append(q.get())
        for p in procs:
            p.join(timeout=30)
        assert len(done) == 6 and all(ok for _, ok, _, _ in done)
        assert b.stats["thetas"] == 6 * 40 * 3 and b.stats["batches"] < b.stats["thetas"]      # requests were batched
        assert max(fit.batches) > 1
        lo = np.array([min(x) for x in FakeFit.bounds]); hi = np.array([max(x) for x in FakeFit.bounds])
        for _, _, th, cube in done:                          # the prior transform of hires_fitter.py:202-209, client side
            want = np.array(cube) * (hi - lo) + lo
            want[1] = int(want[1])
            assert np.array_equal(np.array(th), want)


def test_broker_client_errors():
    name = f"mcalf_test_err_{os.getpid()}"
    with pytest.raises(RuntimeError, match="no likelihood broker"):
        broker.BrokerClient(name + "_absent", 0, timeout=0.05)
    with broker.LikelihoodBroker(FakeFit(), name, slots=2) as b:
        with pytest.raises(ValueError):
            broker.BrokerClient(name, 2)
        cl = broker.BrokerClient(name, 1)
        with pytest.raises(ValueError):
            cl.lnlhood_pc(np.array([1.0, float("nan"), 1, 1, 1]))        # int(nan), as the reference's :428
        b.stop()
        with pytest.raises(RuntimeError, match="stopped"):
            cl.lnlhood_pc(np.ones(5))
        cl.close()


def test_clients_of_a_block_with_mailboxes_speak_the_resident_protocol():
    """The ranks' side of the broker with resident workgroups, without a GPU: a thread plays the workgroups (poll `req`, read
    the row, store the result, then `ack`) on a block laid out as LikelihoodBroker(resident_us=...) lays it out; the same
    BrokerClient recognises the block's kind and speaks the mailbox protocol -- result <- pending pattern, row, req + 1."""
    import threading
    from multiprocessing import shared_memory
    name = f"mcalf_test_res_{os.getpid()}"
    ndim, slots = FakeFit.ndim, 3
    _, size = broker._layout_resident(ndim, slots)
    shm = shared_memory.SharedMemory(name=name, create=True, size=size)
    shm.buf[:size] = bytes(size)
    v = broker._ResidentViews(shm.buf, ndim, slots)
    v.lo[:] = [np.min(b) for b in FakeFit.bounds]
    v.hi[:] = [np.max(b) for b in FakeFit.bounds]
    v.hdr[1], v.hdr[2], v.hdr[3] = ndim, slots, FakeFit.startind
    v.hdr[0] = broker._MAGIC_RESIDENT
    served = [0]

    def workgroups():
        while not v.hdr[4]:
            for s in range(slots):
                req = v.words[s, 0]
                if req != v.words[s, 2]:
                    assert v.res_bits[s] == np.uint64(broker._PENDING)     # the rank filled the slot before it asked
                    row = v.rows[s][:ndim].copy()
                    v.res[s] = -0.5 * (row ** 2).sum() + row[0]
                    v.words[s, 2] = req
                    served[0] += 1
            time.sleep(0.0002)

    th = threading.Thread(target=workgroups)
    th.start()
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_client, args=(name, s, 15, q)) for s in range(slots)]
        for p in procs:
            p.start()
        done = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=30)
        assert sorted(d[0] for d in done) == list(range(slots)) and all(d[1] for d in done)
        assert served[0] == slots * 15 * 3
        lo, hi = np.array(v.lo), np.array(v.hi)
        for _, _, thv, cube in done:
            want = np.array(cube) * (hi - lo) + lo
            want[1] = int(want[1])
            assert thv == want.tolist()
    finally:
        v.hdr[4] = 1
        th.join(timeout=10)
        del v
        shm.close()
        shm.unlink()


def test_a_rank_written_in_c_speaks_the_mailbox_protocol_of_the_header(tmp_path):
    """include/mcalf_hip.h's mcalf_mailbox_call(), compiled with gcc into a stand-alone rank, against the block layout of
    broker.py and a thread that plays the resident workgroups: the C ABI's description of the mailbox and the Python one agree."""
    import shutil
    import subprocess
    import threading
    from multiprocessing import shared_memory
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mailbox_client")
    subprocess.run([gcc, "-O2", "-std=gnu11", "-o", exe, os.path.join(root, "tests", "stubs", "mailbox_client.c"), "-lrt"], check=True)
    name = f"mcalf_test_c_{os.getpid()}"
    ndim, slots = 5, 2
    off, size = broker._layout_resident(ndim, slots)
    shm = shared_memory.SharedMemory(name=name, create=True, size=size)
    shm.buf[:size] = bytes(size)
    v = broker._ResidentViews(shm.buf, ndim, slots)
    v.hdr[1], v.hdr[2] = ndim, slots
    v.hdr[0] = broker._MAGIC_RESIDENT

    def workgroups():
        while not v.hdr[4]:
            for s in range(slots):
                req = v.words[s, 0]
                if req != v.words[s, 2]:
                    assert v.res_bits[s] == np.uint64(broker._PENDING)
                    row = v.rows[s][:ndim].copy()
                    v.res[s] = -0.5 * (row ** 2).sum() + row[0]
                    v.words[s, 2] = req
            time.sleep(0.0002)

    th = threading.Thread(target=workgroups)
    th.start()
    try:
        out = subprocess.run([exe, name, str(off["box"] + broker._BOX), str(ndim), "12", "3"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        got = [float(x) for x in out.stdout.split()]
        want = []
        for c in range(12):
            row = np.array([3 + c + k * 0.25 for k in range(ndim)])
            want.append(float(-0.5 * (row ** 2).sum() + row[0]))
        assert got == want and int(v.words[1, 2]) == 12 and int(v.words[0, 2]) == 0      # slot 1 was the C rank's
    finally:
        v.hdr[4] = 1
        th.join(timeout=10)
        del v
        shm.close()
        shm.unlink()


def _dying_server(name, ready):
    b = broker.LikelihoodBroker(FakeFit(), name, slots=2)
    ready.put(os.getpid())
    time.sleep(0.5)
    os._exit(9)                                             # dies without close(): no stop flag is ever raised


def test_a_rank_does_not_spin_for_ever_when_the_server_dies_or_never_answers():
    """A server process that is killed (GPU fault, OOM) raises no stop flag.  The client watches the server's pid (in the
    header) and its own per-call limit: both end the wait with a RuntimeError instead of a rank at 100 % CPU for ever."""
    from multiprocessing import shared_memory
    name = f"mcalf_test_dead_{os.getpid()}"
    ctx = mp.get_context("spawn")
    ready = ctx.Queue()
    srv = ctx.Process(target=_dying_server, args=(name, ready))
    srv.start()
    try:
        pid = ready.get(timeout=60)
        cl = broker.BrokerClient(name, 0, call_timeout=60.0)
        assert cl.server_pid == pid
        t0 = time.time()
        with pytest.raises(RuntimeError, match="is gone"):
            cl.lnlhood_pc(np.ones(5))                       # nobody polls: the call waits until the server has died
        assert time.time() - t0 < 30
        cl.close()
    finally:
        srv.join(timeout=30)
        try:
            leftover = shared_memory.SharedMemory(name=name)
            leftover.close()
            leftover.unlink()
        except FileNotFoundError:
            pass
    # a server that lives but never serves: the per-call limit
    name2 = name + "_mute"
    with broker.LikelihoodBroker(FakeFit(), name2, slots=2):
        cl = broker.BrokerClient(name2, 1, call_timeout=0.2)
        t0 = time.time()
        with pytest.raises(RuntimeError, match="did not answer within 0.2"):
            cl.lnlhood_dy(np.ones(5))
        assert time.time() - t0 < 20
        cl.close()


def test_an_evaluator_that_raises_leaves_the_stop_flag_up():
    """serve() that ends with an exception serves nobody any more: the flag must be up so that waiting ranks raise too."""
    class Broken(FakeFit):
        def loglike_batch(self, P):
            raise ValueError("evaluator failed")
    name = f"mcalf_test_broken_{os.getpid()}"
    with broker.LikelihoodBroker(Broken(), name, slots=2) as b:
        cl = broker.BrokerClient(name, 0)
        cl._row[:] = np.ones(5)
        cl.v.req[0] += np.uint64(1)                         # an open request, posted without waiting for it
        with pytest.raises(ValueError, match="evaluator failed"):
            b.serve(native=False)
        assert int(b.v.hdr[4]) == 1
        with pytest.raises(RuntimeError, match="stopped"):
            cl.lnlhood_pc(np.ones(5))
        cl.close()


def test_attaching_before_the_header_is_complete_retries_instead_of_crashing():
    """The block exists before its magic is written.  A rank that attaches in that window must release its view before it
    closes the mapping (BufferError otherwise) and try again."""
    import threading
    from multiprocessing import shared_memory
    name = f"mcalf_test_early_{os.getpid()}"
    _, size = broker._layout(FakeFit.ndim, 2)
    shm = shared_memory.SharedMemory(name=name, create=True, size=size)
    shm.buf[:size] = bytes(size)
    v = broker._Views(shm.buf, FakeFit.ndim, 2)
    try:
        def finish_header():
            time.sleep(0.15)
            v.hdr[1], v.hdr[2], v.hdr[3] = FakeFit.ndim, 2, FakeFit.startind
            v.hdr[0] = broker._MAGIC
        th = threading.Thread(target=finish_header)
        th.start()
        cl = broker.BrokerClient(name, 1, timeout=10.0)      # (attaches several times before the magic appears)
        th.join()
        assert (cl.ndim, cl.slots, cl.startind) == (FakeFit.ndim, 2, FakeFit.startind)
        cl.close()
    finally:
        del v
        shm.close()
        shm.unlink()


def test_resident_answer_that_equals_the_pending_pattern_is_recognised_by_its_acknowledgement():
    """Completion of a mailbox call is normally read off the result slot (it no longer holds the pending pattern).  An
    answer that IS that bit pattern -- a NaN carrying the payload through the arithmetic -- is recognised by the
    workgroup's acknowledgement instead of stalling the rank."""
    import threading
    from multiprocessing import shared_memory
    name = f"mcalf_test_pend_{os.getpid()}"
    ndim, slots = FakeFit.ndim, 1
    _, size = broker._layout_resident(ndim, slots)
    shm = shared_memory.SharedMemory(name=name, create=True, size=size)
    shm.buf[:size] = bytes(size)
    v = broker._ResidentViews(shm.buf, ndim, slots)
    v.hdr[1], v.hdr[2], v.hdr[3] = ndim, slots, FakeFit.startind
    v.hdr[0] = broker._MAGIC_RESIDENT

    def workgroup():
        while not v.hdr[4]:
            req = v.words[0, 0]
            if req != v.words[0, 2]:
                v.res_bits[0] = np.uint64(broker._PENDING)  # the "answer" is the pending pattern itself
                v.words[0, 2] = req
            time.sleep(0.0002)

    th = threading.Thread(target=workgroup)
    th.start()
    try:
        cl = broker.BrokerClient(name, 0, call_timeout=20.0)
        t0 = time.time()
        got = cl.lnlhood_dy(np.ones(ndim))
        assert np.isnan(got) and time.time() - t0 < 10
        cl.close()
    finally:
        v.hdr[4] = 1
        th.join(timeout=10)
        del v
        shm.close()
        shm.unlink()
