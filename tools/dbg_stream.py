import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import mcalf_amd
from mcalf_amd import _lib, workloads
from cases import oracle_synth, problem_from_kwargs
from oracle import c_oracle
kw, _, seed = workloads.config("C", oracle_synth)
co = c_oracle.COracle(problem_from_kwargs(kw))
def runs(idx):
    out=[]; s=None; p=None
    for i in idx:
        if s is None: s=p=i
        elif i==p+1: p=i
        else: out.append((s,p)); s=p=i
    if s is not None: out.append((s,p))
    return out
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2600,4096,2601").split(",")]
for n in sizes:
    P = workloads.draw_P(kw, n, np.random.default_rng(seed + 13))
    dP = torch.from_numpy(P).cuda()
    if n == sizes[-1]:
        want = co.loglike_batch(P)
    else:
        with mcalf_amd.als_fitter(None, **kw) as f0:
            f0.set_chunks(1)
            want = f0.loglike_batch(P)
    with mcalf_amd.als_fitter(None, **kw) as fit:
        out = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        whole = out.cpu().numpy()
        print(n, "device entry vs oracle: max", np.abs(whole - want).max(), "dP == P:", bool(np.array_equal(dP.cpu().numpy(), P)))
        _lib.check(fit._lib.mcalf_loglike_batch_device(fit._ctx, dP.data_ptr(), n, out.data_ptr(), st), fit._ctx)
        torch.cuda.synchronize()
        w2 = out.cpu().numpy()
        print(n, "device entry again vs oracle: max", np.abs(w2 - want).max(), "bad rows", runs(np.nonzero(np.abs(w2 - want) > 1e-4)[0].tolist())[:8])
        fit.set_chunks(1)
        w3 = fit.loglike_batch(P)
        fit.set_chunks(0)
        print(n, "pipelined host entry vs oracle: max", np.abs(w3 - want).max())
        for rep in range(3):
            got = fit.loglike_batch(P)
            bad = np.nonzero(~(got == whole))[0]
            print(n, "rep", rep, "mismatches", bad.size, "stream vs oracle bad rows", int((np.abs(got - want) > 1e-4).sum()), "runs", runs(bad.tolist())[:12], "P addr %x" % P.ctypes.data)
