"""Set-up and per-call cost of the constraint elimination at bench scale (Quad4 n x n, PhaseFracture, 3 dofs / node):
Dirichlet rows on two edges (no fill-in, in place) and a periodic left/right pairing (fill-in: output pattern + copy).
usage: python3 scripts/cons_scale.py [n]      prints one line per case"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from falcon_amd import capi, meshgen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
coords, conn = meshgen.grid_quad4(n, n)
nd = 3
asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, young=210.0, poisson=0.3, gc=2.7e-3, l0=0.015)
dev = torch.device("cuda", 0)
st = torch.from_numpy(meshgen.smooth_state(coords, nd, seed=1, amp=1e-3, noise=0.0, l0=0.3)).to(dev)
K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
asm.assemble(st, K, f, flags=capi.OVERWRITE)
torch.cuda.synchronize()
x, y = coords[:, 0], coords[:, 1]
bot = np.nonzero(y == y.min())[0]; top = np.nonzero(y == y.max())[0]
left = np.nonzero(x == x.min())[0]; right = np.nonzero(x == x.max())[0]
left = left[np.argsort(y[left])]; right = right[np.argsort(y[right])]
cases = {"dirichlet": dict(slave=np.concatenate([bot * nd, bot * nd + 1, top * nd + 1])),
         "periodic": dict(slave=np.concatenate([right * nd + t for t in range(nd)]), mdof=np.concatenate([left * nd + t for t in range(nd)]))}
for name, cs in cases.items():
    ns = cs["slave"].size
    t0 = time.perf_counter()
    if "mdof" in cs:
        cons = asm.constraints(cs["slave"], np.full(ns, 1e-3), np.arange(ns + 1), cs["mdof"], np.ones(ns))
    else:
        cons = asm.constraints(cs["slave"], np.full(ns, 1e-3))
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    rhs = (-f).clone()
    Ko = torch.empty(cons.nnz, dtype=torch.float64, device=dev) if cons.fill_in else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 5
    Kw = K.clone()
    for it in range(reps + 1):
        if it == 1:
            ev[0].record()
        if cons.fill_in:
            cons.eliminate_out(K, rhs, out=Ko)
        else:
            cons.eliminate(Kw, rhs)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    print("%s: n=%d ndof=%d nnz=%d slaves=%d fill_in=%d create %.3f s, eliminate %.3f ms%s" % (
        name, n, asm.ndof, asm.nnz, ns, cons.fill_in, t_create, ms,
        " (copy of %.2f GB included: %.2f TB/s)" % (16e-9 * asm.nnz, 16e-9 * asm.nnz / ms) if cons.fill_in else ""), flush=True)
    cons.close()
