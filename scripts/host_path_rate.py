"""PCIe-inclusive rate of fa_assemble_host (host buffers in, host buffers out) next to the device-resident rate, cfg2 / cfg3 at 1 M elements."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from falcon_amd import capi, meshgen
for name, model, mat, kw in (("cfg2", capi.LINEAR_ELASTICITY, capi.HOOKE, {}),
                             ("cfg3", capi.PHASE_FRACTURE, capi.AMOR_PHASE, dict(gc=2.7, l0=0.015))):
    n = 1000
    coords, conn = meshgen.grid_quad4(n, n)
    asm = capi.Assembler(capi.QUAD4_G2, model, mat, coords, conn, young=210e3, poisson=0.3, **kw)
    st = meshgen.smooth_state(coords, asm.ndofpn)
    K = np.zeros(asm.nnz); f = np.zeros(asm.ndof)
    asm.assemble_host(st, K, f, flags=capi.OVERWRITE)
    t = time.perf_counter()
    for _ in range(3): asm.assemble_host(st, K, f, flags=capi.OVERWRITE)
    dt = (time.perf_counter() - t) / 3
    print(f"{name} 1M elements: host path {dt*1e3:.1f} ms/call = {n*n/dt/1e6:.1f} M elements/s, {8*(asm.nnz+2*asm.ndof)/dt/1e9:.1f} GB/s over PCIe")
