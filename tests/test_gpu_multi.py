"""Multi-GPU path with the HIP kernels on ONE device: the blocks of 2 / 3 ranks are assembled one after the other
through the C ABI, their interface contributions exchanged in-process (falcon_amd.distributed.exchange_in_process,
the same pack / apply code the RCCL path runs), and the result compared with the single-domain HIP assembly and
with the oracle: f_int halo-sum, K reduced to owner-complete rows + ghost block, K left additive."""
import numpy as np
import pytest

from falcon_amd import capi, meshgen
from falcon_amd.distributed import (InterfacePlanK, SlabPartition, apply_K, apply_vector, exchange_in_process, pack_K,
                                    pack_vector)
from oracle import pyoracle as po
from tests.conftest import rel_rowmax_err

pytestmark = pytest.mark.gpu
H = 0.05


@pytest.mark.parametrize("case,dims,world", [("quad4", (40, 37), 2), ("quad4", (40, 37), 3), ("hex8", (7, 6, 9), 2), ("hex8", (7, 6, 9), 3)])
def test_blocks_assembled_by_the_hip_path(case, dims, world):
    import torch
    dev = torch.device("cuda", 0)
    shape = capi.QUAD4_G2 if case == "quad4" else capi.HEX8_G2
    nd = len(dims) + 1
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.1, keep_off_diags=True)
    gen = meshgen.grid_quad4 if case == "quad4" else meshgen.grid_hex8
    coords, conn = gen(*dims, *[d * H for d in dims])
    state = meshgen.smooth_state(coords, nd, amp=2e-3, noise=0.0, l0=0.2, phi_noise=0.0)
    # single domain: HIP and oracle
    asm = capi.Assembler(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
    rp, ci = asm.pattern()
    Kg = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); fg = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(state).to(dev), Kg, fg, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    cfg = po.make_config(shape, po.PHASE_FRACTURE, po.AMOR, **kw)
    ref = po.assemble(cfg, coords, conn, rp, ci, state)
    assert rel_rowmax_err(rp, Kg.cpu().numpy(), ref["K"]) < 1e-12
    n = asm.ndof
    dense = lambda rp_, ci_, v, off=0, rows=None: None
    # the ranks, one after the other on the same device
    parts, plans, Ks, fs, pats = [], [], [], [], []
    for r in range(world):
        part = SlabPartition(dims, world, r, nd)
        lc, ln = part.local_mesh(meshgen, H)
        a = capi.Assembler(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, lc, ln, **kw)
        b, e = part.global_dof_range()
        K = torch.zeros(a.nnz, dtype=torch.float64, device=dev); f = torch.zeros(a.ndof, dtype=torch.float64, device=dev)
        a.assemble(torch.from_numpy(state[b:e].copy()).to(dev), K, f, flags=capi.OVERWRITE)
        lrp, lci = a.pattern()
        parts.append(part); plans.append(InterfacePlanK(part, lrp, lci)); Ks.append(K); fs.append(f); pats.append((lrp, lci, b, e))
    torch.cuda.synchronize()
    K_add = [k.clone() for k in Ks]
    # exchange: f_int halo sum, K interface rows to the lower rank
    recv_f = exchange_in_process([pack_vector(f, p) for f, p in zip(fs, parts)])
    recv_K = exchange_in_process([pack_K(k, p, pl) for k, p, pl in zip(Ks, parts, plans)])
    ghosts = []
    for r in range(world):
        apply_vector(fs[r], parts[r], recv_f[r])
        ghosts.append(apply_K(Ks[r], parts[r], plans[r], recv_K[r]))
    torch.cuda.synchronize()
    fscale = np.abs(ref["f"]).max()
    Kd = np.zeros((n, n)); Ko = np.zeros((n, n)); Ka = np.zeros((n, n))
    for i in range(n):
        Kd[i, ci[rp[i]:rp[i + 1]]] = Kg.cpu().numpy()[rp[i]:rp[i + 1]]
    for r in range(world):
        lrp, lci, b, e = pats[r]
        ld = parts[r].layer_dofs
        assert np.abs(fs[r].cpu().numpy() - ref["f"][b:e]).max() < 1e-12 * fscale          # every rank: assembled values
        Kr = Ks[r].cpu().numpy(); Kad = K_add[r].cpu().numpy()
        for i in range(e - b):
            Ka[b + i, b + lci[lrp[i]:lrp[i + 1]]] += Kad[lrp[i]:lrp[i + 1]]
            if r == 0 or i >= ld:
                Ko[b + i, b + lci[lrp[i]:lrp[i + 1]]] += Kr[lrp[i]:lrp[i + 1]]
        if ghosts[r] is not None:
            Ko[plans[r].ghost_rows, plans[r].ghost_cols] = ghosts[r].cpu().numpy()
    tol = 1e-12 * np.abs(Kd).max(axis=1, keepdims=True)
    assert np.all(np.abs(Ka - Kd) <= tol)          # additive subdomain matrices sum to the global matrix
    assert np.all(np.abs(Ko - Kd) <= tol)          # owner-complete rows + ghost block
