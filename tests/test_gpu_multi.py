"""Multi-GPU path with the HIP kernels on ONE device: the blocks of 2 / 3 ranks are assembled one after the other
through the C ABI, their interface contributions exchanged in-process (falcon_amd.distributed.exchange_in_process,
the same pack / apply code the RCCL path runs), and the result compared with the single-domain HIP assembly and
with the oracle: f_int halo-sum, K reduced to owner-complete rows + ghost block, K left additive."""
import numpy as np
import pytest

from falcon_amd import capi, meshgen
from falcon_amd.distributed import (InterfacePlanK, OverlappedStep, SlabPartition, apply_K, apply_vector,
                                    exchange_in_process, pack_K, pack_vector)
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


@pytest.mark.parametrize("case,dims", [("quad4", (150, 130)), ("hex8", (14, 12, 13)), ("quad4-force", (150, 130)), ("hex8-force", (14, 12, 13)),
                                       ("quad4-persistent", (150, 130))])
def test_interface_first_order_gives_the_same_result(case, dims, monkeypatch):
    """a pass issued as two launches -- the tiles that own an interface node first, on a side stream,
    the interior on the caller's stream (fa_set_interface, FA_PART_INTERFACE / FA_PART_INTERIOR) -- writes the same K, f_int
    and history as the one-launch pass, in overwrite and in accumulate mode; and OverlappedStep, with the middle rank of three
    and its neighbours played in-process, ends with the same f_int / K as the sequential pack -> exchange -> apply."""
    import torch
    dev = torch.device("cuda", 0)
    force = case.endswith("-force")
    if case.endswith("-persistent"):                   # the interior launch as 16 persistent workgroups walking the tile list
        monkeypatch.setenv("FA_PAIR_PERSIST", "16")
    shape = capi.QUAD4_G2 if case.startswith("quad4") else capi.HEX8_G2
    nd = len(dims) + 1
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.1, keep_off_diags=False)
    gen = meshgen.grid_quad4 if case.startswith("quad4") else meshgen.grid_hex8
    world, r = 3, 1
    part = SlabPartition(dims, world, r, nd)
    lc, ln = part.local_mesh(meshgen, H)
    lc = meshgen.distort(lc, ln, 0.1, 4)
    st = torch.from_numpy(meshgen.smooth_state(lc, nd, amp=2e-3, noise=1e-5, l0=0.2)).to(dev)
    a = capi.Assembler(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, lc, ln, **kw)
    rp, ci = a.pattern()
    z = lambda n: torch.zeros(n, dtype=torch.float64, device=dev)
    K1 = None if force else z(a.nnz); f1 = z(a.ndof)
    a.assemble(st, K1, f1, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    h1 = a.get_history(1).copy()
    nloc = part.nlayers + 1
    a.set_interface([(0, part.layer_nodes), ((nloc - 1) * part.layer_nodes, nloc * part.layer_nodes)])
    for mode in (capi.OVERWRITE, capi.ACCUMULATE):
        nanfill = lambda n: torch.full((n,), float("nan") if mode == capi.OVERWRITE else 0.0, dtype=torch.float64, device=dev)
        K2 = None if force else nanfill(a.nnz); f2 = nanfill(a.ndof)
        a.set_history(1, np.zeros(a.nip_total))
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        a.assemble(st, K2, f2, flags=mode | capi.PART_INTERFACE, stream=side.cuda_stream)
        a.assemble(st, K2, f2, flags=mode | capi.PART_INTERIOR)
        torch.cuda.synchronize()
        fs = f1.abs().max().item()
        assert (f2 - f1).abs().max().item() < 1e-13 * fs               # every entry written, by exactly one of the two launches
        if not force:
            assert rel_rowmax_err(rp, K2.cpu().numpy(), K1.cpu().numpy()) < 1e-13
        assert np.array_equal(a.get_history(1), h1)
    # the overlapped step against the sequential exchange, neighbours played in-process (they send what this rank would: a
    # function of the interface entries, so the check is exact in structure and value)
    plan = None if force else InterfacePlanK(part, rp, ci)
    fake_f = lambda buf, k: buf * (0.5 + k)
    def transport(sends_f, sends_K):
        recv_f = {peer: fake_f(buf, peer) for peer, buf in sends_f.items()}
        recv_K = {}
        if plan is not None:
            g = torch.Generator(device="cpu").manual_seed(3)
            recv_K[part.rank + 1] = torch.rand(plan.recv_len, dtype=torch.float64, generator=g).to(dev)
        return recv_f, recv_K
    Ks = None if force else z(a.nnz); fsq = z(a.ndof)
    a.assemble(st, Ks, fsq, flags=capi.OVERWRITE)
    rf, rK = transport(pack_vector(fsq, part), {} if force else pack_K(Ks, part, plan))
    apply_vector(fsq, part, rf)
    ghost_s = None if force else apply_K(Ks, part, plan, rK)
    Ko = None if force else z(a.nnz); fo = z(a.ndof)
    stepper = OverlappedStep(a, part, plan, capi=capi, device=dev, transport=transport)
    ghost_o = stepper.step(st, Ko, fo, capi.OVERWRITE)
    torch.cuda.synchronize()
    assert (fo - fsq).abs().max().item() < 1e-13 * fsq.abs().max().item()
    if not force:
        assert rel_rowmax_err(rp, Ko.cpu().numpy(), Ks.cpu().numpy()) < 1e-13
        assert torch.equal(ghost_o, ghost_s)
