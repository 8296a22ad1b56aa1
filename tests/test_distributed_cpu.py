"""Multi-rank decomposition (falcon_amd/distributed.py) on CPU: gloo ranks, each assembling its own block of element
layers with the oracle (the checker stands in for the device call here: no GPU in this test; tests/test_gpu_multi.py
runs the same exchange with the HIP path).  Interface f_int entries are summed with halo_sum, interface K rows are
reduced to the lower rank with halo_sum_K.  The result must equal the single-domain assembly; left additive, the
subdomain matrices must add up to the global one (the representation Jive uses under MPI, build/main.cpp:127-130)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 0.125
CASES = {"quad4": ((7, 9), 3), "hex8": ((3, 4, 5), 4)}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _global_problem(case):
    from falcon_amd import meshgen
    dims, nd = CASES[case]
    if len(dims) == 2:
        coords, conn = meshgen.grid_quad4(dims[0], dims[1], dims[0] * H, dims[1] * H)
    else:
        coords, conn = meshgen.grid_hex8(dims[0], dims[1], dims[2], dims[0] * H, dims[1] * H, dims[2] * H)
    state = meshgen.smooth_state(coords, nd, amp=2e-3, noise=0.0, l0=0.2, phi_noise=0.0)
    return coords, conn, state


def _cfg(case):
    from oracle import pyoracle as po
    shape = po.QUAD4_G2 if case == "quad4" else po.HEX8_G2
    return po.make_config(shape, po.PHASE_FRACTURE, po.AMOR, young=210.0, poisson=0.3, gc=2.7e-3, l0=0.1, keep_off_diags=1)


def _worker(rank, world, port, out, case):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from falcon_amd import meshgen
    from falcon_amd.distributed import SlabPartition, InterfacePlanK, halo_sum, halo_sum_K
    from oracle import pyoracle as po
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims, nd = CASES[case]
        part = SlabPartition(dims, world, rank, nd)
        coords, conn = part.local_mesh(meshgen, H)
        _, _, gstate = _global_problem(case)
        b, e = part.global_dof_range()
        state = gstate[b:e].copy()
        rp, ci = po.pattern(coords.shape[0], conn, nd, len(dims))
        r = po.assemble(_cfg(case), coords, conn, rp, ci, state)
        f = torch.from_numpy(r["f"].copy())
        halo_sum(f, part)
        plan = InterfacePlanK(part, rp, ci)
        K = torch.from_numpy(r["K"].copy())
        ghost = halo_sum_K(K, part, plan)
        dist.barrier()
        np.savez(out % rank, f=f.numpy(), f_local=r["f"], K_local=r["K"], K=K.numpy(), rp=rp, ci=ci, b=b, e=e,
                 ghost=np.zeros(0) if ghost is None else ghost.numpy(),
                 grow=np.zeros(0, np.int64) if ghost is None else plan.ghost_rows,
                 gcol=np.zeros(0, np.int64) if ghost is None else plan.ghost_cols, layer_dofs=part.layer_dofs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("quad4", 2), ("quad4", 3), ("hex8", 2)])
def test_blocks_halo_sum_matches_single_domain(tmp_path, case, world):
    import torch.multiprocessing as mp
    from oracle import pyoracle as po
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, case), nprocs=world, join=True)
    coords, conn, state = _global_problem(case)
    dims, nd = CASES[case]
    rp, ci = po.pattern(coords.shape[0], conn, nd, len(dims))
    ref = po.assemble(_cfg(case), coords, conn, rp, ci, state)
    n = ref["f"].size
    Kg = np.zeros((n, n)); Ks = np.zeros((n, n)); Ko = np.zeros((n, n)); fs = np.zeros(n)
    for r in range(n):
        Kg[r, ci[rp[r]:rp[r + 1]]] = ref["K"][rp[r]:rp[r + 1]]
    scale = np.abs(ref["f"]).max()
    for rank in range(world):
        d = np.load(out % rank)
        b, e, ld = int(d["b"]), int(d["e"]), int(d["layer_dofs"])
        # after the halo sum every rank holds the assembled values on its whole block, interface layers included
        assert np.abs(d["f"] - ref["f"][b:e]).max() < 1e-13 * scale
        fs[b:e] += d["f_local"]
        lrp, lci = d["rp"], d["ci"]
        for r in range(e - b):
            Ks[b + r, b + lci[lrp[r]:lrp[r + 1]]] += d["K_local"][lrp[r]:lrp[r + 1]]      # additive representation
            if rank == 0 or r >= ld:                                                       # rows this rank owns
                Ko[b + r, b + lci[lrp[r]:lrp[r + 1]]] += d["K"][lrp[r]:lrp[r + 1]]
        if d["ghost"].size:
            assert np.all(Ko[d["grow"], d["gcol"]] == 0.0)
            Ko[d["grow"], d["gcol"]] = d["ghost"]                                           # columns in the layer above
    assert np.abs(fs - ref["f"]).max() < 1e-13 * scale
    assert np.abs(Ks - Kg).max() < 1e-13 * np.abs(Kg).max()
    assert np.abs(Ko - Kg).max() < 1e-13 * np.abs(Kg).max()                                # owner-complete rows


def test_partition_covers_all_layers():
    from falcon_amd.distributed import SlabPartition, StripPartition
    for world in (1, 2, 3, 8):
        rows = []
        for r in range(world):
            p = StripPartition(5, 4003, world, r, 3)
            rows += list(range(p.a0, p.a1))
            assert (p.lower is None) == (r == 0) and (p.upper is None) == (r == world - 1)
            if p.upper is not None:
                assert p.upper.stop - p.upper.start == 6 * 3 and p.upper.stop == (p.nlayers + 1) * 18
        assert rows == list(range(4003))
    p = SlabPartition((200, 200, 200), 8, 3, 4)
    assert p.nlayers == 25 and p.layer_dofs == 201 * 201 * 4 and p.global_dof_range() == (75 * p.layer_dofs, 101 * p.layer_dofs)
    with pytest.raises(ValueError):
        SlabPartition((4, 2), 3, 0, 2)


def test_canonical_pattern_matches_the_oracles():
    from falcon_amd import meshgen
    from falcon_amd.distributed import canonical_pattern
    from oracle import pyoracle as po
    for gen, nd, rank in ((lambda: meshgen.grid_quad4(4, 3), 3, 2), (lambda: meshgen.grid_hex8(2, 3, 2), 4, 3),
                          (lambda: meshgen.grid_tri3(3, 3), 2, 2)):
        coords, conn = gen()
        rp, ci = canonical_pattern(coords.shape[0], conn, nd)
        rpo, cio = po.pattern(coords.shape[0], conn, nd, rank)
        assert np.array_equal(rp, rpo) and np.array_equal(ci, cio)
