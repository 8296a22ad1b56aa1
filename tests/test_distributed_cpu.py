"""Multi-rank decomposition (falcon_amd/distributed.py) on CPU: two gloo ranks, each assembling its own strip
with the oracle (the checker stands in for the device call here: no GPU in this test), interface f_int entries
summed with halo_sum.  The result must equal the single-domain assembly, and the subdomain matrices must add up
to the global one (the additive representation Jive uses under MPI, build/main.cpp:127-130)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NX, NY, H, NDOFPN = 7, 9, 0.125, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _global_problem():
    from falcon_amd import meshgen
    coords, conn = meshgen.grid_quad4(NX, NY, NX * H, NY * H)
    state = meshgen.smooth_state(coords, NDOFPN, amp=2e-3, noise=0.0, l0=0.2)
    return coords, conn, state


def _cfg():
    from oracle import pyoracle as po
    return po.make_config(po.QUAD4_G2, po.PHASE_FRACTURE, po.AMOR, young=210.0, poisson=0.3, gc=2.7e-3, l0=0.1,
                          keep_off_diags=1)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from falcon_amd import meshgen
    from falcon_amd.distributed import StripPartition, halo_sum
    from oracle import pyoracle as po
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        part = StripPartition(NX, NY, world, rank, NDOFPN)
        coords, conn = part.local_mesh(meshgen, H)
        _, _, gstate = _global_problem()
        b, e = part.global_dof_range()
        state = gstate[b:e].copy()
        rp, ci = po.pattern(coords.shape[0], conn, NDOFPN, 2)
        r = po.assemble(_cfg(), coords, conn, rp, ci, state)
        f = torch.from_numpy(r["f"].copy())
        halo_sum(f, part)
        dist.barrier()
        np.savez(out % rank, f=f.numpy(), f_local=r["f"], K=r["K"], rp=rp, ci=ci, b=b, e=e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_strips_halo_sum_matches_single_domain(tmp_path, world):
    import torch.multiprocessing as mp
    from oracle import pyoracle as po
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    coords, conn, state = _global_problem()
    rp, ci = po.pattern(coords.shape[0], conn, NDOFPN, 2)
    ref = po.assemble(_cfg(), coords, conn, rp, ci, state)
    n = ref["f"].size
    Kg = np.zeros((n, n)); Ks = np.zeros((n, n)); fs = np.zeros(n)
    for r in range(n):
        Kg[r, ci[rp[r]:rp[r + 1]]] = ref["K"][rp[r]:rp[r + 1]]
    scale = np.abs(ref["f"]).max()
    for rank in range(world):
        d = np.load(out % rank)
        b, e = int(d["b"]), int(d["e"])
        # after the halo sum every rank holds the assembled values on its whole strip, interface rows included
        assert np.abs(d["f"] - ref["f"][b:e]).max() < 1e-13 * scale
        fs[b:e] += d["f_local"]
        lrp, lci = d["rp"], d["ci"]
        for r in range(e - b):
            Ks[b + r, b + lci[lrp[r]:lrp[r + 1]]] += d["K"][lrp[r]:lrp[r + 1]]
    assert np.abs(fs - ref["f"]).max() < 1e-13 * scale
    assert np.abs(Ks - Kg).max() < 1e-13 * np.abs(Kg).max()


def test_strip_partition_covers_all_rows():
    from falcon_amd.distributed import StripPartition
    for world in (1, 2, 3, 8):
        rows = []
        for r in range(world):
            p = StripPartition(5, 4003, world, r, 3)
            rows += list(range(p.y0, p.y1))
            assert (p.lower is None) == (r == 0) and (p.upper is None) == (r == world - 1)
            if p.upper is not None:
                assert p.upper.stop - p.upper.start == 6 * 3 and p.upper.stop == (p.ny + 1) * 18
        assert rows == list(range(4003))
