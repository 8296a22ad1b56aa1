"""Multi-GPU decomposition of the assembly: element-block strips, one process per GPU.

Like Jive under MPI (build/main.cpp:127-130: MPInputModule + PartitionModule), every rank owns a block
of elements and assembles its own subdomain matrix; DOFs on the interface between two blocks receive
contributions from both ranks, and the *vector* quantities (f_int) are made consistent by summing the
interface entries with the neighbouring rank.  Here that is one RCCL send/recv pair per interface
(xGMI is point-to-point, strips touch at most two neighbours), followed by a local add.
"""
import numpy as np


class StripPartition:
    """ny_total element rows split into `world` contiguous strips (lexicographic structured mesh,
    nx elements per row, x fastest).  Rank r holds element rows [y0, y1) and node rows [y0, y1]."""

    def __init__(self, nx, ny_total, world, rank, ndofpn):
        self.nx, self.world, self.rank, self.ndofpn = nx, world, rank, ndofpn
        base, rem = divmod(ny_total, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
        self.y0 = sum(sizes[:rank]); self.y1 = self.y0 + sizes[rank]
        self.ny = sizes[rank]
        row_dofs = (nx + 1) * ndofpn
        nloc_rows = self.ny + 1
        # local dof slices of the interface node rows (node-major dofs => contiguous)
        self.lower = slice(0, row_dofs) if rank > 0 else None                       # shared with rank-1
        self.upper = slice((nloc_rows - 1) * row_dofs, nloc_rows * row_dofs) if rank < world - 1 else None
        self.row_dofs = row_dofs

    def local_mesh(self, meshgen, h):
        """Coordinates / connectivity of this rank's strip (local node numbering, element size h)."""
        coords, conn = meshgen.grid_quad4(self.nx, self.ny, self.nx * h, self.ny * h)
        coords[:, 1] += self.y0 * h
        return coords, conn

    def global_dof_range(self):
        """[begin, end) of this strip's dofs in the global node-major numbering."""
        return self.y0 * self.row_dofs, (self.y1 + 1) * self.row_dofs


def halo_sum(f, part, dist=None, group=None):
    """Sum interface entries of the local vector f with both neighbours, in place.  f: torch tensor (CPU with
    gloo, GPU with nccl == RCCL).  After the call both ranks hold the assembled interface values."""
    import torch
    import torch.distributed as tdist
    dist = dist or tdist
    ops, bufs = [], []
    if part.lower is not None:
        recv = torch.empty_like(f[part.lower])
        ops += [dist.P2POp(dist.isend, f[part.lower].contiguous(), part.rank - 1, group=group),
                dist.P2POp(dist.irecv, recv, part.rank - 1, group=group)]
        bufs.append((part.lower, recv))
    if part.upper is not None:
        recv = torch.empty_like(f[part.upper])
        ops += [dist.P2POp(dist.isend, f[part.upper].contiguous(), part.rank + 1, group=group),
                dist.P2POp(dist.irecv, recv, part.rank + 1, group=group)]
        bufs.append((part.upper, recv))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for sl, recv in bufs:
        f[sl] += recv
    return f
