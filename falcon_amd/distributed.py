"""Multi-GPU decomposition of the assembly: element-block slabs, one process per GPU.

Like Jive under MPI (build/main.cpp:127-130: MPInputModule + PartitionModule), every rank owns a block of
elements and assembles its own subdomain matrix and internal-force vector; DOFs on the interface between two
blocks receive contributions from both ranks.  The structured meshes of the BASELINE configs are numbered
lexicographically (x fastest), so a block of element layers along the last axis -- strips of a 2-D Quad4 mesh
(cfg3), slabs of a 3-D Hex8 mesh (cfg4) -- touches at most two neighbours, and an interface is one contiguous
node layer at the start / end of the local numbering:

  * f_int: the interface entries are summed with the neighbouring rank (`halo_sum`): one send/recv pair per
    interface (96 KB for cfg3, 1.3 MB for cfg4), after which both ranks hold the assembled values;
  * K: either left additive (`"additive"`: every rank keeps its subdomain matrix, the representation Jive's
    distributed matrices use), or reduced to owner-complete rows (`halo_sum_K`): the interface rows belong to the
    LOWER rank; the upper rank sends its partial rows, the owner adds the entries whose column is an
    interface dof into its own rows and keeps the others (columns in the node layer above, which the owner does not
    hold) as a small ghost block with global (row, column) ids.

The exchange is written as pack -> transport -> apply, so that the same code runs over torch.distributed (RCCL on
GPUs, gloo on CPU: `halo_sum`, `halo_sum_K`) and in a single process that plays all ranks one after the other (the
one-GPU parity test: `exchange_in_process`).  xGMI is point-to-point: the two interfaces of a rank use different
links and run concurrently (one batch of isend / irecv).
"""
import numpy as np


class SlabPartition:
    """`dims` = elements per direction of the global structured mesh; the last direction is cut into `world`
    contiguous blocks of element layers.  Rank r holds element layers [a0, a1) and node layers [a0, a1]."""

    def __init__(self, dims, world, rank, ndofpn):
        self.dims, self.world, self.rank, self.ndofpn = tuple(int(d) for d in dims), world, rank, ndofpn
        nlay = self.dims[-1]
        base, rem = divmod(nlay, world)
        sizes = [base + (1 if r < rem else 0) for r in range(world)]
        if min(sizes) < 1:
            raise ValueError("more ranks than element layers")
        self.a0 = sum(sizes[:rank]); self.a1 = self.a0 + sizes[rank]
        self.nlayers = sizes[rank]
        self.layer_nodes = int(np.prod([d + 1 for d in self.dims[:-1]]))
        self.layer_dofs = self.layer_nodes * ndofpn
        nloc = self.nlayers + 1
        self.ndof_local = nloc * self.layer_dofs
        # local dof slices of the interface node layers (node-major dofs => contiguous)
        self.lower = slice(0, self.layer_dofs) if rank > 0 else None                                  # shared with rank-1
        self.upper = slice((nloc - 1) * self.layer_dofs, nloc * self.layer_dofs) if rank < world - 1 else None

    # compatibility with the round-1 name
    @property
    def row_dofs(self):
        return self.layer_dofs

    @property
    def local_dims(self):
        return self.dims[:-1] + (self.nlayers,)

    def local_mesh(self, meshgen, h):
        """Coordinates / connectivity of this rank's block (local node numbering, element size h)."""
        d = self.local_dims
        if len(d) == 2:
            coords, conn = meshgen.grid_quad4(d[0], d[1], d[0] * h, d[1] * h)
        else:
            coords, conn = meshgen.grid_hex8(d[0], d[1], d[2], d[0] * h, d[1] * h, d[2] * h)
        coords[:, -1] += self.a0 * h
        return coords, conn

    def global_dof_range(self):
        """[begin, end) of this block's dofs in the global node-major numbering."""
        return self.a0 * self.layer_dofs, (self.a1 + 1) * self.layer_dofs


def StripPartition(nx, ny_total, world, rank, ndofpn):
    """2-D strips (round-1 interface): SlabPartition of an nx x ny_total Quad4 mesh."""
    return SlabPartition((nx, ny_total), world, rank, ndofpn)


# ---------------------------------------------------------------------------------------------------------
# f_int: interface entries summed with both neighbours
# ---------------------------------------------------------------------------------------------------------
def pack_vector(f, part):
    """{destination rank: the interface entries of the local vector it needs}."""
    out = {}
    cp = (lambda v: v.clone()) if hasattr(f, "clone") else (lambda v: v.copy())   # copies: both sides add to their own slice afterwards
    if part.lower is not None:
        out[part.rank - 1] = cp(f[part.lower])
    if part.upper is not None:
        out[part.rank + 1] = cp(f[part.upper])
    return out


def apply_vector(f, part, received):
    """Add what the neighbours sent ({source rank: entries}) to the interface entries of f, in place."""
    if part.lower is not None:
        f[part.lower] += received[part.rank - 1]
    if part.upper is not None:
        f[part.upper] += received[part.rank + 1]
    return f


def _exchange(sends, like, dist, group):
    """One batch of isend / irecv with the neighbours; returns {source rank: tensor}."""
    import torch
    ops, recvs = [], {}
    for peer, buf in sends.items():
        recvs[peer] = torch.empty(like[peer], dtype=buf.dtype, device=buf.device)
        ops += [dist.P2POp(dist.isend, buf, peer, group=group), dist.P2POp(dist.irecv, recvs[peer], peer, group=group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return recvs


def halo_sum(f, part, dist=None, group=None):
    """Sum interface entries of the local vector f with both neighbours, in place.  f: torch tensor (CPU with
    gloo, GPU with nccl == RCCL).  After the call both ranks hold the assembled interface values."""
    import torch.distributed as tdist
    dist = dist or tdist
    sends = pack_vector(f, part)
    recvs = _exchange(sends, {p: b.numel() for p, b in sends.items()}, dist, group)   # both sides of an interface have layer_dofs entries
    return apply_vector(f, part, recvs)


# ---------------------------------------------------------------------------------------------------------
# K: interface rows reduced to the lower rank (owner-complete rows + ghost block)
# ---------------------------------------------------------------------------------------------------------
def canonical_pattern(nnode, conn, nd, mask=None):
    """Row offsets / column indices of the canonical blocked pattern (dof = node * nd + type, column-sorted rows):
    every dof-type pair (or those of `mask`[nd][nd]) of every node pair sharing an element -- the layout
    fa_create builds (FA_PATTERN_FULL / _WRITTEN).  Host-side numpy, used for the neighbour's interface rows."""
    conn = np.asarray(conn, dtype=np.int64)
    npe = conn.shape[1]
    a = np.repeat(conn, npe, axis=1).ravel(); b = np.tile(conn, (1, npe)).ravel()
    pairs = np.unique(a * nnode + b)
    pa, pb = pairs // nnode, pairs % nnode
    mask = np.ones((nd, nd), bool) if mask is None else np.asarray(mask, bool)
    deg = np.bincount(pa, minlength=nnode)
    cnt = mask.sum(1)                                                  # entries per neighbour in a row of type i
    rp = np.zeros(nnode * nd + 1, dtype=np.int64)
    rp[1:] = np.cumsum((deg[:, None] * cnt[None, :]).ravel())
    ci = np.empty(rp[-1], dtype=np.int64)
    start = np.concatenate([[0], np.cumsum(deg)])[:-1]
    k_in_row = np.arange(pa.size) - start[pa]                          # position of neighbour b in the list of a
    for i in range(nd):
        cols = np.nonzero(mask[i])[0]
        base = rp[pa * nd + i] + k_in_row * cnt[i]
        for q, j in enumerate(cols):
            ci[base + q] = pb * nd + j
    return rp, ci.astype(np.int32)


class InterfacePlanK:
    """Index maps of the K interface reduction of one rank (built once per mesh, host side).

    send (rank > 0):        the value slice of the first node layer's rows, K[0 : send_len], goes to rank - 1
    recv (rank < world-1):  rank + 1's slice of `recv_len` values; entry k is added to own K[own_idx[k]] when
                            in_own[k], else it is ghost entry ghost_of[k] with global ids (ghost_rows, ghost_cols)
    """

    def __init__(self, part, rp, ci, mask=None):
        nd, ld, ln = part.ndofpn, part.layer_dofs, part.layer_nodes
        rp = np.asarray(rp); ci = np.asarray(ci)
        self.send_len = int(rp[ld]) if part.lower is not None else 0
        self.recv_len = 0
        self.own_idx = self.in_own = self.ghost_rows = self.ghost_cols = None
        if part.upper is None:
            return
        # the neighbour's first node layer only touches its first element layer: the pattern of a one-layer block
        from . import meshgen
        d1 = part.dims[:-1] + (1,)
        if len(d1) == 2:
            _, conn1 = meshgen.grid_quad4(d1[0], 1)
        else:
            _, conn1 = meshgen.grid_hex8(d1[0], d1[1], 1)
        rp1, ci1 = canonical_pattern(2 * ln, conn1, nd, mask)
        self.recv_len = int(rp1[ld])
        rows = np.repeat(np.arange(ld), np.diff(rp1[:ld + 1]))        # interface dof (0-based inside the layer)
        cols = ci1[:self.recv_len].astype(np.int64)
        self.in_own = cols < ld                                         # column is an interface dof, too
        top = part.upper.start                                          # first local dof of the owner's interface layer
        own_rows = top + rows[self.in_own]; own_cols = top + cols[self.in_own]
        # position of (own_row, own_col) in the owner's (column-sorted) rows: one sorted key per entry of the owner's
        # interface rows, searched for all received entries at once
        r0, r1 = top, top + ld
        seg0, seg1 = int(rp[r0]), int(rp[r1])
        nloc = np.int64(part.ndof_local)
        own_keys = np.repeat(np.arange(r0, r1, dtype=np.int64), np.diff(rp[r0:r1 + 1])) * nloc + ci[seg0:seg1].astype(np.int64)
        q = own_rows.astype(np.int64) * nloc + own_cols
        pos = np.searchsorted(own_keys, q)
        if pos.size and (pos.max() >= own_keys.size or np.any(own_keys[np.minimum(pos, own_keys.size - 1)] != q)):
            raise ValueError("the owner's pattern lacks an interface entry (column-sorted canonical rows are required)")
        idx = seg0 + pos
        self.own_idx = idx
        gb, _ = part.global_dof_range()
        g_top = gb + top                                                 # global dof of the interface layer's first dof
        self.ghost_rows = g_top + rows[~self.in_own]
        self.ghost_cols = g_top + cols[~self.in_own]                     # layer above: global id = interface + layer_dofs offset

    def device_maps(self, device):
        """(positions of the received entries that go into own rows, positions of the ghost entries, their places in the own K)
        as int64 tensors on `device`, made once and kept."""
        import torch
        cached = getattr(self, "_dev", None)
        if cached is None or cached[0] != device:
            src_own = torch.from_numpy(np.nonzero(self.in_own)[0]).to(device)
            src_ghost = torch.from_numpy(np.nonzero(~self.in_own)[0]).to(device)
            own_idx = torch.from_numpy(np.ascontiguousarray(self.own_idx, dtype=np.int64)).to(device)
            self._dev = cached = (device, src_own, src_ghost, own_idx)
        return cached[1:]


def pack_K(K, part, plan):
    return {part.rank - 1: K[:plan.send_len].contiguous() if hasattr(K, "contiguous") else K[:plan.send_len].copy()} \
        if part.lower is not None else {}


def apply_K(K, part, plan, received):
    """Owner side: add the neighbour's interface x interface entries into the own rows, return the ghost values
    (interface rows x columns of the node layer above; ids in plan.ghost_rows / plan.ghost_cols)."""
    if part.upper is None:
        return None
    R = received[part.rank + 1]
    if hasattr(R, "cpu"):
        # the index maps live on the device from their first use on (round 2 uploaded 10-17 M entries per step for cfg4:
        # more bytes over PCIe than the xGMI message they unpack)
        src_own, src_ghost, own_idx = plan.device_maps(R.device)
        K.index_add_(0, own_idx, R.index_select(0, src_own))
        return R.index_select(0, src_ghost)
    np.add.at(K, plan.own_idx, R[plan.in_own])
    return R[~plan.in_own]


def halo_sum_K(K, part, plan, dist=None, group=None):
    """Interface rows of K reduced to the lower rank, in place; returns the owner's ghost values (or None).
    The upper rank's copies of those rows keep their partial sums; they are no longer the rows of record."""
    import torch
    import torch.distributed as tdist
    dist = dist or tdist
    sends = pack_K(K, part, plan)
    ops, recv = [], {}
    if part.lower is not None:
        ops.append(dist.P2POp(dist.isend, sends[part.rank - 1], part.rank - 1, group=group))
    if part.upper is not None:
        recv[part.rank + 1] = torch.empty(plan.recv_len, dtype=K.dtype, device=K.device)
        ops.append(dist.P2POp(dist.irecv, recv[part.rank + 1], part.rank + 1, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return apply_K(K, part, plan, recv)


class OverlappedStep:
    """One assembly pass of a rank with the interface exchange hidden behind the interior: the node tiles that own
    an interface node are assembled first on a side stream (fa_set_interface, FA_PART_INTERFACE), their f_int entries and -- with a
    plan -- K rows are packed and posted to the neighbours from that stream, and the remaining tiles (FA_PART_INTERIOR) run on
    the caller's stream meanwhile; the received contributions are applied on the side stream, which the caller's stream joins
    at the end.  `transport(sends_f, sends_K)` -> (recv_f, recv_K) defaults to torch.distributed point-to-point operations
    (RCCL); the one-GPU test passes a function that plays the neighbours in-process."""

    def __init__(self, asm, part, plan=None, *, capi, device=None, dist=None, group=None, transport=None):
        import torch
        self.asm, self.part, self.plan, self.capi = asm, part, plan, capi
        self.dist, self.group, self.transport = dist, group, transport
        ln, nloc = part.layer_nodes, part.nlayers + 1
        ranges = []
        if part.lower is not None:
            ranges.append((0, ln))
        if part.upper is not None:
            ranges.append(((nloc - 1) * ln, nloc * ln))
        asm.set_interface(ranges)
        self.side = torch.cuda.Stream(device=device)
        self.ghost = None

    def _p2p(self, sends_f, sends_K):
        import torch
        import torch.distributed as tdist
        dist = self.dist or tdist
        part, plan = self.part, self.plan
        ops, recv_f, recv_K = [], {}, {}
        for peer, buf in sends_f.items():
            recv_f[peer] = torch.empty_like(buf)
            ops += [dist.P2POp(dist.isend, buf, peer, group=self.group), dist.P2POp(dist.irecv, recv_f[peer], peer, group=self.group)]
        if plan is not None:
            for peer, buf in sends_K.items():
                ops.append(dist.P2POp(dist.isend, buf, peer, group=self.group))
            if part.upper is not None:
                recv_K[part.rank + 1] = torch.empty(plan.recv_len, dtype=torch.float64, device=next(iter(sends_f.values())).device)
                ops.append(dist.P2POp(dist.irecv, recv_K[part.rank + 1], part.rank + 1, group=self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return recv_f, recv_K, reqs

    def step(self, state, K, f, flags, **kw):
        import torch
        capi, part, plan = self.capi, self.part, self.plan
        main = torch.cuda.current_stream(self.side.device)
        self.side.wait_stream(main)                                   # the inputs are the caller's
        with torch.cuda.stream(self.side):
            self.asm.assemble(state, K, f, flags=flags | capi.PART_INTERFACE, stream=self.side.cuda_stream, **kw)
            sends_f = pack_vector(f, part)
            sends_K = pack_K(K, part, plan) if (plan is not None and K is not None) else {}
            if self.transport is not None:
                recv_f, recv_K, reqs = (*self.transport(sends_f, sends_K), [])
            else:
                recv_f, recv_K, reqs = self._p2p(sends_f, sends_K)
        self.asm.assemble(state, K, f, flags=flags | capi.PART_INTERIOR, **kw)      # meanwhile, on the caller's stream
        with torch.cuda.stream(self.side):
            for r in reqs:
                r.wait()                                              # RCCL: the side stream waits, not the host
            apply_vector(f, part, recv_f)
            self.ghost = apply_K(K, part, plan, recv_K) if (plan is not None and K is not None) else None
        main.wait_stream(self.side)
        return self.ghost


def exchange_in_process(packs):
    """All ranks played by one process: packs[r] = {destination: buffer}  ->  received[r] = {source: buffer}."""
    world = len(packs)
    received = [dict() for _ in range(world)]
    for src, out in enumerate(packs):
        for dst, buf in out.items():
            received[dst][src] = buf
    return received
