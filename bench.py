#!/usr/bin/env python3
"""bench.py -- assembled elements/s (K + f_int) of the MI355X assembly path, BASELINE.json metric.

A "step" is one GET_MATRIX0 pass (tangent stiffness K into the CSR value array + internal force f_int)
over the whole synthetic mesh, inputs resident in HBM.  N=1 workload = BASELINE.json configs[2]
("cfg3": 4000x4000 Q4, coupled u-phi phase-field fracture, AT2, AmorPhase).  N>1: the mesh is cut
into N blocks of element layers along its last axis (strips of the Quad4 meshes, slabs of the Hex8 cube;
falcon_amd/distributed.py), one per GPU; every step ends with the interface exchange over RCCL -- the f_int
halo-sum and, with --k-halo reduce, the interface rows of K sent to their owner (the lower rank).
--scaling strong (default for N > 1: BASELINE.json quotes the metric on ONE 16 M element mesh "at 1/2/4/8 MI355X"): the one
BASELINE mesh is divided among the GPUs; --scaling weak: every GPU gets a full-size block (N x the BASELINE mesh in
total).  value = all ranks' elements / max time over ranks.  Every rank assembles the node tiles that own an interface node
first, on a side stream, and posts their rows to the neighbours while the interior tiles run (falcon_amd.distributed.
OverlappedStep; --no-overlap: assemble, then exchange).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5|cfg5g2|cfg3micro] [--n SIZE]
                    [--material amor|miehe|bourdin] [--keep-off-diags] [--mode ...] [--pattern ...] [--force-only]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def csrc_sha16():
    """hash of the kernel sources (scripts/make_traffic.py stores it with every traffic figure)"""
    import hashlib
    root = os.path.join(ROOT, "falcon_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode()); h.update(open(os.path.join(root, fn), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(nnz_written, ndof, nnode, rank, npe, nel, nip, phase_field):
    """every input read once, every output written once."""
    b = 8 * nnz_written + 8 * ndof + 8 * ndof + 8 * rank * nnode + 4 * npe * nel
    if phase_field:
        b += 16 * nip * nel
    return b


def workload(name, n, capi, material="amor", keep=False, dims=None):
    mat = {"amor": capi.AMOR_PHASE, "miehe": capi.MIEHE_PHASE, "bourdin": capi.BOURDIN_PHASE, "amor-split": capi.AMOR_PHASE}[material]
    mname = {"amor": "AmorPhase", "miehe": "MiehePhase", "bourdin": "BourdinPhase", "amor-split": "AmorPhase"}[material]
    if name == "cfg2":
        n = n or 1000
        return dict(desc=f"cfg2: {n}x{n} Q4 linear-elastic plate (Hooke, plane strain)", shape=capi.QUAD4_G2,
                    model=capi.LINEAR_ELASTICITY, material=capi.HOOKE, dims=(n, n), h=1.0 / n,
                    written_per_pair=4, kw=dict(young=210e3, poisson=0.3))
    if name == "cfg3":
        n = n or 4000
        return dict(desc=f"cfg3: {n}x{n} Q4 single-edge-notched tension, phase-field fracture (u-phi), AT2, {mname}, "
                         f"keepOffDiags={'true' if keep else 'false'}", shape=capi.QUAD4_G2, model=capi.PHASE_FRACTURE,
                    material=mat, dims=(n, n), h=1.0 / n, written_per_pair=9 if keep else 5,
                    kw=dict(young=210e3, poisson=0.3, gc=2.7, l0=0.015, frac_type=capi.AT2, keep_off_diags=keep))
    if name == "cfg3micro":      # the micromorphic variant on the cfg3 mesh (Miehe / Bourdin, penalty 250)
        n = n or 4000
        m2 = {"bourdin": capi.BOURDIN_PHASE, "miehe": capi.MIEHE_PHASE, "amor": capi.MIEHE_PHASE, "amor-split": capi.AMOR_PHASE}[material]
        mn2 = {capi.BOURDIN_PHASE: "BourdinPhase", capi.MIEHE_PHASE: "MiehePhase", capi.AMOR_PHASE: "AmorPhase"}[m2]
        return dict(desc=f"cfg3micro: {n}x{n} Q4, MicroPhaseFracture (u-phi, local damage by Newton), AT2, "
                         f"{mn2}, keepOffDiags={'true' if keep else 'false'}",
                    shape=capi.QUAD4_G2, model=capi.MICRO_PHASE_FRACTURE, material=m2, dims=(n, n), h=1.0 / n,
                    written_per_pair=9 if keep else 5,
                    kw=dict(young=210e3, poisson=0.3, gc=2.7, l0=0.015, frac_type=capi.AT2, keep_off_diags=keep, beta=250.0))
    if name == "cfg4":
        n = n or 200                                    # BASELINE: 8 M elements = 200^3
        d3 = dims or (n, n, n)
        return dict(desc=f"cfg4: {d3[0]}x{d3[1]}x{d3[2]} hex8 phase-field fracture cube, AT2, {mname}, "
                         f"keepOffDiags={'true' if keep else 'false'}", shape=capi.HEX8_G2,
                    model=capi.PHASE_FRACTURE, material=mat, dims=d3, h=1.0 / max(d3),
                    written_per_pair=16 if keep else 10,
                    kw=dict(young=210e3, poisson=0.3, gc=2.7, l0=0.015, frac_type=capi.AT2, keep_off_diags=keep))
    if name in ("cfg5", "cfg5g2"):
        d3 = dims or ((n, n, n) if n else (125, 125, 128))   # BASELINE: 2 M elements = 125 x 125 x 128
        g3 = name == "cfg5"
        return dict(desc=f"cfg5: {d3[0]}x{d3[1]}x{d3[2]} hex20 linear-elastic block (Hooke; the reference has no Mises material), "
                         f"Gauss{'3' if g3 else '2'}^3",
                    shape=capi.HEX20_G3 if g3 else capi.HEX20_G2, model=capi.LINEAR_ELASTICITY, material=capi.HOOKE,
                    dims=d3, h=1.0 / max(d3), written_per_pair=9, kw=dict(young=210e3, poisson=0.3))
    if name in ("quad8", "quad8g2", "tri6", "tri3", "tri3g3"):   # higher-order 2-D shapes, phase-field fracture like cfg3
        n = n or 1000
        shape, gen, sname = {"quad8": (capi.QUAD8_G3, "grid_quad8", "Quad8 Gauss3*Gauss3"),
                             "quad8g2": (capi.QUAD8_G2, "grid_quad8", "Quad8 Gauss2*Gauss2"),
                             "tri6": (capi.TRI6_G3, "grid_tri6", "Triangle6 Gauss3"),
                             "tri3": (capi.TRI3_G1, "grid_tri3", "Triangle3 Gauss1"),
                             "tri3g3": (capi.TRI3_G3, "grid_tri3", "Triangle3 Gauss3")}[name]
        return dict(desc=f"{name}: {n}x{n} {sname}, phase-field fracture (u-phi), AT2, {mname}, "
                         f"keepOffDiags={'true' if keep else 'false'}", shape=shape, model=capi.PHASE_FRACTURE,
                    material=mat, dims=(n, n), h=1.0 / n, written_per_pair=9 if keep else 5, gen=gen,
                    kw=dict(young=210e3, poisson=0.3, gc=2.7, l0=0.015, frac_type=capi.AT2, keep_off_diags=keep))
    if name in ("gd", "gd3"):    # GradientDamage + Damage (the material of the reference's tests/sm03: exponential1
        # softening; Mazars strain in 2-D on the cfg3 mesh, Energy in 3-D on the cfg4 mesh), all four blocks written
        gdkw = dict(young=40e3, poisson=0.2, softening=capi.SOFT_EXPONENTIAL1, kappa_i=0.02, soft_alpha=0.92, soft_beta=30.0,
                    length_scale=1.0e-6)
        if name == "gd":
            n = n or 4000
            return dict(desc=f"gd: {n}x{n} Q4, GradientDamage (u-eqv), Damage: exponential1 softening, Mazars strain, plane stress",
                        shape=capi.QUAD4_G2, model=capi.GRADIENT_DAMAGE, material=capi.DAMAGE, dims=(n, n), h=1.0 / n,
                        written_per_pair=9, kw=dict(state=capi.PLANE_STRESS, eqv_strain=capi.EQV_MAZARS, **gdkw))
        d3 = dims or ((n or 128),) * 3
        return dict(desc=f"gd3: {d3[0]}x{d3[1]}x{d3[2]} hex8, GradientDamage (u-eqv), Damage: exponential1 softening, Energy strain",
                    shape=capi.HEX8_G2, model=capi.GRADIENT_DAMAGE, material=capi.DAMAGE, dims=d3, h=1.0 / max(d3),
                    written_per_pair=16, kw=dict(eqv_strain=capi.EQV_ENERGY, **gdkw))
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(wl, capi, meshgen, target_elems):
    """The CPU restatement (oracle) of the reference loop, 1 core, on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    rank = len(wl["dims"])
    if rank == 2:
        m = max(8, int(round(target_elems ** 0.5)))
        coords, conn = meshgen.grid_quad4(m, m, m * wl["h"], m * wl["h"])
        sample = f"{m}x{m} Q4 sub-grid of the same mesh (same h, material, state field), one K+f_int pass"
    else:
        m = max(4, int(round(target_elems ** (1.0 / 3.0))))
        gen = meshgen.grid_hex20 if wl["shape"] in (7, 8) else meshgen.grid_hex8
        coords, conn = gen(m, m, m, m * wl["h"], m * wl["h"], m * wl["h"])
        sample = f"{m}^3 sub-grid of the same mesh, one K+f_int pass"
    kw = dict(wl["kw"])
    cfg = po.make_config(wl["shape"], wl["model"], wl["material"], young=kw["young"], poisson=kw["poisson"],
                         gc=kw.get("gc", 2.7), l0=kw.get("l0", 0.015), frac_type=kw.get("frac_type", 1),
                         keep_off_diags=int(kw.get("keep_off_diags", False)), beta=kw.get("beta", 6500.0),
                         **{k: kw[k] for k in ("state", "softening", "kappa_i", "soft_alpha", "soft_beta", "length_scale", "eqv_strain") if k in kw})
    nd = po.ndof_per_node(wl["model"], rank)
    rp, ci = po.pattern(coords.shape[0], conn, nd, rank)
    st = meshgen.smooth_state(coords, nd)
    t = time.perf_counter()
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    dt = time.perf_counter() - t
    one = dict(value=conn.shape[0] / dt, unit="elements/s", cores=1, kind="port", sample=sample, seconds=round(dt, 2))
    # the same loop on all the host cores this job may use: element-block strips in two colours (strips of one colour
    # share no node), one thread per strip -- SURVEY.md 8d asks for it next to the 1-core figure
    import concurrent.futures as cf
    cores = min(16, len(os.sched_getaffinity(0)))
    nel = conn.shape[0]
    per_row = m if rank == 2 else m * m                 # elements per mesh row (2D) / slab (3D)
    nstrip = min(2 * cores, m // 2)
    if cores < 2 or nstrip < 4:
        return one, None
    rows = [m * k // nstrip for k in range(nstrip + 1)]
    K = np.zeros(rp[-1]); f = np.zeros(ref["f"].size)
    hn = np.zeros_like(ref["hist_new"]) if ref["hist_new"] is not None else None
    def run(k):
        po.assemble(cfg, coords, conn, rp, ci, st, K=K, f=f, hist_new=hn, el_range=(rows[k] * per_row, rows[k + 1] * per_row))
    t = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        for colour in (0, 1):
            list(ex.map(run, range(colour, nstrip, 2)))
    dtm = time.perf_counter() - t
    assert np.abs(K - ref["K"]).max() <= 1e-12 * np.abs(ref["K"]).max()
    allc = dict(value=nel / dtm, unit="elements/s", cores=cores, kind="port",
                sample=sample + f"; {nstrip} element strips in two colours on {cores} threads", seconds=round(dtm, 2))
    return one, allc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--n", type=int, default=0, help="elements per side (default: the BASELINE size of the workload: cfg2 1000^2, "
                                                     "cfg3 4000^2, cfg4 200^3, cfg5 125x125x128)")
    ap.add_argument("--dims", default="", help="3-D workloads: elements per direction as nx,ny,nz (overrides --n)")
    ap.add_argument("--mode", default="overwrite", choices=["overwrite", "accumulate"])
    ap.add_argument("--pattern", default="full", choices=["full", "written"])
    ap.add_argument("--material", default="amor", choices=["amor", "miehe", "bourdin", "amor-split"])
    ap.add_argument("--keep-off-diags", action="store_true")
    ap.add_argument("--force-only", action="store_true", help="GET_INT_VECTOR pass instead of GET_MATRIX0")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1: the one mesh divided (default) | one full-size block per GPU")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: assemble the whole block, then exchange (no interface-first split)")
    ap.add_argument("--k-halo", default="additive", choices=["additive", "reduce"],
                    help="N > 1: leave the subdomain matrices additive (Jive's distributed representation) | send the interface rows of K to the lower rank")
    ap.add_argument("--no-f-int-only", action="store_true", help="skip the extra GET_INT_VECTOR measurement (profiler passes: one kernel per run)")
    ap.add_argument("--cpu-elems", type=int, default=6250000, help="elements in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"

    import torch
    import torch.distributed as dist
    from falcon_amd import capi, meshgen
    from falcon_amd.distributed import InterfacePlanK, OverlappedStep, SlabPartition, halo_sum, halo_sum_K

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    dims = tuple(int(x) for x in args.dims.split(",")) if args.dims else None
    wl = workload(args.workload, args.n, capi, args.material, args.keep_off_diags, dims)
    rk = len(wl["dims"])
    t_setup = time.perf_counter()
    part = None
    ndofpn = rk + (1 if wl["model"] else 0)
    if world > 1:
        if wl["shape"] not in (capi.QUAD4_G2, capi.HEX8_G2):
            raise SystemExit("multi-GPU bench: block partitions exist for the structured Quad4 and Hex8 meshes")
        gdims = tuple(wl["dims"])
        if args.scaling == "weak":      # every rank gets a full-size block, stacked along the last axis
            gdims = gdims[:-1] + (gdims[-1] * world,)
        part = SlabPartition(gdims, world, rank, ndofpn)
        coords, conn = part.local_mesh(meshgen, wl["h"])
    elif rk == 2:
        nx, ny = wl["dims"]
        coords, conn = getattr(meshgen, wl.get("gen", "grid_quad4"))(nx, ny, nx * wl["h"], ny * wl["h"])
    else:
        nx, ny, nz = wl["dims"]
        gen = meshgen.grid_hex20 if wl["shape"] in (capi.HEX20_G3, capi.HEX20_G2) else meshgen.grid_hex8
        coords, conn = gen(nx, ny, nz, nx * wl["h"], ny * wl["h"], nz * wl["h"])
    kind = capi.PATTERN_FULL if args.pattern == "full" else capi.PATTERN_WRITTEN
    asm = capi.Assembler(wl["shape"], wl["model"], wl["material"], coords, conn, pattern_kind=kind,
                         device=local_rank, **wl["kw"])
    # N > 1: the state is a function of the coordinates only (no per-rank noise), so interface nodes carry the same
    # values on both ranks that hold them
    state = torch.from_numpy(meshgen.smooth_state(coords, asm.ndofpn, seed=12345) if world == 1 else
                             meshgen.smooth_state(coords, asm.ndofpn, noise=0.0, phi_noise=0.0)).to(dev)
    del coords
    plan = None
    if part is not None and args.k_halo == "reduce" and not args.force_only:
        mask = None
        if kind == capi.PATTERN_WRITTEN and wl["model"]:
            keep = bool(wl["kw"].get("keep_off_diags"))
            mask = np.ones((ndofpn, ndofpn), bool); mask[:rk, rk] = keep; mask[rk, :rk] = keep
        plan = InterfacePlanK(part, *asm.pattern(), mask=mask)
    want_K = not args.force_only
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev) if want_K else None
    f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    flags = capi.OVERWRITE if args.mode == "overwrite" else capi.ACCUMULATE
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    stepper = None
    if part is not None and not args.no_overlap:
        stepper = OverlappedStep(asm, part, plan, capi=capi, device=dev)

    def step():
        if stepper is not None:
            stepper.step(state, K, f, flags)
            return
        asm.assemble(state, K, f, flags=flags)
        if part is not None:
            halo_sum(f, part)
            if plan is not None:
                halo_sum_K(K, part, plan)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        if stepper is not None:
            stepper.step(state, K, f, flags)          # interface tiles + exchange on the side stream, interior here
            ev[i][1].record()                         # (N > 1: the events bracket the whole overlapped step)
            continue
        asm.assemble(state, K, f, flags=flags)
        ev[i][1].record()
        if part is not None:
            halo_sum(f, part)
            if plan is not None:
                halo_sum_K(K, part, plan)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
        nel_total = torch.tensor([asm.nel], dtype=torch.float64, device=dev)
        dist.all_reduce(nel_total)
        nel_total = nel_total.item()
    else:
        nel_total = asm.nel

    # "also report the f_int-only call (it is the majority of calls under BFGS)" -- measured after, and outside,
    # the timed region of the headline metric, on the same context and state; the model must answer GET_INT_VECTOR
    fonly = None
    if world == 1 and want_K and wl["model"] != capi.LINEAR_ELASTICITY and not args.no_f_int_only:
        for _ in range(3):
            asm.assemble(state, None, f, flags=flags)
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev2:
            a.record(); asm.assemble(state, None, f, flags=flags); b.record()
        torch.cuda.synchronize()
        fonly = float(np.mean([a.elapsed_time(b) for a, b in ev2]))

    if rank == 0:
        pf = wl["model"] != capi.LINEAR_ELASTICITY
        pairs = asm.nnz // (asm.ndofpn ** 2 if kind == capi.PATTERN_FULL else wl["written_per_pair"])
        nnz_written = pairs * wl["written_per_pair"] if want_K else 0
        ab = algorithmic_bytes(nnz_written, asm.ndof, asm.nnode, asm.rank, asm.npe, asm.nel, asm.nip, pf)
        achieved = ab / (kern_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the PMC passes of scripts/r03_profiles.sh (FETCH_SIZE x 2 + WRITE_SIZE; they
        # cannot run inside this process).  The figure is reported only if it was measured with THIS library -- the hash of
        # the kernel sources is kept next to it -- and always with the run it comes from; otherwise null and the reason.
        traffic, traffic_source = None, "no PMC measurement for this configuration"
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            key = f"{args.workload}:{asm.nel}:{args.mode}:{args.pattern}:{'f' if args.force_only else 'K'}"
            if args.material != "amor" or args.keep_off_diags:
                key += f":{args.material}:{int(args.keep_off_diags)}"
            td = json.load(open(tf))
            det = td.get("_detail", {}).get(key, {})
            if key in td:
                if det.get("csrc_sha16") == csrc_sha16():
                    traffic, traffic_source = td[key], det.get("source")
                else:
                    traffic_source = f"stale: {det.get('source')} was measured with another build of the kernels"
        out = {
            "metric": "assembled elements/sec (K + f_int)" if want_K else "assembled elements/sec (f_int only)",
            "value": nel_total * args.steps / elapsed, "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["desc"], "elements_per_gpu": asm.nel, "ndof_per_gpu": asm.ndof,
                       "nnz_per_gpu": asm.nnz, "mode": args.mode, "pattern": args.pattern,
                       "partition": (f"{world} blocks of element layers along the last axis ({args.scaling} scaling), f_int interface "
                                     f"halo-sum over RCCL, K {'interface rows reduced to the lower rank' if plan is not None else 'additive'}, "
                                     f"{'interface tiles first, exchange overlapped with the interior' if stepper is not None else 'assemble then exchange'}")
                                    if world > 1 else "single GPU",
                       "setup_seconds": round(t_setup, 2)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": ab, "bytes_per_element": ab / asm.nel},
        }
        if fonly is not None:
            abf = algorithmic_bytes(0, asm.ndof, asm.nnode, asm.rank, asm.npe, asm.nel, asm.nip, pf)
            out["f_int_only"] = {"value": asm.nel / (fonly * 1e-3), "unit": "elements/s", "kernel_ms": fonly,
                                 "roofline_frac": abf / (fonly * 1e-3) / 8e12, "algorithmic_bytes_per_launch": abf}
        if world == 1 and args.cpu_elems > 0:
            out["cpu_baseline"], allc = cpu_baseline(wl, capi, meshgen, args.cpu_elems)
            if allc is not None:
                out["cpu_baseline_all_cores"] = allc
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
