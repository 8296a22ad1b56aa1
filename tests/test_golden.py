"""Committed golden vectors (tests/golden/cases.npz, written by tests/golden/make_goldens.py).

CPU: the oracle reproduces them (regression pin of the checker itself; the analytic pins are in test_oracle_kat.py).
GPU: the HIP path, through the C ABI, reproduces them to the north-star tolerance (1e-12 relative, bit-exact pattern)."""
import os

import numpy as np
import pytest

from tests.conftest import rel_rowmax_err

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cases.npz"))
NAMES = [str(n) for n in G["names"]]


def case(name):
    d = {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(name + "/")}
    kw = {str(k): float(v) for k, v in zip(d["kw_keys"], d["kw_vals"])}
    for k in ("state", "frac_type", "keep_off_diags"):
        kw[k] = int(kw[k])
    return d, kw


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_goldens(name):
    from oracle import pyoracle as po
    d, kw = case(name)
    shape, model, material = int(d["shape"]), int(d["model"]), int(d["material"])
    cfg = po.make_config(shape, model, material, **kw)
    rank = d["coords"].shape[1]
    rp, ci = po.pattern(d["coords"].shape[0], d["conn"], po.ndof_per_node(model, rank), rank, po.PATTERN_FULL, po.keep_code(cfg))
    assert np.array_equal(rp, d["row_ptr"]) and np.array_equal(ci, d["col_idx"])
    ext = model in (po.PHASE_FRACTURE_EXT, po.PHASE_FRACTURE_EXTIT)
    mext = model in (po.MICRO_EXT, po.MICRO_EXTIT)
    r = po.assemble(cfg, d["coords"], d["conn"], rp, ci, d["state"], state0=d["state0"] if ext else None,
                    state00=d["state00"] if (ext or mext) else None, dt=0.7, dt0=0.4,
                    hist_pre=d["hist_pre"] if model in po.HISTORY_MODELS else None, want_strain=True)
    assert np.array_equal(r["K"], d["K"]) and np.array_equal(r["f"], d["f"]) and np.array_equal(r["strain"], d["strain"])
    if model in po.HISTORY_MODELS:
        assert np.array_equal(r["hist_new"], d["hist_new"])
    assert r["err_ext"] == float(d["err_ext"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_reproduces_goldens(name):
    import torch
    from falcon_amd import capi
    d, kw = case(name)
    shape, model, material = int(d["shape"]), int(d["model"]), int(d["material"])
    asm = capi.Assembler(shape, model, material, d["coords"], d["conn"], **kw)
    rp, ci = asm.pattern()
    assert np.array_equal(rp, d["row_ptr"]) and np.array_equal(ci, d["col_idx"])          # bit-exact pattern / dof indexing
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    K = torch.full((asm.nnz,), 7.0, dtype=torch.float64, device=dev); f = torch.full((asm.ndof,), -3.0, dtype=torch.float64, device=dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    hist = model == capi.PHASE_FRACTURE or model >= capi.MICRO_PHASE_FRACTURE
    if hist:
        asm.set_history(0, d["hist_pre"])
    ext = model in (capi.PHASE_FRACTURE_EXT, capi.PHASE_FRACTURE_EXTIT)
    mext = model in (capi.MICRO_PHASE_FRACTURE_EXT, capi.MICRO_PHASE_FRACTURE_EXTIT)
    asm.assemble(t(d["state"]), K, f, state0=t(d["state0"]) if ext else None, state00=t(d["state00"]) if (ext or mext) else None,
                 dt=0.7, dt0=0.4, err_ext=err, flags=capi.OVERWRITE | capi.STORE_STRAIN)
    torch.cuda.synchronize()
    assert rel_rowmax_err(rp, K.cpu().numpy(), d["K"]) < 1e-12
    assert np.abs(f.cpu().numpy() - d["f"]).max() < 1e-12 * np.abs(d["f"]).max()
    assert np.abs(asm.get_strain_cache() - d["strain"]).max() < 1e-12 * np.abs(d["strain"]).max()
    if hist:
        assert np.abs(asm.get_history(1) - d["hist_new"]).max() < 1e-12 * np.abs(d["hist_new"]).max()
    if model in (capi.PHASE_FRACTURE_EXTIT, capi.MICRO_PHASE_FRACTURE_EXTIT):
        assert abs(err.item() - float(d["err_ext"])) < 1e-12 * float(d["err_ext"])
