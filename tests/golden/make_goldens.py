"""Writes tests/golden/cases.npz: small seeded cases (inputs and the oracle's outputs) for every model family.

The reference ships no numeric outputs for this path and cannot be built here, so these vectors
are produced by the CPU restatement in oracle/ -- itself pinned by the analytic known-answer tests -- and serve as a
regression pin for the oracle (tests/test_golden.py, CPU) and as a second target for the HIP path (GPU).
Run from the repository root:  python tests/golden/make_goldens.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from falcon_amd import meshgen            # noqa: E402
from oracle import pyoracle as po         # noqa: E402

CASES = [  # name, shape, mesh, model, material, keep, frac, extra config
    ("le_tri3_plane_stress", po.TRI3_G1, lambda: meshgen.grid_tri3(3, 2), po.LINEAR_ELASTICITY, po.HOOKE, 0, po.AT2, dict(state=po.PLANE_STRESS)),
    ("le_hex20_g3", po.HEX20_G3, lambda: meshgen.grid_hex20(2, 1, 1), po.LINEAR_ELASTICITY, po.HOOKE, 0, po.AT2, {}),
    ("pf_quad4_amor_at2", po.QUAD4_G2, lambda: meshgen.grid_quad4(4, 3), po.PHASE_FRACTURE, po.AMOR, 0, po.AT2, {}),
    ("pf_quad4_amor_keep_czm", po.QUAD4_G2, lambda: meshgen.grid_quad4(3, 3), po.PHASE_FRACTURE, po.AMOR, 1, po.CZM_CORNELISSEN,
     dict(sigma_t=3.0, young=3e4, gc=0.12, l0=0.25)),
    ("pf_quad8_bourdin_at1", po.QUAD8_G3, lambda: meshgen.grid_quad8(2, 2), po.PHASE_FRACTURE, po.BOURDIN, 1, po.AT1, {}),
    ("pf_hex8_miehe", po.HEX8_G2, lambda: meshgen.grid_hex8(2, 2, 1), po.PHASE_FRACTURE, po.MIEHE, 1, po.AT2, {}),
    ("ext_quad4_amor", po.QUAD4_G2, lambda: meshgen.grid_quad4(3, 2), po.PHASE_FRACTURE_EXT, po.AMOR, 0, po.AT2, {}),
    ("extit_tri6_miehe", po.TRI6_G3, lambda: meshgen.grid_tri6(2, 2), po.PHASE_FRACTURE_EXTIT, po.MIEHE, 0, po.AT2, {}),
    ("micro_quad4_miehe", po.QUAD4_G2, lambda: meshgen.grid_quad4(3, 3), po.MICRO_PHASE_FRACTURE, po.MIEHE, 1, po.AT2, dict(beta=250.0)),
    ("micro_hex8_bourdin", po.HEX8_G2, lambda: meshgen.grid_hex8(2, 1, 2), po.MICRO_PHASE_FRACTURE, po.BOURDIN, 0, po.AT2, dict(beta=250.0)),
    ("microext_quad4_miehe", po.QUAD4_G2, lambda: meshgen.grid_quad4(3, 3), po.MICRO_EXT, po.MIEHE, 0, po.AT2, {}),
    ("microextit_tri3_bourdin", po.TRI3_G1, lambda: meshgen.grid_tri3(3, 3), po.MICRO_EXTIT, po.BOURDIN, 0, po.AT1, dict(beta=250.0)),
]


def build(case):
    name, shape, mesh, model, material, keep, frac, extra = case
    coords, conn = mesh()
    linear = conn.shape[1] in (3, 4) or shape == po.HEX8_G2
    coords = meshgen.distort(coords, conn, 0.12 if linear else 0.05, seed=7)
    rank = coords.shape[1]
    nd = po.ndof_per_node(model, rank)
    kw = dict(state=po.PLANE_STRAIN, young=210.0, poisson=0.3, frac_type=frac, gc=2.7e-3, l0=0.2, sigma_t=0.0, beta=6500.0,
              keep_off_diags=keep)
    kw.update(extra)
    cfg = po.make_config(shape, model, material, **kw)
    rng = np.random.default_rng(11)
    st = meshgen.smooth_state(coords, nd, seed=3, amp=5e-3, noise=1e-4, l0=0.3)
    st0 = st.copy(); st00 = st.copy()
    if nd > rank:
        st0[rank::nd] += rng.uniform(-0.02, 0.02, coords.shape[0])
        st00[rank::nd] += rng.uniform(-0.02, 0.02, coords.shape[0])
    info = po.shape_info(shape)
    H0 = rng.uniform(0, 2e-3 if model < po.MICRO_PHASE_FRACTURE else 0.4, conn.shape[0] * info[2])   # H0_ or committed local damage
    rp, ci = po.pattern(coords.shape[0], conn, nd, rank, po.PATTERN_FULL, po.keep_code(cfg))
    ext = model in (po.PHASE_FRACTURE_EXT, po.PHASE_FRACTURE_EXTIT)
    mext = model in (po.MICRO_EXT, po.MICRO_EXTIT)                       # state00 carries stateExt_
    r = po.assemble(cfg, coords, conn, rp, ci, st, state0=st0 if ext else None, state00=st00 if (ext or mext) else None, dt=0.7, dt0=0.4,
                    hist_pre=H0 if model in po.HISTORY_MODELS else None, want_strain=True)
    out = dict(shape=shape, model=model, material=material, coords=coords, conn=conn, state=st, state0=st0, state00=st00,
               hist_pre=H0, row_ptr=rp, col_idx=ci, K=r["K"], f=r["f"], strain=r["strain"],
               hist_new=r["hist_new"] if r["hist_new"] is not None else np.zeros(0), err_ext=np.float64(r["err_ext"]),
               kw_keys=np.array(sorted(kw)), kw_vals=np.array([float(kw[k]) for k in sorted(kw)]))
    return {f"{name}/{k}": np.asarray(v) for k, v in out.items()}


if __name__ == "__main__":
    data = {}
    for c in CASES:
        data.update(build(c))
    data["names"] = np.array([c[0] for c in CASES])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cases.npz"), **data)
    print("wrote", len(CASES), "cases")
