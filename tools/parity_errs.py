#!/usr/bin/env python3
"""Per-element errors of density / pressure / force of every kernel variant against the oracle on the parity scenes
(the numbers the bars in tests/test_gpu_parity.py are set from)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.orc import Oracle
from tests.test_gpu_parity import SCENES, VARIANTS
from tests.util import gpu_solver, oracle_pair, elem_err, force_err, rel_err
orc = Oracle("f32")
for name in sorted(SCENES):
    sc = SCENES[name]()
    for vn, v in sorted(VARIANTS.items()):
        p_o, st = oracle_pair(orc, sc)
        p_g, s = gpu_solver(sc); s.set_variant(v)
        st.sort_particles(p_o); s.sort_particles()
        st.compute_pressure(p_o); s.compute_pressure()
        d_g, d_o, pr_g, pr_o = s.get("density"), st["density"].copy(), s.get("pressure"), st["pressure"].copy()
        rd = st["restDensity"]; K = sc["params"]["pressureK"]
        ok = np.isfinite(pr_o)
        cond = 7 * K * (d_o / rd) ** 7
        s.set("density", d_o); s.set("pressure", pr_o)
        st.compute_force(p_o); s.compute_force()
        f_g, f_o = s.get("force"), st["force"]
        fl = (st["type"] == 0) & np.isfinite(f_o).all(axis=1)
        nf = np.linalg.norm(f_o[fl], axis=1)
        print(f"{name:22s} {vn:8s} dens elem {elem_err(d_g, d_o, 1.0):.2e}  pres elem(floor cond) "
              f"{elem_err(pr_g[ok], pr_o[ok], cond[ok]):.2e} pres elem(floor 1) {elem_err(pr_g[ok], pr_o[ok], 1.0):.2e}  "
              f"force median-floor {force_err(f_g[fl], f_o[fl]):.2e} p99-floor {force_err(f_g[fl], f_o[fl], np.percentile(nf, 99)):.2e} "
              f"old {rel_err(f_g[fl], f_o[fl]):.2e}  |f| median {np.median(nf):.3g} p99 {np.percentile(nf,99):.3g} max {nf.max():.3g}")
        s.close(); st.close()
