#!/usr/bin/env python3
"""fp64 validation run of the Eulerian grid solver (default 256^3): the same scene stepped in fp32 and in
fp64, with tracer particles, and how far the fp32 run is from the fp64 one."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolver_amd import capi
ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, nargs=3, default=[256, 256, 256])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--tracers", type=int, default=100000)
a = ap.parse_args()
rng = np.random.default_rng(7)
g32 = capi.GridSolver(a.dim, precision=capi.PRECISION_F32)
g64 = capi.GridSolver(a.dim, precision=capi.PRECISION_F64)
lo = np.array([g64.xmin[0], 0.0, g64.xmin[2]])
pts = lo + rng.random((a.tracers, 3)) * np.array(a.dim) * 0.02
for k in ("u", "w"):
    f = 0.2 * rng.standard_normal(g64.shape(k))
    g32.set(k, f.astype(np.float32))
    g64.set(k, f.astype(np.float32).astype(np.float64))
g32.set_particles(pts.astype(np.float32))
g64.set_particles(pts.astype(np.float32).astype(np.float64))
out = {"dim": a.dim, "tracers": a.tracers, "steps": []}
for s in range(a.steps):
    row = {}
    for name, g in (("f32", g32), ("f64", g64)):
        g.synchronize(); t0 = time.perf_counter()
        g.advect(); g.advect_particles(); g.body_force(); g.make_rhs()
        it, res = g.solve(-1)
        g.update_u(); div = g.div_velocity(); g.synchronize()
        row[name] = {"ms": (time.perf_counter() - t0) * 1e3, "cg_iterations": it, "residual": res, "div_u_sum": div}
    for k in ("u", "v", "w", "p"):
        x, y = g32.get(k).astype(np.float64), g64.get(k)
        row["max_abs_diff_" + k] = float(np.abs(x - y).max())
        row["scale_" + k] = float(np.abs(y).max())
    p32, _ = g32.get_particles(); p64, _ = g64.get_particles()
    row["max_tracer_diff_in_cells"] = float(np.abs(p32 - p64).max() / 0.02)
    out["steps"].append(row)
print(json.dumps(out))
