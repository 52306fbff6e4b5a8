#!/usr/bin/env python3
"""Per-kernel times of the WCSPH step for several kernel variants on one scene (same state for all), plus the
largest relative difference of density and force between each variant and the first one."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dam16m"); ap.add_argument("--jitter", type=float, default=0.0)
ap.add_argument("--variants", type=int, nargs="+", default=[2, 3]); ap.add_argument("--spinup", type=int, default=100)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
sc = bench.build_scene(a.scene, a.jitter)
p = capi.make_params(0, **sc["params"])
s = capi.Solver(p, len(sc["pos"])); s.upload(sc)
s.set_variant(2)  # the experiment flags of variant 3 (FSA_FAST_DBG) must not disturb the spin-up
s.step(a.spinup); s.sort_particles(); s.synchronize()
ref = None
out = {}
for v in a.variants:
    s.set_variant(v)
    s.compute_pressure(); s.compute_force(); s.synchronize()
    s.timing_enable(True); s.timing_reset()
    for _ in range(a.steps):
        s.compute_pressure(); s.compute_force()
    t = s.timing_read(); s.timing_enable(False)
    d, f = s.get("density"), s.get("force")
    fluid = s.get("type") == 0
    res = {k: round(ms / c, 4) for k, (c, ms) in t.items()}
    if ref is None:
        ref = (d, f)
    else:
        res["density_rel_diff"] = float(np.abs(d - ref[0]).max() / np.abs(ref[0]).max())
        fs = np.abs(ref[1][fluid]).max()
        res["force_rel_diff"] = float(np.abs(f[fluid] - ref[1][fluid]).max() / fs)
    if v == 3:
        hc = s.get("hitCount")[fluid]
        res["record"] = {"walks": float((hc < 0).mean()), "extras_mean": float(hc[hc >= 0].mean()), "extras_max": int(hc.max())}
    out[v] = res
    print(v, json.dumps(res), flush=True)
