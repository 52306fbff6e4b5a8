#!/usr/bin/env python3
"""Neighbour-list statistics of the IISPH and DFSPH timing scenes after a few steps: mean and maximum list length,
and the mean over wavefronts of the longest list among their 64 lanes (what a replay pass loops to)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--steps", type=int, default=13)
a = ap.parse_args()
out = {}
for step in ("iisph", "dfsph"):
    if step == "iisph":
        sc = scenes.dam_break(*a.dims, wall_type=scenes.TYPE_RIGID)
        sc["mass"][sc["type"] == scenes.TYPE_RIGID] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
        sc["group"][sc["type"] == scenes.TYPE_RIGID] = 1
        sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
    else:
        sc = scenes.dam_break(*a.dims)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    if step == "iisph":
        s.setup_iisph(); s.step_iisph(a.steps)
    else:
        s.setup_dfsph(); s.step_dfsph(a.steps)
    cnt = s.get("neighbourCount")
    fl = s.get("type") == 0
    n = len(cnt) // 64 * 64
    wave_max = cnt[:n].reshape(-1, 64).max(1)
    out[step] = {"particles": int(len(cnt)), "mean": float(cnt[cnt >= 0].mean()), "mean_fluid": float(cnt[fl & (cnt >= 0)].mean()),
                 "max": int(cnt.max()), "walks_cells": int((cnt < 0).sum()), "fluid_walks_cells": int(((cnt < 0) & fl).sum()),
                 "blocks_with_fluid_walking": int(((cnt[:n] < 0) & fl[:n]).reshape(-1, 256 if n % 256 == 0 else 64).any(1).sum()), "mean_wave_max": float(wave_max.mean()),
                 "density_max": float(s.get("density")[fl].max())}
    s.close()
print(json.dumps(out))
