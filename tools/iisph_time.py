#!/usr/bin/env python3
"""Time the IISPH step on a dam break whose tank is made of TYPE_RIGID particles (default 1 M fluid
particles): ms/step and per-kernel averages."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--variant", type=int, default=2)
ap.add_argument("--no-lists", action="store_true", help="search in every pass instead of replaying recorded lists")
a = ap.parse_args()
sc = scenes.dam_break(*a.dims, wall_type=scenes.TYPE_RIGID)
sc["mass"][sc["type"] == scenes.TYPE_RIGID] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
sc["group"][sc["type"] == scenes.TYPE_RIGID] = 1
sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
s = capi.Solver(p, len(sc["pos"]))
s.set_variant(a.variant)
s.set_neighbour_lists(not a.no_lists)
s.upload(sc)
s.setup_iisph()
s.step_iisph(3)
s.synchronize()
s.timing_enable(True); s.timing_reset()
t0 = time.perf_counter()
s.step_iisph(a.steps)
s.synchronize()
wall = time.perf_counter() - t0
kt = s.timing_read()
import numpy as np
fl = s.get("type") == 0
print(json.dumps({"scene": a.dims, "particles": len(sc["pos"]), "fluid": sc["n_fluid"], "variant": a.variant,
                  "neighbour_lists": not a.no_lists, "ms_per_step": wall / a.steps * 1e3,
                  "particle_steps_per_s": sc["n_fluid"] * a.steps / wall,
                  "max_density": float(s.get("density")[fl].max()), "finite": bool(np.isfinite(s.get("pos")).all()),
                  "out_of_grid": s.num_undef(), "max_speed": float(np.abs(s.get("vel")[fl]).max()),
                  "fluid_y_range": [float(s.get("pos")[fl][:, 1].min()), float(s.get("pos")[fl][:, 1].max())],
                  "fluid_x_max": float(s.get("pos")[fl][:, 0].max()),
                  "kernels_ms_per_step": {k: round(ms / a.steps, 4) for k, (c, ms) in kt.items()},
                  "launches_per_step": {k: c / a.steps for k, (c, ms) in kt.items()}}))
