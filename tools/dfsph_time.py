#!/usr/bin/env python3
"""Time the DFSPH step on a dam break (default 1 M particles): ms/step and per-kernel averages."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--no-lists", action="store_true", help="search in every pass instead of replaying recorded lists")
a = ap.parse_args()
sc = scenes.dam_break(*a.dims)
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
s = capi.Solver(p, len(sc["pos"]))
s.set_neighbour_lists(not a.no_lists)
s.upload(sc)
s.setup_dfsph()
s.step_dfsph(5)
s.synchronize()
s.timing_enable(True); s.timing_reset()
t0 = time.perf_counter()
s.step_dfsph(a.steps)
s.synchronize()
wall = time.perf_counter() - t0
kt = s.timing_read()
print(json.dumps({"scene": a.dims, "particles": len(sc["pos"]), "fluid": sc["n_fluid"], "neighbour_lists": not a.no_lists, "ms_per_step": wall / a.steps * 1e3,
                  "particle_steps_per_s": sc["n_fluid"] * a.steps / wall,
                  "kernels_ms_per_step": {k: round(ms / a.steps, 4) for k, (c, ms) in kt.items()},
                  "launches_per_step": {k: c / a.steps for k, (c, ms) in kt.items()}}))
