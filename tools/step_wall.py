#!/usr/bin/env python3
"""Wall-clock time per WCSPH step without per-kernel event records, for scenes of any size (the bench times
with an event pair around every launch, which shows at small particle counts)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[64, 32, 64])
ap.add_argument("--steps", type=int, default=2000)
a = ap.parse_args()
sc = scenes.dam_break(*a.dims)
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
s = capi.Solver(p, len(sc["pos"]))
s.upload(sc)
s.step(100)
s.synchronize()
t0 = time.perf_counter()
s.step(a.steps)
s.synchronize()
wall = time.perf_counter() - t0
print(json.dumps({"dims": a.dims, "particles": len(sc["pos"]), "fluid": sc["n_fluid"], "ms_per_step": wall / a.steps * 1e3, "particle_steps_per_s": sc["n_fluid"] * a.steps / wall}))
