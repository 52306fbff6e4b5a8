#!/usr/bin/env python3
"""The IISPH step on z-slabs, all slabs on this one GPU (fsa_slab_iisph_*): ms/step of the whole team beside the
single solver's, with --check every particle compared bit for bit.  Scene as tools/iisph_time.py: a dam break whose
tank is made of TYPE_RIGID particles."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from fluidsolver_amd import capi, scenes, slabs
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--world", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--variant", type=int, default=2)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
sc = scenes.dam_break(*a.dims, wall_type=scenes.TYPE_RIGID)
sc["mass"][sc["type"] == scenes.TYPE_RIGID] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
sc["group"][sc["type"] == scenes.TYPE_RIGID] = 1
sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
n = len(sc["pos"])
ranks = [slabs.SlabSolver(sc, r, a.world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, a.variant),
                          exchange_buffers=False) for r in range(a.world)]
for ss in ranks:
    ss.backend.enable_iisph(sc["group"], sc["normal"])
team = capi.SlabIisphTeam([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                          max(ss.record_cap for ss in ranks), n)
team.setup()
team.step(3)
team.stats()
w0 = capi.load().fsa_debug_host_waits()
t0 = time.perf_counter()
team.step(a.steps)
steps, lost, sent, rounds = team.stats(a.steps)
wall = time.perf_counter() - t0
waits = capi.load().fsa_debug_host_waits() - w0
res = {"scene": a.dims, "particles": n, "fluid": sc["n_fluid"], "world": a.world, "variant": a.variant,
       "cuts": ranks[0].layout.cuts, "ms_per_step_all_slabs": wall / a.steps * 1e3,
       "ms_per_step_per_slab": wall / a.steps * 1e3 / a.world, "particles_held": [ss.backend.n for ss in ranks],
       "lost": lost, "records_sent_per_step": sent / (a.steps + 3 + 1), "pressure_rounds": rounds,
       "host_waits_per_step": waits / a.steps}
# the single solver beside it
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
s = capi.Solver(p, n)
s.set_variant(a.variant)
s.upload(sc)
s.setup_iisph()
s.step_iisph(3)
s.synchronize()
t0 = time.perf_counter()
s.step_iisph(a.steps)
s.synchronize()
res["single_solver_ms_per_step"] = (time.perf_counter() - t0) / a.steps * 1e3
if a.check:
    from slab_worker import owned_iisph
    from test_slabs_cpu import assert_same_iisph_bits
    ref = {k: s.get(k) for k in ("pos", "vel", "density", "pressure", "uniqueId", "type")}
    assert_same_iisph_bits([owned_iisph(ss) for ss in ranks], ref)
    res["check"] = f"every particle equals the single solver's after {a.steps + 3} steps, bit for bit"
print(json.dumps(res))
team.close()
s.close()
for ss in ranks:
    ss.close()
