#!/usr/bin/env python3
"""Two solvers on the same scene, one on kernel variant 3 (one search per step: recorded hits, blocks cut into
segments) and one on variant 2 (both passes search, blocks that cannot be tiled walk the cells), stepped side by side
through a developing flow and compared every --every steps for every particle, bit for bit; also reports how many
fluid particles of the variant-3 solver have no record (walk the cells) at each comparison."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--jitter", type=float, default=0.0, help="noise on the start positions, in spacings")
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--every", type=int, default=250)
a = ap.parse_args()
sc = scenes.dam_break(*a.dims)
if a.jitter > 0:
    sc = scenes.jittered(sc, seed=5, pos_jitter=a.jitter, vel_scale=0.5, fluid_only=True)
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
solvers = {}
for v in (3, 2):
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.set_variant(v)
    solvers[v] = s
names = ("pos", "vel", "force", "density", "pressure", "uniqueId")
log, equal = [], True
for done in range(a.every, a.steps + 1, a.every):
    for s in solvers.values():
        s.step(a.every)
    bad = [k for k in names if solvers[3].get(k).tobytes() != solvers[2].get(k).tobytes()]
    s3 = solvers[3]
    s3.sort_particles(); s3.compute_pressure()   # (the staged calls leave the state as it is: the record of the current positions)
    fluid = s3.get("type") == 0
    hc = s3.get("hitCount")
    vel = s3.get("vel")[fluid]
    log.append({"step": done, "differs": bad, "fluid_without_record": int(((hc < 0) & fluid).sum()),
                "out_of_grid": s3.num_undef(), "max_speed": float(np.linalg.norm(vel, axis=1).max())})
    print(json.dumps(log[-1]), file=sys.stderr, flush=True)
    if bad:
        equal = False
        break
print(json.dumps({"dims": a.dims, "jitter": a.jitter, "particles": len(sc["pos"]), "steps": log[-1]["step"], "every": a.every,
                  "equal_bit_for_bit": equal, "log": log}))
