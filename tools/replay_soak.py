#!/usr/bin/env python3
"""Two solvers on the same scene stepping the DFSPH step side by side, one whose passes replay the hit record of the
variant-3 density pass and one on neighbour lists only, through a developing flow; every --every steps every
particle of the two is compared bit for bit, and the iteration counts of the two correction loops step by step."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--jitter", type=float, default=0.0, help="noise on the start positions, in spacings")
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--every", type=int, default=100)
a = ap.parse_args()
sc = scenes.dam_break(*a.dims)
if a.jitter > 0:
    sc = scenes.jittered(sc, seed=5, pos_jitter=a.jitter, vel_scale=0.5, fluid_only=True)
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
solvers = {}
for mode in ("replay", "lists"):
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.set_neighbour_lists(True, replay=mode == "replay")
    s.setup_dfsph()
    solvers[mode] = s
names = ("pos", "vel", "v_star", "density", "DF_factor", "pstiff", "uniqueId")
log, equal = [], True
for done in range(a.every, a.steps + 1, a.every):
    for s in solvers.values():
        s.step_dfsph(a.every)
    bad = [k for k in names if solvers["replay"].get(k).tobytes() != solvers["lists"].get(k).tobytes()]
    s = solvers["replay"]
    fluid = s.get("type") == 0
    vel = s.get("vel")[fluid]
    log.append({"step": done, "differs": bad, "out_of_grid": s.num_undef(),
                "max_speed": float(np.linalg.norm(vel, axis=1).max())})
    print(json.dumps(log[-1]), file=sys.stderr, flush=True)
    if bad:
        equal = False
        break
print(json.dumps({"dims": a.dims, "jitter": a.jitter, "particles": len(sc["pos"]), "steps": log[-1]["step"],
                  "every": a.every, "equal_bit_for_bit": equal, "log": log}))
