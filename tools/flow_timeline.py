"""The bench scene stepped for a long while: every `--every` steps the out-of-grid count, the largest speed and
density, and the time per step.  Usage: python tools/flow_timeline.py [--scene dam16m] [--steps 2000] [--every 250]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from fluidsolver_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dam16m")
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--every", type=int, default=250)
ap.add_argument("--pressureK", type=float, default=None)
ap.add_argument("--dt", type=float, default=None)
a = ap.parse_args()
sc = bench.build_scene(a.scene)
if a.pressureK: sc["params"]["pressureK"] = a.pressureK
if a.dt: sc["params"]["dt"] = a.dt
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
s = capi.Solver(p, len(sc["pos"]))
s.upload(sc)
rows = []
done = 0
while done < a.steps:
    s.synchronize(); t0 = time.perf_counter()
    s.step(a.every); s.synchronize()
    ms = (time.perf_counter() - t0) / a.every * 1e3
    done += a.every
    fluid = s.get("type") == 0
    v = np.linalg.norm(s.get("vel")[fluid], axis=1)
    s.sort_particles(); s.compute_pressure()
    rho = s.get("density")[s.get("type") == 0]
    pos = s.get("pos")[s.get("type") == 0]
    rows.append(dict(step=done, ms_per_step=ms, out_of_grid=s.num_undef(), max_speed=float(v.max()),
                     p99_speed=float(np.percentile(v, 99)), max_density=float(rho.max()),
                     p99_density=float(np.percentile(rho, 99)), y_max=float(pos[:, 1].max()), x_max=float(pos[:, 0].max())))
    print(json.dumps(rows[-1]), flush=True)
