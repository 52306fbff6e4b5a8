#!/usr/bin/env python3
"""A step of the library on the exact-arithmetic kernels (variant 1) beside the CPU oracle on the same scene, compared
every --every steps: order of the particles and every bit of pos / vel / density / pressure (and the force of the fluid
particles for the WCSPH step).  --step wcsph (fsa_step), dfsph (fsa_step_dfsph) or iisph (fsa_step_iisph, on a dam
break in a tank of rigid particles)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fluidsolver_amd import capi, scenes
from oracle.orc import Oracle
from util import gpu_solver, oracle_pair
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--every", type=int, default=50)
ap.add_argument("--variant", type=int, default=1)
ap.add_argument("--jitter", type=float, default=0.0, help="noise on the fluid particles' start positions, in spacings (and on their velocities)")
ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--f64", action="store_true", help="the fp64 solver (generic kernels) beside the fp64 oracle")
ap.add_argument("--step", choices=["wcsph", "dfsph", "iisph"], default="wcsph")
ap.add_argument("--scene", choices=["dam", "ddam"], default="dam", help="dam break, or double dam break with obstacles")
a = ap.parse_args()
if a.step == "iisph":
    sc = scenes.dam_break(*a.dims, wall_type=scenes.TYPE_RIGID)
    rigid = sc["type"] == scenes.TYPE_RIGID
    sc["mass"][rigid] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
    sc["group"][rigid] = 1
    sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
else:
    sc = scenes.dam_break(*a.dims) if a.scene == "dam" else scenes.double_dam_break(*a.dims)
if a.jitter > 0:
    sc = scenes.jittered(sc, seed=a.seed, pos_jitter=a.jitter, vel_scale=0.5, fluid_only=True)
orc = Oracle("f64" if a.f64 else "f32")
orc.set_threads(min(16, os.cpu_count() or 1))
p_o, st = oracle_pair(orc, sc)
p_g, s = gpu_solver(sc, 1 if a.f64 else 0)
if not a.f64:
    s.set_variant(a.variant)
real = np.float64 if a.f64 else np.float32
if a.step == "dfsph":
    st.sort_particles(p_o); st.compute_density_alpha(p_o)
    s.setup_dfsph()
elif a.step == "iisph":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_iisph import setup_both
    setup_both(p_o, st, s)
rounds = []
log, t_cpu = [], 0.0
for done in range(a.every, a.steps + 1, a.every):
    t0 = time.perf_counter()
    if a.step == "wcsph":
        st.solve_sph(p_o, a.every)
    elif a.step == "dfsph":
        st.solve_dfsph(p_o, a.every)
    else:
        st.solve_iisph(p_o, a.every, mean_in_double=True)
    t_cpu += time.perf_counter() - t0
    if a.step == "wcsph":
        s.step(a.every)
    elif a.step == "dfsph":
        s.step_dfsph(a.every)
    else:
        s.step_iisph(a.every)
    fl = st["type"] == 0
    bad = [] if np.array_equal(s.get("uniqueId"), st["uniqueId"]) else ["order"]
    if not bad:
        bad = [k for k in ("pos", "vel", "density", "pressure") if s.get(k).tobytes() != np.ascontiguousarray(st[k], real).tobytes()]
        if a.step == "wcsph" and s.get("force")[fl].tobytes() != np.ascontiguousarray(st["force"], real)[fl].tobytes():
            bad.append("force")
    v = np.linalg.norm(st["vel"][fl], axis=1).max()
    log.append({"step": done, "differs": bad, "max_speed": float(v), "out_of_grid": s.num_undef()})
    print(json.dumps(log[-1]), file=sys.stderr, flush=True)
    if bad:
        break
print(json.dumps({"step": a.step, "scene": a.scene, "dims": a.dims, "particles": len(sc["pos"]), "variant": a.variant, "f64": a.f64, "jitter": a.jitter, "seed": a.seed, "steps": log[-1]["step"],
                  "equal_bit_for_bit": not log[-1]["differs"], "oracle_seconds": round(t_cpu, 1), "log": log}))
