#!/usr/bin/env python3
"""Do the density pass (bound by instruction issue) and the force replay (bound by latency) of the WCSPH step run
faster side by side than one after the other?  Two solvers with half the scene each, on their own streams: the
force pass of one queued together with the density pass of the other, against the two run apart."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[256, 128, 256])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
sol = []
for k in range(2):
    sc = scenes.dam_break(*a.dims)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.step(30)
    s.sort_particles()
    s.compute_pressure()
    s.compute_force()
    s.synchronize()
    sol.append(s)
A, B = sol


def timed(fn):
    for s in sol:
        s.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    for s in sol:
        s.synchronize()
    return (time.perf_counter() - t0) / a.reps * 1e3


def apart():
    A.compute_force(); A.synchronize(); B.compute_pressure(); B.synchronize()


def together():
    A.compute_force(); B.compute_pressure(); A.synchronize(); B.synchronize()


def two_density():
    A.compute_pressure(); B.compute_pressure(); A.synchronize(); B.synchronize()


def two_force():
    A.compute_force(); B.compute_force(); A.synchronize(); B.synchronize()


out = {"dims": a.dims, "particles_each": A.n}
out["force_alone_ms"] = timed(lambda: (A.compute_force(), A.synchronize()))
out["density_alone_ms"] = timed(lambda: (B.compute_pressure(), B.synchronize()))
out["force_then_density_ms"] = timed(apart)
out["force_with_density_ms"] = timed(together)
out["density_with_density_ms"] = timed(two_density)
out["force_with_force_ms"] = timed(two_force)
print(json.dumps(out))
