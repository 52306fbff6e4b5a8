#!/usr/bin/env python3
"""Do the density pass (bound by instruction issue) and the force replay (bound by latency) of the WCSPH step run
faster side by side than one after the other?  Two solvers with half the scene each, on their own streams: the
force pass of one queued together with the density pass of the other, against the two run apart.  Then, with a
measurement build of the library (tools/build_variant.sh ablate "-DFSA_ABLATE" fsa_sph_fast.hip fsa_capi.hip and
FSA_LIB_PATH), both passes of one solver in ONE launch with their workgroups mixed on every CU."""
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

# the two passes of ONE solver in one launch, their workgroups alternating on every XCD (fsa_debug_fused_probe)
import ctypes as C
lib = capi.load()
lib.fsa_debug_fused_probe.restype = C.c_int
lib.fsa_debug_fused_probe.argtypes = [C.c_void_p, C.c_int]


def fused(mode):
    capi.check(lib.fsa_debug_fused_probe(A._h, mode))


sol = [A]
if lib.fsa_debug_fused_probe(A._h, 1) != 0:
    out["fused"] = "not in this build"
    print(json.dumps(out))
    sys.exit(0)
A.compute_pressure(); A.compute_force(); A.synchronize()
out["one_solver_density_then_force_ms"] = timed(lambda: (A.compute_pressure(), A.compute_force()))
for stride in (1, 2, 3, 4, 5, 8, 16, 64):
    out[f"fused_stride{stride}_ms"] = [round(timed(lambda: fused(m + 16 * stride)), 4) for m in (0, 1, 2)]  # both, density, force
f0 = A.get("force").copy()
fused(0); A.synchronize()
import numpy as np
out["fused_force_equals"] = bool(np.array_equal(f0, A.get("force")))
print(json.dumps(out))
