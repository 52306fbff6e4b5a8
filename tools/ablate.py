#!/usr/bin/env python3
"""Time the tuned density / force kernels with parts switched off (state untouched)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dam16m"); ap.add_argument("--jitter", type=float, default=0.2)
ap.add_argument("--variant", type=int, default=2); ap.add_argument("--spinup", type=int, default=5)
ap.add_argument("--ablate", type=int, nargs="+", default=[0, 1, 2, 3, 4, 5, 7])
a = ap.parse_args()
import bench
sc = bench.build_scene(a.scene, a.jitter)
p = capi.make_params(0, **sc["params"])
s = capi.Solver(p, len(sc["pos"])); s.set_variant(a.variant); s.upload(sc)
s.step(a.spinup); s.sort_particles(); s.compute_pressure(); s.synchronize()
lib = capi.load(); lib.fsa_debug_time_pass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
for which, name in ((0, "density"), (1, "force")):
    for ab in a.ablate:
        s.timing_enable(True); s.timing_reset()
        capi.check(lib.fsa_debug_time_pass(s._h, which, ab, 5))
        t = s.timing_read(); s.timing_enable(False)
        k = name + "_dbg"
        print(f"{name} ablate={ab} (1=no drain 2=no append 4=no cull 128=every block walks the cells): {t[k][1]/t[k][0]:.3f} ms")
