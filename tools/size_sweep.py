#!/usr/bin/env python3
"""Per-kernel times of the WCSPH step for dam breaks of several shapes (fluid lattice nx x ny x nz), as ns per particle:
shows how the two neighbour passes depend on the size of the scene and on the length of its grid rows (x extent)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", nargs="*", default=["128,64,128", "256,64,128", "64,64,256", "128,128,128", "256,128,128",
                                                "512,64,64", "256,256,64", "256,256,256"])
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
for shape in a.shapes:
    dims = [int(v) for v in shape.split(",")]
    sc = scenes.dam_break(*dims)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.step(100)
    s.synchronize()
    s.timing_enable(True)
    s.timing_reset()
    s.step(a.steps)
    s.synchronize()
    kt = s.timing_read()
    n, nf = len(sc["pos"]), sc["n_fluid"]
    ms = {k: v[1] / v[0] for k, v in kt.items() if v[0]}
    out = {"dims": dims, "particles": n, "fluid": nf, "grid": list(p.gridres), "step_ms": sum(v[1] for v in kt.values()) / a.steps,
           "density_ns": ms.get("density", 0) * 1e6 / n, "force_ns": ms.get("force", 0) * 1e6 / nf,
           "rest_ns": sum(v for k, v in ms.items() if k not in ("density", "force")) * 1e6 / n}
    print(json.dumps(out), flush=True)
    s.close()
