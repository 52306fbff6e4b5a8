#!/usr/bin/env python3
"""Per-kernel times of the WCSPH step for fluid blocks of the same particle count and different shapes: how much
the neighbour passes depend on how far apart (in sorted order) the rows of a particle's 27 cells lie.  The sorted
order is (y, z, x): a scene with a small x-z cross-section has its y-neighbour rows a few workgroups away."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, nargs="+", default=[256, 256, 256, 64, 4096, 64, 1024, 16, 1024])
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
out = []
for k in range(0, len(a.shapes), 3):
    dims = a.shapes[k:k + 3]
    sc = scenes.dam_break(*dims)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.step(20)
    s.synchronize()
    s.timing_enable(True)
    s.timing_reset()
    s.step(a.steps)
    s.synchronize()
    kt = s.timing_read()
    nf = sc["n_fluid"]
    row = {"dims": dims, "particles": len(sc["pos"]), "fluid": nf, "gridres": list(p.gridres)}
    for name, (c, ms) in sorted(kt.items()):
        row[name + "_ms"] = round(ms / max(c, 1), 4)
        row[name + "_ns_per_fluid_particle"] = round(ms / max(c, 1) * 1e6 / nf, 4)
    out.append(row)
    s.close()
print(json.dumps(out))
