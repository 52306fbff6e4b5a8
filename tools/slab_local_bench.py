#!/usr/bin/env python3
"""Rehearsal of an N-rank slab run on one GPU: the bench scene cut into N z-slabs, all slabs in this process (local
transport).  Prints the wall time per step of the whole group and the kernel times of one slab -- what one rank of
an N-GPU run has to do per step, without the link."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, slabs
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dam16m"); ap.add_argument("--world", type=int, default=8)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--spinup", type=int, default=100)
ap.add_argument("--dims", type=int, nargs=3, default=None, help="dam break of nx ny nz particles instead of --scene")
ap.add_argument("--check", action="store_true", help="compare every owned particle with a plain single-solver run, bit for bit")
ap.add_argument("--show", type=int, default=-1, help="slab whose kernel times are printed (default: the middle one)")
a = ap.parse_args()
if a.dims:
    from fluidsolver_amd import scenes
    sc = scenes.dam_break(*a.dims)
else:
    sc = bench.build_scene(a.scene, 0.0)
ranks = [slabs.SlabSolver(sc, r, a.world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                          exchange_buffers=False) for r in range(a.world)]
g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                         max(ss.record_cap for ss in ranks))
g.step(a.spinup); g.sync()
show = a.show if a.show >= 0 else a.world // 2
for ss in ranks:
    ss.backend.solver.timing_enable(True); ss.backend.solver.timing_reset()
t0 = time.perf_counter()
g.step(a.steps); st = g.sync()
wall = time.perf_counter() - t0
res = {"world": a.world, "cuts": ranks[0].layout.cuts, "ms_per_step_all_slabs": wall / a.steps * 1e3,
       "ms_per_step_per_slab": wall / a.steps * 1e3 / a.world, "particles_held": [ss.backend.n for ss in ranks],
       "lost": int(st.lost)}
kt = ranks[show].backend.solver.timing_read()
res["slab"] = show
res["kernels_ms"] = {k: round(ms / c, 4) for k, (c, ms) in kt.items() if c}
res["kernels_sum_ms"] = round(sum(ms / c for k, (c, ms) in kt.items() if c), 4)
if a.check:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_slabs_cpu import assert_same_bits, merged
    total = a.spinup + a.steps
    # the group leaves the particles integrated; the plain run's results for the same instant
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    ps = capi.Solver(p, len(sc["pos"]))
    ps.upload(sc)
    ps.step(total - 1)
    ps.sort_particles(); ps.compute_pressure(); ps.compute_force()
    ref = {k: ps.get(k) for k in ("density", "pressure", "force")}
    ps.advect()
    for k in ("pos", "vel", "uniqueId", "type"):
        ref[k] = ps.get(k)
    ps.close()
    uid, got = merged([ss.owned() for ss in ranks])
    assert_same_bits(uid, got, ref)
    res["check"] = "every owned particle equals the plain run bit for bit"
print(json.dumps(res))
g.close()
for ss in ranks:
    ss.close()
