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
ap.add_argument("--recut-every", type=int, default=0, help="balance the cuts afresh every so many steps")
ap.add_argument("--lopsided", action="store_true", help="start from cuts that put most particles into the first slabs")
ap.add_argument("--shift-cuts", type=int, default=0, help="start from the balanced cuts moved by this many layers")
a = ap.parse_args()
if a.dims:
    from fluidsolver_amd import scenes
    sc = scenes.dam_break(*a.dims)
else:
    sc = bench.build_scene(a.scene, 0.0)
cuts0 = None
if a.lopsided:
    gz = slabs.derive_grid(sc["params"])["gridres"][2]
    cuts0 = [0] + [gz // 2 + 3 * r for r in range(a.world - 1)] + [gz]
if a.shift_cuts:
    geo = slabs.derive_grid(sc["params"])
    lay = slabs.cell_layer_host(np.asarray(sc["pos"], np.float32), geo["vmin"], geo["vmax"], geo["dx"], geo["gridres"])
    bal = slabs.balanced_cuts(np.bincount(lay[lay >= 0], minlength=geo["gridres"][2]), a.world)
    cuts0 = [bal[0]] + [c + a.shift_cuts for c in bal[1:-1]] + [bal[-1]]
ranks = [slabs.SlabSolver(sc, r, a.world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                          exchange_buffers=False, cuts=cuts0) for r in range(a.world)]


def make_group():
    return capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                                max(ss.record_cap for ss in ranks))


g = make_group()
state = {"done": 0, "sent": 0, "lost": 0, "recuts": 0, "imbalance": []}


def run(nsteps, timed=False):
    global g
    left = nsteps
    while left > 0:
        k = left
        if a.recut_every:
            k = min(k, a.recut_every - state["done"] % a.recut_every)
        g.step(k)
        left -= k
        state["done"] += k
        if a.recut_every and state["done"] % a.recut_every == 0 and left > 0:
            st = g.sync()
            state["sent"] += int(st.records_sent); state["lost"] += int(st.lost)
            g.close()
            held = [ss.backend.n for ss in ranks]
            if slabs.rebalance_local(ranks):
                state["recuts"] += 1
            state["imbalance"].append({"step": state["done"], "held_before": held, "held_after": [ss.backend.n for ss in ranks],
                                       "cuts": list(ranks[0].layout.cuts)})
            if timed:
                for ss in ranks:
                    ss.backend.solver.timing_enable(True); ss.backend.solver.timing_reset()
            g = make_group()


run(a.spinup); g.sync()
show = a.show if a.show >= 0 else a.world // 2
for ss in ranks:
    ss.backend.solver.timing_enable(True); ss.backend.solver.timing_reset()
state["sent"] = -int(g.sync().records_sent)  # (the group counts from its creation)
sent0 = 0
t0 = time.perf_counter()
run(a.steps, timed=True); st = g.sync()
wall = time.perf_counter() - t0
state["sent"] += int(st.records_sent); state["lost"] += int(st.lost)
res = {"world": a.world, "cuts": ranks[0].layout.cuts, "ms_per_step_all_slabs": wall / a.steps * 1e3,
       "ms_per_step_per_slab": wall / a.steps * 1e3 / a.world, "particles_held": [ss.backend.n for ss in ranks],
       "lost": state["lost"], "scene": a.scene, "steps": a.steps, "spinup": a.spinup, "recut_every": a.recut_every,
       "recuts": state["recuts"], "records_sent_per_step": (state["sent"] - sent0) / max(a.steps, 1)}
held = np.array(res["particles_held"], float)
res["max_over_mean_particles_held"] = float(held.max() / held.mean())
if state["imbalance"]:
    res["recut_log"] = state["imbalance"]
if a.recut_every:
    res["ms_per_step_all_slabs_note"] = "wall time includes the re-cuts (export, new solvers, upload)"
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
