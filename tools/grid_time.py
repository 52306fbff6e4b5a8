#!/usr/bin/env python3
"""Time the Eulerian grid solver (default 256^3, the size of the fp64 validation configuration): per step
and per CG round, with the bandwidth the round's algorithmic traffic corresponds to.  --slabs N: the same on a
team of N z-slabs on this GPU (a rehearsal of the multi-GPU form: every slab's kernels and the copies of the
neighbours' planes run one after the other on one stream); --check: the fields after the timed steps against a
single solver that did the same steps."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolver_amd import capi
ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, nargs=3, default=[256, 256, 256])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
ap.add_argument("--maxiter", type=int, default=-1)
ap.add_argument("--slabs", type=int, default=0)
ap.add_argument("--reach", type=int, default=1)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
prec = capi.PRECISION_F32 if a.precision == "f32" else capi.PRECISION_F64
ref = None
if a.slabs:
    cuts = [round(r * a.dim[2] / a.slabs) for r in range(a.slabs + 1)]
    g = capi.GridTeam(a.dim, cuts, reach=a.reach, padding=1, h=0.02, dt=0.001, rho=1.0, precision=prec)
    g.synchronize = g.sync
    if a.check:
        ref = capi.GridSolver(a.dim, padding=1, h=0.02, dt=0.001, rho=1.0, precision=prec)
else:
    g = capi.GridSolver(a.dim, padding=1, h=0.02, dt=0.001, rho=1.0, precision=prec)
cells = a.dim[0] * a.dim[1] * a.dim[2]
R = 4 if a.precision == "f32" else 8
rng = np.random.default_rng(1)
# a swirl so that the advection has work to do
for k in ("u", "w"):
    f = (0.2 * rng.standard_normal(g.shape(k))).astype(g.dtype)
    g.set(k, f)
    if ref:
        ref.set(k, f)
g.step(1, a.maxiter)
g.synchronize()
out = {"dim": a.dim, "cells": cells, "precision": a.precision, "slabs": a.slabs, "steps": []}
if a.slabs:
    out["cuts"], out["reach"] = cuts, a.reach
for s in range(a.steps):
    t = {}
    def timed(name, fn):
        g.synchronize(); t0 = time.perf_counter(); r = fn(); g.synchronize(); t[name] = (time.perf_counter() - t0) * 1e3
        return r
    timed("advect_ms", g.advect)
    timed("body_force_ms", g.body_force)
    timed("make_rhs_ms", g.make_rhs)
    it, res = timed("solve_ms", lambda: g.solve(a.maxiter))
    timed("update_u_ms", g.update_u)
    div = timed("div_ms", g.div_velocity)
    rounds = it + 1
    t.update(iterations=it, residual=res, div_u_sum=div, ms_per_round=t["solve_ms"] / rounds,
             # per round and cell: matvec reads q, r and writes Aq; the update reads p, q, r, Aq and writes p, r;
             # the direction pass reads r, q and writes q = 12 reals
             round_GBps=12 * R * cells / (t["solve_ms"] / rounds * 1e-3) / 1e9,
             step_ms=sum(v for k, v in t.items() if k.endswith("_ms")))
    out["steps"].append(t)
if ref:
    ref.step(1 + a.steps, a.maxiter)
    cmp = {}
    for k in ("u", "v", "w", "p"):
        x, y = g.get(k), ref.get(k)
        cmp[k] = {"equal_bits": bool(np.array_equal(x, y)),
                  "max_abs_diff_over_scale": float(np.abs(x.astype(np.float64) - y).max() / max(np.abs(y).max(), 1e-300))}
    out["against_single_solver"] = cmp
print(json.dumps(out))
