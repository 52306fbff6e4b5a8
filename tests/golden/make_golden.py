#!/usr/bin/env python3
"""Generates the fixtures in this directory from the CPU oracle (oracle/sph_oracle.c).

    python tests/golden/make_golden.py

The reference ships no golden vectors for this path and cannot be built here, so these files pin the
ORACLE (and, through the GPU tests, the HIP path) against silent change; they are not outputs of the
reference.  One file per precision: the inputs of a small noisy dam break with wall particles and the
outputs of every stage of one step, plus the positions after five whole steps; and, in iisph_tank_*.npz,
the same for the IISPH step on a small tank whose wall uses the viscosity / friction branch; and, in
grid_16x32x16_*.npz, the stages of one step of the Eulerian grid solver on its shipped grid.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fluidsolver_amd import scenes  # noqa: E402
from oracle.orc import Oracle  # noqa: E402
from util import oracle_pair  # noqa: E402


def golden_scene():
    sc = scenes.dam_break(14, 12, 10, tank_x_factor=2.0)
    return scenes.jittered(sc, seed=4242, pos_jitter=0.2, vel_scale=1.0)


def generate(precision):
    orc = Oracle(precision)
    orc.set_threads(1)
    sc = golden_scene()
    p, st = oracle_pair(orc, sc)
    out = {"params_json": np.frombuffer(json.dumps(sc["params"]).encode(), dtype=np.uint8)}
    for k in ("pos", "vel", "type", "mass", "restDensity", "color", "normal", "group", "uniqueId"):
        out["in_" + k] = np.asarray(sc[k])
    st.calc_hash(p)
    out["hash_unsorted"] = st["particleHash"].copy()
    st.sort_particle()
    out["particleHash"] = st["particleHash"].copy()
    out["particleIndex"] = st["particleIndex"].copy()
    st.reorder(p)
    st.copy_sorted_to_primary()
    out["gridCellStart"] = st["gridCellStart"].copy()
    ge = st["gridCellEnd"].copy()
    ge[st["gridCellStart"] == -1] = 0  # undefined for empty cells
    out["gridCellEnd"] = ge
    out["sorted_uniqueId"] = st["uniqueId"].copy()
    out["indexTable"] = st["indexTable"].copy()
    st.compute_pressure(p)
    out["density"] = st["density"].copy()
    out["pressure"] = st["pressure"].copy()
    st.compute_force(p)
    out["force"] = st["force"].copy()
    st.advect(p)
    out["pos_after"] = st["pos"].copy()
    out["vel_after"] = st["vel"].copy()
    out["v_star_after"] = st["v_star"].copy()
    st.solve_sph(p, 4)
    order = np.argsort(st["uniqueId"])
    out["pos_step5_by_id"] = st["pos"][order].copy()
    st.close()
    path = os.path.join(HERE, f"small_dam_{precision}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def iisph_scene():
    sc = scenes.iisph_tank(7, 9, 6, wall_group=0, fluid_groups=(1, 2), boundary_visc=0.1, boundary_friction=1.0)
    return scenes.jittered(sc, seed=777, pos_jitter=0.15, vel_scale=0.4, fluid_only=True)


def generate_iisph(precision):
    orc = Oracle(precision)
    orc.set_threads(1)
    sc = iisph_scene()
    p, st = oracle_pair(orc, sc)
    out = {"params_json": np.frombuffer(json.dumps(sc["params"]).encode(), dtype=np.uint8)}
    for k in ("pos", "vel", "type", "mass", "restDensity", "color", "normal", "group", "uniqueId"):
        out["in_" + k] = np.asarray(sc[k])
    st.sort_particles(p)
    st.iisph_factor(p)
    st.rigid_particle_volume(p)
    out["sorted_uniqueId"] = st["uniqueId"].copy()
    out["density0"] = st["density"].copy()
    out["dii0"] = st["dii"].copy()
    out["restDensity0"] = st["restDensity"].copy()
    st.non_pressure_force_multiphase(p)
    out["v_star"] = st["v_star"].copy()
    st.iisph_predict_density(p)
    out["aii"] = st["aii"].copy()
    out["density_star"] = st["density_star"].copy()
    rounds, mean = st.iisph_solve_pressure(p, 10, mean_in_double=True)
    out["rounds"] = np.int32(rounds)
    out["pressure"] = st["pressure"].copy()
    out["error"] = st["error"].copy()
    st.iisph_update(p)
    out["pressureForce"] = st["pressureForce"].copy()
    out["pos_after"] = st["pos"].copy()
    out["vel_after"] = st["vel"].copy()
    st.sort_particles(p)
    st.iisph_factor(p)
    st.solve_iisph(p, 2, mean_in_double=True)
    out["pos_step3_by_id"] = st["pos"][np.argsort(st["uniqueId"])].copy()
    st.close()
    path = os.path.join(HERE, f"iisph_tank_{precision}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def grid_inputs(precision):
    """a noisy velocity field on the shipped 16 x 32 x 16 grid and 64 tracers"""
    from oracle.orc import GridOracle
    g = GridOracle(precision, (16, 32, 16), padding=1, h=0.02, dt=0.001, rho=1.0)
    rng = np.random.Generator(np.random.PCG64(2024))
    fields = {k: (0.3 * rng.standard_normal(g[k].shape)).astype(g.dtype) for k in ("u", "v", "w")}
    lo = np.array([g.xmin[0], 0.0, g.xmin[2]])
    pts = (lo + rng.random((64, 3)) * np.array([16, 32, 16]) * 0.02).astype(g.dtype)
    return g, fields, pts


def generate_grid(precision):
    g, fields, pts = grid_inputs(precision)
    out = {"in_" + k: v for k, v in fields.items()}
    out["in_tracers"] = pts
    for k, v in fields.items():
        g[k][...] = v
    g.set_particles(pts)
    g.advect()
    for k in ("u", "v", "w"):
        out["advected_" + k] = g[k].copy()
    g.advect_particles()
    out["tracers_after"] = g["pos"].copy()
    g.body_force()
    g.make_rhs()
    out["divU"] = g["divU"].copy()
    out["b"] = g["b"].copy()
    it, res = g.solve(2000)
    out["cg_iterations"] = np.int32(it)
    out["p"] = g["p"].copy()
    g.update_u()
    for k in ("u", "v", "w"):
        out["projected_" + k] = g[k].copy()
    g.div_velocity()
    g.step(2000)
    g.step(2000)
    out["p_step3"] = g["p"].copy()
    g.close()
    path = os.path.join(HERE, f"grid_16x32x16_{precision}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    generate_grid("f32")
    generate_grid("f64")
    generate("f32")
    generate("f64")
    generate_iisph("f32")
    generate_iisph("f64")
