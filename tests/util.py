"""Helpers shared by the parity tests: build an oracle state and a GPU solver from one scene."""
import ctypes as C

import numpy as np


def oracle_pair(orc, scene, **over):
    """(oracle params, oracle state) for a scene dict."""
    P = dict(scene["params"])
    P.update(over)
    p = orc.make_params(P["spacing"], P["smoothratio"], P["vmin"], P["vmax"], gravity=P["gravity"], dt=P["dt"],
                        viscosity=P["viscosity"], pressureK=P["pressureK"], boundary_visc=P["boundary_visc"],
                        acceleration_limit=P["acceleration_limit"], boundary_mode=P["boundary_mode"],
                        surface_tension=P.get("surface_tension", 0.0),
                        boundary_friction=P.get("boundary_friction", 0.0))
    nc = p.gridres[0] * p.gridres[1] * p.gridres[2]
    st = orc.new_state(len(scene["pos"]), nc)
    sc = dict(scene)
    if orc.precision == "f64":
        for k in ("pos", "vel", "color", "normal", "mass", "restDensity"):
            sc[k] = np.asarray(scene[k], np.float64)
    st.load(sc)
    return p, st


def gpu_solver(scene, precision=0, capacity=None, **over):
    from fluidsolver_amd import capi
    P = dict(scene["params"])
    P.update(over)
    p = capi.make_params(precision, **P)
    s = capi.Solver(p, capacity or len(scene["pos"]), precision)
    s.upload(scene)
    return p, s


def rel_err(a, b, scale=None):
    """max |a-b| / max(|b|, scale) -- scale defaults to the largest magnitude in b."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    if scale is None:
        scale = np.abs(b).max() if b.size else 1.0
    scale = max(float(scale), 1e-300)
    return float((np.abs(a - b) / np.maximum(np.abs(b), scale)).max())


def ulp_diff(a, b):
    """largest distance in units in the last place between two float32 arrays"""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.abs(a - b).max()) if a.size else 0


def elem_err(a, b, floor):
    """Per-element relative error: max_i |a_i - b_i| / max(|b_i|, floor_i).  `floor` (a scalar or an array) is the
    magnitude below which an element is compared absolutely; nothing is scaled by the array's largest entry."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    with np.errstate(invalid="ignore", divide="ignore"):
        floor = np.asarray(floor, np.float64)
        floor = np.where(np.isnan(floor), np.inf, floor)  # no scale for that element (rest density 0): not compared
        return float((np.abs(a - b) / np.maximum(np.maximum(np.abs(b), floor), 1e-300)).max())


def force_err(f_a, f_b, floor=None):
    """Per-particle relative error of force vectors: max_i |f_a_i - f_b_i| / max(|f_b_i|, floor), Euclidean norms;
    floor defaults to the median |f_b_i| (a particle whose pair terms cancel is compared against a typical force)."""
    f_a = np.asarray(f_a, np.float64)
    f_b = np.asarray(f_b, np.float64)
    if f_a.size == 0:
        return 0.0
    nb = np.linalg.norm(f_b, axis=1)
    if floor is None:
        floor = float(np.median(nb))
    return float((np.linalg.norm(f_a - f_b, axis=1) / np.maximum(nb, max(floor, 1e-300))).max())


def force_term_sums(pos, vel, mass, density, pressure, typ, sr, kspikydiff, viscosity, chunk=256):
    """S_i = sum over the fluid neighbours j of fluid particle i of the MAGNITUDES of the pair terms of the force
    pass (pressure term and viscosity term of ForceCell, solver/sph_kernel_impl.cuh:295-312), in float64, by brute
    force: what a per-term relative error is multiplied with when the terms themselves cancel.  0 for other types."""
    pos = np.asarray(pos, np.float64)
    vel = np.asarray(vel, np.float64)
    fl = np.where(np.asarray(typ) == 0)[0]
    with np.errstate(invalid="ignore", divide="ignore"):  # wall particles without fluid neighbours have density 0
        a = np.asarray(pressure, np.float64) / np.asarray(density, np.float64) ** 2
    mass = np.asarray(mass, np.float64)
    dens = np.asarray(density, np.float64)
    out = np.zeros(len(pos))
    pj, vj = pos[fl], vel[fl]
    for s0 in range(0, len(fl), chunk):
        i = fl[s0:s0 + chunk]
        dx = pos[i][:, None, :] - pj[None, :, :]
        d2 = (dx * dx).sum(-1)
        d = np.sqrt(d2)
        ok = (d2 < sr * sr) & (i[:, None] != fl[None, :]) & (d > 0)
        dd = np.where(ok, d, 1.0)
        nablaw = kspikydiff * (sr - dd) ** 2 / dd
        pterm = dd * nablaw * mass[fl][None, :] * (a[i][:, None] + a[fl][None, :])
        dv = np.linalg.norm(vel[i][:, None, :] - vj[None, :, :], axis=-1)
        vterm = dv * nablaw * d2 / (d2 + 0.01 * sr * sr) * mass[fl][None, :] / dens[fl][None, :] * 2 * viscosity
        out[i] = np.where(ok, np.abs(pterm) + np.abs(vterm), 0.0).sum(1)
    return out
