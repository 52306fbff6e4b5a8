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
