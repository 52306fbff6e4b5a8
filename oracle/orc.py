"""ctypes front end of the CPU oracle (oracle/sph_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg as the checker / timed baseline.  The product
package never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TYPE_FLUID, TYPE_CLOTH, TYPE_BOUNDARY, TYPE_EMITTER, TYPE_RIGID = 0, 1, 2, 3, 4
GRID_UNDEF = 99999999

_VEC3 = ("pos", "vel", "normal", "force", "v_star",
         "sortedPos", "sortedVel", "sortedNormal", "sortedV_star", "dii", "dijpjl", "pressureForce")
_VEC4 = ("color", "sortedColor")
_REAL1 = ("mass", "density", "pressure", "restDensity",
          "sortedMass", "sortedRestDensity", "sorted_pressure", "DF_factor", "pstiff", "error",
          "aii", "density_star")
_INT1 = ("type", "group", "uniqueId", "indexTable", "sortedType", "sortedGroup",
         "sortedUniqueId", "particleHash", "particleIndex")
_CELL = ("gridCellStart", "gridCellEnd")

# field order of struct orc_state in sph_oracle.h
_STATE_FIELDS = (
    "pos", "vel", "normal", "force", "v_star", "color",
    "mass", "density", "pressure", "restDensity",
    "type", "group", "uniqueId", "indexTable",
    "sortedPos", "sortedVel", "sortedNormal", "sortedV_star", "sortedColor",
    "sortedMass", "sortedRestDensity", "sorted_pressure",
    "sortedType", "sortedGroup", "sortedUniqueId",
    "particleHash", "particleIndex", "gridCellStart", "gridCellEnd",
    "DF_factor", "pstiff", "error",
    "dii", "dijpjl", "pressureForce", "aii", "density_star",
)


def build(force: bool = False) -> None:
    """Compile liborc32.so / liborc64.so with the committed Makefile."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _make_types(real):
    class Params(C.Structure):
        _fields_ = [
            ("dx", real), ("gridxmin", real * 3), ("gridxmax", real * 3), ("gridres", C.c_int * 3),
            ("gravity", real * 3), ("dt", real), ("viscosity", real), ("spacing", real),
            ("smoothradius", real), ("kspikydiff", real), ("kernel_cubic", real),
            ("kernel_cubic_gradient", real), ("pressureK", real), ("boundary_visc", real),
            ("acceleration_limit", real), ("boundary_mode", C.c_int), ("kpoly6", real),
            ("surface_tension", real), ("boundary_friction", real),
        ]

    class State(C.Structure):
        _fields_ = [("n", C.c_int), ("num_cells", C.c_int)] + [(f, C.c_void_p) for f in _STATE_FIELDS]

    return Params, State


class Oracle:
    """One precision of the oracle: Oracle('f32') or Oracle('f64')."""

    def __init__(self, precision: str = "f32"):
        assert precision in ("f32", "f64")
        self.precision = precision
        self.real = C.c_float if precision == "f32" else C.c_double
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.prefix = "orc_" if precision == "f32" else "orc64_"
        path = os.path.join(_HERE, "liborc32.so" if precision == "f32" else "liborc64.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        self.Params, self.State = _make_types(self.real)
        L, P, S = self.lib, C.POINTER(self.Params), C.POINTER(self.State)
        self._fn("create", S, [C.c_int, C.c_int])
        self._fn("destroy", None, [S])
        self._fn("derive_params", None, [P, self.real, self.real, C.POINTER(self.real), C.POINTER(self.real)])
        self._fn("set_threads", None, [C.c_int])
        self._fn("get_threads", C.c_int, [])
        for name in ("calc_hash", "reorder_data_and_find_cell_start", "sort_particles", "compute_pressure",
                     "compute_force", "advect", "rigid_particle_volume"):
            self._fn(name, None, [P, S])
        self._fn("sort_particle", None, [S])
        self._fn("copy_sorted_to_primary", None, [S])
        self._fn("solve_sph", None, [P, S, C.c_int])
        for name in ("compute_density_alpha", "compute_non_pressure_force", "solve_density_stiff",
                     "solve_divergence_stiff", "apply_pstiff", "update_position", "update_velocities"):
            self._fn(name, None, [P, S])
        for name in ("correct_density_error", "correct_divergence_error"):
            self._fn(name, C.c_int, [P, S, C.c_int, self.real, C.POINTER(self.real)])
        self._fn("solve_dfsph", None, [P, S, C.c_int])
        for name in ("non_pressure_force_multiphase", "iisph_factor", "iisph_predict_density", "iisph_calc_dijpjl",
                     "iisph_calc_new_pressure", "iisph_calc_pressure_force", "iisph_update"):
            self._fn(name, None, [P, S])
        self._fn("effective_mass", None, [P, S, C.c_int, C.POINTER(self.real), C.POINTER(self.real)])
        self._fn("iisph_solve_pressure", C.c_int, [P, S, C.c_int, C.c_int, C.POINTER(self.real)])
        self._fn("solve_iisph", None, [P, S, C.c_int, C.c_int])

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        setattr(self, "_" + name, f)

    # -- parameters -------------------------------------------------
    def make_params(self, spacing, smoothratio, vmin, vmax, *, gravity=(0.0, -10.0, 0.0), dt=0.0004,
                    viscosity=0.05, pressureK=1119.0, boundary_visc=0.0, acceleration_limit=0.0,
                    boundary_mode=0, surface_tension=0.0, boundary_friction=0.0):
        p = self.Params()
        r3 = self.real * 3
        self._derive_params(C.byref(p), spacing, smoothratio, r3(*vmin), r3(*vmax))
        p.gravity = r3(*gravity)
        p.dt = dt
        p.viscosity = viscosity
        p.pressureK = pressureK
        p.boundary_visc = boundary_visc
        p.acceleration_limit = acceleration_limit
        p.boundary_mode = boundary_mode
        p.surface_tension = surface_tension
        p.boundary_friction = boundary_friction
        return p

    def set_threads(self, n: int) -> None:
        self._set_threads(int(n))

    def new_state(self, n: int, num_cells: int) -> "OracleState":
        return OracleState(self, n, num_cells)


class OracleState:
    """Owns one orc_state; every array is exposed as a numpy view (no copy)."""

    def __init__(self, orc: Oracle, n: int, num_cells: int):
        self.orc = orc
        self.n = n
        self.num_cells = num_cells
        self._s = orc._create(n, num_cells)
        if not self._s:
            raise MemoryError("orc_create failed")
        st = self._s.contents
        self.a = {}
        for f in _STATE_FIELDS:
            ptr = getattr(st, f)
            if f in _VEC3:
                shape, ct = (n, 3), orc.real
            elif f in _VEC4:
                shape, ct = (n, 4), orc.real
            elif f in _REAL1:
                shape, ct = (n,), orc.real
            elif f in _INT1:
                shape, ct = (n,), C.c_int
            else:
                shape, ct = (num_cells,), C.c_int
            count = int(np.prod(shape))
            if count == 0:
                self.a[f] = np.zeros(shape, dtype=np.dtype(ct))
            else:
                buf = (ct * count).from_address(ptr)
                self.a[f] = np.frombuffer(buf, dtype=np.dtype(ct)).reshape(shape)

    def __getitem__(self, k):
        return self.a[k]

    def load(self, scene: dict) -> None:
        """Copy a scene dict (pos, vel, color, type, group, uniqueId, mass, restDensity, normal) in."""
        n = self.n
        for k in ("pos", "vel", "normal", "color", "mass", "restDensity", "type", "group", "uniqueId",
                  "v_star", "pressure", "density", "force", "dii", "dijpjl", "pressureForce", "aii",
                  "density_star"):
            if k in scene:
                self.a[k][...] = np.asarray(scene[k]).reshape(self.a[k].shape)
        if "uniqueId" not in scene:
            self.a["uniqueId"][...] = np.arange(n, dtype=np.int32)

    def close(self):
        if self._s:
            self.a = {}
            self.orc._destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # stage calls ----------------------------------------------------
    def calc_hash(self, p): self.orc._calc_hash(C.byref(p), self._s)
    def sort_particle(self): self.orc._sort_particle(self._s)
    def reorder(self, p): self.orc._reorder_data_and_find_cell_start(C.byref(p), self._s)
    def copy_sorted_to_primary(self): self.orc._copy_sorted_to_primary(self._s)
    def sort_particles(self, p): self.orc._sort_particles(C.byref(p), self._s)
    def compute_pressure(self, p): self.orc._compute_pressure(C.byref(p), self._s)
    def compute_force(self, p): self.orc._compute_force(C.byref(p), self._s)
    def advect(self, p): self.orc._advect(C.byref(p), self._s)
    def rigid_particle_volume(self, p): self.orc._rigid_particle_volume(C.byref(p), self._s)
    def solve_sph(self, p, nsteps=1): self.orc._solve_sph(C.byref(p), self._s, int(nsteps))

    # DFSPH ------------------------------------------------------------
    def compute_density_alpha(self, p): self.orc._compute_density_alpha(C.byref(p), self._s)
    def compute_non_pressure_force(self, p): self.orc._compute_non_pressure_force(C.byref(p), self._s)
    def solve_density_stiff(self, p): self.orc._solve_density_stiff(C.byref(p), self._s)
    def solve_divergence_stiff(self, p): self.orc._solve_divergence_stiff(C.byref(p), self._s)
    def apply_pstiff(self, p): self.orc._apply_pstiff(C.byref(p), self._s)
    def update_position(self, p): self.orc._update_position(C.byref(p), self._s)
    def update_velocities(self, p): self.orc._update_velocities(C.byref(p), self._s)

    def correct_density_error(self, p, maxiter=5, ethres=1.0):
        e = self.orc.real()
        it = self.orc._correct_density_error(C.byref(p), self._s, int(maxiter), ethres, C.byref(e))
        return it, e.value

    def correct_divergence_error(self, p, maxiter=5, ethres=1.0):
        e = self.orc.real()
        it = self.orc._correct_divergence_error(C.byref(p), self._s, int(maxiter), ethres, C.byref(e))
        return it, e.value

    def solve_dfsph(self, p, nsteps=1): self.orc._solve_dfsph(C.byref(p), self._s, int(nsteps))

    # IISPH ------------------------------------------------------------
    def effective_mass(self, p, dens_arr, vfrac):
        """vfrac: (n, typenum) in uniqueId order"""
        vf = np.ascontiguousarray(vfrac, self.orc.dtype)
        da = np.ascontiguousarray(dens_arr, self.orc.dtype)
        RP = C.POINTER(self.orc.real)
        self.orc._effective_mass(C.byref(p), self._s, vf.shape[1], da.ctypes.data_as(RP), vf.ctypes.data_as(RP))

    def non_pressure_force_multiphase(self, p): self.orc._non_pressure_force_multiphase(C.byref(p), self._s)
    def iisph_factor(self, p): self.orc._iisph_factor(C.byref(p), self._s)
    def iisph_predict_density(self, p): self.orc._iisph_predict_density(C.byref(p), self._s)
    def iisph_calc_dijpjl(self, p): self.orc._iisph_calc_dijpjl(C.byref(p), self._s)
    def iisph_calc_new_pressure(self, p): self.orc._iisph_calc_new_pressure(C.byref(p), self._s)
    def iisph_calc_pressure_force(self, p): self.orc._iisph_calc_pressure_force(C.byref(p), self._s)
    def iisph_update(self, p): self.orc._iisph_update(C.byref(p), self._s)

    def iisph_solve_pressure(self, p, maxiter=10, mean_in_double=False):
        e = self.orc.real()
        rounds = self.orc._iisph_solve_pressure(C.byref(p), self._s, int(maxiter), int(mean_in_double), C.byref(e))
        return rounds, e.value

    def solve_iisph(self, p, nsteps=1, mean_in_double=False):
        self.orc._solve_iisph(C.byref(p), self._s, int(nsteps), int(mean_in_double))


# ---------------------------------------------------------------------------------------------
# Eulerian grid solver (oracle/grid_oracle.c)
# ---------------------------------------------------------------------------------------------
_GRID_FACE = ("u", "v", "w", "uadv", "vadv", "wadv")
_GRID_CELL = ("p", "divU", "b", "Aq", "r", "q")


def _grid_struct(real):
    class Grid(C.Structure):
        _fields_ = ([("dim", C.c_int * 3), ("padding", C.c_int)] +
                    [(k, C.c_int) for k in ("xlen", "ylen", "zlen", "ulen", "vlen", "wlen", "Size", "dimSize",
                                            "uSize", "vSize", "wSize")] +
                    [("xmin", real * 3), ("h", real), ("dt", real), ("rho", real)] +
                    [(k, C.c_void_p) for k in _GRID_FACE + _GRID_CELL] +
                    [("divUsum", real), ("frame", C.c_int), ("n_particles", C.c_int), ("pos", C.c_void_p),
                     ("color", C.c_void_p)])
    return Grid


class GridOracle:
    """One grid of the grid-solver oracle: GridOracle('f32', dim, padding, h, dt, rho)."""

    def __init__(self, precision, dim, padding=1, h=0.02, dt=0.001, rho=1.0):
        assert precision in ("f32", "f64")
        self.real = C.c_float if precision == "f32" else C.c_double
        self.dtype = np.float32 if precision == "f32" else np.float64
        prefix = "gorc_" if precision == "f32" else "gorc64_"
        path = os.path.join(_HERE, "libgorc32.so" if precision == "f32" else "libgorc64.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        self.Grid = _grid_struct(self.real)
        G, RP = C.POINTER(self.Grid), C.POINTER(self.real)

        def fn(name, restype, argtypes):
            f = getattr(self.lib, prefix + name)
            f.restype, f.argtypes = restype, argtypes
            return f
        self._create = fn("create", G, [C.POINTER(C.c_int), C.c_int, self.real, self.real, self.real])
        self._destroy = fn("destroy", None, [G])
        self._set_particles = fn("set_particles", None, [G, C.c_int, RP])
        self._mvproduct = fn("mvproduct", None, [G, RP, RP])
        self._dotproduct = fn("dotproduct", self.real, [G, RP, RP])
        self._solve = fn("solve", C.c_int, [G, C.c_int, RP])
        self._step = fn("step", None, [G, C.c_int])
        for name in ("div_velocity", "make_rhs", "update_u", "advect", "body_force", "testcase", "advect_particles"):
            setattr(self, "_" + name, fn(name, None, [G]))
        for name in ("sample_u", "sample_v", "sample_w"):
            setattr(self, "_" + name, fn(name, self.real, [G, RP]))
        self._g = self._create((C.c_int * 3)(*dim), int(padding), h, dt, rho)
        if not self._g:
            raise MemoryError("gorc_create failed")
        self.dim = tuple(int(d) for d in dim)
        self.padding = int(padding)

    # -- views ------------------------------------------------------------------
    def _view(self, ptr, count, shape):
        buf = (self.real * count).from_address(ptr)
        return np.frombuffer(buf, dtype=self.dtype).reshape(shape)

    def __getitem__(self, k):
        """numpy view (no copy) of a field; face fields as (z, y, x) padded arrays, cell fields as (z, y, x).
        The views of u/v/w change identity after advect (the buffers swap): fetch them again."""
        g = self._g.contents
        if k in _GRID_FACE:
            shape = {"u": (g.zlen, g.ylen, g.ulen), "v": (g.zlen, g.vlen, g.xlen), "w": (g.wlen, g.ylen, g.xlen)}[k[0]]
            return self._view(getattr(g, k), int(np.prod(shape)), shape)
        if k in _GRID_CELL:
            return self._view(getattr(g, k), g.dimSize, (g.dim[2], g.dim[1], g.dim[0]))
        if k == "pos":
            return self._view(g.pos, g.n_particles * 3, (g.n_particles, 3)) if g.n_particles else np.zeros((0, 3), self.dtype)
        if k == "color":
            return self._view(g.color, g.n_particles * 4, (g.n_particles, 4)) if g.n_particles else np.zeros((0, 4), self.dtype)
        raise KeyError(k)

    @property
    def xmin(self): return np.array(self._g.contents.xmin[:], self.dtype)
    @property
    def h(self): return self._g.contents.h
    @property
    def dt(self): return self._g.contents.dt
    @property
    def rho(self): return self._g.contents.rho
    @property
    def div_u_sum(self): return self._g.contents.divUsum

    def _p(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        return a, a.ctypes.data_as(C.POINTER(self.real))

    def set_particles(self, pos):
        a, ptr = self._p(pos)
        self._set_particles(self._g, len(a), ptr)

    def mvproduct(self, v):
        a, pa = self._p(v)
        out = np.empty_like(a)
        self._mvproduct(self._g, pa, out.ctypes.data_as(C.POINTER(self.real)))
        return out

    def dotproduct(self, a, b):
        a, pa = self._p(a)
        b, pb = self._p(b)
        return self._dotproduct(self._g, pa, pb)

    def sample(self, which, point):
        a, pa = self._p(point)
        return getattr(self, "_sample_" + which)(self._g, pa)

    def solve(self, maxiter=-1):
        r = self.real()
        it = self._solve(self._g, int(maxiter), C.byref(r))
        return it, r.value

    def div_velocity(self): self._div_velocity(self._g)
    def make_rhs(self): self._make_rhs(self._g)
    def update_u(self): self._update_u(self._g)
    def advect(self): self._advect(self._g)
    def body_force(self): self._body_force(self._g)
    def testcase(self): self._testcase(self._g)
    def advect_particles(self): self._advect_particles(self._g)
    def step(self, maxiter=-1): self._step(self._g, int(maxiter))

    def close(self):
        if self._g:
            self._destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
