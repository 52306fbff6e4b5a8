"""ctypes binding of libfluidsolver_amd.so (the C ABI declared in include/fluidsolver_amd.h).

There is no fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FSA_LIB_PATH: another build of the same library, for measurements that compare builds)
LIB_PATH = os.environ.get("FSA_LIB_PATH") or os.path.join(_HERE, "lib", "libfluidsolver_amd.so")

FSA_OK = 0
PRECISION_F32, PRECISION_F64 = 0, 1
BOUNDARY_VERBATIM, BOUNDARY_LEGACY_CLAMP = 0, 1
TYPE_FLUID, TYPE_CLOTH, TYPE_BOUNDARY, TYPE_EMITTER, TYPE_RIGID = 0, 1, 2, 3, 4
GRID_UNDEF = 99999999
KERNEL_NAME_MAX = 32

# every symbol include/fluidsolver_amd.h declares
EXPORTS = (
    "fsa_last_error", "fsa_version", "fsa_derive_params", "fsa_create", "fsa_destroy",
    "fsa_copy_param_to_device", "fsa_set_stream", "fsa_synchronize", "fsa_upload", "fsa_num_particles",
    "fsa_num_undef", "fsa_calc_hash", "fsa_sort_particle", "fsa_reorder_data_and_find_cell_start",
    "fsa_sort_particles", "fsa_compute_pressure", "fsa_compute_force", "fsa_advect",
    "fsa_rigid_particle_volume", "fsa_step", "fsa_readback_pos_color", "fsa_get_buffer", "fsa_set_buffer",
    "fsa_timing_enable", "fsa_timing_reset", "fsa_timing_read", "fsa_set_variant", "fsa_set_neighbour_lists",
    "fsa_slab_record_bytes", "fsa_slab_pack", "fsa_slab_unpack", "fsa_slab_set_owned", "fsa_slab_export_owned",
    "fsa_set_scene_has_rigid", "fsa_sort_particle_static",
    "fsa_slab_enable_iisph", "fsa_slab_pack_carry", "fsa_slab_unpack_carry", "fsa_slab_layer_indices",
    "fsa_slab_gather_field", "fsa_slab_scatter_field", "fsa_slab_owned_sum",
    "fsa_slab_iisph_create_local", "fsa_slab_iisph_setup", "fsa_slab_iisph_step", "fsa_slab_iisph_stats",
    "fsa_slab_iisph_destroy",
    "fsa_slab_group_create_local", "fsa_slab_group_unique_id", "fsa_slab_group_create_rccl", "fsa_slab_group_step",
    "fsa_slab_group_sync", "fsa_slab_group_rccl_loopback", "fsa_slab_group_destroy",
    "fsa_compute_density_alpha", "fsa_compute_non_pressure_force", "fsa_solve_density_stiff",
    "fsa_solve_divergence_stiff", "fsa_apply_pstiff", "fsa_update_position", "fsa_update_velocities",
    "fsa_correct_density_error", "fsa_correct_divergence_error", "fsa_setup_dfsph", "fsa_step_dfsph",
    "fsa_non_pressure_force_multiphase", "fsa_iisph_factor", "fsa_iisph_predict_density", "fsa_iisph_calc_dijpjl",
    "fsa_iisph_calc_new_pressure", "fsa_iisph_solve_pressure", "fsa_iisph_calc_pressure_force", "fsa_iisph_update",
    "fsa_setup_iisph", "fsa_step_iisph", "fsa_set_volume_fractions", "fsa_effective_mass",
    "fsa_grid_create", "fsa_grid_destroy", "fsa_grid_sizes", "fsa_grid_synchronize", "fsa_grid_get_field",
    "fsa_grid_set_field", "fsa_grid_set_particles", "fsa_grid_get_particles", "fsa_grid_advect",
    "fsa_grid_advect_particles", "fsa_grid_body_force", "fsa_grid_testcase", "fsa_grid_div_velocity",
    "fsa_grid_make_rhs", "fsa_grid_solve", "fsa_grid_update_u", "fsa_grid_mvproduct", "fsa_grid_dotproduct",
    "fsa_grid_step",
    "fsa_grid_team_create_local", "fsa_grid_team_create_rccl", "fsa_grid_team_destroy", "fsa_grid_team_get_field",
    "fsa_grid_team_set_field", "fsa_grid_team_set_particles", "fsa_grid_team_get_particles", "fsa_grid_team_advect",
    "fsa_grid_team_advect_particles", "fsa_grid_team_body_force", "fsa_grid_team_make_rhs",
    "fsa_grid_team_div_velocity", "fsa_grid_team_solve", "fsa_grid_team_update_u", "fsa_grid_team_step",
    "fsa_grid_team_sync", "fsa_grid_team_create_hosted", "fsa_slab_group_create_hosted", "fsa_slab_enable_dfsph",
    "fsa_slab_owned_max", "fsa_slab_dfsph_create_local", "fsa_slab_dfsph_setup", "fsa_slab_dfsph_step",
    "fsa_slab_dfsph_stats", "fsa_slab_dfsph_destroy",
)


class FsaError(RuntimeError):
    pass


class Params(C.Structure):
    """struct fsa_params"""
    _fields_ = [
        ("dx", C.c_double), ("gridxmin", C.c_double * 3), ("gridxmax", C.c_double * 3),
        ("gridres", C.c_int * 3), ("gravity", C.c_double * 3), ("dt", C.c_double),
        ("viscosity", C.c_double), ("spacing", C.c_double), ("smoothradius", C.c_double),
        ("kspikydiff", C.c_double), ("kernel_cubic", C.c_double), ("kernel_cubic_gradient", C.c_double),
        ("pressureK", C.c_double), ("boundary_visc", C.c_double), ("acceleration_limit", C.c_double),
        ("boundary_mode", C.c_int), ("slab_z0", C.c_int), ("slab_nz", C.c_int), ("kpoly6", C.c_double),
        ("surface_tension", C.c_double), ("boundary_friction", C.c_double),
    ]

    @property
    def num_cells(self) -> int:
        nz = self.slab_nz if self.slab_nz > 0 else self.gridres[2]
        return self.gridres[0] * self.gridres[1] * nz


SLAB_GROUP_ID_BYTES = 128


class SlabGroupStats(C.Structure):
    _fields_ = [("steps", C.c_longlong), ("particles", C.c_longlong), ("max_particles", C.c_longlong),
                ("lost", C.c_longlong), ("records_sent", C.c_longlong), ("error_bits", C.c_int)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * KERNEL_NAME_MAX), ("launches", C.c_int), ("total_ms", C.c_double)]


_lib = None


def _bind_to_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels ship their own libamdhip64 / libhsa-runtime64.  Two copies of the runtime in
    one process cannot both open the GPU, so when torch is installed but not imported yet, its copy is
    loaded first and this library's libamdhip64.so.7 dependency resolves to it (same SONAME), whichever
    of the two gets imported first.  FSA_SYSTEM_HIP=1 keeps the system runtime."""
    if "torch" in sys.modules or os.environ.get("FSA_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load the shared library once and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FsaError(f"{LIB_PATH} is missing: build it with `make -C fluidsolver_amd/csrc` "
                       "(or __graft_entry__.build()); there is no CPU fallback")
    _bind_to_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    H, P = C.c_void_p, C.POINTER(Params)
    lib.fsa_last_error.restype = C.c_char_p
    lib.fsa_last_error.argtypes = []
    lib.fsa_version.restype = C.c_char_p
    lib.fsa_version.argtypes = []
    protos = {
        "fsa_derive_params": [P, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "fsa_create": [P, C.c_int, C.c_int, C.c_int, C.POINTER(H)],
        "fsa_destroy": [H],
        "fsa_copy_param_to_device": [H, P],
        "fsa_set_stream": [H, C.c_void_p],
        "fsa_synchronize": [H],
        "fsa_upload": [H, C.c_int] + [C.c_void_p] * 9,
        "fsa_num_particles": [H, C.POINTER(C.c_int)],
        "fsa_num_undef": [H, C.POINTER(C.c_int)],
        "fsa_calc_hash": [H], "fsa_sort_particle": [H], "fsa_reorder_data_and_find_cell_start": [H],
        "fsa_sort_particles": [H], "fsa_compute_pressure": [H], "fsa_compute_force": [H], "fsa_advect": [H],
        "fsa_rigid_particle_volume": [H],
        "fsa_step": [H, C.c_int],
        "fsa_readback_pos_color": [H, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)],
        "fsa_get_buffer": [H, C.c_char_p, C.c_void_p, C.c_size_t],
        "fsa_set_buffer": [H, C.c_char_p, C.c_void_p, C.c_size_t],
        "fsa_timing_enable": [H, C.c_int], "fsa_timing_reset": [H],
        "fsa_timing_read": [H, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)],
        "fsa_set_variant": [H, C.c_int],
        "fsa_set_neighbour_lists": [H, C.c_int],
        "fsa_slab_record_bytes": [H, C.POINTER(C.c_int)],
        "fsa_slab_pack": [H] + [C.POINTER(C.c_int)] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                           C.POINTER(C.c_int)],
        "fsa_slab_unpack": [H, C.c_void_p, C.c_int],
        "fsa_slab_set_owned": [H, C.POINTER(C.c_int)],
        "fsa_slab_export_owned": [H, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int)],
        "fsa_set_scene_has_rigid": [H, C.c_int],
        "fsa_slab_enable_iisph": [H, C.c_int, C.c_void_p, C.c_void_p],
        "fsa_slab_pack_carry": [H] + [C.POINTER(C.c_int)] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                                 C.c_void_p, C.c_int, C.POINTER(C.c_int)],
        "fsa_slab_unpack_carry": [H, C.c_void_p, C.c_void_p, C.c_int],
        "fsa_slab_layer_indices": [H, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)],
        "fsa_slab_gather_field": [H, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p],
        "fsa_slab_scatter_field": [H, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p],
        "fsa_slab_owned_sum": [H, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)],
        "fsa_slab_iisph_create_local": [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p)],
        "fsa_slab_iisph_setup": [H], "fsa_slab_iisph_step": [H, C.c_int],
        "fsa_slab_iisph_stats": [H, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_int), C.c_int],
        "fsa_slab_iisph_destroy": [H],
        "fsa_slab_group_create_local": [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p)],
        "fsa_slab_group_unique_id": [C.c_char_p],
        "fsa_slab_group_create_rccl": [H, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)],
        "fsa_slab_group_step": [H, C.c_int],
        "fsa_slab_group_sync": [H, C.POINTER(SlabGroupStats)],
        "fsa_slab_group_rccl_loopback": [H, C.c_void_p, C.c_void_p, C.c_size_t],
        "fsa_slab_group_destroy": [H],
        "fsa_sort_particle_static": [P, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
        "fsa_compute_density_alpha": [H], "fsa_compute_non_pressure_force": [H], "fsa_solve_density_stiff": [H],
        "fsa_solve_divergence_stiff": [H], "fsa_apply_pstiff": [H], "fsa_update_position": [H],
        "fsa_update_velocities": [H], "fsa_setup_dfsph": [H], "fsa_step_dfsph": [H, C.c_int],
        "fsa_correct_density_error": [H, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)],
        "fsa_correct_divergence_error": [H, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)],
        "fsa_non_pressure_force_multiphase": [H], "fsa_iisph_factor": [H], "fsa_iisph_predict_density": [H],
        "fsa_iisph_calc_dijpjl": [H], "fsa_iisph_calc_new_pressure": [H], "fsa_iisph_calc_pressure_force": [H],
        "fsa_iisph_update": [H], "fsa_setup_iisph": [H], "fsa_step_iisph": [H, C.c_int],
        "fsa_iisph_solve_pressure": [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)],
        "fsa_set_volume_fractions": [H, C.c_int, C.POINTER(C.c_double), C.c_void_p], "fsa_effective_mass": [H],
        "fsa_grid_create": [C.POINTER(C.c_int), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                            C.POINTER(C.c_void_p)],
        "fsa_grid_destroy": [H], "fsa_grid_sizes": [H, C.POINTER(C.c_int), C.POINTER(C.c_double)],
        "fsa_grid_synchronize": [H], "fsa_grid_get_field": [H, C.c_char_p, C.c_void_p, C.c_size_t],
        "fsa_grid_set_field": [H, C.c_char_p, C.c_void_p, C.c_size_t],
        "fsa_grid_set_particles": [H, C.c_int, C.c_void_p],
        "fsa_grid_get_particles": [H, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)],
        "fsa_grid_advect": [H], "fsa_grid_advect_particles": [H], "fsa_grid_body_force": [H], "fsa_grid_testcase": [H],
        "fsa_grid_div_velocity": [H, C.POINTER(C.c_double)], "fsa_grid_make_rhs": [H],
        "fsa_grid_solve": [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)], "fsa_grid_update_u": [H],
        "fsa_grid_mvproduct": [H, C.c_char_p, C.c_char_p],
        "fsa_grid_dotproduct": [H, C.c_char_p, C.c_char_p, C.POINTER(C.c_double)],
        "fsa_grid_step": [H, C.c_int, C.c_int],
        "fsa_grid_team_create_local": [C.POINTER(C.c_int), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                       C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)],
        "fsa_grid_team_create_rccl": [C.POINTER(C.c_int), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)],
        "fsa_grid_team_create_hosted": [C.POINTER(C.c_int), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)],
        "fsa_slab_enable_dfsph": [H],
        "fsa_slab_dfsph_create_local": [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.POINTER(C.c_void_p)],
        "fsa_slab_dfsph_setup": [H], "fsa_slab_dfsph_step": [H, C.c_int], "fsa_slab_dfsph_destroy": [H],
        "fsa_slab_dfsph_stats": [H, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int],
        "fsa_slab_owned_max": [H, C.c_char_p, C.POINTER(C.c_int), C.c_double, C.POINTER(C.c_double)],
        "fsa_slab_group_create_hosted": [H, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p)],
        "fsa_grid_team_destroy": [H], "fsa_grid_team_get_field": [H, C.c_char_p, C.c_void_p, C.c_size_t],
        "fsa_grid_team_set_field": [H, C.c_char_p, C.c_void_p, C.c_size_t],
        "fsa_grid_team_set_particles": [H, C.c_int, C.c_void_p],
        "fsa_grid_team_get_particles": [H, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)],
        "fsa_grid_team_advect": [H], "fsa_grid_team_advect_particles": [H], "fsa_grid_team_body_force": [H],
        "fsa_grid_team_make_rhs": [H], "fsa_grid_team_div_velocity": [H, C.POINTER(C.c_double)],
        "fsa_grid_team_solve": [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)], "fsa_grid_team_update_u": [H],
        "fsa_grid_team_step": [H, C.c_int, C.c_int], "fsa_grid_team_sync": [H, C.POINTER(C.c_int)],
    }
    for name, args in protos.items():
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = args
    for name in ("fsa_debug_host_waits", "fsa_debug_event_waits"):  # not in the public header
        getattr(lib, name).restype = C.c_longlong
        getattr(lib, name).argtypes = []
    lib.fsa_debug_one_mass_flavour.restype = C.c_int
    lib.fsa_debug_one_mass_flavour.argtypes = [H]
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != FSA_OK:
        raise FsaError(f"fluidsolver_amd error {rc}: {load().fsa_last_error().decode(errors='replace')}")


def make_params(precision=PRECISION_F32, *, spacing, smoothratio, vmin, vmax, gravity=(0.0, -10.0, 0.0),
                dt=0.0004, viscosity=0.05, pressureK=1119.0, boundary_visc=0.0, acceleration_limit=0.0,
                boundary_mode=BOUNDARY_VERBATIM, slab_z0=0, slab_nz=0, surface_tension=0.0,
                boundary_friction=0.0, **_ignored) -> Params:
    """Parameter block as MultiphaseSPHSolver::ParseParam/LoadParam fill it."""
    lib = load()
    p = Params()
    d3 = C.c_double * 3
    check(lib.fsa_derive_params(C.byref(p), precision, spacing, smoothratio, d3(*vmin), d3(*vmax)))
    p.gravity = d3(*gravity)
    p.dt = dt
    p.viscosity = viscosity
    p.pressureK = pressureK
    p.boundary_visc = boundary_visc
    p.acceleration_limit = acceleration_limit
    p.boundary_mode = boundary_mode
    p.slab_z0 = slab_z0
    p.slab_nz = slab_nz
    p.surface_tension = surface_tension
    p.boundary_friction = boundary_friction
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


SORT_STATIC_MAX_PARTICLES = 1 << 20
SORT_STATIC_MAX_CELLS = 1 << 20


def sort_particle_static(params: Params, n: int, pos3_ptr: int, hash_ptr: int, index_ptr: int, device: int = 0,
                         stream_ptr: int = 0) -> None:
    """fsa_sort_particle_static: the pointers are device addresses (n x 3 floats in, n ints out twice)."""
    check(load().fsa_sort_particle_static(C.byref(params), int(device), C.c_void_p(stream_ptr), int(n),
                                          C.c_void_p(pos3_ptr), C.c_void_p(hash_ptr), C.c_void_p(index_ptr)))


# name -> (kind, per-particle width); kind r = real, i = int, c = per-cell int
BUFFERS = {
    "pos": ("r", 3), "vel": ("r", 3), "force": ("r", 3), "v_star": ("r", 3), "normal": ("r", 3),
    "color": ("r", 4), "mass": ("r", 1), "density": ("r", 1), "pressure": ("r", 1), "restDensity": ("r", 1),
    "type": ("i", 1), "group": ("i", 1), "uniqueId": ("i", 1), "indexTable": ("i", 1),
    "particleHash": ("i", 1), "particleIndex": ("i", 1),
    "gridCellStart": ("c", 0), "gridCellEnd": ("c", 0), "cellStartDense": ("c", 1),
    "orderKey": ("l", 1), "neighbourCount": ("i", 1), "hitCount": ("i", 1),
    "DF_factor": ("r", 1), "pstiff": ("r", 1), "error": ("r", 1),
    "dii": ("r", 3), "dijpjl": ("r", 3), "pressureForce": ("r", 3), "aii": ("r", 1), "density_star": ("r", 1),
}


class Solver:
    """Thin object wrapper over one fsa_solver handle."""

    def __init__(self, params: Params, capacity: int, precision=PRECISION_F32, device: int = 0):
        self.lib = load()
        self.params = params
        self.precision = precision
        self.dtype = np.float32 if precision == PRECISION_F32 else np.float64
        self.capacity = int(capacity)
        self.num_cells = params.num_cells
        self._h = C.c_void_p()
        check(self.lib.fsa_create(C.byref(params), precision, self.capacity, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.fsa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data --------------------------------------------------------------------------
    @property
    def n(self) -> int:
        n = C.c_int()
        check(self.lib.fsa_num_particles(self._h, C.byref(n)))
        return n.value

    def upload(self, scene: dict) -> None:
        dt = self.dtype
        keep = []

        def real(k, w, required=True):
            if k not in scene or scene[k] is None:
                if required:
                    raise KeyError(k)
                return None
            a = np.ascontiguousarray(scene[k], dtype=dt)
            keep.append(a)
            return a

        def integer(k, required=True):
            if k not in scene or scene[k] is None:
                if required:
                    raise KeyError(k)
                return None
            a = np.ascontiguousarray(scene[k], dtype=np.int32)
            keep.append(a)
            return a

        pos = real("pos", 3)
        n = 0 if pos is None else pos.reshape(-1, 3).shape[0]
        check(self.lib.fsa_upload(self._h, n, _ptr(pos), _ptr(real("vel", 3)), _ptr(real("color", 4, False)),
                                  _ptr(real("normal", 3, False)), _ptr(integer("type")),
                                  _ptr(integer("group", False)), _ptr(integer("uniqueId", False)),
                                  _ptr(real("mass", 1)), _ptr(real("restDensity", 1))))

    def get(self, name: str) -> np.ndarray:
        kind, w = BUFFERS[name]
        n = self.n
        if kind == "r":
            out = np.empty((n, w) if w > 1 else (n,), dtype=self.dtype)
        elif kind == "i":
            out = np.empty((n,), dtype=np.int32)
        elif kind == "l":
            out = np.empty((n,), dtype=np.int64)
        else:
            out = np.empty((self.num_cells + w,), dtype=np.int32)
        check(self.lib.fsa_get_buffer(self._h, name.encode(), _ptr(out), out.nbytes))
        return out

    def set(self, name: str, value) -> None:
        kind, _ = BUFFERS[name]
        a = np.ascontiguousarray(value, dtype=self.dtype if kind == "r" else np.int32)
        check(self.lib.fsa_set_buffer(self._h, name.encode(), _ptr(a), a.nbytes))

    def readback_pos_color(self):
        n = self.n
        pos = np.empty((n, 3), self.dtype)
        col = np.empty((n, 4), self.dtype)
        m = C.c_int()
        check(self.lib.fsa_readback_pos_color(self._h, _ptr(pos), _ptr(col), C.byref(m)))
        return pos, col

    def num_undef(self) -> int:
        n = C.c_int()
        check(self.lib.fsa_num_undef(self._h, C.byref(n)))
        return n.value

    # -- the operator seam ------------------------------------------------------------
    def copy_param_to_device(self, params: Params): check(self.lib.fsa_copy_param_to_device(self._h, C.byref(params))); self.params = params
    def calc_hash(self): check(self.lib.fsa_calc_hash(self._h))
    def sort_particle(self): check(self.lib.fsa_sort_particle(self._h))
    def reorder(self): check(self.lib.fsa_reorder_data_and_find_cell_start(self._h))
    def sort_particles(self): check(self.lib.fsa_sort_particles(self._h))
    def compute_pressure(self): check(self.lib.fsa_compute_pressure(self._h))
    def compute_force(self): check(self.lib.fsa_compute_force(self._h))
    def advect(self): check(self.lib.fsa_advect(self._h))
    def rigid_particle_volume(self): check(self.lib.fsa_rigid_particle_volume(self._h))
    def step(self, nsteps: int = 1): check(self.lib.fsa_step(self._h, int(nsteps)))
    def synchronize(self): check(self.lib.fsa_synchronize(self._h))
    def set_stream(self, stream_ptr): check(self.lib.fsa_set_stream(self._h, C.c_void_p(stream_ptr)))
    def set_variant(self, v: int): check(self.lib.fsa_set_variant(self._h, int(v)))
    def set_neighbour_lists(self, on, replay: bool = True):
        """on: the passes that repeat between two sorts replay recorded neighbours instead of searching;
        replay=False: from neighbour lists only, never from the hit record of the variant-3 density pass"""
        check(self.lib.fsa_set_neighbour_lists(self._h, (1 if replay else 2) if on else 0))

    # -- DFSPH ------------------------------------------------------------------------
    def compute_density_alpha(self): check(self.lib.fsa_compute_density_alpha(self._h))
    def compute_non_pressure_force(self): check(self.lib.fsa_compute_non_pressure_force(self._h))
    def solve_density_stiff(self): check(self.lib.fsa_solve_density_stiff(self._h))
    def solve_divergence_stiff(self): check(self.lib.fsa_solve_divergence_stiff(self._h))
    def apply_pstiff(self): check(self.lib.fsa_apply_pstiff(self._h))
    def update_position(self): check(self.lib.fsa_update_position(self._h))
    def update_velocities(self): check(self.lib.fsa_update_velocities(self._h))
    def setup_dfsph(self): check(self.lib.fsa_setup_dfsph(self._h))
    def step_dfsph(self, nsteps: int = 1): check(self.lib.fsa_step_dfsph(self._h, int(nsteps)))

    def correct_density_error(self, maxiter=5, ethres=1.0):
        it, e = C.c_int(), C.c_double()
        check(self.lib.fsa_correct_density_error(self._h, int(maxiter), float(ethres), C.byref(it), C.byref(e)))
        return it.value, e.value

    def correct_divergence_error(self, maxiter=5, ethres=1.0):
        it, e = C.c_int(), C.c_double()
        check(self.lib.fsa_correct_divergence_error(self._h, int(maxiter), float(ethres), C.byref(it), C.byref(e)))
        return it.value, e.value

    # -- IISPH ------------------------------------------------------------------------
    def non_pressure_force_multiphase(self): check(self.lib.fsa_non_pressure_force_multiphase(self._h))
    def iisph_factor(self): check(self.lib.fsa_iisph_factor(self._h))
    def iisph_predict_density(self): check(self.lib.fsa_iisph_predict_density(self._h))
    def iisph_calc_dijpjl(self): check(self.lib.fsa_iisph_calc_dijpjl(self._h))
    def iisph_calc_new_pressure(self): check(self.lib.fsa_iisph_calc_new_pressure(self._h))
    def iisph_calc_pressure_force(self): check(self.lib.fsa_iisph_calc_pressure_force(self._h))
    def iisph_update(self): check(self.lib.fsa_iisph_update(self._h))
    def setup_iisph(self): check(self.lib.fsa_setup_iisph(self._h))
    def step_iisph(self, nsteps: int = 1): check(self.lib.fsa_step_iisph(self._h, int(nsteps)))

    def effective_mass(self): check(self.lib.fsa_effective_mass(self._h))

    def set_volume_fractions(self, dens_arr, vfrac):
        """vfrac: (n, typenum) rows in the current device order"""
        vf = np.ascontiguousarray(vfrac, self.dtype)
        da = (C.c_double * vf.shape[1])(*[float(d) for d in dens_arr])
        check(self.lib.fsa_set_volume_fractions(self._h, vf.shape[1], da, _ptr(vf)))

    def iisph_solve_pressure(self, maxiter=10):
        it, e = C.c_int(), C.c_double()
        check(self.lib.fsa_iisph_solve_pressure(self._h, int(maxiter), C.byref(it), C.byref(e)))
        return it.value, e.value

    # -- z-slab decomposition --------------------------------------------------------
    @property
    def slab_record_bytes(self) -> int:
        b = C.c_int()
        check(self.lib.fsa_slab_record_bytes(self._h, C.byref(b)))
        return b.value

    def slab_pack(self, own, keep, left, right, send_left_ptr, cap_left, send_right_ptr, cap_right):
        """-> (remaining, records_left, records_right, lost); the pointers are device addresses."""
        i2 = C.c_int * 2
        counts = (C.c_int * 4)()
        check(self.lib.fsa_slab_pack(self._h, i2(*own), i2(*keep), i2(*left), i2(*right),
                                     C.c_void_p(send_left_ptr), int(cap_left), C.c_void_p(send_right_ptr),
                                     int(cap_right), counts))
        return tuple(counts)

    def slab_unpack(self, records_ptr, count):
        check(self.lib.fsa_slab_unpack(self._h, C.c_void_p(records_ptr), int(count)))

    def slab_export_owned(self, own, records_ptr, cap) -> int:
        n = C.c_int()
        check(self.lib.fsa_slab_export_owned(self._h, (C.c_int * 2)(*own), C.c_void_p(records_ptr), int(cap),
                                             C.byref(n)))
        return n.value

    def slab_set_owned(self, own):
        check(self.lib.fsa_slab_set_owned(self._h, (C.c_int * 2)(*own)))

    def set_scene_has_rigid(self, flag: bool):
        check(self.lib.fsa_set_scene_has_rigid(self._h, 1 if flag else 0))

    # -- the IISPH step on slab solvers (pointers are device addresses) ---------------------
    def slab_enable_iisph(self, group, normal):
        """group / normal of the whole scene, indexed by the global unique ids"""
        g = np.ascontiguousarray(group, np.int32)
        nrm = np.ascontiguousarray(normal, self.dtype).reshape(-1, 3)
        assert len(g) == len(nrm)
        check(self.lib.fsa_slab_enable_iisph(self._h, len(g), _ptr(g), _ptr(nrm)))

    def slab_pack_carry(self, own, keep, left, right, send_left_ptr, extra_left_ptr, cap_left, send_right_ptr,
                        extra_right_ptr, cap_right):
        i2 = C.c_int * 2
        counts = (C.c_int * 4)()
        check(self.lib.fsa_slab_pack_carry(self._h, i2(*own), i2(*keep), i2(*left), i2(*right),
                                           C.c_void_p(send_left_ptr), C.c_void_p(extra_left_ptr), int(cap_left),
                                           C.c_void_p(send_right_ptr), C.c_void_p(extra_right_ptr), int(cap_right),
                                           counts))
        return tuple(counts)

    def slab_unpack_carry(self, records_ptr, extras_ptr, count):
        check(self.lib.fsa_slab_unpack_carry(self._h, C.c_void_p(records_ptr), C.c_void_p(extras_ptr), int(count)))

    def slab_layer_indices(self, lo, hi, idx_ptr, cap) -> int:
        n = C.c_int()
        check(self.lib.fsa_slab_layer_indices(self._h, int(lo), int(hi), C.c_void_p(idx_ptr), int(cap), C.byref(n)))
        return n.value

    def slab_gather_field(self, name, idx_ptr, count, out_ptr):
        check(self.lib.fsa_slab_gather_field(self._h, name.encode(), C.c_void_p(idx_ptr), int(count),
                                             C.c_void_p(out_ptr)))

    def slab_scatter_field(self, name, idx_ptr, count, in_ptr):
        check(self.lib.fsa_slab_scatter_field(self._h, name.encode(), C.c_void_p(idx_ptr), int(count),
                                              C.c_void_p(in_ptr)))

    def slab_owned_sum(self, name, own) -> float:
        v = C.c_double()
        check(self.lib.fsa_slab_owned_sum(self._h, name.encode(), (C.c_int * 2)(*own), C.byref(v)))
        return v.value

    def slab_owned_max(self, name, own, start) -> float:
        v = C.c_double()
        check(self.lib.fsa_slab_owned_max(self._h, name.encode(), (C.c_int * 2)(*own), float(start), C.byref(v)))
        return v.value

    def slab_enable_dfsph(self): check(self.lib.fsa_slab_enable_dfsph(self._h))

    # -- measurement ------------------------------------------------------------------
    def timing_enable(self, on=True): check(self.lib.fsa_timing_enable(self._h, 1 if on else 0))
    def timing_reset(self): check(self.lib.fsa_timing_reset(self._h))

    def timing_read(self) -> dict:
        cap = 64
        arr = (KernelTime * cap)()
        cnt = C.c_int()
        check(self.lib.fsa_timing_read(self._h, arr, cap, C.byref(cnt)))
        return {arr[k].name.decode(): (arr[k].launches, arr[k].total_ms) for k in range(min(cnt.value, cap))}


class GridSolver:
    """Thin object wrapper over one fsa_grid handle (the Eulerian grid solver)."""

    FACE = ("u", "v", "w", "uadv", "vadv", "wadv")
    CELL = ("p", "divU", "b", "Aq", "r", "q")

    def __init__(self, dim, padding=1, h=0.02, dt=0.001, rho=1.0, precision=PRECISION_F32, device: int = 0):
        self.lib = load()
        self.precision = precision
        self.dtype = np.float32 if precision == PRECISION_F32 else np.float64
        self.dim = tuple(int(d) for d in dim)
        self.padding = int(padding)
        h_ = C.c_void_p()
        check(self.lib.fsa_grid_create((C.c_int * 3)(*self.dim), self.padding, float(h), float(dt), float(rho),
                                       precision, device, C.byref(h_)))
        self._h = h_
        sizes, xmin = (C.c_int * 4)(), (C.c_double * 3)()
        check(self.lib.fsa_grid_sizes(self._h, sizes, xmin))
        self.sizes = tuple(sizes)
        self.xmin = np.array(xmin[:])
        nx, ny, nz = self.dim
        pd = self.padding
        self.shapes = {"u": (nz + 2 * pd, ny + 2 * pd, nx + 2 * pd + 1), "v": (nz + 2 * pd, ny + 2 * pd + 1, nx + 2 * pd),
                       "w": (nz + 2 * pd + 1, ny + 2 * pd, nx + 2 * pd)}

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fsa_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shape(self, name):
        return self.shapes[name[0]] if name in self.FACE else (self.dim[2], self.dim[1], self.dim[0])

    def get(self, name: str) -> np.ndarray:
        out = np.empty(self.shape(name), self.dtype)
        check(self.lib.fsa_grid_get_field(self._h, name.encode(), _ptr(out), out.nbytes))
        return out

    def set(self, name: str, value) -> None:
        a = np.ascontiguousarray(value, self.dtype)
        if a.shape != self.shape(name):
            raise ValueError(f"field {name}: shape {a.shape}, expected {self.shape(name)}")
        check(self.lib.fsa_grid_set_field(self._h, name.encode(), _ptr(a), a.nbytes))

    def set_particles(self, pos):
        a = np.ascontiguousarray(pos, self.dtype).reshape(-1, 3)
        check(self.lib.fsa_grid_set_particles(self._h, len(a), _ptr(a)))

    def get_particles(self):
        n = C.c_int()
        check(self.lib.fsa_grid_get_particles(self._h, None, None, C.byref(n)))
        pos, col = np.empty((n.value, 3), self.dtype), np.empty((n.value, 4), self.dtype)
        check(self.lib.fsa_grid_get_particles(self._h, _ptr(pos), _ptr(col), C.byref(n)))
        return pos, col

    def advect(self): check(self.lib.fsa_grid_advect(self._h))
    def advect_particles(self): check(self.lib.fsa_grid_advect_particles(self._h))
    def body_force(self): check(self.lib.fsa_grid_body_force(self._h))
    def testcase(self): check(self.lib.fsa_grid_testcase(self._h))
    def make_rhs(self): check(self.lib.fsa_grid_make_rhs(self._h))
    def update_u(self): check(self.lib.fsa_grid_update_u(self._h))
    def synchronize(self): check(self.lib.fsa_grid_synchronize(self._h))
    def step(self, nsteps=1, maxiter=-1): check(self.lib.fsa_grid_step(self._h, int(nsteps), int(maxiter)))
    def mvproduct(self, src, dst): check(self.lib.fsa_grid_mvproduct(self._h, src.encode(), dst.encode()))

    def div_velocity(self) -> float:
        s = C.c_double()
        check(self.lib.fsa_grid_div_velocity(self._h, C.byref(s)))
        return s.value

    def dotproduct(self, a, b) -> float:
        s = C.c_double()
        check(self.lib.fsa_grid_dotproduct(self._h, a.encode(), b.encode(), C.byref(s)))
        return s.value

    def solve(self, maxiter=-1):
        it, r = C.c_int(), C.c_double()
        check(self.lib.fsa_grid_solve(self._h, int(maxiter), C.byref(it), C.byref(r)))
        return it.value, r.value


class HostTransport(C.Structure):
    """fsa_host_transport"""
    SENDRECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
    ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
    _fields_ = [("ctx", C.c_void_p), ("sendrecv", SENDRECV), ("allreduce", ALLREDUCE)]


def torch_host_transport(group=None):
    """an fsa_host_transport over torch.distributed (any backend that moves CPU tensors: gloo)"""
    import torch
    import torch.distributed as dist

    def view(ptr, nbytes, dtype=torch.uint8):
        return torch.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype)

    def sendrecv(_ctx, peer, send, send_bytes, recv, recv_bytes):
        try:
            reqs = [dist.isend(view(send, send_bytes), peer, group=group), dist.irecv(view(recv, recv_bytes), peer, group=group)]
            for r in reqs:
                r.wait()
            return 0
        except Exception as exc:  # noqa: BLE001  (an exception must not cross the C frames)
            print(f"host transport: {exc}", file=sys.stderr)
            return 1

    def allreduce(_ctx, buf, count, what):
        try:
            dt, size, op = ((torch.float64, 8, dist.ReduceOp.SUM), (torch.float32, 4, dist.ReduceOp.SUM),
                            (torch.int32, 4, dist.ReduceOp.MAX))[what]
            dist.all_reduce(view(buf, count * size, dt), op=op, group=group)
            return 0
        except Exception as exc:  # noqa: BLE001
            print(f"host transport: {exc}", file=sys.stderr)
            return 1

    tr = HostTransport(None, HostTransport.SENDRECV(sendrecv), HostTransport.ALLREDUCE(allreduce))
    return tr



class GridTeam:
    """fsa_grid_team: the grid solver on z-slabs.  Fields go in and out in the single solver's layout (the whole
    grid).  GridTeam(dim, cuts, ...) holds all slabs in this process; GridTeam.rccl(...) holds slab `rank`, and so
    does GridTeam.hosted(...) with a caller-supplied transport (torch_host_transport)."""

    FACE, CELL = GridSolver.FACE, GridSolver.CELL

    def __init__(self, dim, cuts, reach=1, padding=1, h=0.02, dt=0.001, rho=1.0, precision=PRECISION_F32, device: int = 0,
                 rank=None, uid: bytes = None, transport=None):
        self.lib = load()
        self._transport = transport  # (the callbacks must outlive the team)
        self.precision = precision
        self.dtype = np.float32 if precision == PRECISION_F32 else np.float64
        self.dim = tuple(int(d) for d in dim)
        self.padding = int(padding)
        self.cuts = [int(c) for c in cuts]
        world = len(self.cuts) - 1
        h_ = C.c_void_p()
        cdim, ccuts = (C.c_int * 3)(*self.dim), (C.c_int * len(self.cuts))(*self.cuts)
        if rank is None:
            check(self.lib.fsa_grid_team_create_local(cdim, self.padding, float(h), float(dt), float(rho), precision,
                                                      device, world, ccuts, int(reach), C.byref(h_)))
        elif transport is not None:
            check(self.lib.fsa_grid_team_create_hosted(cdim, self.padding, float(h), float(dt), float(rho), precision,
                                                       device, int(rank), world, C.addressof(transport), ccuts, int(reach),
                                                       C.byref(h_)))
        else:
            check(self.lib.fsa_grid_team_create_rccl(cdim, self.padding, float(h), float(dt), float(rho), precision,
                                                     device, int(rank), world, uid, ccuts, int(reach), C.byref(h_)))
        self._h = h_
        nx, ny, nz = self.dim
        pd = self.padding
        self.shapes = {"u": (nz + 2 * pd, ny + 2 * pd, nx + 2 * pd + 1), "v": (nz + 2 * pd, ny + 2 * pd + 1, nx + 2 * pd),
                       "w": (nz + 2 * pd + 1, ny + 2 * pd, nx + 2 * pd)}

    @classmethod
    def rccl(cls, dim, cuts, rank, uid, **kw):
        return cls(dim, cuts, rank=rank, uid=uid, **kw)

    @classmethod
    def hosted(cls, dim, cuts, rank, transport, **kw):
        return cls(dim, cuts, rank=rank, transport=transport, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fsa_grid_team_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shape(self, name):
        return self.shapes[name[0]] if name in self.FACE else (self.dim[2], self.dim[1], self.dim[0])

    def get(self, name: str) -> np.ndarray:
        out = np.zeros(self.shape(name), self.dtype)
        check(self.lib.fsa_grid_team_get_field(self._h, name.encode(), _ptr(out), out.nbytes))
        return out

    def set(self, name: str, value) -> None:
        a = np.ascontiguousarray(value, self.dtype)
        if a.shape != self.shape(name):
            raise ValueError(f"field {name}: shape {a.shape}, expected {self.shape(name)}")
        check(self.lib.fsa_grid_team_set_field(self._h, name.encode(), _ptr(a), a.nbytes))

    def set_particles(self, pos):
        a = np.ascontiguousarray(pos, self.dtype).reshape(-1, 3)
        check(self.lib.fsa_grid_team_set_particles(self._h, len(a), _ptr(a)))

    def get_particles(self):
        n = C.c_int()
        check(self.lib.fsa_grid_team_get_particles(self._h, None, None, C.byref(n)))
        pos, col = np.empty((n.value, 3), self.dtype), np.empty((n.value, 4), self.dtype)
        check(self.lib.fsa_grid_team_get_particles(self._h, _ptr(pos), _ptr(col), C.byref(n)))
        return pos, col

    def advect(self): check(self.lib.fsa_grid_team_advect(self._h))
    def advect_particles(self): check(self.lib.fsa_grid_team_advect_particles(self._h))
    def body_force(self): check(self.lib.fsa_grid_team_body_force(self._h))
    def make_rhs(self): check(self.lib.fsa_grid_team_make_rhs(self._h))
    def update_u(self): check(self.lib.fsa_grid_team_update_u(self._h))
    def step(self, nsteps=1, maxiter=-1): check(self.lib.fsa_grid_team_step(self._h, int(nsteps), int(maxiter)))

    def sync(self) -> int:
        """waits for the team's work; raises when a back-trace left the planes a slab holds"""
        bits = C.c_int()
        check(self.lib.fsa_grid_team_sync(self._h, C.byref(bits)))
        return bits.value

    def div_velocity(self) -> float:
        s = C.c_double()
        check(self.lib.fsa_grid_team_div_velocity(self._h, C.byref(s)))
        return s.value

    def solve(self, maxiter=-1):
        it, r = C.c_int(), C.c_double()
        check(self.lib.fsa_grid_team_solve(self._h, int(maxiter), C.byref(it), C.byref(r)))
        return it.value, r.value


class SlabGroup:
    """fsa_slab_group: the z-slab step of several solvers without host round trips (see the header)."""

    def __init__(self, handle, solvers):
        self.lib = load()
        self._h = handle
        self.solvers = solvers  # kept alive

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(SLAB_GROUP_ID_BYTES)
        check(load().fsa_slab_group_unique_id(buf))
        return buf.raw

    @classmethod
    def local(cls, solvers, cuts, halo, record_cap):
        lib = load()
        hs = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
        h = C.c_void_p()
        check(lib.fsa_slab_group_create_local(len(solvers), hs, (C.c_int * len(cuts))(*cuts), int(halo),
                                              int(record_cap), C.byref(h)))
        return cls(h, list(solvers))

    @classmethod
    def hosted(cls, solver, rank, world, transport, cuts, halo, record_cap):
        lib = load()
        h = C.c_void_p()
        check(lib.fsa_slab_group_create_hosted(solver._h, int(rank), int(world), C.addressof(transport),
                                               (C.c_int * len(cuts))(*cuts), int(halo), int(record_cap), C.byref(h)))
        g = cls(h, [solver])
        g._transport = transport  # (the callbacks must outlive the group)
        return g

    @classmethod
    def rccl(cls, solver, rank, world, uid: bytes, cuts, halo, record_cap):
        lib = load()
        h = C.c_void_p()
        check(lib.fsa_slab_group_create_rccl(solver._h, int(rank), int(world), uid, (C.c_int * len(cuts))(*cuts),
                                             int(halo), int(record_cap), C.byref(h)))
        return cls(h, [solver])

    def step(self, nsteps=1):
        check(self.lib.fsa_slab_group_step(self._h, int(nsteps)))

    def sync(self) -> SlabGroupStats:
        st = SlabGroupStats()
        check(self.lib.fsa_slab_group_sync(self._h, C.byref(st)))
        return st

    def rccl_loopback(self, src_ptr, dst_ptr, nbytes):
        check(self.lib.fsa_slab_group_rccl_loopback(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), int(nbytes)))

    def close(self):
        if self._h:
            self.lib.fsa_slab_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class SlabIisphTeam:
    """fsa_slab_iisph_*: the IISPH step on slabs that all live in this process (the solvers must have been prepared
    with Solver.slab_enable_iisph and hold the particles they own)."""

    def __init__(self, solvers, cuts, halo, record_cap, n_global):
        self.lib = load()
        self.solvers = list(solvers)
        arr = (C.c_void_p * len(self.solvers))(*[s._h for s in self.solvers])
        self._h = C.c_void_p()
        check(self.lib.fsa_slab_iisph_create_local(len(self.solvers), arr, (C.c_int * len(cuts))(*cuts), int(halo),
                                                   int(record_cap), int(n_global), C.byref(self._h)))

    def setup(self): check(self.lib.fsa_slab_iisph_setup(self._h))
    def step(self, nsteps=1): check(self.lib.fsa_slab_iisph_step(self._h, int(nsteps)))

    def stats(self, last_rounds=0):
        """-> (steps, lost, records_sent, [rounds of the last `last_rounds` steps]); waits for the device"""
        st, lo, se = C.c_longlong(), C.c_longlong(), C.c_longlong()
        r = (C.c_int * max(last_rounds, 1))()
        check(self.lib.fsa_slab_iisph_stats(self._h, C.byref(st), C.byref(lo), C.byref(se), r, int(last_rounds)))
        return st.value, lo.value, se.value, list(r)[:last_rounds]

    def close(self):
        if self._h:
            self.lib.fsa_slab_iisph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlabDfsphTeam:
    """fsa_slab_dfsph_*: the DFSPH step on slabs that all live in this process (the solvers must have been prepared
    with Solver.slab_enable_dfsph and hold the particles they own)."""

    def __init__(self, solvers, cuts, halo, record_cap, maxiter=5, ethres=1.0):
        self.lib = load()
        self.solvers = list(solvers)
        arr = (C.c_void_p * len(self.solvers))(*[s._h for s in self.solvers])
        self._h = C.c_void_p()
        check(self.lib.fsa_slab_dfsph_create_local(len(self.solvers), arr, (C.c_int * len(cuts))(*cuts), int(halo),
                                                   int(record_cap), int(maxiter), float(ethres), C.byref(self._h)))

    def setup(self): check(self.lib.fsa_slab_dfsph_setup(self._h))
    def step(self, nsteps=1): check(self.lib.fsa_slab_dfsph_step(self._h, int(nsteps)))

    def stats(self, last_rounds=0):
        """-> (steps, lost, records_sent, [density-loop rounds], [divergence-loop rounds]) of the last steps; waits"""
        st, lo, se = C.c_longlong(), C.c_longlong(), C.c_longlong()
        a, b = (C.c_int * max(last_rounds, 1))(), (C.c_int * max(last_rounds, 1))()
        check(self.lib.fsa_slab_dfsph_stats(self._h, C.byref(st), C.byref(lo), C.byref(se), a, b, int(last_rounds)))
        return st.value, lo.value, se.value, list(a)[:last_rounds], list(b)[:last_rounds]

    def close(self):
        if self._h:
            self.lib.fsa_slab_dfsph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
