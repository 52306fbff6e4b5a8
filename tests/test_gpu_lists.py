"""Recorded neighbour lists: the passes of the IISPH and DFSPH steps that replay them give, bit for bit, what the
searching (tiled) kernels give, including particles whose list overflows and blocks that cannot be staged."""
import numpy as np
import pytest

from fluidsolver_amd import capi, scenes
from util import gpu_solver

pytestmark = pytest.mark.gpu

IISPH_OUT = ("pos", "vel", "pressure", "density", "dii", "aii", "density_star", "dijpjl", "pressureForce", "v_star",
             "uniqueId")
DFSPH_OUT = ("pos", "vel", "density", "DF_factor", "pstiff", "v_star", "uniqueId")


def iisph_scene(crowded=False):
    sc = scenes.iisph_tank(12, 14, 10, wall_group=1, fluid_groups=(0,), boundary_visc=0.1, boundary_friction=1.0)
    sc = scenes.jittered(sc, seed=41, pos_jitter=0.15, vel_scale=0.3, fluid_only=True)
    if crowded:  # squeeze part of the fluid: more than 64 neighbours for some particles
        pos = np.array(sc["pos"])
        fl = np.where(np.asarray(sc["type"]) == 0)[0][:400]
        c = pos[fl].mean(0)
        pos[fl] = (c + (pos[fl] - c) * np.float32(0.45)).astype(np.float32)
        sc = dict(sc, pos=pos)
    return sc


def run_iisph(sc, variant, lists, steps):
    p, s = gpu_solver(sc)
    s.set_variant(variant)
    s.set_neighbour_lists(lists)
    s.setup_iisph()
    s.step_iisph(steps)
    out = {k: s.get(k) for k in IISPH_OUT}
    s.close()
    return out


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("crowded", [False, True], ids=["plain", "crowded"])
def test_iisph_with_lists_equals_iisph_with_search(variant, crowded):
    sc = iisph_scene(crowded)
    steps = 1 if crowded else 3  # the squeezed blob explodes after a step; one step covers the overflow path
    a = run_iisph(sc, variant, True, steps)
    b = run_iisph(sc, variant, False, steps)
    for k in IISPH_OUT:
        assert a[k].tobytes() == b[k].tobytes(), k


def test_crowded_scene_does_overflow_a_list():
    """the scene above really has particles with more than 64 neighbours in the support"""
    sc = iisph_scene(True)
    pos = np.asarray(sc["pos"], np.float64)
    fl = np.where(np.asarray(sc["type"]) == 0)[0][:400]
    sr = float(sc["params"]["spacing"]) * float(sc["params"]["smoothratio"])
    d = np.linalg.norm(pos[fl][:, None, :] - pos[None, fl, :], axis=2)
    assert ((d < sr).sum(1) > 64).any()


@pytest.mark.parametrize("variant", [1, 2])
def test_dfsph_with_lists_equals_dfsph_with_search(variant):
    sc = scenes.jittered(scenes.dam_break(16, 12, 10), seed=43, pos_jitter=0.15, vel_scale=0.3, fluid_only=True)

    def run(lists):
        p, s = gpu_solver(sc)
        s.set_variant(variant)
        s.set_neighbour_lists(lists)
        s.setup_dfsph()
        s.step_dfsph(3)
        out = {k: s.get(k) for k in DFSPH_OUT}
        s.close()
        return out

    a, b = run(True), run(False)
    for k in DFSPH_OUT:
        assert a[k].tobytes() == b[k].tobytes(), k


def test_lists_are_rebuilt_after_positions_change_without_a_sort():
    """a pass after fsa_set_buffer("pos") must not replay the list of the old positions"""
    sc = iisph_scene()
    outs = []
    for lists in (True, False):
        p, s = gpu_solver(sc)
        s.set_neighbour_lists(lists)
        s.setup_iisph()
        s.non_pressure_force_multiphase()  # records (when enabled)
        pos = s.get("pos")
        rng = np.random.default_rng(3)
        moved = (pos + rng.normal(0, 0.0005, pos.shape)).astype(np.float32)
        s.set("pos", moved)  # same cells for nearly all, different neighbourhoods
        s.sort_particles()
        s.iisph_factor()
        outs.append({k: s.get(k) for k in ("density", "dii", "uniqueId")})
        s.close()
    for k in outs[0]:
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k
