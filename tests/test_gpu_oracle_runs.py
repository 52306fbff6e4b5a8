"""Whole runs beside the oracle, bit for bit: the three steps of the library on the exact-arithmetic kernels (variant 1:
tiled passes, recorded neighbour lists, the reference's expression order) against the plain loops of the CPU oracle,
through a flow that develops -- the same order of particles and the same bits in every array at every check.  Not a
tolerance anywhere: one rounding that differed in one of the millions of sums would show, and stay.
(tools/oracle_soak.py runs the same comparison longer and larger.)"""
import numpy as np
import pytest

from fluidsolver_amd import scenes
from test_gpu_iisph import setup_both
from util import gpu_solver, oracle_pair

pytestmark = pytest.mark.gpu


def scene_for(step):
    if step == "iisph":  # a dam break in a tank of rigid particles, the parameters of the shipped IISPH scene
        sc = scenes.dam_break(40, 24, 24, wall_type=scenes.TYPE_RIGID)
        rigid = sc["type"] == scenes.TYPE_RIGID
        sc["mass"][rigid] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
        sc["group"][rigid] = 1
        sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
        return sc
    if step == "ddam":
        return scenes.double_dam_break(24, 24, 24)
    return scenes.dam_break(40, 24, 24)


@pytest.mark.parametrize("step,steps,every", [("wcsph", 400, 100), ("ddam", 300, 100), ("dfsph", 150, 50),
                                              ("iisph", 150, 50)])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_run_equals_the_oracle_bit_for_bit(orc32, orc64, precision, step, steps, every):
    """f64: the fp64 solvers (generic kernels; the reference has no fp64 path) against the fp64 build of the oracle --
    the criterion there is 1e-10, what is measured is equality"""
    sc = scene_for(step)
    f64 = precision == "f64"
    real = np.float64 if f64 else np.float32
    p_o, st = oracle_pair(orc64 if f64 else orc32, sc)
    p_g, s = gpu_solver(sc, 1 if f64 else 0)
    if not f64:
        s.set_variant(1)
    if step == "dfsph":
        st.sort_particles(p_o)
        st.compute_density_alpha(p_o)
        s.setup_dfsph()
    elif step == "iisph":
        setup_both(p_o, st, s)
    moved = 0.0
    for done in range(every, steps + 1, every):
        if step in ("wcsph", "ddam"):
            st.solve_sph(p_o, every)
            s.step(every)
        elif step == "dfsph":
            st.solve_dfsph(p_o, every)
            s.step_dfsph(every)
        else:
            st.solve_iisph(p_o, every, mean_in_double=True)
            s.step_iisph(every)
        assert np.array_equal(s.get("uniqueId"), st["uniqueId"]), (step, done)
        for k in ("pos", "vel", "density", "pressure"):
            assert s.get(k).tobytes() == np.ascontiguousarray(st[k], real).tobytes(), (step, done, k)
        fl = st["type"] == 0
        if step in ("wcsph", "ddam"):  # (force[] of other types is never written by either side)
            assert s.get("force")[fl].tobytes() == np.ascontiguousarray(st["force"], real)[fl].tobytes(), (step, done)
        moved = max(moved, float(np.linalg.norm(st["vel"][fl], axis=1).max()))
    assert moved > 0.3, "the flow was meant to develop"
    s.close()
    st.close()
