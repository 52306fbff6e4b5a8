"""The plumbing configuration: a ball of ~33 K fluid particles (ball.raw scaled by 0.1) dropped
onto a floor plate in the shipped grid box, 1000 steps.  The reference ships no loader for ball.raw on the
particle path, so the scene is this build's own construction (fluidsolver_amd.scenes.ball_drop)."""
import numpy as np
import pytest

from fluidsolver_amd import scenes
from util import gpu_solver, oracle_pair

G = 9.8  # the literal of the force pass


def centre_of_mass(pos, types):
    return pos[types == 0].astype(np.float64).mean(0)


def test_ball_drop_1000_steps_on_the_oracle(orc32):
    sc = scenes.ball_drop()
    assert 32000 < sc["n_fluid"] < 35000
    p, st = oracle_pair(orc32, sc)
    n = len(sc["pos"])
    y0 = centre_of_mass(st["pos"], st["type"])[1]
    st.solve_sph(p, 400)  # the ball touches the floor after 0.2 s = 500 steps
    t = 400 * sc["params"]["dt"]
    com = centre_of_mass(st["pos"], st["type"])
    assert abs(com[1] - (y0 - 0.5 * G * t * t)) < 2e-3, "free fall"
    assert abs(com[0]) < 1e-3 and abs(com[2]) < 1e-3
    st.solve_sph(p, 600)
    assert np.isfinite(st["pos"]).all() and np.isfinite(st["vel"]).all()
    assert np.array_equal(np.sort(st["uniqueId"]), np.arange(n))
    com = centre_of_mass(st["pos"], st["type"])
    assert -0.05 < com[1] < 0.12, "the ball has landed and spread on the plate"
    # With the shipped PressureK (sound speed ~2.8 m/s) the 2 m/s impact squeezes part of the ball through
    # the one-layer penalty plate; those particles end on the bottom of the grid box (legacy clamp) and
    # leave the grid.  Most of the ball must still be there.
    assert (st["particleHash"] == 99999999).sum() < 0.4 * sc["n_fluid"]
    st.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 2, 3])
def test_ball_drop_gpu_follows_oracle(orc32, variant):
    sc = scenes.ball_drop()
    p_o, st = oracle_pair(orc32, sc)
    p_g, s = gpu_solver(sc)
    s.set_variant(variant)
    st.solve_sph(p_o, 400)
    s.step(400)
    a = s.get("pos")[np.argsort(s.get("uniqueId"))].astype(np.float64)
    b = st["pos"][np.argsort(st["uniqueId"])].astype(np.float64)
    # 400 steps of free fall: rounding differences stay at rounding level (1e-5 of the 1.1 m box)
    assert np.abs(a - b).max() < 1.1e-5
    # after the impact trajectories separate particle by particle; the bulk must agree
    st.solve_sph(p_o, 600)
    s.step(600)
    cg = centre_of_mass(s.get("pos"), s.get("type"))
    co = centre_of_mass(st["pos"], st["type"])
    assert np.abs(cg - co).max() < 5e-3
    assert abs(s.num_undef() - int((st["particleHash"] == 99999999).sum())) < 0.02 * sc["n_fluid"]
    s.close()
    st.close()
