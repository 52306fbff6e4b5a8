"""Known answers for the DFSPH part of the oracle, worked out by hand from the reference text
(solver/sph_kernel_impl.cuh:582-1018 under /root/reference/particle_solver) in float32 numpy.
The reference holds no test or fixture for this step either: parity unpinned, anchored here."""
import numpy as np

from fluidsolver_amd import scenes
from util import oracle_pair

F = np.float32


def tiny_scene(pos, vel=None, types=None, mass=1.0e-4, rest=1000.0):
    pos = np.asarray(pos, F).reshape(-1, 3)
    n = len(pos)
    sc = {"pos": pos, "vel": np.zeros((n, 3), F) if vel is None else np.asarray(vel, F).reshape(n, 3),
          "color": np.ones((n, 4), F), "normal": np.zeros((n, 3), F),
          "type": np.zeros(n, np.int32) if types is None else np.asarray(types, np.int32),
          "group": np.zeros(n, np.int32), "mass": np.full(n, mass, F), "restDensity": np.full(n, rest, F),
          "uniqueId": np.arange(n, dtype=np.int32), "params": dict(scenes.REF_PARAMS)}
    sc["params"].update(vmin=scenes.REF_DOMAIN_MIN, vmax=scenes.REF_DOMAIN_MAX, boundary_mode=0)
    return sc


def test_kpoly6_constant(orc32):
    p, st = oracle_pair(orc32, tiny_scene([[0, 0, 0]]))
    sr = F(F(0.005) * F(2))
    want = F(np.float64(F(315.0)) / (np.float64(F(64.0)) * 3.141592 * np.float64(F(sr) ** F(9))))
    assert abs(p.kpoly6 / want - 1) < 1e-6
    st.close()


def test_isolated_particle_density_and_factor(orc32):
    p, st = oracle_pair(orc32, tiny_scene([[0.0, 0.2, 0.0]]))
    st.sort_particles(p)
    st.compute_density_alpha(p)
    sr2 = F(p.smoothradius) * F(p.smoothradius)
    dens = F(F(F(sr2 * sr2) * sr2) * F(1.0e-4)) * F(p.kpoly6)
    assert st["density"][0] == dens
    assert st["DF_factor"][0] == dens / F(0.000001)  # the denominator clamp
    st.close()


def test_pair_density_alpha_by_hand(orc32):
    a, b = np.array([0.0, 0.2, 0.0], F), np.array([0.004, 0.203, 0.0], F)
    p, st = oracle_pair(orc32, tiny_scene([a, b]))
    st.sort_particles(p)
    st.compute_density_alpha(p)
    sr, k, m = F(p.smoothradius), F(p.kspikydiff), F(1.0e-4)
    sr2 = sr * sr
    for i, (xi, xj) in enumerate(((a, b), (b, a))):
        x = xi - xj
        d2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2]
        d = np.sqrt(d2)
        c2, c = sr2 - d2, sr - d
        nablaw = k * c * c / d
        dens = F(0) + c2 * c2 * c2 * m
        av = x * nablaw * m
        mw = av[0] * av[0] + av[1] * av[1] + av[2] * av[2]
        dens = (dens + sr2 * sr2 * sr2 * m) * F(p.kpoly6)
        denom = (av[0] * av[0] + av[1] * av[1] + av[2] * av[2]) + mw
        assert st["density"][i] == dens
        assert st["DF_factor"][i] == dens / denom
    st.close()


def test_non_pressure_force_isolated_and_non_fluid(orc32):
    sc = tiny_scene([[0.0, 0.2, 0.0], [0.2, 0.2, 0.2]], vel=[[0.1, 0.2, 0.3], [1, 1, 1]], types=[0, 4])
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.compute_density_alpha(p)
    st.compute_non_pressure_force(p)
    i = int(np.where(st["type"] == 0)[0][0])
    dt = F(p.dt)
    want = np.array([F(0.1) + (F(0) + F(0)) * dt, F(0.2) + (F(0) + F(-9.8)) * dt, F(0.3) + (F(0) + F(0)) * dt], F)
    assert np.array_equal(st["v_star"][i], want)
    assert np.array_equal(st["v_star"][1 - i], np.zeros(3, F))
    st.close()


def test_update_velocities_with_zero_limit(orc32):
    """AccelerationLimit reads as 0 from the shipped file: the drift is scaled to zero and v_star ends
    as the gravity parameter (0, -10, 0), not the force literal"""
    sc = tiny_scene([[0.0, 0.2, 0.0]], vel=[[0.5, 0.0, 0.0]])
    p, st = oracle_pair(orc32, sc)
    st["v_star"][0] = (0.7, -0.1, 0.2)
    st.update_velocities(p)
    assert np.array_equal(st["vel"][0], np.array([0.7, -0.1, 0.2], F))
    assert np.array_equal(st["v_star"][0], np.array([0.0, -10.0, 0.0], F))
    st.close()


def test_density_solver_reduces_compression(orc32):
    """a compressed block: the Jacobi rounds must bring the predicted density error down"""
    sc = scenes.dam_break(10, 10, 10, tank_x_factor=2.0, walls=False, boundary_mode=0)
    sc["pos"] = (np.asarray(sc["pos"], F) * F(0.97)).astype(F)  # 9 % denser than the rest lattice
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.compute_density_alpha(p)
    st.compute_non_pressure_force(p)
    st.solve_density_stiff(p)
    e0 = st["error"].max()
    for _ in range(5):
        st.apply_pstiff(p)
        st.solve_density_stiff(p)
    assert st["error"].max() < 0.5 * e0
    st.close()
