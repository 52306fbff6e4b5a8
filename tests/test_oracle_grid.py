"""The oracle's restatement of the Eulerian grid solver against answers that can be worked out by hand.
(The reference ships no fixtures for this solver and cannot be built here: these anchor the restatement,
they do not pin it to the reference.)"""
import numpy as np
import pytest

from oracle.orc import GridOracle


@pytest.fixture(params=["f32", "f64"])
def grid(request):
    g = GridOracle(request.param, (6, 8, 5), padding=1, h=0.02, dt=0.001, rho=1.0)
    yield g
    g.close()


def test_sizes_and_low_corner_follow_set_size_and_load_config(grid):
    assert grid["u"].shape == (7, 10, 9) and grid["v"].shape == (7, 11, 8) and grid["w"].shape == (8, 10, 8)
    assert grid["p"].shape == (5, 8, 6)
    # -dim/2*h with integer division: -6/2 = -3, -5/2 = -2
    assert np.allclose(grid.xmin, [-3 * 0.02, 0.0, -2 * 0.02])


def test_divergence_of_the_testcase_field_is_nine_over_h(grid):
    grid.testcase()  # u = 2 i, v = 3 j, w = 4 k
    grid.div_velocity()
    want = (2 + 3 + 4) / grid.h
    assert np.allclose(grid["divU"], want, rtol=1e-6)
    assert abs(grid.div_u_sum - want * want * 6 * 8 * 5) <= 1e-5 * grid.div_u_sum


def test_laplacian_of_a_constant_counts_the_open_sides(grid):
    one = np.ones((5, 8, 6), grid.dtype)
    out = grid.mvproduct(one)
    nz, ny, nx = out.shape
    want = np.zeros_like(out)
    want[:, :, 0] += 1; want[:, :, -1] += 1     # x sides: p = 0 outside
    want[:, -1, :] += 1                          # top; the floor (j = 0) is a wall and adds nothing
    want[0, :, :] += 1; want[-1, :, :] += 1      # z sides
    assert np.array_equal(out, want)


def test_laplacian_is_symmetric(grid):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((5, 8, 6)).astype(grid.dtype)
    b = rng.standard_normal((5, 8, 6)).astype(grid.dtype)
    lhs = float((grid.mvproduct(a).astype(np.float64) * b).sum())
    rhs = float((grid.mvproduct(b).astype(np.float64) * a).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


def test_sampling_returns_face_values_at_face_positions_and_means_between(grid):
    rng = np.random.default_rng(2)
    grid["u"][...] = rng.standard_normal(grid["u"].shape)
    h, x0 = grid.h, grid.xmin
    pd = grid.padding
    i, j, k = 2, 3, 1
    # face (i, j, k) of u sits at x = x0 + i h, y = (j + 0.5) h, z = z0 + (k + 0.5) h
    at = [x0[0] + i * h, (j + 0.5) * h, x0[2] + (k + 0.5) * h]
    want = grid["u"][k + pd, j + pd, i + pd]
    assert abs(grid.sample("u", at) - want) <= 2e-5 * max(abs(want), 1.0)
    mid = [x0[0] + (i + 0.5) * h, (j + 0.5) * h, x0[2] + (k + 0.5) * h]
    want = 0.5 * (grid["u"][k + pd, j + pd, i + pd] + grid["u"][k + pd, j + pd, i + 1 + pd])
    assert abs(grid.sample("u", mid) - want) <= 2e-5 * max(abs(want), 1.0)
    # outside the effective grid: 0
    assert grid.sample("u", [x0[0] - 0.5 * h, 0.01, 0.0]) == 0
    assert grid.sample("v", [0.0, 8 * h + 0.5 * h, 0.0]) == 0


def test_advection_of_a_uniform_x_velocity_keeps_it_in_the_interior(grid):
    c = 0.3
    grid["u"][...] = c
    grid.advect()
    pd = grid.padding
    u = grid["u"]
    # faces whose back-traced point stays inside the grid sample a constant field
    assert np.allclose(u[pd:-pd, pd:-pd, pd + 1:-pd - 1], c, rtol=1e-5)
    assert np.all(grid["v"] == 0) and np.all(grid["w"] == 0)


def test_body_force_adds_ten_dt_to_every_v_face_of_the_effective_grid(grid):
    grid.body_force()
    pd = grid.padding
    v = grid["v"]
    inner = v[pd:-pd, pd:-pd, pd:-pd]
    assert inner.shape == (5, 9, 6)
    assert np.all(inner == grid.dtype(10 * grid.dtype(grid.dt)))
    v2 = v.copy()
    v2[pd:-pd, pd:-pd, pd:-pd] = 0
    assert np.all(v2 == 0)


def test_projection_solves_the_system_and_removes_the_divergence(grid):
    rng = np.random.default_rng(3)
    pd = grid.padding
    for k, shape in (("u", (5, 8, 7)), ("v", (5, 9, 6)), ("w", (6, 8, 6))):
        f = grid[k]
        f[pd:pd + shape[0], pd:pd + shape[1], pd:pd + shape[2]] = 0.1 * rng.standard_normal(shape)
    grid["v"][pd:-pd, pd, pd:-pd] = 0  # no flow through the floor
    grid.div_velocity()
    before = grid.div_u_sum
    grid.make_rhs()
    it, res = grid.solve(500)
    assert res < 1e-6 and 0 < it < 500
    Ap = grid.mvproduct(grid["p"].copy())
    scale = np.abs(grid["b"]).max()
    assert np.abs(Ap - grid["b"]).max() <= (2e-4 if grid.dtype == np.float32 else 1e-4) * scale
    grid.update_u()
    grid.div_velocity()
    assert grid.div_u_sum < 1e-6 * before


def test_update_u_applies_the_pressure_gradient_per_face(grid):
    rng = np.random.default_rng(4)
    p = rng.standard_normal((5, 8, 6)).astype(grid.dtype)
    grid["p"][...] = p
    grid["v"][...] = 1.0
    grid.update_u()
    pd = grid.padding
    c = grid.dt / grid.rho / grid.h
    u = grid["u"][pd:-pd, pd:-pd, pd:-pd]  # (5, 8, 7)
    assert np.allclose(u[:, :, 0], -p[:, :, 0] * c, rtol=1e-5)
    assert np.allclose(u[:, :, 1:-1], -(p[:, :, 1:] - p[:, :, :-1]) * c, rtol=1e-4, atol=1e-7)
    assert np.allclose(u[:, :, -1], p[:, :, -1] * c, rtol=1e-5)
    v = grid["v"][pd:-pd, pd:-pd, pd:-pd]  # (5, 9, 6)
    assert np.all(v[:, 0, :] == 0)  # the floor
    assert np.allclose(v[:, -1, :], 1.0 + p[:, -1, :] * c, rtol=1e-5)


def test_tracers_move_with_the_sampled_velocity(grid):
    grid["u"][...] = 0.2
    grid["v"][...] = -0.1
    grid["w"][...] = 0.05
    pts = np.array([[0.0, 0.05, 0.0], [0.01, 0.08, -0.01]], grid.dtype)
    grid.set_particles(pts)
    grid.advect_particles()
    dt = grid.dt
    assert np.allclose(grid["pos"], pts + np.array([0.2, -0.1, 0.05]) * dt, rtol=1e-5, atol=1e-8)
    assert np.allclose(grid["color"][:, 0], 0.2 / 0.01, rtol=1e-5)
    assert np.all(grid["color"][:, 1] == 0) and np.allclose(grid["color"][:, 2], 0.05, rtol=1e-5)


def test_the_shipped_configuration_runs_ten_steps_and_stays_divergence_free():
    g = GridOracle("f32", (16, 32, 16), padding=1, h=0.02, dt=0.001, rho=1.0)  # config/Eulerian.xml
    for _ in range(10):
        g.step(2000)
    assert np.isfinite(g["v"]).all()
    assert g.div_u_sum < 1e-5
    pd = 1
    assert np.all(g["v"][pd:-pd, pd, pd:-pd] == 0)  # nothing crosses the floor
    g.close()


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_oracle_reproduces_the_grid_golden(precision):
    import os
    from golden.make_golden import grid_inputs
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"grid_16x32x16_{precision}.npz")
    gold = dict(np.load(path))
    g, fields, pts = grid_inputs(precision)
    for k, v in fields.items():
        assert v.tobytes() == gold["in_" + k].tobytes()
        g[k][...] = v
    g.set_particles(pts)
    g.advect()
    for k in ("u", "v", "w"):
        assert g[k].tobytes() == gold["advected_" + k].tobytes(), k
    g.advect_particles()
    assert g["pos"].tobytes() == gold["tracers_after"].tobytes()
    g.body_force()
    g.make_rhs()
    assert g["divU"].tobytes() == gold["divU"].tobytes() and g["b"].tobytes() == gold["b"].tobytes()
    it, _ = g.solve(2000)
    assert it == int(gold["cg_iterations"])
    assert g["p"].tobytes() == gold["p"].tobytes()
    g.update_u()
    for k in ("u", "v", "w"):
        assert g[k].tobytes() == gold["projected_" + k].tobytes(), k
    g.div_velocity()
    g.step(2000)
    g.step(2000)
    assert g["p"].tobytes() == gold["p_step3"].tobytes()
    g.close()
