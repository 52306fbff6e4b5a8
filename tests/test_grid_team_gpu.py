"""The grid solver on z-slabs (fsa_grid_team_*) against the single grid solver from the same state: every
per-cell and per-face pass bit for bit, the conjugate-gradient solve and whole steps bit for bit in fp32 (the
totals of a round are added up in double slab after slab, which rounds to the same float) and to 1e-10 of the
field's scale in fp64."""
import numpy as np
import pytest

from fluidsolver_amd import capi
from util import rel_err

pytestmark = pytest.mark.gpu

# (dim, padding, precision, cuts, reach): the shipped configuration, a grid whose rows take the one-cell stencil,
# one with the four-cell stencil, one with the z-marching stencil and uneven slabs, a wider padding and reach
CASES = [((16, 32, 16), 1, "f32", [0, 16], 1), ((16, 32, 16), 1, "f32", [0, 7, 16], 1),
         ((16, 32, 16), 1, "f64", [0, 5, 10, 16], 1), ((5, 7, 9), 2, "f32", [0, 4, 9], 1),
         ((12, 5, 10), 1, "f64", [0, 3, 6, 10], 1), ((64, 10, 20), 1, "f32", [0, 9, 20], 2),
         ((64, 37, 19), 1, "f32", [0, 4, 8, 12, 19], 2)]


@pytest.fixture(params=CASES, ids=lambda c: f"{c[0][0]}x{c[0][1]}x{c[0][2]}-pad{c[1]}-{c[2]}-{len(c[3]) - 1}slabs")
def pair(request):
    dim, pad, prec, cuts, reach = request.param
    precision = capi.PRECISION_F32 if prec == "f32" else capi.PRECISION_F64
    g = capi.GridSolver(dim, padding=pad, h=0.02, dt=0.001, rho=1.3, precision=precision)
    t = capi.GridTeam(dim, cuts, reach=reach, padding=pad, h=0.02, dt=0.001, rho=1.3, precision=precision)
    yield g, t, prec
    t.close()
    g.close()


def randomise(g, t, seed, scale=0.5, pad_too=False):
    rng = np.random.default_rng(seed)
    pd = g.padding
    for k in ("u", "v", "w"):
        f = (scale * rng.standard_normal(g.shape(k))).astype(g.dtype)
        if not pad_too:  # the padding around the grid stays zero, as the solvers leave it
            inner = np.zeros_like(f)
            inner[pd:-pd, pd:-pd, pd:-pd] = f[pd:-pd, pd:-pd, pd:-pd]
            f = inner
        g.set(k, f)
        t.set(k, f)


def same(a, b, prec, exact=True):
    a, b = np.asarray(a), np.asarray(b)
    if a.tobytes() == b.tobytes():
        return True
    return (not exact or prec == "f64") and rel_err(a, b) <= 1e-10


def test_fields_come_back_as_they_went_in(pair):
    g, t, prec = pair
    randomise(g, t, 3, pad_too=True)
    for k in ("u", "v", "w"):
        assert np.array_equal(t.get(k), g.get(k)), k
    p = np.random.default_rng(4).standard_normal(g.shape("p")).astype(g.dtype)
    t.set("p", p)
    assert np.array_equal(t.get("p"), p)


def test_advection_tracers_and_gravity(pair):
    g, t, prec = pair
    randomise(g, t, 11, scale=2.0)
    rng = np.random.default_rng(12)
    lo = np.array([g.xmin[0], 0.0, g.xmin[2]])
    pts = (lo + rng.random((700, 3)) * np.array(g.dim) * 0.02).astype(g.dtype)
    pts[:30] += 3 * 0.02  # a few outside the grid, above the last slab among them
    pts[30:40, 2] -= 0.02 * g.dim[2]  # and below the first
    for s in (g, t):
        s.set_particles(pts)
        s.advect()
        s.advect_particles()
        s.body_force()
    for k in ("u", "v", "w"):
        assert np.array_equal(t.get(k), g.get(k)), k
    (pg, cg), (pt, ct) = g.get_particles(), t.get_particles()
    assert np.array_equal(pt, pg) and np.array_equal(ct, cg)
    # a second round: the copies of the neighbours' planes were refreshed, gravity included
    for s in (g, t):
        s.advect()
        s.advect_particles()
    for k in ("u", "v", "w"):
        assert np.array_equal(t.get(k), g.get(k)), k
    assert np.array_equal(t.get_particles()[0], g.get_particles()[0])
    assert t.sync() == 0


def test_divergence_right_hand_side_and_velocity_update(pair):
    g, t, prec = pair
    randomise(g, t, 21)
    for s in (g, t):
        s.make_rhs()
    assert np.array_equal(t.get("divU"), g.get("divU")) and np.array_equal(t.get("b"), g.get("b"))
    sg, st = g.div_velocity(), t.div_velocity()
    assert abs(sg - st) <= 1e-13 * sg
    p = np.random.default_rng(22).standard_normal(g.shape("p")).astype(g.dtype)
    for s in (g, t):
        s.set("p", p)
        s.update_u()
    for k in ("u", "v", "w"):
        assert np.array_equal(t.get(k), g.get(k)), k
    # and the refreshed copies feed the next advection
    for s in (g, t):
        s.advect()
    for k in ("u", "v", "w"):
        assert np.array_equal(t.get(k), g.get(k)), k


def test_pressure_solve(pair):
    g, t, prec = pair
    randomise(g, t, 41, scale=0.1)
    for s in (g, t):
        s.make_rhs()
    it_g, res_g = g.solve(2000)
    it_t, res_t = t.solve(2000)
    assert res_t < 1e-6
    if prec == "f32":
        assert it_t == it_g and res_t == res_g
    else:
        assert abs(it_t - it_g) <= 1
    for k in ("p", "r", "q", "Aq"):
        assert same(t.get(k), g.get(k), prec, exact=k == "p" or it_t == it_g), k


def test_solve_with_an_iteration_cap_and_with_nothing_to_do(pair):
    g, t, prec = pair
    for s in (g, t):
        s.make_rhs()
    assert t.solve(50)[0] == 0 and np.all(t.get("p") == 0)
    randomise(g, t, 51, scale=0.1)
    for s in (g, t):
        s.make_rhs()
    assert g.solve(3)[0] == 3 and t.solve(3)[0] == 3
    for k in ("p", "q"):
        assert same(t.get(k), g.get(k), prec), k


def test_five_steps(pair):
    g, t, prec = pair
    randomise(g, t, 61, scale=0.3)
    rng = np.random.default_rng(62)
    pts = (np.array([g.xmin[0], 0.0, g.xmin[2]]) + rng.random((300, 3)) * np.array(g.dim) * 0.02).astype(g.dtype)
    for s in (g, t):
        s.set_particles(pts)
        s.step(5, 2000)
    assert t.sync() == 0
    if prec == "f32":
        for k in ("u", "v", "w", "p", "divU"):
            assert np.array_equal(t.get(k), g.get(k)), k
        assert np.array_equal(t.get_particles()[0], g.get_particles()[0])
    else:
        # a round more or less in one of the solves (the mean residual crosses 1e-6 between two rounds whose
        # totals differ in the last bits) moves the result by about the solve's own tolerance
        for k in ("u", "v", "w", "p"):
            assert rel_err(t.get(k), g.get(k)) <= 1e-5, k
    assert t.div_velocity() < 1e-5


def test_a_back_trace_that_leaves_the_slab_is_reported():
    dim, cuts = (16, 8, 16), [0, 8, 16]
    t = capi.GridTeam(dim, cuts, reach=1)
    w = np.zeros(t.shape("w"), np.float32)
    w[:] = 100.0  # 100 m/s x 0.001 s = 5 cells per step
    t.set("w", w)
    t.advect()
    with pytest.raises(capi.FsaError, match="reach"):
        t.sync()
    t.close()
    # the same field with a reach of six planes is fine
    t = capi.GridTeam(dim, cuts, reach=6)
    g = capi.GridSolver(dim)
    for s in (g, t):
        s.set("w", w)
        s.advect()
    assert t.sync() == 0
    for k in ("u", "v", "w"):
        assert np.array_equal(t.get(k), g.get(k)), k
    t.close()
    g.close()


def test_one_rank_over_rccl_equals_the_single_solver():
    """the all-reduce and the stream ordering of the RCCL kind with the one rank a one-GPU box allows"""
    dim = (16, 12, 8)
    g = capi.GridSolver(dim)
    t = capi.GridTeam.rccl(dim, [0, 8], 0, capi.SlabGroup.unique_id(), reach=1)
    randomise(g, t, 71, scale=0.3)
    pts = (np.array([g.xmin[0], 0.0, g.xmin[2]]) + np.random.default_rng(72).random((50, 3)) * np.array(dim) * 0.02)
    for s in (g, t):
        s.set_particles(pts.astype(np.float32))
        s.step(3, 500)
    for k in ("u", "v", "w", "p"):
        assert np.array_equal(t.get(k), g.get(k)), k
    assert np.array_equal(t.get_particles()[0], g.get_particles()[0])
    assert np.array_equal(t.get_particles()[1], g.get_particles()[1])
    assert t.sync() == 0
    t.close()
    g.close()


def _assemble(parts, name, cuts, pad, shape):
    """the whole grid's field from what every rank wrote for the planes it owns"""
    out = np.zeros(shape, parts[0][name].dtype)
    world = len(cuts) - 1
    face = name in ("u", "v", "w")
    for r, part in enumerate(parts):
        lo, hi = cuts[r], cuts[r + 1]
        if face:
            lo, hi = lo + pad, hi + pad
            if r == 0:
                lo = 0
            if r == world - 1:
                hi = shape[0]
        out[lo:hi] = part[name][lo:hi]
        rest = part[name].copy()
        rest[lo:hi] = 0
        assert not rest.any(), f"rank {r} wrote planes of {name} that it does not own"
    return out


@pytest.mark.parametrize("dim,cuts,prec", [((16, 12, 16), [0, 7, 16], "f32"), ((16, 12, 18), [0, 5, 11, 18], "f32"),
                                           ((12, 6, 12), [0, 4, 8, 12], "f64")])
def test_ranks_in_separate_processes_equal_the_single_solver(tmp_path, dim, cuts, prec):
    """one process per slab, all on this GPU, the bytes between them through a caller-supplied transport
    (torch.distributed over gloo): what RCCL teams do between GPUs, rank by rank -- first, inner and last slab,
    sums over the ranks, the same loop decisions everywhere"""
    import os
    import socket
    import subprocess
    import sys
    from grid_team_worker import initial_fields, tracers
    world = len(cuts) - 1
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "grid_team_worker.py"), "--rank", str(r), "--world",
                               str(world), "--port", str(port), "--out", str(tmp_path), "--dim", *map(str, dim), "--cuts",
                               *map(str, cuts), "--prec", prec], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=240)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    parts = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world)]
    precision = capi.PRECISION_F32 if prec == "f32" else capi.PRECISION_F64
    g = capi.GridSolver(dim, precision=precision)
    for k, f in initial_fields(g.shape, g.dtype, 91, 0.3).items():
        g.set(k, f)
    g.set_particles(tracers(dim, 0.02, g.dtype, 92))
    g.step(4, 2000)
    it, res = g.solve(7)
    for r in range(world):
        assert int(parts[r]["bits"]) == 0
        assert parts[r]["cg"][0] == it
        # every rank holds all tracers
        assert np.array_equal(parts[r]["pos"], parts[0]["pos"]) and np.array_equal(parts[r]["color"], parts[0]["color"])
    for k in ("u", "v", "w", "p", "q", "divU"):
        got, want = _assemble(parts, k, cuts, 1, g.shape(k)), g.get(k)
        if prec == "f32":
            assert np.array_equal(got, want), k
        else:
            assert rel_err(got, want) <= 1e-5, k
    pos, col = g.get_particles()
    if prec == "f32":
        assert np.array_equal(parts[0]["pos"], pos) and np.array_equal(parts[0]["color"], col)
        assert parts[0]["cg"][1] == res
    g.close()


def test_bad_teams_are_rejected():
    with pytest.raises(capi.FsaError, match="span"):
        capi.GridTeam((8, 8, 8), [0, 4, 7])
    with pytest.raises(capi.FsaError, match="planes"):
        capi.GridTeam((8, 8, 8), [0, 2, 8], reach=1)  # a slab of two planes cannot feed a neighbour's three copies
    with pytest.raises(capi.FsaError):
        capi.GridTeam((8, 8, 8), [0, 8], reach=0)
    t = capi.GridTeam((8, 8, 8), [0, 4, 8])
    with pytest.raises(capi.FsaError):  # a size that is not the whole grid's
        capi.check(t.lib.fsa_grid_team_set_field(t._h, b"p", np.zeros(4, np.float32).ctypes.data, 16))
    with pytest.raises(capi.FsaError):
        capi.check(t.lib.fsa_grid_team_get_field(t._h, b"nope", None, 0))
    t.close()
