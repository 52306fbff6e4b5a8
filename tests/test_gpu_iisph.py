"""IISPH step on the GPU, stage by stage through the C ABI, against the oracle from the same state:
1e-5 relative in fp32, 1e-10 in fp64 (relative to the largest magnitude of each array); the integration
kernel, the carried pressure and the number of relaxation rounds exact."""
import numpy as np
import pytest

from fluidsolver_amd import capi, scenes
from util import gpu_solver, oracle_pair, rel_err

pytestmark = pytest.mark.gpu


def scene(name):
    if name == "shipped_groups":  # fluid group 0, wall object id 1: the arrangement of the shipped scene
        sc = scenes.iisph_tank(12, 14, 10, wall_group=1, fluid_groups=(0,))
        return scenes.jittered(sc, seed=31, pos_jitter=0.15, vel_scale=0.3, fluid_only=True)
    # two fluid bodies of groups 1 and 2 and a wall of group 0: the 0.5 XSPH branch, surface tension inside
    # each body only, and the wall viscosity / friction branch
    sc = scenes.iisph_tank(8, 12, 9, wall_group=0, fluid_groups=(1, 2), boundary_visc=0.1, boundary_friction=1.0)
    return scenes.jittered(sc, seed=32, pos_jitter=0.15, vel_scale=0.5, fluid_only=True)


CASES = [("shipped_groups", "f32", 2), ("shipped_groups", "f32", 0), ("wall_friction", "f32", 2),
         ("wall_friction", "f32", 0), ("wall_friction", "f64", 2)]


@pytest.fixture(params=CASES, ids=lambda p: f"{p[0]}-{p[1]}-v{p[2]}")
def pair(request, orc32, orc64):
    name, prec, variant = request.param
    sc = scene(name)
    if prec == "f64":
        sc = dict(sc)
        for k in ("pos", "vel", "color", "normal", "mass", "restDensity"):
            sc[k] = np.asarray(sc[k], np.float64)
    orc = orc32 if prec == "f32" else orc64
    p_o, st = oracle_pair(orc, sc)
    p_g, s = gpu_solver(sc, precision=capi.PRECISION_F32 if prec == "f32" else capi.PRECISION_F64)
    s.set_variant(variant)
    tol = 1e-5 if prec == "f32" else 1e-10
    yield sc, p_o, st, s, tol, variant
    s.close()
    st.close()


def sync_inputs(s, st, names):
    for k in names:
        s.set(k, st[k])


def setup_both(p, st, s):
    st.sort_particles(p)
    st.iisph_factor(p)
    st.rigid_particle_volume(p)
    s.setup_iisph()
    assert np.array_equal(s.get("uniqueId"), st["uniqueId"])


def test_iisph_stages_from_the_same_state(pair):
    sc, p, st, s, tol, variant = pair
    setup_both(p, st, s)
    fluid = st["type"] == 0
    assert rel_err(s.get("density")[fluid], st["density"][fluid]) <= tol
    assert rel_err(s.get("dii")[fluid], st["dii"][fluid]) <= tol
    assert rel_err(s.get("restDensity"), st["restDensity"]) <= tol
    sync_inputs(s, st, ("density", "dii", "restDensity"))

    st.non_pressure_force_multiphase(p)
    s.non_pressure_force_multiphase()
    vs_g, vs_o = s.get("v_star"), st["v_star"]
    assert np.isfinite(vs_o).all()
    assert rel_err(vs_g, vs_o) <= tol
    assert np.all(vs_g[~fluid] == 0)
    sync_inputs(s, st, ("v_star",))

    # a warm-start pressure for the predictor to halve
    rng = np.random.default_rng(5)
    pr = (rng.random(len(fluid)) * 200.0 * fluid).astype(st["pressure"].dtype)
    st["pressure"][...] = pr
    s.set("pressure", pr)
    st.iisph_predict_density(p)
    s.iisph_predict_density()
    assert s.get("pressure").tobytes() == st["pressure"].tobytes()
    assert rel_err(s.get("density_star")[fluid], st["density_star"][fluid]) <= tol
    assert rel_err(s.get("aii")[fluid], st["aii"][fluid]) <= tol
    sync_inputs(s, st, ("density_star", "aii"))

    st.iisph_calc_dijpjl(p)
    s.iisph_calc_dijpjl()
    assert rel_err(s.get("dijpjl")[fluid], st["dijpjl"][fluid]) <= tol
    sync_inputs(s, st, ("dijpjl",))

    st.iisph_calc_new_pressure(p)
    s.iisph_calc_new_pressure()
    # error = corrected density: compare on the scale of the density
    assert np.abs(s.get("error").astype(np.float64) - st["error"]).max() <= tol * float(st["density"].max())
    assert np.all(s.get("error")[~fluid] == 0)
    # the new pressure divides a small density difference by a_ii: absolute scale = density / |a_ii|
    pg, po = s.get("pressure").astype(np.float64), st["pressure"].astype(np.float64)
    aii = np.abs(st["aii"].astype(np.float64))
    ok = aii > 1e-8
    bound = 4 * tol * float(st["density"].max()) / aii[ok] + tol * np.abs(po[ok])
    assert np.all(np.abs(pg[ok] - po[ok]) <= bound)
    assert np.all(pg[~ok] == 0) and np.all(po[~ok] == 0)
    sync_inputs(s, st, ("pressure", "error"))

    st.iisph_calc_pressure_force(p)
    s.iisph_calc_pressure_force()
    assert rel_err(s.get("pressureForce")[fluid], st["pressureForce"][fluid]) <= tol
    sync_inputs(s, st, ("pressureForce",))

    # IISPHUpdate = force + integration; from synchronised forces the integration itself is exact
    st.iisph_update(p)
    s.iisph_update()
    scale = float(np.max(np.array(sc["params"]["vmax"]) - np.array(sc["params"]["vmin"])))
    assert rel_err(s.get("vel"), st["vel"]) <= tol
    assert np.abs(s.get("pos").astype(np.float64) - st["pos"]).max() <= tol * scale


def test_pressure_solve_rounds_and_result(pair):
    sc, p, st, s, tol, variant = pair
    setup_both(p, st, s)
    st.non_pressure_force_multiphase(p)
    s.non_pressure_force_multiphase()
    st.iisph_predict_density(p)
    s.iisph_predict_density()
    # the library reduces the mean of error[] in double; the oracle can do either
    r_o, m_o = st.iisph_solve_pressure(p, 10, mean_in_double=True)
    r_g, m_g = s.iisph_solve_pressure(10)
    assert r_g == r_o
    assert abs(m_g - m_o) <= 10 * tol * abs(m_o)
    fluid = st["type"] == 0
    # ten relaxed-Jacobi rounds amplify the per-round differences a little
    assert rel_err(s.get("pressure")[fluid], st["pressure"][fluid]) <= 100 * tol


@pytest.mark.parametrize("maxiter", [1, 2, 3, 7])
def test_pressure_solve_round_limits(pair, maxiter):
    """the rounds are queued at once and each swaps the two pressure arrays: whatever the number of rounds, asked for
    or run, "pressure" must be the array the last round wrote"""
    sc, p, st, s, tol, variant = pair
    setup_both(p, st, s)
    st.non_pressure_force_multiphase(p)
    s.non_pressure_force_multiphase()
    st.iisph_predict_density(p)
    s.iisph_predict_density()
    r_o, m_o = st.iisph_solve_pressure(p, maxiter, mean_in_double=True)
    r_g, m_g = s.iisph_solve_pressure(maxiter)
    assert r_g == r_o
    fluid = st["type"] == 0
    assert rel_err(s.get("pressure")[fluid], st["pressure"][fluid]) <= 100 * tol


def test_pressure_solve_stops_on_the_device(orc32):
    """particles too far apart to see each other: the mean of error[] is the same in every round, so the loop stops
    in its second round while ten were queued; rounds, mean and the pressure array must be the oracle's"""
    sc = scenes.iisph_tank(6, 6, 6, wall_group=1, fluid_groups=(0,))
    sc = dict(sc)
    fluid = sc["type"] == 0
    centre = sc["pos"][fluid].mean(0)
    pos = sc["pos"].copy()
    pos[fluid] = centre + (pos[fluid] - centre) * np.float32(2.6)  # 2.6 spacings apart > the support of 2
    sc["pos"] = pos
    p, st = oracle_pair(orc32, sc)
    _, s = gpu_solver(sc)
    setup_both(p, st, s)
    st.non_pressure_force_multiphase(p)
    s.non_pressure_force_multiphase()
    st.iisph_predict_density(p)
    s.iisph_predict_density()
    r_o, m_o = st.iisph_solve_pressure(p, 10, mean_in_double=True)
    r_g, m_g = s.iisph_solve_pressure(10)
    assert r_o < 10, "the scene was meant to stop the loop early"
    assert r_g == r_o
    assert abs(m_g - m_o) <= 1e-4 * max(abs(m_o), 1e-12)
    fl = st["type"] == 0
    assert rel_err(s.get("pressure")[fl], st["pressure"][fl]) <= 1e-4
    s.iisph_update()  # reads the array the last round wrote
    st.iisph_update(p)
    assert rel_err(s.get("vel"), st["vel"]) <= 1e-4
    s.close()
    st.close()


def test_five_iisph_steps(pair):
    sc, p, st, s, tol, variant = pair
    setup_both(p, st, s)
    for _ in range(5):
        # per step, so that a different number of rounds in one step would show up as such
        st.solve_iisph(p, 1, mean_in_double=True)
        s.step_iisph(1)
    assert np.array_equal(np.sort(s.get("uniqueId")), np.arange(len(st["uniqueId"])))
    a = s.get("pos")[np.argsort(s.get("uniqueId"))].astype(np.float64)
    b = st["pos"][np.argsort(st["uniqueId"])].astype(np.float64)
    scale = float(np.max(np.array(sc["params"]["vmax"]) - np.array(sc["params"]["vmin"])))
    assert np.isfinite(a).all()
    assert np.abs(a - b).max() <= 10 * tol * scale
    pa = s.get("pressure")[np.argsort(s.get("uniqueId"))]
    pb = st["pressure"][np.argsort(st["uniqueId"])]
    assert rel_err(pa, pb) <= 1e3 * tol


def test_the_sort_carries_pressure_once_the_iisph_step_is_in_use(orc32):
    sc = scene("shipped_groups")
    p_o, st = oracle_pair(orc32, sc)
    p_g, s = gpu_solver(sc)
    setup_both(p_o, st, s)
    rng = np.random.default_rng(9)
    pr = rng.random(len(sc["pos"])).astype(np.float32)
    st["pressure"][...] = pr
    s.set("pressure", pr)
    mv = (st["pos"] + rng.normal(0, 0.004, st["pos"].shape)).astype(np.float32)
    st["pos"][...] = mv
    s.set("pos", mv)
    st.sort_particles(p_o)
    s.sort_particles()
    assert np.array_equal(s.get("uniqueId"), st["uniqueId"])
    assert s.get("pressure").tobytes() == st["pressure"].tobytes()
    assert np.array_equal(s.get("group"), st["group"])
    s.close()
    st.close()


def test_iisph_entries_reject_slab_solvers_and_unsorted_states():
    sc = scene("shipped_groups")
    p_g, s = gpu_solver(sc)
    with pytest.raises(capi.FsaError):
        s.iisph_factor()  # nothing sorted yet
    s.close()
    P = dict(sc["params"])
    p = capi.make_params(capi.PRECISION_F32, **P)
    p.slab_z0, p.slab_nz = 0, p.gridres[2]
    slab = capi.Solver(p, 16)
    with pytest.raises(capi.FsaError):
        slab.setup_iisph()
    slab.close()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_effective_mass_from_volume_fractions(prec, orc32, orc64):
    """mass, rest density and colour of the non-rigid particles from a two-phase mix; rigid ones untouched"""
    sc = scene("shipped_groups")
    n = len(sc["pos"])
    rng = np.random.default_rng(17)
    a = rng.random(n)
    vf = np.stack([a, 1 - a, np.zeros(n)], 1)
    vf[::7] = (0.0, 0.0, 1.0)  # a vanishing fraction is skipped (mass_frac < EPSILON)
    dens = (500.0, 1000.0, 1500.0)
    if prec == "f64":
        sc = dict(sc)
        for k in ("pos", "vel", "color", "normal", "mass", "restDensity"):
            sc[k] = np.asarray(sc[k], np.float64)
    orc = orc32 if prec == "f32" else orc64
    p_o, st = oracle_pair(orc, sc)
    p_g, s = gpu_solver(sc, precision=capi.PRECISION_F32 if prec == "f32" else capi.PRECISION_F64)
    with pytest.raises(capi.FsaError):
        s.effective_mass()  # no fractions yet
    st.sort_particles(p_o)
    s.sort_particles()
    uid = s.get("uniqueId")
    s.set_volume_fractions(dens, vf[uid])  # rows in the current device order
    st.effective_mass(p_o, dens, vf)
    s.effective_mass()
    for k in ("mass", "restDensity", "color"):
        assert s.get(k).tobytes() == st[k].tobytes(), k
    rigid = st["type"] == 4
    assert np.all(s.get("restDensity")[rigid] == 0)
    # the table stays with the particles through a sort
    mv = (st["pos"] + rng.normal(0, 0.004, st["pos"].shape)).astype(st["pos"].dtype)
    st["pos"][...] = mv
    s.set("pos", mv)
    st.sort_particles(p_o)
    s.sort_particles()
    st["mass"][...] = 0
    s.set("mass", np.zeros(n, st["mass"].dtype))
    st.effective_mass(p_o, dens, vf)
    s.effective_mass()
    assert s.get("mass").tobytes() == st["mass"].tobytes()
    s.close()
    st.close()


def test_pressure_solve_equals_the_stage_calls_round_by_round(pair):
    """fsa_iisph_solve_pressure lets the two passes of a round hand their packed neighbour records on to each other
    (the packing kernels run once per solve); the stage calls pack before every pass.  Same bits."""
    sc, p, st, s, tol, variant = pair
    setup_both(p, st, s)
    s.non_pressure_force_multiphase()
    s.iisph_predict_density()
    start = {k: s.get(k) for k in ("pressure", "dijpjl", "error")}
    rounds, _ = s.iisph_solve_pressure(4)
    solved = {k: s.get(k) for k in ("pressure", "dijpjl", "error")}
    for k, v in start.items():
        s.set(k, v)
    for _ in range(rounds):
        s.iisph_calc_dijpjl()
        s.iisph_calc_new_pressure()
    for k in solved:
        assert np.array_equal(s.get(k), solved[k], equal_nan=True), k
