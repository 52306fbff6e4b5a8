"""The DFSPH step (SPHSolver::SolveDFSPH) on z-slabs, on the GPU through the C ABI: all slabs in this process
(LocalComm), and two ranks over gloo sharing the one GPU of the test box, against the plain single-solver run: every
owned particle bit for bit, the same number of rounds in both correction loops of every step."""
import numpy as np
import pytest

from fluidsolver_amd import capi, slabs
from slab_worker import SCENES, owned_dfsph
from test_slabs_cpu import run_ranks
from util import gpu_solver

pytestmark = pytest.mark.gpu

STEPS = 6
FIELDS = ("pos", "vel", "density", "v_star", "DF_factor", "type")


def plain_dfsph_run(sc, steps, variant):
    p, s = gpu_solver(sc)
    s.set_variant(variant)
    s.setup_dfsph()
    dens, div = [], []
    for _ in range(steps):
        s.compute_non_pressure_force()
        dens.append(s.correct_density_error(5, 1.0)[0])
        s.sort_particles()
        s.compute_density_alpha()
        div.append(s.correct_divergence_error(5, 1.0)[0])
    res = {k: s.get(k) for k in FIELDS + ("uniqueId",)}
    s.close()
    return res, dens, div


def assert_same_dfsph_bits(parts, ref):
    uid = np.concatenate([p["uniqueId"] for p in parts])
    order = np.argsort(uid, kind="stable")
    ro = np.argsort(ref["uniqueId"], kind="stable")
    assert np.array_equal(uid[order], ref["uniqueId"][ro]), "every particle must be owned by exactly one rank"
    for k in FIELDS:
        a, b = np.concatenate([p[k] for p in parts])[order], ref[k][ro]
        assert np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes(), f"{k} differs from the undivided run"


@pytest.mark.parametrize("variant", [1, 3])
@pytest.mark.parametrize("world,scene", [(1, "dam"), (2, "dam"), (3, "dam"), (3, "ddam")])
def test_dfsph_on_local_slabs_equals_plain_solver(world, scene, variant):
    sc = SCENES[scene]()
    ref, dens, div = plain_dfsph_run(sc, STEPS, variant)
    ranks = [slabs.SlabSolver(sc, r, world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, variant),
                              exchange_buffers=False) for r in range(world)]
    team = slabs.DfsphSlabs(ranks, slabs.LocalComm())
    for ss in ranks:
        ss.backend.solver.timing_enable(True)
    team.setup()
    team.step(STEPS)
    assert sum(ss.lost for ss in ranks) == 0
    if world > 1:
        assert sum(ss.sent_records for ss in ranks) > 0
    # on variant 3 the slabs' passes replay the hit record (made for the layers a slab computes on), else the lists
    ran = [set(ss.backend.solver.timing_read()) for ss in ranks]
    assert all(("record_hits" in r) == (variant == 3) and ("list_build" in r) == (variant != 3) for r in ran)
    assert team.density_rounds == dens and team.divergence_rounds == div
    assert max(dens) + max(div) > 0  # (the loops really ran)
    assert_same_dfsph_bits([owned_dfsph(ss) for ss in ranks], ref)
    for ss in ranks:
        ss.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_library_team_equals_plain_solver(world):
    """the same step driven inside the library (fsa_slab_dfsph_*: what the host program uses for FSA_SLABS=N with the
    DFSPH step), refreshes ordered by events instead of host waits"""
    sc = SCENES["ddam"]()
    ref, dens, div = plain_dfsph_run(sc, STEPS, 3)
    ranks = [slabs.SlabSolver(sc, r, world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                              exchange_buffers=False) for r in range(world)]
    for ss in ranks:
        ss.backend.enable_dfsph()
    team = capi.SlabDfsphTeam([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                              max(ss.record_cap for ss in ranks))
    team.setup()
    team.step(2)
    team.step(STEPS - 2)
    steps, lost, sent, got_dens, got_div = team.stats(STEPS)
    assert steps == STEPS and lost == 0 and got_dens == dens and got_div == div
    if world > 1:
        assert sent > 0
    assert_same_dfsph_bits([owned_dfsph(ss) for ss in ranks], ref)
    team.close()
    for ss in ranks:
        ss.close()


def test_dfsph_two_ranks_over_gloo_equal_plain_solver(tmp_path):
    parts = run_ranks(2, "hip", STEPS, tmp_path, variant=3, scene="dam", mode="dfsph")
    assert sum(int(p["lost"]) for p in parts) == 0
    ref, dens, div = plain_dfsph_run(SCENES["dam"](), STEPS, 3)
    for p in parts:
        assert p["density_rounds"].tolist() == dens and p["divergence_rounds"].tolist() == div
    assert_same_dfsph_bits(parts, ref)


def test_dfsph_calls_need_enabling_on_a_slab_solver():
    sc = SCENES["dam"]()
    ss = slabs.SlabSolver(sc, 0, 2, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                          exchange_buffers=False)
    ss.backend.solver.sort_particles()
    with pytest.raises(capi.FsaError, match="fsa_slab_enable_dfsph"):
        ss.backend.solver.compute_density_alpha()
    ss.backend.enable_dfsph()
    ss.backend.solver.compute_density_alpha()
    with pytest.raises(capi.FsaError, match="DFSPH"):
        ss.backend.enable_iisph(np.zeros(len(sc["pos"]), np.int32), np.zeros((len(sc["pos"]), 3), np.float32))
    ss.close()
