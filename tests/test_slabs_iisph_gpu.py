"""The IISPH step (what the reference's main.cpp runs: MultiphaseSPHSolver::SolveIISPH) on z-slabs, on the GPU
through the C ABI: all slabs in this process (LocalComm), and two ranks over gloo sharing the one GPU of the test box,
against the plain single-solver run: every owned particle bit for bit, the same number of pressure rounds."""
import numpy as np
import pytest

from fluidsolver_amd import capi, slabs
from slab_worker import SCENES, owned_iisph
from test_slabs_cpu import assert_same_iisph_bits, run_ranks
from util import gpu_solver

pytestmark = pytest.mark.gpu

STEPS = 6


def plain_iisph_run(sc, steps, variant):
    p, s = gpu_solver(sc)
    s.set_variant(variant)
    s.setup_iisph()
    rounds = []
    for _ in range(steps):
        s.non_pressure_force_multiphase()
        s.iisph_predict_density()
        rounds.append(s.iisph_solve_pressure(10)[0])
        s.iisph_update()
        s.sort_particles()
        s.iisph_factor()
    res = {k: s.get(k) for k in ("pos", "vel", "density", "pressure", "uniqueId", "type")}
    s.close()
    return res, rounds


@pytest.mark.parametrize("variant", [1, 3])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_iisph_on_local_slabs_equals_plain_solver(world, variant):
    sc = SCENES["itank"]()
    ref, rounds = plain_iisph_run(sc, STEPS, variant)
    ranks = [slabs.SlabSolver(sc, r, world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, variant),
                              exchange_buffers=False) for r in range(world)]
    team = slabs.IisphSlabs(ranks, slabs.LocalComm(), len(sc["pos"]), sc["group"], sc["normal"])
    team.setup()
    team.step(STEPS)
    assert sum(ss.lost for ss in ranks) == 0
    if world > 1:
        assert sum(ss.sent_records for ss in ranks) > 0
    assert team.rounds == rounds and max(rounds) > 1
    assert_same_iisph_bits([owned_iisph(ss) for ss in ranks], ref)
    for ss in ranks:
        ss.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_library_team_equals_plain_solver(world):
    """the same step driven inside the library (fsa_slab_iisph_*: what the host program uses for FSA_SLABS=N with
    the IISPH step), refreshes ordered by events instead of host waits"""
    sc = SCENES["itank"]()
    ref, rounds = plain_iisph_run(sc, STEPS, 3)
    ranks = [slabs.SlabSolver(sc, r, world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                              exchange_buffers=False) for r in range(world)]
    for ss in ranks:
        ss.backend.enable_iisph(sc["group"], sc["normal"])
    team = capi.SlabIisphTeam([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                              max(ss.record_cap for ss in ranks), len(sc["pos"]))
    team.setup()
    team.step(2)
    team.step(STEPS - 2)
    steps, lost, sent, got_rounds = team.stats(STEPS)
    assert steps == STEPS and lost == 0 and got_rounds == rounds
    if world > 1:
        assert sent > 0
    assert_same_iisph_bits([owned_iisph(ss) for ss in ranks], ref)
    team.close()
    for ss in ranks:
        ss.close()


def test_iisph_two_ranks_over_gloo_equal_plain_solver(tmp_path):
    parts = run_ranks(2, "hip", STEPS, tmp_path, variant=3, scene="itank", mode="iisph")
    assert sum(int(p["lost"]) for p in parts) == 0
    ref, rounds = plain_iisph_run(SCENES["itank"](), STEPS, 3)
    for p in parts:
        assert p["rounds"].tolist() == rounds
    assert_same_iisph_bits(parts, ref)


def test_iisph_calls_need_the_tables_on_a_slab_solver():
    sc = SCENES["itank"]()
    ss = slabs.SlabSolver(sc, 0, 2, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                          exchange_buffers=False)
    ss.backend.solver.sort_particles()
    with pytest.raises(capi.FsaError, match="fsa_slab_enable_iisph"):
        ss.backend.solver.iisph_factor()
    ss.close()
