"""Native slab groups (fsa_slab_group_*): the slab step with the exchange inside the library and no host round
trip per step.  All slabs in one process on the one GPU of the test box (the `local` transport, one stream per
solver), against the plain single-solver run, bit for bit; the RCCL transport with the one rank a single GPU
allows, plus its loop-back self test."""
import numpy as np
import pytest

from fluidsolver_amd import capi, slabs
from test_slabs_cpu import assert_same_bits, merged
from test_slabs_gpu import plain_gpu_run
from slab_worker import DRIFT_CUTS, SCENES

pytestmark = pytest.mark.gpu

STEPS = 25


def make_ranks(sc, world, cuts=None):
    return [slabs.SlabSolver(sc, r, world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                             cuts=cuts, exchange_buffers=False) for r in range(world)]


@pytest.mark.parametrize("scene", ["dam", "ddam", "drift"])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_local_group_equals_plain_solver(scene, world):
    """(`drift`: with particles outside the grid that change owner as they drift along z past the cuts)"""
    sc = SCENES[scene]()
    ref = plain_gpu_run(sc, STEPS, 3)
    ranks = make_ranks(sc, world, cuts=DRIFT_CUTS[world] if scene == "drift" else None)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    g.step(7)
    g.step(STEPS - 7)  # no sync in between: the second call queues behind the first
    st = g.sync()
    assert st.steps == STEPS and st.lost == 0 and st.error_bits == 0
    if world > 1:
        assert st.records_sent > 0
    parts = [ss.owned() for ss in ranks]
    uid, got = merged(parts)
    assert_same_bits(uid, got, ref)
    assert st.particles == sum(ss.backend.n for ss in ranks)
    g.close()
    # the solvers live on with a host-side count
    n0 = ranks[0].backend.n
    ranks[0].backend.solver.sort_particles()
    assert ranks[0].backend.n == n0
    for ss in ranks:
        ss.close()


def test_local_group_lopsided_cuts_and_stepwise_sync():
    """three slabs with very uneven cuts, synchronising (and reading buffers) every few steps"""
    sc = SCENES["dam"]()
    gz = slabs.derive_grid(sc["params"])["gridres"][2]
    ranks = make_ranks(sc, 3, cuts=[0, 3, 6, gz])
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], [0, 3, 6, gz], slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    done = 0
    for k in (1, 2, 5, 9, 8):
        g.step(k)
        done += k
        _ = ranks[1].backend.get("pos")  # any entry point on a member waits for the queued steps
    assert done == STEPS
    st = g.sync()
    assert st.lost == 0
    uid, got = merged([ss.owned() for ss in ranks])
    assert_same_bits(uid, got, plain_gpu_run(sc, STEPS, 3))
    g.close()
    for ss in ranks:
        ss.close()


def test_group_steps_make_no_host_synchronisation():
    """queueing steps must not make the host wait for the stream or the device, not even once (the library counts
    its own waits); all it may do is wait for an event recorded three steps earlier, which keeps the host from
    running further ahead while the device still has work queued"""
    lib = capi.load()
    sc = SCENES["ddam"]()
    ranks = make_ranks(sc, 3)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    g.step(5)
    g.sync()
    before = lib.fsa_debug_host_waits()
    g.step(60)
    assert lib.fsa_debug_host_waits() == before
    st = g.sync()
    assert lib.fsa_debug_host_waits() > before  # the sync itself is one
    assert st.steps == 65 and st.lost == 0 and st.error_bits == 0
    g.close()
    for ss in ranks:
        ss.close()


def test_group_reports_an_outgrown_message():
    sc = SCENES["dam"]()
    ranks = make_ranks(sc, 2)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO, 16)
    g.step(2)
    with pytest.raises(capi.FsaError, match="outgrew"):
        g.sync()
    g.close()
    for ss in ranks:
        ss.close()


@pytest.mark.parametrize("tight", [False, True])
def test_both_ends_size_the_messages_alike_also_when_counts_are_cut(tight, monkeypatch):
    """a message is sized from the count of three steps before on both ends (the sender from its own count, the
    receiver from the header it got); between processes a disagreement would be a send and a receive of different
    sizes.  The local transport derives the receiver's number too and fails the step if the two differ: run with no
    head room, and with messages that are too small on purpose (a fifth of a typical count short), so that counts
    outgrow them and are cut"""
    sc = SCENES["ddam"]()

    def run(slack):
        monkeypatch.setenv("FSA_SLAB_MSG_SLACK", str(slack))
        ranks = make_ranks(sc, 3)
        g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                                 max(ss.record_cap for ss in ranks))
        try:
            g.step(40)             # (a disagreement raises here: "sized ... by the sender and ... by the receiver")
            return g.sync()
        finally:
            g.close()
            for ss in ranks:
                ss.close()

    st = run(4096)
    per_message = st.records_sent / (st.steps * 4)   # three slabs: four messages per step
    assert per_message > 50
    if not tight:                  # counts may or may not outgrow a message without head room
        try:
            assert run(0).error_bits == 0
        except capi.FsaError as e:
            assert "outgrew" in str(e)
    else:
        with pytest.raises(capi.FsaError, match="outgrew"):
            run(-int(per_message * 0.125 + per_message * 0.2))


def test_group_refuses_unfit_solvers():
    sc = SCENES["dam"]()
    ranks = make_ranks(sc, 2)
    ranks[0].backend.solver.set_variant(2)
    with pytest.raises(capi.FsaError, match="variant 3"):
        capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO, 1000)
    ranks[0].backend.solver.set_variant(3)
    with pytest.raises(capi.FsaError, match="holds layers"):
        capi.SlabGroup.local([ss.backend.solver for ss in ranks], [0, 4, ranks[0].layout.cuts[-1]], slabs.HALO, 1000)
    for ss in ranks:
        ss.close()


def test_rccl_group_single_rank_and_loopback():
    """the RCCL transport as far as one GPU goes: communicator of one rank, the group step on it (no neighbours, so
    nothing travels), and a send to itself through the communicator on the solver's stream"""
    import torch
    sc = SCENES["dam"]()
    ref = plain_gpu_run(sc, STEPS, 3)
    ranks = make_ranks(sc, 1)
    ss = ranks[0]
    g = capi.SlabGroup.rccl(ss.backend.solver, 0, 1, capi.SlabGroup.unique_id(), ss.layout.cuts, slabs.HALO, 4096)
    g.step(STEPS)
    st = g.sync()
    assert st.lost == 0 and st.error_bits == 0
    uid, got = merged([ss.owned()])
    assert_same_bits(uid, got, ref)
    src = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
    dst = torch.zeros_like(src)
    torch.cuda.synchronize()
    g.rccl_loopback(src.data_ptr(), dst.data_ptr(), src.numel() * 4)
    assert torch.equal(src, dst)
    g.close()
    ss.close()


def test_group_without_overlap_gives_the_same_bits(monkeypatch):
    """FSA_SLAB_OVERLAP=0 (read when the group is made): transport on the solver's stream, same results"""
    monkeypatch.setenv("FSA_SLAB_OVERLAP", "0")
    sc = SCENES["ddam"]()
    ranks = make_ranks(sc, 3)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    g.step(STEPS)
    st = g.sync()
    assert st.lost == 0 and st.error_bits == 0
    uid, got = merged([ss.owned() for ss in ranks])
    assert_same_bits(uid, got, plain_gpu_run(sc, STEPS, 3))
    g.close()
    for ss in ranks:
        ss.close()


def test_group_reports_a_particle_that_jumps_layers():
    """a particle far from the cuts that lands in a neighbour's layers within one step was not looked at when the
    records were written: the group flags it (error bit 32) instead of losing it silently"""
    sc = SCENES["dam"]()
    geo = slabs.derive_grid(sc["params"])
    ranks = make_ranks(sc, 2)
    cut = ranks[0].layout.cuts[1]
    # a fluid particle of rank 0, at least five layers below the cut, thrown across it
    layer = slabs.cell_layer_host(sc["pos"], geo["vmin"], geo["vmax"], geo["dx"], geo["gridres"])
    cand = np.nonzero((sc["type"] == 0) & (layer < cut - 5) & (layer >= 0))[0]
    assert len(cand)
    i = cand[0]
    dist = (cut + 1 - layer[i]) * float(geo["dx"])
    sc = dict(sc)
    sc["vel"] = sc["vel"].copy()
    sc["vel"][i] = (0.0, 0.0, dist / sc["params"]["dt"])
    for ss in ranks:
        ss.close()
    ranks = make_ranks(sc, 2)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    g.step(2)  # the first pack (before any sort) sends everything through the edge pass; the second one does not
    with pytest.raises(capi.FsaError, match="crossed a whole cell layer"):
        g.sync()
    g.close()
    for ss in ranks:
        ss.close()


def test_group_refuses_members_changed_between_steps():
    sc = SCENES["dam"]()
    ranks = make_ranks(sc, 2)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    g.step(3)
    g.sync()
    s0 = ranks[0].backend.solver
    s0.set("vel", s0.get("vel"))  # any change of the particles invalidates what the group has prepared
    with pytest.raises(capi.FsaError, match="changed between two steps"):
        g.step(1)
    g.close()
    for ss in ranks:
        ss.close()


@pytest.mark.parametrize("mode", ["0", "1"])
def test_group_with_either_wall_sum_placement(mode, monkeypatch):
    """large slab solvers sum the wall neighbours in a pass of their own, which the group step runs once before the
    forces next to the cuts; small ones (all the test scenes) in the tail of the density pass.  Both forced here."""
    monkeypatch.setenv("FSA_WALL_PASS", mode)
    sc = SCENES["ddam"]()
    ref = plain_gpu_run(sc, STEPS, 3)
    ranks = make_ranks(sc, 3)
    g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], ranks[0].layout.cuts, slabs.HALO,
                             max(ss.record_cap for ss in ranks))
    g.step(STEPS)
    st = g.sync()
    assert st.lost == 0 and st.error_bits == 0
    uid, got = merged([ss.owned() for ss in ranks])
    assert_same_bits(uid, got, ref)
    g.close()
    for ss in ranks:
        ss.close()
