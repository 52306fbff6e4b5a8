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


@pytest.mark.parametrize("crowded", [False, True], ids=["plain", "crowded"])
def test_recorded_counts_are_the_neighbour_counts(crowded):
    """the length of every list = the number of particles within the support (brute force), and the crowded
    scene really has lists that overflow"""
    sc = iisph_scene(crowded)
    p, s = gpu_solver(sc)
    s.setup_iisph()
    cnt = s.get("neighbourCount")
    pos = s.get("pos").astype(np.float64)
    sr = float(np.float32(sc["params"]["spacing"]) * np.float32(sc["params"]["smoothratio"]))
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    inner = np.array([len(v) for v in tree.query_ball_point(pos, sr * (1 - 1e-5))])
    outer = np.array([len(v) for v in tree.query_ball_point(pos, sr * (1 + 1e-5))])
    listed = cnt >= 0
    assert np.all((cnt[listed] >= inner[listed]) & (cnt[listed] <= outer[listed]))
    # a particle is unlisted (-1: it walks the cells) when its list would overflow or its block cannot be staged
    assert not listed[inner > 64].any()
    assert (inner > 64).any() == crowded
    s.close()


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


def block_classes(s, sc):
    """class of every block of 256 sorted particles as the row table defines it, from the sorted hashes and the
    dense cell table: 1 = all nine rows at most 384 long, 2 = at most 640, 0 = longer or out of the grid"""
    hs = s.get("particleHash").astype(np.int64)
    start = s.get("cellStartDense").astype(np.int64)
    C = len(start) - 1
    gx, gz = int(s.params.gridres[0]), int(s.params.gridres[2])
    out = []
    for b0 in range(0, len(hs), 256):
        lo, hi = hs[b0], hs[min(b0 + 256, len(hs)) - 1]
        if hi == capi.GRID_UNDEF:
            out.append(0)
            continue
        longest = 0
        for r in range(9):
            off = (r % 3 - 1) * gx * gz + (r // 3 - 1) * gx
            a, e = np.clip(lo + off - 1, 0, C), np.clip(hi + off + 2, 0, C)
            longest = max(longest, int(start[e] - start[a]))
        out.append(1 if longest <= 384 else 2 if longest <= 640 else 0)
    return np.array(out)


def test_a_lattice_lined_up_with_the_cells_runs_in_large_tiles():
    """an undisturbed lattice has cells with 12, 18 and 27 particles: rows beyond 384 entries.  Those blocks take the
    large tiles (not the cell walk), are listed, and give the bits of the cell walk (variant 0 = variant 1)."""
    sc = scenes.dam_break(32, 24, 32)
    p, s = gpu_solver(sc)
    s.set_variant(1)
    s.sort_particles()
    cls = block_classes(s, sc)
    fluid = s.get("type") == 0
    all_fluid = np.array([fluid[b0:b0 + 256].all() for b0 in range(0, len(fluid), 256)])
    # (a run of wall particles, four to a cell, spans twice as many cells: its rows can exceed any tile)
    assert (cls == 2).sum() > 10 and (cls[all_fluid] == 0).sum() == 0
    cnt = s.get("neighbourCount")
    # blocks of class 1 and 2 are listed over their rows; a class-0 block gets segments of its own where its particles
    # lie in a few stretches of cells (then all of it is listed) or walks the cells (none of it is)
    listed, of_class = cnt >= 0, np.repeat(cls, 256)[:len(cnt)]
    assert listed[of_class != 0].all()
    whole = [listed[b0:b0 + 256] for b0 in range(0, len(cnt), 256)]
    assert all(w.all() or not w.any() for w in whole)
    assert (cls == 0).sum() > 0 and any(w.all() for w, c in zip(whole, cls) if c == 0)
    assert listed[fluid].mean() > 0.995
    s.compute_pressure()
    s.compute_force()
    tiled = {k: s.get(k) for k in ("density", "pressure", "force")}
    s.set_variant(0)
    s.compute_pressure()
    s.compute_force()
    for k, v in tiled.items():
        assert s.get(k).tobytes() == v.tobytes(), k
    s.close()


@pytest.mark.parametrize("step", ["iisph", "dfsph"])
@pytest.mark.parametrize("variant", [1, 2])
def test_blocks_with_segments_of_their_own_replay_the_same_bits(step, variant):
    """in a long tank some blocks hold the end of a wall row, a nearly empty row and the start of a fluid row: their
    lists are built over segments of their own (class 0 by the row table, listed all the same), and the passes that
    replay them give the bits of the searching passes, which walk the cells for such blocks"""
    if step == "iisph":
        sc = scenes.dam_break(48, 24, 32, wall_type=scenes.TYPE_RIGID)
        rigid = sc["type"] == scenes.TYPE_RIGID
        sc["mass"][rigid] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
        sc["group"][rigid] = 1
        sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
    else:
        sc = scenes.dam_break(48, 24, 32)
    names = IISPH_OUT if step == "iisph" else DFSPH_OUT
    got = []
    for lists in (True, False):
        p, s = gpu_solver(sc)
        s.set_variant(variant)
        s.set_neighbour_lists(lists)
        if step == "iisph":
            s.setup_iisph()
        else:
            s.setup_dfsph()
        if lists:
            cls, cnt = block_classes(s, sc), s.get("neighbourCount")
            fluid = s.get("type") == 0
            own = [b for b in np.where(cls == 0)[0] if (cnt[256 * b:256 * b + 256] >= 0).all()]
            assert len(own) > 0 and any(fluid[256 * b:256 * b + 256].any() for b in own)
            assert (cnt[fluid] >= 0).all()
        if step == "iisph":
            s.step_iisph(2)
        else:
            s.step_dfsph(2)
        got.append({k: s.get(k) for k in names})
        s.close()
    for k in names:
        assert got[0][k].tobytes() == got[1][k].tobytes(), k


# ---- the hit record of the variant-3 density pass, replayed by the DFSPH passes ---------------------------------

def run_dfsph(sc, mode, steps, variant=3):
    """mode: "replay" (the default of variant 3), "lists" (lists only) or "search"; returns the results and the
    names of the kernels that ran"""
    p, s = gpu_solver(sc)
    s.set_variant(variant)
    s.set_neighbour_lists(mode != "search", replay=mode == "replay")
    s.timing_enable(True)
    s.setup_dfsph()
    s.step_dfsph(steps)
    out = {k: s.get(k) for k in DFSPH_OUT}
    ran = set(s.timing_read())
    s.close()
    return out, ran


def crowded_dam():
    sc = scenes.jittered(scenes.dam_break(16, 12, 10), seed=43, pos_jitter=0.15, vel_scale=0.3, fluid_only=True)
    pos = np.array(sc["pos"])
    fl = np.where(np.asarray(sc["type"]) == 0)[0][:500]
    c = pos[fl].mean(0)
    pos[fl] = (c + (pos[fl] - c) * np.float32(0.4)).astype(np.float32)  # ranges of far more than 57 candidates
    return dict(sc, pos=pos)


REPLAY_SCENES = {
    "jittered": lambda: scenes.jittered(scenes.dam_break(16, 12, 10), seed=43, pos_jitter=0.15, vel_scale=0.3,
                                        fluid_only=True),
    "at_walls": lambda: scenes.jittered(scenes.dam_break(14, 12, 10, tank_x_factor=2.0), seed=21, pos_jitter=0.2,
                                        vel_scale=1.0),
    "crowded": crowded_dam,
    "large_tiles": lambda: scenes.dam_break(32, 24, 32),    # rows beyond a tile: two tiles per row
    "own_segments": lambda: scenes.dam_break(48, 24, 32),   # blocks with segments of their own
}


@pytest.mark.parametrize("name", list(REPLAY_SCENES))
def test_dfsph_replaying_the_hit_record_equals_dfsph_with_lists_and_with_search(name):
    """on variant 3 the passes of the DFSPH step (density and factor, non-pressure force with its wall neighbours, the
    density-change passes, applyPStiff) take their neighbours from the hit record of the WCSPH density pass, made once
    per sort: the bits of the list replay and of the searching kernels, particles without a record (crowded
    neighbourhoods, blocks that cannot be tiled) included"""
    sc = REPLAY_SCENES[name]()
    steps = 3 if len(sc["pos"]) < 20000 else 2
    a, ran_a = run_dfsph(sc, "replay", steps)
    b, ran_b = run_dfsph(sc, "lists", steps)
    c, ran_c = run_dfsph(sc, "search", steps)
    assert "record_hits" in ran_a and "list_build" not in ran_a
    assert "list_build" in ran_b and "record_hits" not in ran_b
    assert "list_build" not in ran_c and "record_hits" not in ran_c
    for k in DFSPH_OUT:
        assert a[k].tobytes() == b[k].tobytes(), k
        assert a[k].tobytes() == c[k].tobytes(), k
    if name == "crowded":  # some particles have no record and walk the cells inside the replay kernels
        p, s = gpu_solver(sc)
        s.sort_particles()
        s.compute_pressure()
        assert (s.get("hitCount") < 0).sum() > 0
        s.close()


def test_hit_record_is_made_again_after_positions_change_without_a_sort():
    sc = REPLAY_SCENES["jittered"]()
    outs = []
    for mode in ("replay", "search"):
        p, s = gpu_solver(sc)
        s.set_neighbour_lists(mode != "search")
        s.setup_dfsph()
        s.compute_non_pressure_force()  # records (when enabled)
        pos = s.get("pos")
        rng = np.random.default_rng(3)
        s.set("pos", (pos + rng.normal(0, 0.0005, pos.shape)).astype(np.float32))
        s.sort_particles()
        s.compute_density_alpha()
        outs.append({k: s.get(k) for k in ("density", "DF_factor", "uniqueId")})
        s.close()
    for k in outs[0]:
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k


def test_wcsph_force_after_dfsph_passes_still_finds_its_record():
    """the DFSPH passes keep their packed records out of the buffers of the WCSPH record while they replay it: a WCSPH
    force pass on the same sorted state gives what it gives without them"""
    sc = REPLAY_SCENES["at_walls"]()
    p, a = gpu_solver(sc)
    p, b = gpu_solver(sc)
    for s in (a, b):
        s.sort_particles()
        s.compute_pressure()
    a.compute_density_alpha()
    a.compute_non_pressure_force()
    a.solve_density_stiff()
    a.apply_pstiff()
    for s in (a, b):
        s.compute_force()
    assert a.get("force").tobytes() == b.get("force").tobytes()
    a.close()
    b.close()


def test_neighbour_mode_values_and_a_solver_whose_size_is_no_multiple_of_a_block():
    """fsa_set_neighbour_lists takes 0, 1 or 2 and nothing else; a solver of 1000 particles (three full blocks and a
    part) and one of 7 (less than a wavefront) replay the record like any other"""
    sc = scenes.jittered(scenes.dam_break(10, 10, 10, walls=False), seed=5, pos_jitter=0.1, vel_scale=0.2)
    p, s = gpu_solver(sc)
    for bad in (-1, 3):
        with pytest.raises(capi.FsaError):
            capi.check(s.lib.fsa_set_neighbour_lists(s._h, bad))
    s.close()
    for n in (1000, 7):
        sub = {k: (np.asarray(v)[:n] if k != "params" and hasattr(v, "__len__") and len(v) == len(sc["pos"]) else v)
               for k, v in sc.items()}
        a, ran = run_dfsph(sub, "replay", 3)
        b, _ = run_dfsph(sub, "search", 3)
        assert "record_hits" in ran
        for k in DFSPH_OUT:
            assert a[k].tobytes() == b[k].tobytes(), (n, k)
