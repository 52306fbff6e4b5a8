"""z-slab driver on CPU: world_size 2 and 3 over gloo, the per-rank solver replaced by the oracle.

What is checked is the decomposition logic of fluidsolver_amd/slabs.py (cuts, ownership, the single
migration + halo exchange per step, order keys): after K steps the owned particles of all ranks together
must be every particle exactly once, and carry bit-for-bit the values of the undivided oracle run.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from fluidsolver_amd import slabs
from util import oracle_pair

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world, backend, steps, out, timeout=600, variant=2, recut_at=0, scene="dam", mode="wcsph", group="none"):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "slab_worker.py"), "--rank", str(r), "--world",
                               str(world), "--port", str(port), "--out", str(out), "--backend", backend, "--steps",
                               str(steps), "--variant", str(variant), "--recut-at", str(recut_at), "--scene", scene,
                               "--mode", mode, "--group", group])
             for r in range(world)]
    try:
        codes = [p.wait(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world, codes
    return [dict(np.load(os.path.join(out, f"rank{r}.npz"))) for r in range(world)]


def undivided(orc, steps, scene="dam"):
    from slab_worker import SCENES
    sc = SCENES[scene]()
    p, st = oracle_pair(orc, sc)
    for _ in range(steps - 1):
        st.solve_sph(p, 1)
    st.sort_particles(p)
    st.compute_pressure(p)
    st.compute_force(p)
    res = {k: st[k].copy() for k in ("density", "pressure", "force")}
    st.advect(p)
    for k in ("pos", "vel", "uniqueId", "type"):
        res[k] = st[k].copy()
    st.close()
    return sc, res


def merged(parts):
    uid = np.concatenate([p["uniqueId"] for p in parts])
    order = np.argsort(uid, kind="stable")
    return uid[order], {k: np.concatenate([p[k] for p in parts])[order]
                        for k in ("pos", "vel", "density", "pressure", "force", "type")}


def assert_same_bits(uid, got, ref):
    ro = np.argsort(ref["uniqueId"], kind="stable")
    assert np.array_equal(uid, ref["uniqueId"][ro]), "every particle must be owned by exactly one rank"
    fluid = got["type"] == 0
    for k in ("pos", "vel", "density", "pressure", "force", "type"):
        a, b = got[k], ref[k][ro]
        if k == "force":  # computeF writes force[] for fluid particles only; the other entries are stale
            a, b = a[fluid], b[fluid]
        assert np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes(), \
            f"{k} differs from the undivided run"


@pytest.mark.parametrize("world", [2, 3])
def test_out_of_grid_particles_move_on_with_their_z_layer(orc32, tmp_path, world):
    """a particle outside the grid is nobody's neighbour but still sums over the cells around it: it has to live
    with the slab that owns the layer of its z coordinate, and change owner when it drifts past a cut"""
    steps = 25
    parts = run_ranks(world, "oracle", steps, tmp_path, scene="drift")
    assert sum(int(p["lost"]) for p in parts) == 0
    sc, ref = undivided(orc32, steps, scene="drift")
    uid, got = merged(parts)
    assert_same_bits(uid, got, ref)
    # the clusters really are outside at the end, on every rank, and feel their neighbours
    for p in parts:
        out = p["hash_layer"] < 0
        assert out.sum() >= 2
        assert np.abs(p["force"][out]).max() > 0  # (there is no gravity in this scene)


@pytest.mark.parametrize("world", [2, 3])
def test_slab_run_equals_undivided_run(orc32, tmp_path, world):
    steps = 25
    parts = run_ranks(world, "oracle", steps, tmp_path)
    assert sum(int(p["lost"]) for p in parts) == 0
    # the scene is noisy enough that particles really change owner
    assert sum(int(p["sent"]) for p in parts) > 0
    sc, ref = undivided(orc32, steps)
    uid, got = merged(parts)
    assert_same_bits(uid, got, ref)
    cuts = parts[0]["cuts"]
    assert len(cuts) == world + 1 and all(np.diff(cuts) >= slabs.HALO + 1)


def test_recut_mid_run_keeps_the_run_bit_identical(orc32, tmp_path):
    """start from lopsided cuts, re-cut after 10 of 25 steps (particles move to other ranks, some not to a
    neighbour): still the undivided run, and the ranks end up balanced"""
    steps = 25
    parts = run_ranks(3, "oracle", steps, tmp_path, recut_at=10)
    assert sum(int(p["lost"]) for p in parts) == 0
    sc, ref = undivided(orc32, steps)
    uid, got = merged(parts)
    assert_same_bits(uid, got, ref)
    counts = np.array([int(p["owned_count"]) for p in parts], float)
    assert counts.max() / counts.mean() < 1.35
    assert list(parts[0]["cuts"]) != [0, 3, 6, int(parts[0]["cuts"][-1])]


def test_recut_with_particles_outside_the_grid(orc32, tmp_path):
    """the re-cut hands particles outside the grid to the rank that owns the layer of their z coordinate"""
    steps = 25
    parts = run_ranks(3, "oracle", steps, tmp_path, recut_at=10, scene="drift")
    assert sum(int(p["lost"]) for p in parts) == 0
    sc, ref = undivided(orc32, steps, scene="drift")
    uid, got = merged(parts)
    assert_same_bits(uid, got, ref)
    assert sum(int((p["hash_layer"] < 0).sum()) for p in parts) >= 6


def test_double_dam_with_obstacles_recut(orc32, tmp_path):
    """the double dam break with box obstacles (uneven, shifting particle count per layer) over two ranks, from
    lopsided cuts, with one rebalance() on the way: still the undivided run bit for bit"""
    steps = 25
    parts = run_ranks(2, "oracle", steps, tmp_path, recut_at=10, scene="ddam")
    assert sum(int(p["lost"]) for p in parts) == 0
    assert sum(int(p["sent"]) for p in parts) > 0
    sc, ref = undivided(orc32, steps, scene="ddam")
    assert (sc["type"] != 0).sum() > 0 and len(np.unique(sc["group"])) >= 4  # walls, obstacles, two fluid blocks
    uid, got = merged(parts)
    assert_same_bits(uid, got, ref)
    counts = np.array([int(p["owned_count"]) for p in parts], float)
    assert counts.max() / counts.mean() < 1.35


def test_balanced_cuts():
    hist = np.array([0, 0, 10, 10, 10, 10, 10, 10, 10, 10, 0, 0])
    assert slabs.balanced_cuts(hist, 2) == [0, 6, 12]
    c = slabs.balanced_cuts(hist, 4)
    assert c[0] == 0 and c[-1] == 12 and all(np.diff(c) >= 3)
    with pytest.raises(ValueError):
        slabs.balanced_cuts(hist, 5)
    # everything in one layer: the minimum thickness still holds
    one = np.zeros(20, int)
    one[3] = 100
    c = slabs.balanced_cuts(one, 3)
    assert all(np.diff(c) >= 3) and c[-1] == 20


def test_cell_layer_host_matches_oracle_hash(orc32):
    from slab_worker import test_scene
    sc = test_scene()
    p, st = oracle_pair(orc32, sc)
    st.calc_hash(p)
    h = st["particleHash"].astype(np.int64)
    gx, gz = p.gridres[0], p.gridres[2]
    want = np.where(h == 99999999, -1, (h // gx) % gz)
    geo = slabs.derive_grid(sc["params"])
    assert geo["gridres"] == [p.gridres[0], p.gridres[1], p.gridres[2]]
    got = slabs.cell_layer_host(sc["pos"], geo["vmin"], geo["vmax"], geo["dx"], geo["gridres"])
    assert np.array_equal(got, want)
    st.close()


def test_layout_ranges():
    lay = slabs.SlabLayout(20, [0, 7, 13, 20], halo=2)
    assert lay.own(1) == (7, 13) and lay.held(1) == (5, 15)
    assert lay.held(0) == (0, 9) and lay.held(2) == (11, 20)
    assert lay.held(-1) == (0, 0) and lay.held(3) == (0, 0)


def test_windowed_scene_gives_each_rank_its_particles_of_the_whole_scene():
    """a rank that builds only the particles of its own layers (scenes.*(z_select=...)) uploads exactly what it
    would take from the whole scene, unique ids included, and sizes its buffers the same"""
    from fluidsolver_amd import scenes

    class Capture:
        record_bytes = 48

        def __init__(self, P, held, cap):
            self.held, self.cap = held, cap

        def upload(self, sub):
            self.sub = sub

        def set_scene_has_rigid(self, flag):
            self.rigid = flag

        def set_owned(self, own):
            self.own = own

    whole = scenes.double_dam_break(8, 8, 20, tank_x_factor=3.0)
    probe = scenes.double_dam_break(8, 8, 20, tank_x_factor=3.0, z_select=lambda z: np.zeros(len(z), bool))
    assert len(probe["pos"]) == 0 and probe["n_total"] == len(whole["pos"]) and probe["n_fluid"] == whole["n_fluid"]
    geo = slabs.derive_grid(whole["params"])
    gz = geo["gridres"][2]
    layer_of = lambda z: slabs.z_layer_host(z, geo["vmin"][2], geo["vmax"][2], geo["dx"], gz)  # noqa: E731
    zv, zc = probe["z_counts"]
    hist = np.bincount(layer_of(zv), weights=zc, minlength=gz).astype(np.int64)
    full_layers = slabs.cell_layer_host(whole["pos"], geo["vmin"], geo["vmax"], geo["dx"], geo["gridres"])
    assert np.array_equal(hist, np.bincount(full_layers, minlength=gz))
    world = 3
    cuts = slabs.balanced_cuts(hist, world)
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        part = scenes.double_dam_break(8, 8, 20, tank_x_factor=3.0,
                                       z_select=lambda z: (layer_of(z) >= lo) & (layer_of(z) < hi))
        a = slabs.SlabSolver(whole, r, world, None, Capture, exchange_buffers=False)
        b = slabs.SlabSolver(part, r, world, None, Capture, exchange_buffers=False, cuts=cuts, layer_hist=hist,
                             has_rigid=False)
        assert a.layout.cuts == b.layout.cuts and a.capacity == b.capacity and a.record_cap == b.record_cap
        assert sorted(a.backend.sub) == sorted(b.backend.sub)
        for k in a.backend.sub:
            assert np.array_equal(a.backend.sub[k], b.backend.sub[k]), k


# ---- the IISPH step on slabs ---------------------------------------------------------------------------------
def assert_same_iisph_bits(parts, ref):
    uid = np.concatenate([p["uniqueId"] for p in parts])
    order = np.argsort(uid, kind="stable")
    ro = np.argsort(ref["uniqueId"], kind="stable")
    assert np.array_equal(uid[order], ref["uniqueId"][ro]), "every particle must be owned by exactly one rank"
    typ = np.concatenate([p["type"] for p in parts])[order]
    for k in ("pos", "vel", "type", "pressure", "density"):
        a, b = np.concatenate([p[k] for p in parts])[order], ref[k][ro]
        if k == "density":  # (never written for TYPE_RIGID particles: whatever the array slot held before)
            a, b = a[typ == 0], b[typ == 0]
        assert np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes(), f"{k} differs from the undivided run"


@pytest.mark.parametrize("world", [2, 3])
def test_dfsph_slab_run_equals_undivided_run(orc32, tmp_path, world):
    """SolveDFSPH on slabs: v_star riding with the records, a field refresh after every neighbour pass, the maximum
    of error[] taken over the slabs.  Every owned particle must carry the bits of the undivided oracle run, and both
    correction loops must run the same number of rounds in every step"""
    from slab_worker import SCENES
    steps = 6
    parts = run_ranks(world, "oracle", steps, tmp_path, scene="dam", mode="dfsph")
    assert sum(int(p["lost"]) for p in parts) == 0
    assert sum(int(p["sent"]) for p in parts) > 0
    sc = SCENES["dam"]()
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.compute_density_alpha(p)
    dens, div = [], []
    for _ in range(steps):
        st.compute_non_pressure_force(p)
        dens.append(st.correct_density_error(p, 5, 1.0)[0])
        st.sort_particles(p)
        st.compute_density_alpha(p)
        div.append(st.correct_divergence_error(p, 5, 1.0)[0])
    for part in parts:
        assert part["density_rounds"].tolist() == dens and part["divergence_rounds"].tolist() == div
    uid = np.concatenate([q["uniqueId"] for q in parts])
    order = np.argsort(uid, kind="stable")
    ro = np.argsort(st["uniqueId"], kind="stable")
    assert np.array_equal(uid[order], st["uniqueId"][ro])
    for k in ("pos", "vel", "density", "v_star", "DF_factor", "type"):
        a, b = np.concatenate([q[k] for q in parts])[order], st[k][ro]
        assert np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes(), f"{k} differs from the undivided run"
    st.close()


@pytest.mark.parametrize("world", [2, 3])
def test_iisph_slab_run_equals_undivided_run(orc32, tmp_path, world):
    """the step main.cpp runs (SolveIISPH) on slabs: a field refresh after every neighbour pass, the pressure riding
    with the records, the stop test summed over the slabs; tank walls (TYPE_RIGID) in every slab.  Every owned
    particle must carry the bits of the undivided oracle run, and the rounds must be the same in number"""
    from slab_worker import SCENES
    steps = 6
    parts = run_ranks(world, "oracle", steps, tmp_path, scene="itank", mode="iisph")
    assert sum(int(p["lost"]) for p in parts) == 0
    assert sum(int(p["sent"]) for p in parts) > 0
    sc = SCENES["itank"]()
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.iisph_factor(p)
    st.rigid_particle_volume(p)
    rounds = []
    for _ in range(steps):
        st.non_pressure_force_multiphase(p)
        st.iisph_predict_density(p)
        rounds.append(st.iisph_solve_pressure(p, 10, mean_in_double=True)[0])
        st.iisph_update(p)
        st.sort_particles(p)
        st.iisph_factor(p)
    ref = {k: st[k].copy() for k in ("pos", "vel", "density", "pressure", "uniqueId", "type")}
    st.close()
    for part in parts:
        assert part["rounds"].tolist() == rounds
    assert max(rounds) > 1
    assert_same_iisph_bits(parts, ref)
