"""One rank of a slab run, started by the slab tests as a separate process:

    python tests/slab_worker.py --rank R --world N --port P --out DIR --backend oracle|hip --steps K

Writes DIR/rank<R>.npz with the results for the particles the rank owns.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def test_scene():
    """small dam break with noise on positions and velocities: particles cross the slab cuts"""
    from fluidsolver_amd import scenes
    sc = scenes.dam_break(12, 10, 28)
    n_fluid = sc["n_fluid"]
    sc = scenes.jittered(sc, seed=77, pos_jitter=0.2, vel_scale=1.5, fluid_only=True)
    sc["n_fluid"] = n_fluid
    return sc


def ddam_scene():
    """small double dam break with box obstacles (two fluid blocks, obstacles staggered along z): the particle
    count per layer is uneven and shifts as the blocks collapse"""
    from fluidsolver_amd import scenes
    sc = scenes.double_dam_break(8, 8, 20, tank_x_factor=3.0)
    n_fluid = sc["n_fluid"]
    sc = scenes.jittered(sc, seed=78, pos_jitter=0.2, vel_scale=1.5, fluid_only=True)
    sc["n_fluid"] = n_fluid
    return sc


# cuts used with drift_scene (its grid has 22 cell layers)
DRIFT_CUTS = {1: None, 2: [0, 11, 22], 3: [0, 7, 14, 22]}


def drift_scene():
    """the dam scene plus small clusters of fluid particles between the tank wall and the x-min face of the grid
    box: half of each cluster starts just beyond that face and is held on it by the position clamp (out of the
    grid: nobody's neighbour, but still feeling its mates next to it through viscosity), and all of them drift
    along z across the cuts of DRIFT_CUTS, four cell layers in 25 steps"""
    from fluidsolver_amd import slabs
    sc = test_scene()
    geo = slabs.derive_grid(sc["params"])
    assert geo["gridres"][2] == 22
    dx, xmin, zmin = float(geo["dx"]), float(geo["vmin"][0]), float(geo["vmin"][2])
    pos, vel = np.array(sc["pos"], np.float32), np.array(sc["vel"], np.float32)
    movers = np.nonzero(np.asarray(sc["type"]) == 0)[0][-24:]
    k = 0
    for cut in (7, 11, 14):
        for side in (-1.0, 1.0):  # starting left of the cut and moving right, and the other way round
            for j in range(4):
                outside = j >= 2
                i = movers[k]
                k += 1
                pos[i] = (xmin + (-0.2 if outside else 0.5) * dx, (0.05 if side < 0 else 0.09) + 0.5 * dx * (j % 2),
                          zmin + (cut + 0.4 * side) * dx)
                vel[i] = (0.0, 0.0 if outside else 0.3, -4.0 * side)
    sc["pos"], sc["vel"] = pos, vel
    return sc


def itank_scene():
    """a column of fluid in a tank of TYPE_RIGID particles (the kind of scene the IISPH step runs), long in z, with
    noise on positions and velocities: particles cross the slab cuts, and the walls are part of every slab"""
    from fluidsolver_amd import scenes
    sc = scenes.iisph_tank(8, 10, 26)
    rng = np.random.Generator(np.random.PCG64(79))
    fl = np.asarray(sc["type"]) == 0
    pos, vel = np.array(sc["pos"], np.float32), np.array(sc["vel"], np.float32)
    pos[fl] += rng.uniform(-0.15, 0.15, (int(fl.sum()), 3)).astype(np.float32) * np.float32(0.005)
    vel[fl] = rng.uniform(-1.0, 1.0, (int(fl.sum()), 3)).astype(np.float32) * np.float32([0.3, 0.3, 1.5])
    sc["pos"], sc["vel"] = pos, vel
    return sc


SCENES = {"dam": test_scene, "ddam": ddam_scene, "drift": drift_scene, "itank": itank_scene}


def owned_iisph(ss):
    """what a slab owns after IISPH steps (each ends with the sort and the density of the next step)"""
    b = ss.backend
    layer = b.global_layer_of_sorted()
    m = ((layer >= ss.own[0]) & (layer < ss.own[1])) | (layer < 0)
    out = {k: np.asarray(b.get(k))[m] for k in ("pos", "vel", "density", "pressure", "uniqueId", "type")}
    out["hash_layer"] = layer[m]
    return out


def owned_dfsph(ss):
    """what a slab owns after DFSPH steps (each ends with the velocity update that follows the sort)"""
    b = ss.backend
    layer = b.global_layer_of_sorted()
    m = ((layer >= ss.own[0]) & (layer < ss.own[1])) | (layer < 0)
    out = {k: np.asarray(b.get(k))[m] for k in ("pos", "vel", "density", "v_star", "DF_factor", "uniqueId", "type")}
    out["hash_layer"] = layer[m]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", choices=sorted(SCENES), default="dam")
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", choices=["oracle", "hip"], required=True)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--variant", type=int, default=3)
    ap.add_argument("--recut-at", type=int, default=0, help="re-cut the slabs after this many steps")
    ap.add_argument("--mode", choices=["wcsph", "iisph", "dfsph"], default="wcsph")
    ap.add_argument("--group", choices=["none", "hosted"], default="none",
                    help="hosted: the library's one-slab-per-process group, its messages over this job's gloo group")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fluidsolver_amd import slabs

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
    sc = SCENES[a.scene]()
    if a.backend == "oracle":
        from oracle.orc import Oracle
        from slab_oracle_backend import OracleDfsphSlabBackend, OracleIisphSlabBackend, OracleSlabBackend
        orc = Oracle("f32")
        orc.set_threads(2)
        device = torch.device("cpu")
        cls = {"iisph": OracleIisphSlabBackend, "dfsph": OracleDfsphSlabBackend}.get(a.mode, OracleSlabBackend)
        factory = lambda P, held, cap: cls(orc, P, held, cap)  # noqa: E731
    else:
        device = torch.device("cuda", 0)
        factory = lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, a.variant)  # noqa: E731
    tr = slabs.NeighbourTransport(a.rank, a.world, device)
    # with --recut-at the run starts from deliberately lopsided cuts, so that the re-cut has work to do
    cuts = None
    if a.recut_at:
        gz = slabs.derive_grid(sc["params"])["gridres"][2]
        cuts = [0] + [3 * (r + 1) for r in range(a.world - 1)] + [gz]
    if a.scene == "drift" and not a.recut_at:
        cuts = DRIFT_CUTS[a.world]
    if a.mode == "dfsph":
        ss = slabs.SlabSolver(sc, a.rank, a.world, tr, factory, cuts=cuts, exchange_buffers=False)
        team = slabs.DfsphSlabs([ss], slabs.DistComm(tr))
        team.setup()
        team.step(a.steps)
        out = owned_dfsph(ss)
        out.update(cuts=np.asarray(ss.layout.cuts), lost=np.asarray(ss.lost), sent=np.asarray(ss.sent_records),
                   density_rounds=np.asarray(team.density_rounds), divergence_rounds=np.asarray(team.divergence_rounds))
        np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **out)
        ss.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if a.mode == "iisph":
        ss = slabs.SlabSolver(sc, a.rank, a.world, tr, factory, cuts=cuts, exchange_buffers=False)
        team = slabs.IisphSlabs([ss], slabs.DistComm(tr), len(sc["pos"]), sc["group"], sc["normal"])
        team.setup()
        team.step(a.steps)
        out = owned_iisph(ss)
        out.update(cuts=np.asarray(ss.layout.cuts), lost=np.asarray(ss.lost), sent=np.asarray(ss.sent_records),
                   rounds=np.asarray(team.rounds))
        np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **out)
        ss.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if a.group == "hosted":
        from fluidsolver_amd import capi
        ss = slabs.SlabSolver(sc, a.rank, a.world, tr, factory, cuts=cuts, exchange_buffers=False)
        g = capi.SlabGroup.hosted(ss.backend.solver, a.rank, a.world, capi.torch_host_transport(), ss.layout.cuts, ss.halo,
                                  ss.record_cap)
        g.step(7)
        g.step(a.steps - 7)
        st = g.sync()
        out = ss.owned()
        out.update(cuts=np.asarray(ss.layout.cuts), lost=np.asarray(int(st.lost)), sent=np.asarray(int(st.records_sent)),
                   owned_count=np.asarray(len(out["uniqueId"])), error_bits=np.asarray(int(st.error_bits)))
        np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **out)
        g.close()
        ss.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    ss = slabs.SlabSolver(sc, a.rank, a.world, tr, factory, cuts=cuts)
    if a.recut_at:
        ss.step(a.recut_at)
        changed = ss.rebalance()
        assert changed, "the lopsided cuts should have been replaced"
        ss.step(a.steps - a.recut_at)
    else:
        ss.step(a.steps)
    out = ss.owned()
    out["cuts"] = np.asarray(ss.layout.cuts)
    out["lost"] = np.asarray(ss.lost)
    out["sent"] = np.asarray(ss.sent_records)
    out["owned_count"] = np.asarray(len(out["uniqueId"]))
    np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **out)
    ss.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
