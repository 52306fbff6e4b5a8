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


SCENES = {"dam": test_scene, "ddam": ddam_scene}


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
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fluidsolver_amd import slabs

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
    sc = SCENES[a.scene]()
    if a.backend == "oracle":
        from oracle.orc import Oracle
        from slab_oracle_backend import OracleSlabBackend
        orc = Oracle("f32")
        orc.set_threads(2)
        device = torch.device("cpu")
        factory = lambda P, held, cap: OracleSlabBackend(orc, P, held, cap)  # noqa: E731
    else:
        device = torch.device("cuda", 0)
        factory = lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, a.variant)  # noqa: E731
    tr = slabs.NeighbourTransport(a.rank, a.world, device)
    # with --recut-at the run starts from deliberately lopsided cuts, so that the re-cut has work to do
    cuts = None
    if a.recut_at:
        gz = slabs.derive_grid(sc["params"])["gridres"][2]
        cuts = [0] + [3 * (r + 1) for r in range(a.world - 1)] + [gz]
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
