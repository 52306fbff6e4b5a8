"""One rank of a grid-solver team whose ranks talk through torch.distributed (gloo), started by
tests/test_grid_team_gpu.py as a separate process:

    python tests/grid_team_worker.py --rank R --world N --port P --out DIR --dim X Y Z --cuts ... --prec f32|f64

Every rank builds the same initial fields from the same seed, takes its planes, runs the steps and writes what it owns
(fields in the whole grid's layout, zero outside its planes) to DIR/rank<R>.npz."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def initial_fields(shape_of, dtype, seed, scale):
    rng = np.random.default_rng(seed)
    return {k: (scale * rng.standard_normal(shape_of(k))).astype(dtype) for k in ("u", "v", "w")}


def tracers(dim, h, dtype, seed, n=300):
    """the same points on every rank and in the test: inside the grid box, whose low corner is (-dim.x / 2 * h, 0,
    -dim.z / 2 * h)"""
    rng = np.random.default_rng(seed)
    low = np.array([-(dim[0] // 2) * h, 0.0, -(dim[2] // 2) * h])
    return (low + rng.random((n, 3)) * np.array(dim) * h).astype(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--dim", type=int, nargs=3, required=True)
    ap.add_argument("--cuts", type=int, nargs="+", required=True)
    ap.add_argument("--prec", choices=["f32", "f64"], default="f32")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--reach", type=int, default=1)
    ap.add_argument("--scale", type=float, default=0.3)
    a = ap.parse_args()

    import torch.distributed as dist
    from fluidsolver_amd import capi

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
    precision = capi.PRECISION_F32 if a.prec == "f32" else capi.PRECISION_F64
    tr = capi.torch_host_transport()
    t = capi.GridTeam.hosted(a.dim, a.cuts, a.rank, tr, reach=a.reach, precision=precision)
    for k, f in initial_fields(t.shape, t.dtype, 91, a.scale).items():
        t.set(k, f)
    t.set_particles(tracers(a.dim, 0.02, t.dtype, 92))
    t.step(a.steps, 2000)
    it, res = t.solve(7)  # one capped solve more on the final state: the loop's scalars are the same on every rank
    bits = t.sync()
    out = {k: t.get(k) for k in ("u", "v", "w", "p", "q", "divU")}
    out["pos"], out["color"] = t.get_particles()
    out["cg"] = np.array([it, res])
    out["bits"] = np.array(bits)
    np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **out)
    t.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
