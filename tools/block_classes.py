#!/usr/bin/env python3
"""Which blocks of 256 consecutive sorted particles of a dam-break scene can the tiled kernels stage, computed on the
CPU from the scene's own cell hashes (no GPU needed): the class of every block as the row table defines it (1: every
neighbour row at most 384 long as one range of cells, 2: at most 640, 0: longer), and for the class-0 blocks what the
segmentation of csrc/fsa_tile.h (sparse_segments: runs of adjacent cells, segments merged per neighbour row while
they fit a tile) makes of them.  Used to find the blocks that walked the cells particle by particle."""
import argparse, collections, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--max-runs", type=int, default=16)
ap.add_argument("--max-segments", type=int, default=24)
a = ap.parse_args()
ROWCAP, ROWCAP_BIG, BS = 384, 640, 256

sc = scenes.dam_break(*a.dims)
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
gx, gy, gz = (int(v) for v in p.gridres)
pos, typ = np.asarray(sc["pos"], np.float32), np.asarray(sc["type"])
cell = np.floor((pos - np.array(p.gridxmin, np.float32)) / np.float32(p.dx)).astype(np.int64)
h = cell[:, 1] * gx * gz + cell[:, 2] * gx + cell[:, 0]
order = np.argsort(h, kind="stable")
hs, fluid = h[order], typ[order] == 0
C, n = gx * gy * gz, len(h)
start = np.searchsorted(hs, np.arange(C + 1))
offs = [(r % 3 - 1) * gx * gz + (r // 3 - 1) * gx for r in range(9)]


def segments(hb):
    """(outcome, runs, segments) of the device's segmentation for the sorted cell hashes of one block"""
    heads = np.r_[True, np.diff(hb) > 2]
    lo, hi = hb[heads], hb[np.r_[heads[1:], True]]
    if len(lo) > a.max_runs:
        return "too many runs", len(lo), 0
    total = 0
    for off in offs:
        A, B = start[np.clip(lo + off - 1, 0, C)], start[np.clip(hi + off + 2, 0, C)]
        ca, cb = A[0], B[0]
        for k in range(1, len(lo)):
            if B[k] - ca <= ROWCAP:
                cb = B[k]
            else:
                total += -(-(cb - ca) // ROWCAP)
                ca, cb = A[k], B[k]
        total += -(-(cb - ca) // ROWCAP)
    return ("segments" if 1 <= total <= a.max_segments else "too many segments"), len(lo), int(total)


classes, outcome = collections.Counter(), collections.Counter()
fluid_in = collections.Counter()
for b0 in range(0, n, BS):
    hb = hs[b0:b0 + BS]
    longest = max(int(start[min(max(hb[-1] + off + 2, 0), C)] - start[min(max(hb[0] + off - 1, 0), C)]) for off in offs)
    cls = 1 if longest <= ROWCAP else 2 if longest <= ROWCAP_BIG else 0
    classes[cls] += 1
    nf = int(fluid[b0:b0 + BS].sum())
    fluid_in[f"class {cls}"] += nf
    if cls == 0:
        what, runs, segs = segments(hb)
        outcome[what] += 1
        fluid_in[what] += nf
print(json.dumps({"dims": a.dims, "grid": [gx, gy, gz], "particles": n, "fluid": int(fluid.sum()), "blocks": (n + BS - 1) // BS,
                  "blocks_by_class": dict(classes), "class0_blocks_by_outcome": dict(outcome),
                  "fluid_particles_in": dict(fluid_in)}))
