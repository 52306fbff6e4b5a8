#!/usr/bin/env python3
"""Neighbour-count statistics of a bench scene state (after spin-up): hits per fluid particle, cell occupancy."""
import argparse, json, os, sys
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="dam1m"); ap.add_argument("--jitter", type=float, default=0.0)
ap.add_argument("--spinup", type=int, default=100)
a = ap.parse_args()
sc = bench.build_scene(a.scene, a.jitter)
p = capi.make_params(0, **sc["params"])
s = capi.Solver(p, len(sc["pos"])); s.upload(sc)
s.step(a.spinup); s.sort_particles(); s.synchronize()
pos = s.get("pos")[:, :3].astype(np.float64); typ = s.get("type")
sr = None
sr = float(p.smoothradius)
fl = np.where(typ == 0)[0]
t = cKDTree(pos[fl])
sample = fl[np.random.default_rng(0).choice(len(fl), 200000, replace=False)]
cnt = t.query_ball_point(pos[sample], sr * 0.999999, return_length=True, workers=8)
h = s.get("particleHash")
occ = np.bincount(np.bincount(h[h >= 0].astype(np.int64))) if h is not None else None
print(json.dumps({"jitter": a.jitter, "hits_mean": float(cnt.mean()), "hits_p99": float(np.percentile(cnt, 99)),
                  "hits_max": int(cnt.max()), "cell_occupancy_hist": occ.tolist()[:40] if occ is not None else None}))
