#!/usr/bin/env python3
"""Long slab-group run on one GPU beside the plain single-solver run of the same scene, compared every --every steps
for every owned particle, bit for bit.  Stops at the first difference and says where it is."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fluidsolver_amd import capi, scenes, slabs
from test_slabs_cpu import merged
ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=6)
ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
ap.add_argument("--scene", choices=["dam", "ddam"], default="dam", help="dam break, or double dam break with obstacles")
ap.add_argument("--jitter", type=float, default=0.0, help="noise on the start positions, in spacings")
ap.add_argument("--steps", type=int, default=4000)
ap.add_argument("--every", type=int, default=100)
ap.add_argument("--start-fine", type=int, default=-1, help="from this step on compare after every step")
a = ap.parse_args()
sc = scenes.dam_break(*a.dims) if a.scene == "dam" else scenes.double_dam_break(*a.dims)
if a.jitter > 0:
    n_fluid = sc["n_fluid"]
    sc = scenes.jittered(sc, seed=5, pos_jitter=a.jitter, vel_scale=0.5, fluid_only=True)
    sc["n_fluid"] = n_fluid
ranks = [slabs.SlabSolver(sc, r, a.world, None, lambda P, held, cap: slabs.HipSlabBackend(P, held, cap, 0, 3),
                          exchange_buffers=False) for r in range(a.world)]
cuts = ranks[0].layout.cuts
g = capi.SlabGroup.local([ss.backend.solver for ss in ranks], cuts, slabs.HALO, max(ss.record_cap for ss in ranks))
p = capi.make_params(capi.PRECISION_F32, **sc["params"])
ps = capi.Solver(p, len(sc["pos"]))
ps.upload(sc)
done = 0
while done < a.steps:
    k = 1 if 0 <= a.start_fine <= done else min(a.every, a.steps - done)
    if 0 <= a.start_fine and done < a.start_fine:
        k = min(k, a.start_fine - done)
    g.step(k)
    try:
        st = g.sync()
    except capi.FsaError as exc:
        # how fast is the fastest particle of the plain run over these steps?
        worst = 0.0
        for _ in range(k):
            ps.step(1)
            v = ps.get("vel")
            worst = max(worst, float(np.abs(v).max()) * sc["params"]["dt"] / float(slabs.derive_grid(sc["params"])["dx"]))
        print(json.dumps({"step": done + k, "group_error": str(exc), "plain_run_max_cell_layers_per_step": worst}))
        break
    if k > 1:
        ps.step(k - 1)
    ps.sort_particles(); ps.compute_pressure(); ps.compute_force()
    ref = {f: ps.get(f) for f in ("density", "pressure", "force")}
    ps.advect()
    for f in ("pos", "vel", "uniqueId", "type"):
        ref[f] = ps.get(f)
    done += k
    parts = [ss.owned() for ss in ranks]
    uid, got = merged(parts)
    ro = np.argsort(ref["uniqueId"], kind="stable")
    if not np.array_equal(uid, ref["uniqueId"][ro]):
        print(json.dumps({"step": done, "problem": "owned sets differ", "owned": int(len(uid)), "plain": int(len(ro)),
                          "lost": int(st.lost)}))
        break
    layer = np.concatenate([q["hash_layer"] for q in parts])[np.argsort(np.concatenate([q["uniqueId"] for q in parts]), kind="stable")]
    bad = {}
    fluid = got["type"] == 0
    for f in ("density", "pressure", "force", "pos", "vel"):
        x, y = got[f], ref[f][ro]
        ne = (x != y)
        if ne.ndim > 1:
            ne = ne.any(axis=1)
        if f == "force":
            ne &= fluid
        if ne.any():
            idx = np.nonzero(ne)[0]
            bad[f] = {"count": int(ne.sum()), "uids": uid[idx[:8]].tolist(), "layers": layer[idx[:8]].tolist(),
                      "types": got["type"][idx[:8]].tolist(),
                      "max_abs": float(np.abs(x[idx].astype(np.float64) - y[idx].astype(np.float64)).max()),
                      "slab_val": np.asarray(x[idx[0]]).tolist(), "plain_val": np.asarray(y[idx[0]]).tolist()}
    if bad:
        print(json.dumps({"step": done, "cuts": cuts, "lost": int(st.lost), "differences": bad}))
        break
    speed = float(np.abs(ref["vel"]).max()) * sc["params"]["dt"] / float(slabs.derive_grid(sc["params"])["dx"])
    print(json.dumps({"step": done, "ok": True, "lost": int(st.lost), "held": [ss.backend.n for ss in ranks],
                      "max_cell_layers_per_step": round(speed, 4)}), flush=True)
g.close()
for ss in ranks:
    ss.close()
ps.close()
