import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from fluidsolver_amd import capi
out = {}
for scene, spin in (("dam16m", 0), ("dam16m", 105), ("dam1m", 105)):
    sc = bench.build_scene(scene)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.step(spin)
    s.sort_particles()
    cnt = s.get("neighbourCount")
    fl = s.get("type") == 0
    n = len(cnt) // 256 * 256
    blk = (cnt[:n] < 0).reshape(-1, 256).any(1)
    out[f"{scene}@{spin}"] = {"unlisted": int((cnt < 0).sum()), "unlisted_fluid": int(((cnt < 0) & fl).sum()),
                              "fluid": int(fl.sum()), "blocks": int(len(blk)), "blocks_unstaged": int(blk.sum()),
                              "max_count": int(cnt.max())}
    s.close()
print(json.dumps(out))
