import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests.test_gpu_parity import SCENES
from tests.util import gpu_solver
for name in ("edge", "small_dam_walls"):
    sc = SCENES[name]()
    res = {}
    for v in (2, 3):
        p, s = gpu_solver(sc)
        s.set_variant(v)
        s.sort_particles(); s.compute_pressure(); s.compute_force(); s.synchronize()
        res[v] = (s.get("density"), s.get("force"), s.get("type"), s.get("particleHash"), s.get("pos"))
        s.close()
    d2, f2, t, h, pos = res[2]; d3, f3 = res[3][0], res[3][1]
    bad = np.where(d2 != d3)[0]
    print(name, "n", len(d2), "density mismatches", len(bad), "gres", list(p.gridres))
    for i in bad[:12]:
        print("  i", i, "type", t[i], "hash", h[i], "pos", pos[i], "d2", d2[i], "d3", d3[i])
    fl = t == 0
    badf = np.where((f2 != f3).any(axis=1) & fl)[0]
    print("  force mismatches", len(badf), "of fluid", fl.sum())
    for i in badf[:12]:
        print("  i", i, "hash", h[i], "f2", f2[i], "f3", f3[i])
