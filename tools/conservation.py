#!/usr/bin/env python3
"""Conserved-quantity drift of the GPU step over many steps (default: 1000 steps of the 1 M-particle dam
break).  Prints one JSON object; run on the GPU box:  python tools/conservation.py > gpurun_out/cons.json"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsolver_amd import capi, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[128, 64, 128])
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--variant", type=int, default=2)
    a = ap.parse_args()
    sc = scenes.dam_break(*a.dims)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.set_variant(a.variant)
    s.upload(sc)
    g = 9.8  # the literal of the force pass
    rows = []

    def sample(step):
        uid = s.get("uniqueId") if step else np.arange(len(sc["pos"]), dtype=np.int32)
        fl = sc["type"][uid] == 0
        m = s.get("mass").astype(np.float64)[fl]
        x = s.get("pos").astype(np.float64)[fl]
        v = s.get("vel").astype(np.float64)[fl]
        mom = (m[:, None] * v).sum(0)
        rows.append({"step": step, "time": step * sc["params"]["dt"], "particles": int(s.n),
                     "fluid_particles": int(fl.sum()), "mass": float(m.sum()),
                     "momentum": mom.tolist(),
                     # y momentum minus what gravity alone would have put in (walls take up the rest)
                     "momentum_y_minus_gravity": float(mom[1] + m.sum() * g * step * sc["params"]["dt"]),
                     "kinetic": float((0.5 * m * (v * v).sum(1)).sum()),
                     "potential": float((m * g * x[:, 1]).sum()),
                     "max_speed": float(np.sqrt((v * v).sum(1)).max()),
                     "out_of_grid": int(s.num_undef()) if step else 0,
                     "centre_of_mass": ((m[:, None] * x).sum(0) / m.sum()).tolist()})

    sample(0)
    done = 0
    while done < a.steps:
        k = min(a.every, a.steps - done)
        s.step(k)
        done += k
        sample(done)
    e0 = rows[0]["kinetic"] + rows[0]["potential"]
    e1 = rows[-1]["kinetic"] + rows[-1]["potential"]
    print(json.dumps({"scene": f"dam break {a.dims}", "variant": a.variant, "steps": a.steps,
                      "energy_change_rel": (e1 - e0) / abs(e0), "mass_change": rows[-1]["mass"] - rows[0]["mass"],
                      "particles_change": rows[-1]["particles"] - rows[0]["particles"], "samples": rows}, indent=1))
    s.close()


if __name__ == "__main__":
    main()
