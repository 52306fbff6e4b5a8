#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the WCSPH step on the 16M-particle dam break.

    python bench.py --gpus N --steps K --warmup W

One "step" = one SPHSolver::SolveSPH pass (cell hash -> stable counting sort -> reorder -> density /
pressure -> force -> integration) over all particles, device resident, no host read-back inside
the timed region.  Rank 0 prints ONE JSON line.  For N > 1 the driver launches one process per GPU
with torch.distributed.run; the grid is cut into z-slabs (fluidsolver_amd/slabs.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# algorithmic bytes per particle and launch (fp32, minimal SoA hot set)
ALGO_BYTES = {
    "hash_count": 20,   # R pos 12, W hash 4 + slot 4
    "scatter": 12,      # R hash 4 + slot 4, W index 4
    "rank": 12,
    "reorder": 88,      # R perm 4 + gather 40, W 40 + index table 4
    "rank_reorder": 96,  # the two in one pass: the permutation is not read back (-4)
    "density": 32,      # R pos 12 + type 4 + mass 4 + restDensity 4, W density 4 + pressure 4
    "force": 52,        # R pos 12 + vel 12 + type 4 + mass 4 + density 4 + pressure 4, W force 12
    "advect": 76,       # R force 12 + vel 12 + pos 12 + type 4, W vel 12 + pos 12 + v_star 12
    "advect_hash": 84,  # advect + W hash 4 + slot 4 (the position is not read again)
}


VARIANT_NAMES = {
    0: "0: one thread per particle, the reference's expression order and double promotions",
    1: "1: LDS-tiled kernels, exact arithmetic (bit-equal to the CPU oracle over whole runs)",
    2: "2: LDS-tiled kernels, fast fp32 pair arithmetic (within 1e-5 per element of the oracle)",
    3: "3: as 2 with one neighbour search per step (record / replay); bit-equal to 2, within 1e-5 per element of "
       "the oracle",
}
DEVELOPED_FLOW_STEPS = int(os.environ.get("FSA_BENCH_FLOW_STEPS", 2000))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup", type=int, default=int(os.environ.get("FSA_BENCH_SPINUP", 100)),
                    help="untimed steps before warm-up so that the dam has started to collapse")
    ap.add_argument("--scene", default=os.environ.get("FSA_BENCH_SCENE", "dam16m"),
                    choices=["dam16m", "dam1m", "dam128k", "ddam64m"])
    ap.add_argument("--jitter", type=float, default=float(os.environ.get("FSA_BENCH_JITTER", 0.0)),
                    help="uniform position noise in units of the particle spacing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recut-every", type=int, default=int(os.environ.get("FSA_BENCH_RECUT_EVERY", 0)),
                    help="slab runs: re-balance the z-cuts every k steps (scenes whose mass moves along z)")
    ap.add_argument("--variant", type=int, default=None)
    return ap.parse_args()


SCENES = {"dam16m": (256, 256, 256), "dam16m_flat": (512, 64, 512), "dam1m": (128, 64, 128), "dam128k": (64, 32, 64), "ddam64m": (320, 320, 320)}


def build_scene(name, jitter=0.0, z_select=None):
    """`z_select`: build only the particles whose z coordinate it selects (fluidsolver_amd/scenes.py)"""
    from fluidsolver_amd import scenes
    nx, ny, nz = SCENES[name]
    sc = scenes.double_dam_break(nx, ny, nz, z_select=z_select) if name == "ddam64m" else \
        scenes.dam_break(nx, ny, nz, z_select=z_select)
    if jitter > 0:
        n_fluid = sc["n_fluid"]
        sc = scenes.jittered(sc, seed=2024, pos_jitter=jitter, vel_scale=0.0, fluid_only=True)
        sc["n_fluid"] = n_fluid
    return sc


def host_threads():
    """The cores this process may use: the job's OMP_NUM_THREADS where the launcher sets it (a GPU box hands each
    job a share of the machine and says so that way), else the scheduler affinity."""
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    return max(1, min(env, aff) if env else aff)


def time_oracle(sc, threads, seconds_target):
    from oracle.orc import Oracle
    orc = Oracle("f32")
    orc.set_threads(threads)
    P = sc["params"]
    p = orc.make_params(P["spacing"], P["smoothratio"], P["vmin"], P["vmax"], gravity=P["gravity"], dt=P["dt"],
                        viscosity=P["viscosity"], pressureK=P["pressureK"], boundary_visc=P["boundary_visc"],
                        acceleration_limit=P["acceleration_limit"], boundary_mode=P["boundary_mode"])
    n = len(sc["pos"])
    st = orc.new_state(n, p.gridres[0] * p.gridres[1] * p.gridres[2])
    st.load(sc)
    st.solve_sph(p, 1)  # first touch
    t0 = time.perf_counter()
    st.solve_sph(p, 1)
    one = time.perf_counter() - t0
    steps = max(2, min(200, int(seconds_target / max(one, 1e-3))))
    t0 = time.perf_counter()
    st.solve_sph(p, steps)
    dt = time.perf_counter() - t0
    st.close()
    return sc["n_fluid"] * steps / dt, steps, dt, n


def cpu_baseline(sc, scene_name):
    """The CPU oracle (a port of the reference step, OpenMP over particles) timed on this host: all the cores the
    job has on the bench workload itself for a bounded number of steps, and one thread on the 1M-particle dam
    break of the same construction."""
    from fluidsolver_amd import scenes
    threads = host_threads()
    value, steps, dt, n = time_oracle(sc, threads, 10.0)
    small = scenes.dam_break(128, 64, 128)
    v1, s1, d1, n1 = time_oracle(small, 1, 4.0)
    return {"value": value, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/sph_oracle.c, the bench workload ({scene_name}: {n} particles incl. walls, "
                      f"{sc['n_fluid']} fluid counted), {steps} steps, {dt:.1f} s",
            "host_cpus_visible": os.cpu_count(),
            "single_thread": {"value": v1, "cores": 1,
                              "sample": f"1M-particle dam break ({n1} particles), {s1} steps, {d1:.1f} s"}}


def run_single(args, jitter=None, keep_scene=False, variant=None, spinup=None, record_stats=False, scene=None):
    import torch
    from fluidsolver_amd import capi

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(0)
    sc = build_scene(scene or args.scene, args.jitter if jitter is None else jitter)
    n, n_fluid = len(sc["pos"]), sc["n_fluid"]
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, n)
    variant = args.variant if variant is None else variant
    if variant is not None:
        s.set_variant(variant)
    s.upload(sc)
    s.step(args.spinup if spinup is None else spinup)
    s.step(args.warmup)
    s.synchronize()
    s.timing_enable(True)
    s.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.step(args.steps)
    s.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kt = s.timing_read()
    s.timing_enable(False)
    n_undef = s.num_undef() if args.steps + args.warmup + args.spinup > 0 else 0
    stats = None
    if record_stats:  # how many fluid particles the last step's record kernels could not serve from the tiles
        s.sort_particles()
        s.compute_pressure()
        fluid = s.get("type") == 0
        hc = s.get("hitCount")
        stats = {"fluid_particles_without_record": int(((hc < 0) & fluid).sum()),
                 "fraction_without_record": float(((hc < 0) & fluid).sum() / max(int(fluid.sum()), 1)),
                 "max_speed": float(np.linalg.norm(s.get("vel")[fluid], axis=1).max())}
    s.close()
    return dict(n=n, n_fluid=n_fluid, wall=wall, kernels=kt, num_cells=p.num_cells, n_undef=n_undef,
                gridres=list(p.gridres), scene=sc if keep_scene else None, stats=stats,
                variant=3 if variant is None else variant)


def next_rows(args):
    """The steps beside the WCSPH one, each timed briefly on the GPU (not part of `value`): the DFSPH step and the
    IISPH step at the bench scene's size, one step of the Eulerian grid solver at 256^3."""
    from fluidsolver_amd import capi, scenes
    out = {}
    dims = {"dam16m": (256, 256, 256), "dam1m": (128, 64, 128), "dam128k": (64, 32, 64)}.get(args.scene)
    if dims:
        sc = scenes.dam_break(*dims)
        s = capi.Solver(capi.make_params(capi.PRECISION_F32, **sc["params"]), len(sc["pos"]))
        s.upload(sc)
        s.setup_dfsph()
        s.step_dfsph(3)
        s.synchronize()
        t0 = time.perf_counter()
        s.step_dfsph(5)
        s.synchronize()
        out["dfsph"] = {"scene": f"dam break {dims[0]}x{dims[1]}x{dims[2]} (SolveDFSPH, two correction loops, stop tests on the device)",
                        "ms_per_step": (time.perf_counter() - t0) / 5 * 1e3, "particles": len(sc["pos"])}
        s.close()
        sc = scenes.dam_break(*dims, wall_type=scenes.TYPE_RIGID)
        sc["mass"][sc["type"] == scenes.TYPE_RIGID] = scenes.rigid_mass(sc["params"]["spacing"], sc["params"]["restdensity"])
        sc["group"][sc["type"] == scenes.TYPE_RIGID] = 1
        sc["params"] = dict(sc["params"], viscosity=0.05, **scenes.IISPH_PARAMS)
        s = capi.Solver(capi.make_params(capi.PRECISION_F32, **sc["params"]), len(sc["pos"]))
        s.upload(sc)
        s.setup_iisph()
        s.step_iisph(3)
        s.synchronize()
        t0 = time.perf_counter()
        s.step_iisph(5)
        s.synchronize()
        out["iisph"] = {"scene": f"dam break {dims[0]}x{dims[1]}x{dims[2]} in a tank of TYPE_RIGID particles (SolveIISPH, up to 10 "
                                 "pressure rounds)",
                        "ms_per_step": (time.perf_counter() - t0) / 5 * 1e3, "particles": len(sc["pos"])}
        s.close()
    g = capi.GridSolver((256, 256, 256), padding=1, h=0.02, dt=0.001, rho=1.0)
    rng = np.random.default_rng(1)
    for k in ("u", "w"):
        g.set(k, (0.2 * rng.standard_normal(g.shape(k))).astype(g.dtype))
    g.step(1)
    g.make_rhs()
    g.synchronize()
    t0 = time.perf_counter()
    it, _ = g.solve(200)
    g.synchronize()
    ms_round = (time.perf_counter() - t0) * 1e3 / (it + 1)
    out["grid"] = {"scene": "Eulerian grid solver, 256^3 cells, fp32: conjugate-gradient rounds of the pressure projection",
                   "ms_per_cg_round": ms_round, "GB_per_s": 12 * 4 * 256 ** 3 / (ms_round * 1e-3) / 1e9,
                   "bytes_per_cell_and_round": 48}
    g.close()
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("FSA_FORCE_SLABS") == "1":  # (env: slab path on one GPU)
        from fluidsolver_amd import slabs
        res = slabs.bench_distributed(args, build_scene)
    else:
        res = run_single(args, keep_scene=not args.no_cpu_baseline)
    if rank != 0:
        return
    wall, steps = res["wall"], args.steps
    value = res["n_fluid"] * steps / wall
    kernels = {k: {"launches": c, "avg_ms": (ms / c if c else 0.0)} for k, (c, ms) in res["kernels"].items()}
    # the dominant kernel of the step
    cost = {k: kernels[k]["avg_ms"] * kernels[k]["launches"] for k in kernels if k in ALGO_BYTES}
    dom = max(cost, key=cost.get, default=None)
    # the two neighbour passes cost about the same; the force pass is the one the target is stated on
    if dom and "force" in cost and cost["force"] >= 0.9 * cost[dom]:
        dom = "force"

    def units_of(k):
        if "local_n" in res:  # slab run: the kernel timings are rank 0's, over the particles it holds
            return res["local_n_fluid"] if k == "force" else res["local_n"]
        return res["n_fluid"] if k == "force" else res["n"]

    # every kernel's algorithmic bytes per launch over its average launch time, against the HBM peak
    roof_all = {}
    for k in cost:
        if kernels[k]["avg_ms"] > 0:
            gbs = ALGO_BYTES[k] * units_of(k) / (kernels[k]["avg_ms"] * 1e-3) / 1e9
            roof_all[k] = {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms": kernels[k]["avg_ms"],
                           "bytes_per_particle": ALGO_BYTES[k]}
    roof = None
    if dom:
        units = units_of(dom)
        byts = ALGO_BYTES[dom] * units
        avg_s = kernels[dom]["avg_ms"] * 1e-3
        achieved = byts / avg_s / 1e9 if avg_s > 0 else 0.0
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": byts, "avg_launch_ms": kernels[dom]["avg_ms"]}
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                t = json.load(open(prof))
                key = f"{args.scene}:{dom}"
                if key in t:
                    roof["traffic"] = t[key]
            except Exception:
                pass
        # second view of the same kernel: VALU issue slots (the pass is instruction-bound, not HBM-bound)
        vprof = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vprof) and avg_s > 0:
            try:
                v = json.load(open(vprof))
                key = f"{args.scene}:{dom}"
                if key in v:
                    peak = v["simds"] * v["clock_ghz"] * 1e9 / v["cycles_per_wave_inst"]
                    rate = v[key]["valu_wave_insts"] / avg_s
                    roof["valu_issue"] = {"wave_insts_per_launch": v[key]["valu_wave_insts"], "achieved_per_s": rate,
                                          "peak_per_s": peak, "frac": rate / peak}
            except Exception:
                pass
    out = {
        "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s", "n_gpus": args.gpus,
        "steps": steps, "warmup": args.warmup, "ms_per_step": wall / steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"dam16m": "dam break, 256^3 = 16 777 216 fluid particles + one layer of wall "
                                          "particles, WCSPH fp32, shipped SceneId-2 parameters (viscosity 0.01)",
                                "dam16m_flat": "dam break 512x64x512 = 16 777 216 fluid particles, column 0.32 m high",
                                "dam1m": "dam break 128x64x128 = 1 048 576 fluid particles",
                                "dam128k": "dam break 64x32x64 fluid particles",
                                "ddam64m": "double dam break, 2 x 320^3 = 65 536 000 fluid particles, box obstacles, "
                                           "one layer of wall particles"}[args.scene],
                   "particles_total": res["n"], "particles_counted": res["n_fluid"], "grid": res["gridres"],
                   "spinup_steps": args.spinup, "parallelism": f"z-slabs x{args.gpus}" if args.gpus > 1 else "1 GPU",
                   "out_of_grid_particles": res["n_undef"],
                   "kernel_variant": VARIANT_NAMES.get(res.get("variant", 3 if args.variant is None else args.variant)),
                   **{k: res[k] for k in ("comm_backend", "slab_step", "slab_cuts", "halo_layers",
                                          "max_local_particles", "mean_local_particles", "lost_particles",
                                          "records_sent_per_step", "exchange_ms_per_step_max",
                                          "host_syncs_in_timed_steps", "recuts", "recut_every",
                                          "sum_owned_particles") if k in res}},
        "roofline": roof, "roofline_all": {"unit": "GB/s", "peak": HBM_PEAK_GBS, "kernels": roof_all},
        "kernels": kernels,
    }
    if res.get("sum_owned_particles", res["n"]) != res["n"] or res.get("lost_particles", 0):
        # a slab run whose ranks do not own every particle exactly once between them has timed something else
        raise SystemExit(f"bench: {res.get('sum_owned_particles')} of {res['n']} particles owned after the run, "
                         f"{res.get('lost_particles')} dropped by a full exchange buffer")
    single = args.gpus == 1 and world == 1 and "local_n" not in res
    if single and args.jitter == 0 and not os.environ.get("FSA_BENCH_NO_DEVELOPED"):
        # the headline state is a lattice that has barely moved; the same scene in a disordered state, the fluid
        # in contact with the walls, costs more per step
        dev = run_single(args, jitter=0.2)
        out["config"]["developed"] = {
            "state": "same scene, positions jittered by +-0.2 spacing before the spin-up steps: disordered "
                     "neighbourhoods, fluid in contact with the walls",
            "ms_per_step": dev["wall"] / steps * 1e3, "value": dev["n_fluid"] * steps / dev["wall"],
            "unit": "particle-steps/s",
            "kernels_ms": {k: (ms / c if c else 0.0) for k, (c, ms) in dev["kernels"].items()}}
        # ... and the flow itself: the column after DEVELOPED_FLOW_STEPS steps of collapse
        flow = run_single(args, spinup=DEVELOPED_FLOW_STEPS, record_stats=True)
        out["config"]["developed_flow"] = {
            "state": f"same scene after {DEVELOPED_FLOW_STEPS} steps of collapse "
                     f"({DEVELOPED_FLOW_STEPS * 4e-4:.2f} s of flow)",
            "ms_per_step": flow["wall"] / steps * 1e3, "value": flow["n_fluid"] * steps / flow["wall"],
            "unit": "particle-steps/s", "out_of_grid_particles": flow["n_undef"], **flow["stats"],
            "kernels_ms": {k: (ms / c if c else 0.0) for k, (c, ms) in flow["kernels"].items()}}
        if args.scene == "dam16m":
            # The 1.28 m column of the headline scene is far outside what the shipped stiffness and one layer of
            # penalty wall particles hold: the bottom compresses by 40 % and from step 400 on fluid is pressed
            # through the floor onto the grid-box clamp (out_of_grid_particles above).  The same number of
            # particles as a column 0.32 m high (BASELINE config 2's height, four times its footprint) stays in
            # its tank: the developed flow of a scene the parameters suit.
            flat = run_single(args, spinup=DEVELOPED_FLOW_STEPS, record_stats=True, scene="dam16m_flat")
            out["config"]["developed_flow_flat"] = {
                "state": f"512 x 64 x 512 = 16 777 216 fluid particles (column 0.32 m high) after "
                         f"{DEVELOPED_FLOW_STEPS} steps of collapse",
                "particles_total": flat["n"], "grid": flat["gridres"],
                "ms_per_step": flat["wall"] / steps * 1e3, "value": flat["n_fluid"] * steps / flat["wall"],
                "unit": "particle-steps/s", "out_of_grid_particles": flat["n_undef"], **flat["stats"],
                "kernels_ms": {k: (ms / c if c else 0.0) for k, (c, ms) in flat["kernels"].items()}}
        # ... and the flavour whose whole runs equal the CPU oracle's bit for bit (variant 1), same state as `value`
        if res.get("variant", 3) != 1:
            ex = run_single(args, variant=1)
            out["value_exact"] = {
                "kernel_variant": VARIANT_NAMES[1], "ms_per_step": ex["wall"] / steps * 1e3,
                "value": ex["n_fluid"] * steps / ex["wall"], "unit": "particle-steps/s",
                "kernels_ms": {k: (ms / c if c else 0.0) for k, (c, ms) in ex["kernels"].items()}}
    if single and args.jitter == 0 and not os.environ.get("FSA_BENCH_NO_DEVELOPED") and not os.environ.get("FSA_BENCH_NO_NEXT"):
        out["next_rows"] = next_rows(args)
    if single and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(res["scene"], args.scene)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
