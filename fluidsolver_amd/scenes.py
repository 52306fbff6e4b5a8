"""Scene generators: host-side restatement of the reference's particle emitters.

The reference builds every scene from axis-aligned lattices whose loop counters
are accumulated in fp32 (`for (float x = xmin; x < xmax; x += spacing)`), so
particle counts and coordinates depend on fp32 rounding; `lattice_axis`
reproduces that loop literally.  Citations are relative to
/root/reference/particle_solver.

A scene is a dict of numpy arrays in upload order:
  pos (n,3) f32, vel (n,3) f32, color (n,4) f32, normal (n,3) f32,
  type (n,) i32, group (n,) i32, uniqueId (n,) i32, mass (n,) f32,
  restDensity (n,) f32
plus 'params' (keyword arguments for the solver parameter block).

Windows.  The generators of the benchmark scenes take `z_select`, a function from an array of z coordinates to a
boolean mask: only the selected particles are built (a rank of a z-slab run builds its own layers, not the scene),
with the unique ids they have in the whole scene; 'n_total' is then the size of the whole scene and
'z_counts' = (z values, particles per value) of the whole scene, from which the layer histogram follows.
"""
from __future__ import annotations

import numpy as np

TYPE_FLUID, TYPE_CLOTH, TYPE_BOUNDARY, TYPE_EMITTER, TYPE_RIGID = 0, 1, 2, 3, 4
TYPE_DEFORMABLE, TYPE_GRANULAR = 5, 6

F32 = np.float32

# config/sph_scene.xml:1-10 and MultiScene id=2 (:57-86)
REF_DOMAIN_MIN = (-0.4, -0.1, -0.4)
REF_DOMAIN_MAX = (0.4, 1.0, 0.4)
REF_PARAMS = dict(spacing=0.005, smoothratio=2.0, dt=0.0004, viscosity=0.05, pressureK=1119.0,
                  gravity=(0.0, -10.0, 0.0), restdensity=1000.0,
                  # <BoundInfo> is absent from sph_scene.xml and the reader returns 0 for a missing tag
                  boundary_visc=0.0,
                  # the code asks for "AccelerationLimit", the file spells "AcceleractionLimit" -> 0
                  acceleration_limit=0.0)


def lattice_axis(lo, hi, spacing, inclusive=False) -> np.ndarray:
    """`for (float v = lo; v < hi; v += spacing)` (or `<=`), all in fp32."""
    lo, hi, spacing = F32(lo), F32(hi), F32(spacing)
    out = []
    v = lo
    if not spacing > 0:
        raise ValueError("spacing must be positive")
    while (v <= hi) if inclusive else (v < hi):
        out.append(v)
        v = F32(v + spacing)
    return np.asarray(out, dtype=F32)


def _grid3(ax, ay, az) -> np.ndarray:
    """x outermost, z innermost: the nesting of MultiphaseSPHSolver.cpp:714-716."""
    nx, ny, nz = len(ax), len(ay), len(az)
    pos = np.empty((nx, ny, nz, 3), dtype=F32)
    pos[..., 0] = ax[:, None, None]
    pos[..., 1] = ay[None, :, None]
    pos[..., 2] = az[None, None, :]
    return pos.reshape(-1, 3)


def _finish(parts, params) -> dict:
    n = sum(len(p["pos"]) for p in parts)
    # AddDefaultParticle: unique_id = index at creation, MultiphaseSPHSolver.cpp:489 (a windowed part knows which
    # of its indices it kept)
    uid, base = [], 0
    for p in parts:
        local = p.get("uid_local")
        uid.append(base + (np.arange(len(p["pos"]), dtype=np.int64) if local is None else local))
        base += p.get("n_full", len(p["pos"]))
    sc = {
        "pos": np.concatenate([p["pos"] for p in parts]).astype(F32),
        "vel": np.zeros((n, 3), F32),
        "normal": np.concatenate([p.get("normal", np.zeros((len(p["pos"]), 3), F32)) for p in parts]).astype(F32),
        "color": np.concatenate([p["color"] for p in parts]).astype(F32),
        "type": np.concatenate([np.full(len(p["pos"]), p["type"], np.int32) for p in parts]),
        "group": np.concatenate([np.full(len(p["pos"]), p.get("group", 0), np.int32) for p in parts]),
        "mass": np.concatenate([np.full(len(p["pos"]), p["mass"], F32) for p in parts]),
        "restDensity": np.concatenate([np.full(len(p["pos"]), p["restDensity"], F32) for p in parts]),
        "uniqueId": np.concatenate(uid).astype(np.int32) if parts else np.zeros(0, np.int32),
        "params": params,
    }
    if any("n_full" in p for p in parts):  # a windowed scene says what the whole one looks like
        sc["n_total"] = int(base)
        zs = [p["z_counts"] for p in parts]
        sc["z_counts"] = (np.concatenate([z for z, _ in zs]), np.concatenate([c for _, c in zs]))
    return sc


def fluid_block(xmin, xmax, spacing, density, color=(1, 1, 1, 1), group=0, ptype=TYPE_FLUID, z_select=None) -> dict:
    """Solid block branch of AddMultiphaseFluidVolumes, MultiphaseSPHSolver.cpp:714-735.
    mass = spacing^3 * density in fp32 (:539)."""
    s = F32(spacing)
    ax, ay, az = (lattice_axis(xmin[a], xmax[a], s) for a in range(3))
    mass = F32(F32(F32(s * s) * s) * F32(density))
    out = {"type": ptype, "mass": mass, "restDensity": F32(density), "group": group,
           "z_counts": (az, np.full(len(az), len(ax) * len(ay), np.int64))}
    if z_select is not None:
        kz = np.nonzero(z_select(az))[0]
        nx, ny, nz = len(ax), len(ay), len(az)
        out["n_full"] = nx * ny * nz
        # index of lattice point (ix, iy, iz) in the whole block: x outermost, z innermost
        out["uid_local"] = ((np.arange(nx, dtype=np.int64)[:, None, None] * ny + np.arange(ny, dtype=np.int64)[None, :, None])
                            * nz + kz[None, None, :]).reshape(-1)
        az = az[kz]
    out["pos"] = _grid3(ax, ay, az)
    out["color"] = np.tile(np.asarray(color, F32), (len(out["pos"]), 1))
    return out


def _window(part, z_select) -> dict:
    """The selected particles of a part that was built whole (walls and obstacles: a thin shell of the scene)."""
    z = part["pos"][:, 2]
    vals, counts = np.unique(z, return_counts=True)
    part["z_counts"] = (vals.astype(F32), counts.astype(np.int64))
    if z_select is None:
        return part
    keep = np.nonzero(z_select(z))[0]
    part["n_full"] = len(z)
    part["uid_local"] = keep.astype(np.int64)
    for k in ("pos", "normal", "color"):
        if k in part:
            part[k] = part[k][keep]
    return part


def open_box(xmin, xmax, spacing, group=1, ptype=TYPE_RIGID, lid=False, mass=0.0, rest_density=0.0) -> dict:
    """BoundaryGenerator 'OpenBox' (five faces) / 'Box' (six), solver/ParticleGenerator.h:44-141.
    Inclusive fp32 loops; faces are emitted pairwise in the order x-y, y-z, x-z."""
    s = F32(spacing)
    ax = lattice_axis(xmin[0], xmax[0], s, True)
    ay = lattice_axis(xmin[1], xmax[1], s, True)
    az = lattice_axis(xmin[2], xmax[2], s, True)
    pos, nrm = [], []

    def pair(a, b, make, n0, n1, both=True):
        A, B = np.meshgrid(a, b, indexing="ij")
        A, B = A.reshape(-1), B.reshape(-1)
        p0, p1 = make(A, B, 0), make(A, B, 1)
        if both:
            p = np.empty((2 * len(A), 3), F32)
            p[0::2], p[1::2] = p0, p1
            nn = np.empty((2 * len(A), 3), F32)
            nn[0::2], nn[1::2] = n0, n1
        else:
            p, nn = p0, np.tile(np.asarray(n0, F32), (len(A), 1))
        pos.append(p)
        nrm.append(nn)

    lo, hi = [F32(v) for v in xmin], [F32(v) for v in xmax]
    col = lambda v, like: np.full_like(like, v)
    # x-y faces at z = zmin / zmax (loops: x outer, y inner)
    pair(ax, ay, lambda X, Y, k: np.stack([X, Y, col(hi[2] if k else lo[2], X)], 1), (0, 0, 1), (0, 0, -1))
    # y-z faces at x = xmin / xmax (loops: z outer, y inner)
    pair(az, ay, lambda Z, Y, k: np.stack([col(hi[0] if k else lo[0], Z), Y, Z], 1), (1, 0, 0), (-1, 0, 0))
    # x-z faces at y = ymin (/ ymax for a closed Box) (loops: x outer, z inner)
    pair(ax, az, lambda X, Z, k: np.stack([X, col(hi[1] if k else lo[1], X), Z], 1), (0, 1, 0), (0, -1, 0),
         both=lid)
    pos = np.concatenate(pos).astype(F32)
    return {"pos": pos, "normal": np.concatenate(nrm).astype(F32), "type": ptype, "mass": F32(mass),
            "restDensity": F32(rest_density), "group": group,
            "color": np.tile(np.asarray((1, 1, 1, 0.5), F32), (len(pos), 1))}


def cube(xmin, xmax, spacing, normal=(0, 1, 0), group=0, ptype=TYPE_RIGID, mass=0.0, rest_density=0.0) -> dict:
    """BoundaryGenerator 'Cube': a filled inclusive lattice, solver/ParticleGenerator.h:27-43."""
    s = F32(spacing)
    pos = _grid3(lattice_axis(xmin[0], xmax[0], s, True), lattice_axis(xmin[1], xmax[1], s, True),
                 lattice_axis(xmin[2], xmax[2], s, True))
    return {"pos": pos, "normal": np.tile(np.asarray(normal, F32), (len(pos), 1)), "type": ptype,
            "mass": F32(mass), "restDensity": F32(rest_density), "group": group,
            "color": np.tile(np.asarray((1, 1, 1, 0.5), F32), (len(pos), 1))}


# The shipped Viscosity (0.05) belongs to the IISPH default step.  In the WCSPH force pass the viscous
# term is integrated explicitly and on a rest lattice its damping rate is about 8000 / s per particle
# at 0.05, i.e. rate * DT = 3.2 > 2: velocity differences grow every step.  The WCSPH scenes below
# therefore run with 0.01 (rate * DT = 0.64); everything else is the shipped parameter set.
WCSPH_VISCOSITY = 0.01


def _params(vmin, vmax, boundary_mode=0, **over):
    p = dict(REF_PARAMS)
    p.update(vmin=tuple(float(v) for v in vmin), vmax=tuple(float(v) for v in vmax), boundary_mode=boundary_mode)
    p.update(over)
    return p


def rigid_mass(spacing, restdensity):
    """LoadPO: mp = spacing^3 * restdensity * 2 in fp32, MultiphaseSPHSolver.cpp:751-753."""
    s = F32(spacing)
    return F32(F32(F32(F32(s * s) * s) * F32(restdensity)) * F32(2))


def reference_scene(with_boundary=True) -> dict:
    """The shipped scene: config/sph_scene.xml SceneId 2 'Single Dam Break' (:57-86),
    52 272 fluid particles (33 x 48 x 33) with VolFrac 0.5,0.5,0 of DensityArray 500,1000,1500
    (density 750), plus the OpenBox of config/big box.xml:29-34 as TYPE_RIGID (LoadPO, :750-768)."""
    sp = REF_PARAMS["spacing"]
    density = F32(F32(500) * F32(0.5) + F32(1000) * F32(0.5) + F32(1500) * F32(0))
    parts = [fluid_block((-0.08, 0.015, -0.08), (0.081, 0.25, 0.081), sp, density, color=(0.5, 0.5, 0, 1))]
    if with_boundary:
        parts.append(open_box((-0.25, -0.006, -0.25), (0.25, 0.45, 0.25), sp, group=1, ptype=TYPE_RIGID,
                              mass=rigid_mass(sp, REF_PARAMS["restdensity"]), rest_density=0.0))
    return _finish(parts, _params(REF_DOMAIN_MIN, REF_DOMAIN_MAX))


def dam_break_geometry(nx, ny, nz, tank_x_factor=4.0, head_room=1.5, spacing=0.005, boundary_mode=1) -> dict:
    """Boxes of the benchmark dam break (all float64; the particle generators round them to fp32):
    grid box, fluid block corners for the half-open lattice loops, tank corners, parameters."""
    s = F32(spacing)
    sr = float(s) * 2.0
    L = np.array([nx, ny, nz], dtype=np.float64) * float(s)
    inset = 2 * sr                        # wall plane sits two cells inside the grid box
    # The penalty walls repel out to two smoothing radii; four spacings is where that force starts, so a resting
    # block put there settles onto the floor instead of being shot off it.
    gap = 4.0 * float(s)
    tank = np.array([L[0] * tank_x_factor, L[1] * head_room, L[2]]) + 2 * gap
    vmin = np.array([-inset, -inset, -inset])
    # round the box up to a whole number of cells
    ncell = np.ceil((tank + 2 * inset) / sr - 1e-9)
    vmax = vmin + ncell * sr
    f_lo = np.array([gap, gap, gap])
    eps = 0.25 * float(s)
    f_hi = f_lo + (np.array([nx, ny, nz]) - 1) * float(s) + eps
    # every box travels through text files too (host/config): keep what survives a decimal round trip
    rt = lambda v: tuple(float(repr(float(x))) for x in v)
    return {"fluid_lo": rt(f_lo), "fluid_hi": rt(f_hi), "tank_lo": (0.0, 0.0, 0.0), "tank_hi": rt(tank),
            "params": _params(rt(vmin), rt(vmax), boundary_mode=boundary_mode, viscosity=WCSPH_VISCOSITY)}


def dam_break(nx, ny, nz, tank_x_factor=4.0, head_room=1.5, wall_type=TYPE_BOUNDARY, spacing=0.005,
              boundary_mode=1, walls=True, z_select=None) -> dict:
    """Benchmark dam break: an nx*ny*nz fluid lattice (rest density 1000) standing in one corner of
    a tank `tank_x_factor` times as long in x.  Tank walls and floor are one layer of wall particles
    (TYPE_BOUNDARY, so ForceCell's penalty branch acts) placed two cells inside the grid box;
    boundary_mode 1 adds the legacy clamp to the grid box as a safety net."""
    s = F32(spacing)
    g = dam_break_geometry(nx, ny, nz, tank_x_factor, head_room, spacing, boundary_mode)
    fluid = fluid_block(g["fluid_lo"], g["fluid_hi"], s, 1000.0, color=(0.2, 0.4, 1, 1), z_select=z_select)
    parts = [fluid]
    if walls:
        parts.append(_window(open_box(g["tank_lo"], g["tank_hi"], s, group=1, ptype=wall_type,
                                      mass=rigid_mass(s, 1000.0), rest_density=1000.0), z_select))
    sc = _finish(parts, g["params"])
    sc["n_fluid"] = fluid.get("n_full", len(fluid["pos"]))  # of the whole scene
    return sc


def double_dam_break(nx, ny, nz, tank_x_factor=4.0, head_room=1.5, spacing=0.005, obstacles=2, z_select=None) -> dict:
    """Double dam break in small: two nx*ny*nz fluid blocks at the two x ends of the tank and `obstacles`
    box obstacles between them (filled TYPE_BOUNDARY lattices, the generator of BoundaryGenerator 'Cube',
    solver/ParticleGenerator.h:27-43), staggered in z so that the flow is not symmetric along z."""
    s = F32(spacing)
    g = dam_break_geometry(nx, ny, nz, tank_x_factor, head_room, spacing, 1)
    lo, hi = np.array(g["fluid_lo"]), np.array(g["fluid_hi"])
    tank = np.array(g["tank_hi"])
    ext = hi - lo
    a = fluid_block(lo, hi, s, 1000.0, color=(0.2, 0.4, 1, 1), group=0, z_select=z_select)
    shift = np.array([tank[0] - 2 * lo[0] - (nx - 1) * float(s), 0.0, 0.0])
    b = fluid_block(lo + shift, hi + shift, s, 1000.0, color=(1, 0.4, 0.2, 1), group=1, z_select=z_select)
    parts = [a, b, _window(open_box(g["tank_lo"], g["tank_hi"], s, group=2, ptype=TYPE_BOUNDARY,
                                    mass=rigid_mass(s, 1000.0), rest_density=1000.0), z_select)]
    side = max(4 * float(s), 0.15 * ext[2])
    for k in range(obstacles):
        cx = tank[0] * (k + 1) / (obstacles + 1)
        cz = lo[2] + ext[2] * (0.25 if k % 2 == 0 else 0.75)
        o_lo = (cx - side / 2, lo[1], cz - side / 2)
        o_hi = (cx + side / 2, lo[1] + 0.5 * ext[1], cz + side / 2)
        parts.append(_window(cube(o_lo, o_hi, s, normal=(0, 1, 0), group=3 + k, ptype=TYPE_BOUNDARY,
                                  mass=rigid_mass(s, 1000.0), rest_density=1000.0), z_select))
    sc = _finish(parts, g["params"])
    sc["n_fluid"] = a.get("n_full", len(a["pos"])) + b.get("n_full", len(b["pos"]))  # of the whole scene
    return sc


def ball_drop(center=(0.0, 0.30, 0.0), radius=0.10, floor=True) -> dict:
    """Ball drop: the lattice points (spacing 0.005) inside a sphere (ball.raw is a unit sphere; x0.1),
    in the shipped grid box with the shipped scene parameters, above a TYPE_BOUNDARY floor plate."""
    sp = F32(REF_PARAMS["spacing"])
    c = np.asarray(center, F32)
    r = F32(radius)
    blk = fluid_block(c - r, c + r + F32(0.5) * sp, sp, 1000.0, color=(0.2, 0.4, 1, 1))
    d = blk["pos"] - c
    keep = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]) < r * r
    blk["pos"] = blk["pos"][keep]
    blk["color"] = blk["color"][keep]
    parts = [blk]
    if floor:
        parts.append(cube((-0.25, 0.0, -0.25), (0.25, 0.0, 0.25), sp, ptype=TYPE_BOUNDARY,
                          mass=rigid_mass(sp, 1000.0), rest_density=1000.0))
    sc = _finish(parts, _params(REF_DOMAIN_MIN, REF_DOMAIN_MAX, boundary_mode=1, viscosity=WCSPH_VISCOSITY))
    sc["n_fluid"] = int(keep.sum())
    return sc


# <SurfaceTension> of the shipped scene 2 (config/sph_scene.xml:71); <BoundInfo> / <Friction> exist only in
# the last scene of that file (:333-334), the shipped scene reads 0 for both
IISPH_PARAMS = dict(surface_tension=100.0, boundary_visc=0.0, boundary_friction=0.0)


def iisph_tank(nx, ny, nz, wall_group=1, fluid_groups=(0,), spacing=0.005, margin=4, head_room=2.0, **over) -> dict:
    """A scene for the IISPH step: nx x ny x nz fluid particles per entry of `fluid_groups` (blocks side by
    side along x, each with its own group id) standing in an OpenBox of TYPE_RIGID particles (LoadPO,
    MultiphaseSPHSolver.cpp:750-768: mass = 2 spacing^3 restdensity, group = the object's id, normals
    pointing inwards), as SetupFluidScene :815-835 builds the shipped scene."""
    s = float(spacing)
    parts = []
    x0 = 0.0
    for g in fluid_groups:
        lo = (x0, 0.0, 0.0)
        hi = (x0 + (nx - 0.5) * s, (ny - 0.5) * s, (nz - 0.5) * s)
        parts.append(fluid_block(lo, hi, s, REF_PARAMS["restdensity"], group=int(g)))
        x0 += nx * s
    b_lo = (-1.2 * s, -1.2 * s, -1.2 * s)
    b_hi = (x0 + 0.2 * s, ny * s * head_room, (nz + 0.2) * s)
    parts.append(open_box(b_lo, b_hi, s, group=int(wall_group), ptype=TYPE_RIGID,
                          mass=rigid_mass(s, REF_PARAMS["restdensity"]), rest_density=0.0))
    sr = s * REF_PARAMS["smoothratio"]
    vmin = tuple(v - margin * sr for v in b_lo)
    # a whole number of cells per axis
    vmax = tuple(lo + np.ceil((hi + margin * sr - lo) / sr) * sr for lo, hi in zip(vmin, b_hi))
    prm = dict(IISPH_PARAMS)
    prm.update(over)
    return _finish(parts, _params(vmin, vmax, **prm))


def read_raw_bounding_sphere(path):
    """RAW triangle soup, 9 floats per line (the format ObjReader::readraw parses,
    catpaw/objbuilder.cpp:150-187): returns (centre, radius) of the vertex cloud."""
    v = np.loadtxt(path, dtype=np.float64).reshape(-1, 3)
    c = 0.5 * (v.min(0) + v.max(0))
    return c, float(np.sqrt(((v - c) ** 2).sum(1)).max())


def jittered(scene: dict, seed=12345, pos_jitter=0.2, vel_scale=1.0, fluid_only=False) -> dict:
    """Randomised parity input: lattice + U(-j,j)*spacing, velocities U(-v,v); PCG64 stream `seed`.
    With fluid_only the wall particles stay where they are."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sc = dict(scene)
    sp = scene["params"]["spacing"]
    n = len(scene["pos"])
    dp = rng.uniform(-pos_jitter, pos_jitter, (n, 3)).astype(F32) * F32(sp)
    dv = rng.uniform(-vel_scale, vel_scale, (n, 3)).astype(F32) if vel_scale > 0 else np.zeros((n, 3), F32)
    if fluid_only:
        keep = (scene["type"] != TYPE_FLUID)[:, None]
        dp = np.where(keep, F32(0), dp)
        dv = np.where(keep, F32(0), dv)
    sc["pos"] = (scene["pos"] + dp).astype(F32)
    sc["vel"] = dv.astype(F32)
    return sc
