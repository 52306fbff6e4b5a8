"""C++ host layer (fluidsolver_amd/host): the plugin classes main.cpp drives, the config readers, the
headless GUI stand-in.

CPU part: the scene the C++ code builds from host/config equals, bit for bit, the scene of
fluidsolver_amd.scenes that the oracle parity tests use; the reference's own main.cpp compiles and links
against these headers unchanged (only where the reference tree is present).
GPU part: the built program runs the scene through the C ABI and ends in the same state as the Python
binding driving the same library.
"""
import os
import subprocess

import numpy as np
import pytest

from fluidsolver_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "fluidsolver_amd", "host")
REF = "/root/reference/particle_solver"


def have_bins():
    return all(os.path.exists(os.path.join(HOST, "bin", b)) for b in ("particle_solver", "scene_dump"))


def env_for(scene_id):
    e = dict(os.environ)
    e.update(FSA_CONFIG_DIR=os.path.join(HOST, "config"), FSA_SCENE_ID=str(scene_id))
    return e


def read_scene_dump(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    res = raw[4:16].view(np.int32).tolist()
    consts = raw[16:40].view(np.float64).tolist()
    o = 40
    out = {"n": n, "gridres": res, "dx": consts[0], "kernel_cubic": consts[1], "kspikydiff": consts[2]}
    for name, dt, w in (("pos", np.float32, 3), ("color", np.float32, 4), ("normal", np.float32, 3),
                        ("type", np.int32, 1), ("group", np.int32, 1), ("mass", np.float32, 1),
                        ("restDensity", np.float32, 1)):
        nb = n * w * 4
        a = raw[o:o + nb].view(dt)
        out[name] = a.reshape(n, w) if w > 1 else a
        o += nb
    assert o == raw.size
    return out


@pytest.mark.skipif(not have_bins(), reason="host layer not built (make -C fluidsolver_amd/host)")
@pytest.mark.parametrize("scene_id,build", [(2, lambda: scenes.reference_scene()),
                                            (20, lambda: scenes.dam_break(64, 32, 64))])
def test_cpp_scene_setup_equals_python_scene(tmp_path, scene_id, build):
    out = tmp_path / "scene.bin"
    subprocess.check_call([os.path.join(HOST, "bin", "scene_dump"), str(out)], env=env_for(scene_id),
                          stdout=subprocess.DEVNULL)
    got = read_scene_dump(out)
    sc = build()
    assert got["n"] == len(sc["pos"])
    for k in ("pos", "normal", "type", "group", "mass", "restDensity"):
        assert np.ascontiguousarray(got[k]).tobytes() == np.ascontiguousarray(sc[k]).tobytes(), k
    if scene_id == 2:
        assert np.array_equal(got["color"], sc["color"])
    from fluidsolver_amd import capi
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    assert got["gridres"] == list(p.gridres)
    assert (got["dx"], got["kernel_cubic"], got["kspikydiff"]) == (p.dx, p.kernel_cubic, p.kspikydiff)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_reference_main_compiles_against_host_headers(tmp_path):
    """main.cpp of the reference, where it lies, with its three includes resolved to this layer."""
    subprocess.check_call(["make", "-C", HOST, "refmain", f"REF={REF}", f"OUT={tmp_path}"],
                          stdout=subprocess.DEVNULL)
    exe = tmp_path / "particle_solver_refmain"
    assert exe.exists()
    syms = subprocess.check_output(["nm", "-C", "--undefined-only", str(exe)], text=True)
    assert "fsa_step" in syms or "fsa_compute_force" in syms  # it calls into the C ABI, nothing else does the work


@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_xml_reader_conventions(tmp_path):
    """missing tags read as 0, comments are skipped, several top-level elements are fine"""
    cfg = tmp_path / "config"
    cfg.mkdir()
    (cfg / "sph_scene.xml").write_text("""<?xml version="1.0"?>
<!-- a comment with <tags> inside -->
<Fluid><Gravity> 0, -10, 0 </Gravity><VolMin> -0.1,-0.1,-0.1 </VolMin><VolMax> 0.1, 0.1, 0.1 </VolMax>
<SceneId> 7 </SceneId></Fluid>
<MultiScene id="3"><PSpacing> 0.5 </PSpacing></MultiScene>
<MultiScene id="7"><TypeNum> 2 </TypeNum><DT> 0.001 </DT><PSpacing> 0.01 </PSpacing><SmoothRatio> 2 </SmoothRatio>
 <DensityArray> 800,1200 </DensityArray><ViscosityArray> 1,1 </ViscosityArray>
 <!--Vol><VolMin> 9,9,9 </VolMin></Vol-->
 <Vol><VolMin> 0,0,0 </VolMin><VolMax> 0.0301, 0.0201, 0.0101 </VolMax><VolFrac> 0.25, 0.75 </VolFrac></Vol>
</MultiScene>""")
    (cfg / "big box.xml").write_text("<Boundary><Cube><xmin>0,0,0</xmin><xmax>0.01,0,0</xmax>"
                                     "<spacing>0.01</spacing><normal>0,1,0</normal><id>5</id></Cube></Boundary>")
    e = dict(os.environ, FSA_CONFIG_DIR=str(cfg))
    e.pop("FSA_SCENE_ID", None)
    out = tmp_path / "s.bin"
    subprocess.check_call([os.path.join(HOST, "bin", "scene_dump"), str(out)], env=e, stdout=subprocess.DEVNULL)
    got = read_scene_dump(out)
    nx = len(scenes.lattice_axis(0, 0.0301, 0.01))
    ny = len(scenes.lattice_axis(0, 0.0201, 0.01))
    nz = len(scenes.lattice_axis(0, 0.0101, 0.01))
    n_fluid = nx * ny * nz
    assert got["n"] == n_fluid + 2
    assert got["gridres"] == [10, 10, 10]
    f32 = np.float32
    dens = f32(f32(800) * f32(0.25)) + f32(f32(1200) * f32(0.75))
    assert np.all(got["restDensity"][:n_fluid] == dens)
    assert np.all(got["type"][:n_fluid] == 0) and np.all(got["type"][n_fluid:] == 4)
    assert np.all(got["group"][n_fluid:] == 5)
    assert np.all(got["mass"][n_fluid:] == 0)  # RestDensity tag is missing: reads as 0


def read_final(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    pos = raw[4:4 + n * 12].view(np.float32).reshape(n, 3)
    col = raw[4 + n * 12:4 + n * 28].view(np.float32).reshape(n, 4)
    return pos, col


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
@pytest.mark.parametrize("scene_id", [2, 20])
def test_headless_program_matches_python_binding(tmp_path, scene_id):
    from fluidsolver_amd import capi
    frames = 12
    final = tmp_path / "final.bin"
    dump_dir = tmp_path / "dump"
    dump_dir.mkdir()
    e = env_for(scene_id)
    e["FSA_DUMP_DIR"] = str(dump_dir)
    subprocess.check_call([os.path.join(HOST, "bin", "particle_solver"), "--frames", str(frames), "--save-final",
                           str(final), "--keys", "b"], env=e, stdout=subprocess.DEVNULL, timeout=300)
    pos, col = read_final(final)

    sc = scenes.reference_scene() if scene_id == 2 else scenes.dam_break(64, 32, 64)
    if scene_id == 20:
        sc = dict(sc)
        sc["color"] = np.tile(np.asarray((1, 0, 0, 1), np.float32), (len(sc["pos"]), 1))
        sc["color"][sc["type"] != 0] = (1, 1, 1, 0.5)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.sort_particles()
    s.rigid_particle_volume()
    s.step(frames)
    want_pos, want_col = s.readback_pos_color()
    assert pos.tobytes() == want_pos.tobytes()
    assert col.tobytes() == want_col.tobytes()

    # the text checkpoint written by key 'b' holds the same state (6 decimals)
    lines = (dump_dir / f"{frames:03d}.txt").read_text().splitlines()
    assert int(lines[0]) == len(pos)
    rows = np.array([l.split() for l in lines[1:]], dtype=np.float64)
    assert np.abs(rows[:, 0:3] - want_pos).max() < 1e-6
    assert np.abs(rows[:, 3:6] - s.get("vel")).max() < 1e-6
    assert np.array_equal(rows[:, 6].astype(int), s.get("type"))
    assert np.array_equal(rows[:, 9].astype(int), s.get("uniqueId"))
    s.close()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_text_checkpoint_resume(tmp_path):
    """10 frames in one go against 5 frames, text checkpoint (key 'b'), resume, 5 more frames.  The dump
    prints six decimals, so the resumed run agrees to that precision, not bit for bit."""
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = env_for(20)
    dump = tmp_path / "dump"
    dump.mkdir()
    e["FSA_DUMP_DIR"] = str(dump)
    a, c = tmp_path / "a.bin", tmp_path / "c.bin"
    subprocess.check_call([exe, "--frames", "10", "--save-final", str(a)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    subprocess.check_call([exe, "--frames", "5", "--keys", "b"], env=e, stdout=subprocess.DEVNULL, timeout=300)
    ckpt = dump / "005.txt"
    assert ckpt.exists()
    e2 = dict(e, FSA_RESUME=str(ckpt))
    subprocess.check_call([exe, "--frames", "5", "--save-final", str(c)], env=e2, stdout=subprocess.DEVNULL, timeout=300)
    pa, _ = read_final(a)
    pc, _ = read_final(c)
    assert pa.shape == pc.shape
    # both runs end in their own sorted order; compare as sets of points through a lexicographic sort
    ka = np.lexsort((pa[:, 2], pa[:, 1], pa[:, 0]))
    kc = np.lexsort((pc[:, 2], pc[:, 1], pc[:, 0]))
    assert np.abs(np.sort(pa[:, 1]) - np.sort(pc[:, 1])).max() < 2e-5
    assert np.abs(pa[ka].mean(0) - pc[kc].mean(0)).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
@pytest.mark.parametrize("slabs", [2, 3])
def test_headless_program_on_slabs_equals_single_solver(tmp_path, slabs):
    """FSA_SLABS=N: the same host program, the scene on N z-slab solvers driven as one group of the library (all on
    the one GPU here): positions and colours of every frame's read-back are those of the single-solver run, byte for
    byte"""
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = env_for(20)
    a, b = tmp_path / "single.bin", tmp_path / "slabs.bin"
    subprocess.check_call([exe, "--frames", "15", "--save-final", str(a)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    r = subprocess.run([exe, "--frames", "15", "--save-final", str(b)], env=dict(e, FSA_SLABS=str(slabs)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    assert f"on {slabs} z-slabs" in r.stdout
    assert a.read_bytes() == b.read_bytes()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_headless_program_ends_loudly_when_a_slab_outgrows_its_capacity(tmp_path):
    """the slabs of FSA_SLABS are sized at the start and never re-cut: with no head room at all (factor 1, no spare
    particles) the first slab that gains a particle ends the run at the next read-back, non-zero, naming the switch"""
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = dict(env_for(20), FSA_SLABS="3", FSA_SLAB_HEADROOM="1.0", FSA_SLAB_SPARE="0")
    r = subprocess.run([exe, "--frames", "200", "--save-final", str(tmp_path / "x.bin")], env=e, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "FSA_SLAB_HEADROOM" in r.stderr and "outgrew" in r.stderr
    assert not (tmp_path / "x.bin").exists()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_corrupted_checkpoint_is_refused(tmp_path):
    """ids outside 0 .. n-1, repeated ids, a particle count far beyond the file and a truncated file: each must end
    in "malformed checkpoint" and leave the set-up scene running, not write through a bad index"""
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = env_for(20)
    dump = tmp_path / "dump"
    dump.mkdir()
    e["FSA_DUMP_DIR"] = str(dump)
    good = tmp_path / "good.bin"
    subprocess.check_call([exe, "--frames", "2", "--keys", "b", "--save-final", str(good)], env=e,
                          stdout=subprocess.DEVNULL, timeout=300)
    lines = (dump / "002.txt").read_text().splitlines()
    n = int(lines[0])

    def with_id(k, new_id):
        f = lines[k].split()
        f[9] = str(new_id)
        return " ".join(f)

    cases = {
        "negative id": [lines[0], with_id(1, -7)] + lines[2:],
        "id beyond n": [lines[0], with_id(1, n + 1000000)] + lines[2:],
        "repeated id": [lines[0], lines[1], with_id(2, lines[1].split()[9])] + lines[3:],
        "count beyond file": [str(2000000000)] + lines[1:50],
        "truncated": lines[: n // 2],
    }
    for name, text in cases.items():
        bad = tmp_path / "bad.txt"
        bad.write_text("\n".join(text) + "\n")
        out = tmp_path / "out.bin"
        r = subprocess.run([exe, "--frames", "2", "--save-final", str(out)], env=dict(e, FSA_RESUME=str(bad)),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stderr[-500:])
        assert "malformed checkpoint" in r.stdout, name
        # the run went on from the scene the set-up built
        assert out.read_bytes() == good.read_bytes(), name


@pytest.mark.skipif(not have_bins(), reason="host layer not built")
@pytest.mark.parametrize("mode", ["IISPH", "DFSPH"])
def test_resume_is_refused_in_modes_that_cannot_honour_it(tmp_path, mode):
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = dict(env_for(20), FSA_RUN_MODE=mode, FSA_RESUME=str(tmp_path / "none.txt"))
    r = subprocess.run([exe, "--frames", "1"], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "FSA_RESUME" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_headless_program_dfsph_mode(tmp_path):
    from fluidsolver_amd import capi
    frames = 6
    final = tmp_path / "final.bin"
    e = env_for(20)
    e["FSA_RUN_MODE"] = "DFSPH"
    subprocess.check_call([os.path.join(HOST, "bin", "particle_solver"), "--frames", str(frames), "--save-final",
                           str(final)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    pos, _ = read_final(final)
    sc = scenes.dam_break(64, 32, 64)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    s.setup_dfsph()
    s.step_dfsph(frames)
    want, _ = s.readback_pos_color()
    assert pos.tobytes() == want.tobytes()
    s.close()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_headless_program_iisph_mode_on_the_shipped_scene(tmp_path):
    """the step the reference's Setup() selects, on the scene it ships: fluid block of scene 2 in the
    TYPE_RIGID box of 'big box.xml'"""
    from fluidsolver_amd import capi
    frames = 6
    final = tmp_path / "final.bin"
    e = env_for(2)
    e["FSA_RUN_MODE"] = "IISPH"
    subprocess.check_call([os.path.join(HOST, "bin", "particle_solver"), "--frames", str(frames), "--save-final",
                           str(final)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    pos, _ = read_final(final)
    sc = scenes.reference_scene()
    sc["params"] = dict(sc["params"], surface_tension=100.0)  # <SurfaceTension> of scene 2
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    s = capi.Solver(p, len(sc["pos"]))
    s.upload(sc)
    # SetupIISPH: Sort, EffectiveMass (VolFrac 0.5,0.5,0 of DensityArray 500,1000,1500), IISPHFactor, RigidParticleVolume
    s.sort_particles()
    vf = np.tile(np.asarray((0.5, 0.5, 0.0), np.float32), (len(sc["pos"]), 1))
    vf[sc["type"][s.get("uniqueId")] != 0] = 0  # LoadPO leaves the wall particles' fractions at zero
    s.set_volume_fractions((500.0, 1000.0, 1500.0), vf)
    s.effective_mass()
    s.iisph_factor()
    s.rigid_particle_volume()
    s.step_iisph(frames)
    want, _ = s.readback_pos_color()
    assert np.isfinite(want).all()
    assert pos.tobytes() == want.tobytes()
    fluid = s.get("type") == 0
    assert s.get("density")[fluid].max() < 1.05 * 750.0  # the block's rest density
    s.close()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
@pytest.mark.parametrize("slabs", [2, 3])
def test_headless_program_iisph_on_slabs_equals_single_solver(tmp_path, slabs):
    """FSA_SLABS=N with the step the reference's Setup() selects (IISPH), on the scene it ships (fluid block in the
    TYPE_RIGID box): the read-back of the last frame is that of the single-solver run, byte for byte"""
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = env_for(2)
    e["FSA_RUN_MODE"] = "IISPH"
    a, b = tmp_path / "single.bin", tmp_path / "slabs.bin"
    subprocess.check_call([exe, "--frames", "8", "--save-final", str(a)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    r = subprocess.run([exe, "--frames", "8", "--save-final", str(b)], env=dict(e, FSA_SLABS=str(slabs)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    assert f"on {slabs} z-slabs" in r.stdout and "IISPH step" in r.stdout
    assert a.read_bytes() == b.read_bytes()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
@pytest.mark.parametrize("slabs", [2, 3])
def test_headless_program_dfsph_on_slabs_equals_single_solver(tmp_path, slabs):
    """FSA_SLABS=N FSA_RUN_MODE=DFSPH: the read-back of the last frame is that of the single-solver run, byte for byte"""
    exe = os.path.join(HOST, "bin", "particle_solver")
    e = env_for(20)
    e["FSA_RUN_MODE"] = "DFSPH"
    a, b = tmp_path / "single.bin", tmp_path / "slabs.bin"
    subprocess.check_call([exe, "--frames", "8", "--save-final", str(a)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    r = subprocess.run([exe, "--frames", "8", "--save-final", str(b)], env=dict(e, FSA_SLABS=str(slabs)),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    assert f"on {slabs} z-slabs" in r.stdout and "DFSPH step" in r.stdout
    assert a.read_bytes() == b.read_bytes()


def grid_tracers():
    """addParticles (solver/grid_solver.cpp:625-637 of the reference): float loop counters, double bounds"""
    f = np.float32
    out = []
    i = f(0)
    while float(i) < 0.2:
        j = f(0)
        while float(j) < 0.1:
            k = f(0)
            while float(k) < 0.2:
                out.append((f(-0.1) + i, f(0.01) + j, f(-0.1) + k))
                k = f(float(k) + 0.01)
            j = f(float(j) + 0.01)
        i = f(float(i) + 0.01)
    return np.array(out, np.float32)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "bin", "grid_solver")), reason="host layer not built")
def test_headless_grid_solver_matches_python_binding(tmp_path):
    """GridSolver::setup + Step through the plugin class on the shipped Eulerian.xml values"""
    from fluidsolver_amd import capi
    frames = 4
    final = tmp_path / "grid.bin"
    e = dict(os.environ, FSA_CONFIG_DIR=os.path.join(HOST, "config"))
    subprocess.check_call([os.path.join(HOST, "bin", "grid_solver"), "--frames", str(frames), "--save-final", str(final)],
                          env=e, stdout=subprocess.DEVNULL, timeout=300)
    raw = np.fromfile(final, dtype=np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    off = 4
    pos = raw[off:off + n * 12].view(np.float32).reshape(n, 3)
    off += n * 12
    fields = {}
    for name in ("u", "v", "w", "p"):
        m = int(raw[off:off + 4].view(np.int32)[0])
        fields[name] = raw[off + 4:off + 4 + 4 * m].view(np.float32)
        off += 4 + 4 * m
    g = capi.GridSolver((16, 32, 16), padding=1, h=0.02, dt=0.001, rho=1.0)
    tr = grid_tracers()
    assert len(tr) == n
    g.set_particles(tr)
    g.step(frames)
    want_pos, _ = g.get_particles()
    assert pos.tobytes() == want_pos.tobytes()
    for name in ("u", "v", "w", "p"):
        assert fields[name].tobytes() == g.get(name).tobytes(), name
    assert np.abs(fields["v"]).max() > 0
    g.close()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(HOST, "bin", "grid_solver")), reason="host layer not built")
@pytest.mark.parametrize("slabs", [2, 4])
def test_headless_grid_solver_on_slabs_equals_the_single_grid(tmp_path, slabs):
    """FSA_GRID_SLABS=N: the plugin class on a team of z-slabs writes the same file, byte for byte"""
    e = dict(os.environ, FSA_CONFIG_DIR=os.path.join(HOST, "config"))
    exe = os.path.join(HOST, "bin", "grid_solver")
    a, b = tmp_path / "one.bin", tmp_path / "slabs.bin"
    subprocess.check_call([exe, "--frames", "4", "--save-final", str(a)], env=e, stdout=subprocess.DEVNULL, timeout=300)
    out = subprocess.run([exe, "--frames", "4", "--save-final", str(b)], env=dict(e, FSA_GRID_SLABS=str(slabs)),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert f"grid on {slabs} z-slabs" in out.stdout
    assert a.read_bytes() == b.read_bytes()


@pytest.mark.gpu
@pytest.mark.skipif(not have_bins(), reason="host layer not built")
def test_run_mode_tag_of_the_scene_selects_the_step(tmp_path):
    """<RunMode> IISPH </RunMode> inside the scene does what FSA_RUN_MODE=IISPH does; an unknown word fails loudly"""
    import shutil
    cfg = tmp_path / "config"
    shutil.copytree(os.path.join(HOST, "config"), cfg)
    text = (cfg / "sph_scene.xml").read_text()
    marker = '<MultiScene id="2">'
    assert marker in text
    exe = os.path.join(HOST, "bin", "particle_solver")

    def run(config_dir, extra_env, out):
        e = dict(os.environ, FSA_CONFIG_DIR=str(config_dir), FSA_SCENE_ID="2", **extra_env)
        e.pop("FSA_RUN_MODE", None) if "FSA_RUN_MODE" not in extra_env else None
        return subprocess.run([exe, "--frames", "3", "--save-final", str(out)], env=e, stdout=subprocess.DEVNULL,
                              stderr=subprocess.PIPE, timeout=300)

    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    assert run(os.path.join(HOST, "config"), {"FSA_RUN_MODE": "IISPH"}, a).returncode == 0
    (cfg / "sph_scene.xml").write_text(text.replace(marker, marker + "\n\t<RunMode> IISPH </RunMode>", 1))
    assert run(cfg, {}, b).returncode == 0
    assert read_final(a)[0].tobytes() == read_final(b)[0].tobytes()
    (cfg / "sph_scene.xml").write_text(text.replace(marker, marker + "\n\t<RunMode> PBF </RunMode>", 1))
    bad = run(cfg, {}, tmp_path / "c.bin")
    assert bad.returncode != 0 and b"RunMode" in bad.stderr
