"""The oracle against the only known answers this path has.

The reference ships no tests or fixtures.  What exists are a few values recorded while the reference
kernels were studied (shipped scene, config/sph_scene.xml SceneId 2): particle (0,0,0) lies in cell
(40,10,40) = hash 67 240; (0.4,0,0) is outside the grid; the fluid block holds 33 x 48 x 33 = 52 272
particles; with mass = spacing^3 * 1000 the first density pass gives mean 975.8, max 999.975.
"""
import numpy as np

from fluidsolver_amd import scenes
from tests.util import oracle_pair


def test_shipped_grid_and_hash(orc32):
    sc = scenes.reference_scene(with_boundary=False)
    p, st = oracle_pair(orc32, sc)
    assert list(p.gridres) == [80, 110, 80]
    assert st.n == 52272
    two = orc32.new_state(2, 80 * 110 * 80)
    two["pos"][0] = (0, 0, 0)
    two["pos"][1] = (0.4, 0, 0)
    two.calc_hash(p)
    assert two["particleHash"].tolist() == [10 * 80 * 80 + 40 * 80 + 40, 99999999]
    assert two["particleIndex"].tolist() == [0, 1]


def test_shipped_block_first_density(orc32):
    sc = scenes.reference_scene(with_boundary=False)
    n = len(sc["pos"])
    sc["mass"] = np.full(n, np.float32(0.005) ** 3 * np.float32(1000), np.float32)
    sc["restDensity"] = np.full(n, 1000, np.float32)
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.compute_pressure(p)
    assert abs(st["density"].mean(dtype=np.float64) - 975.8) < 0.05
    assert abs(float(st["density"].max()) - 999.975) < 0.001
    # below rest density everywhere: the clamped Tait pressure is zero and the block is in free fall
    assert float(st["pressure"].max()) == 0.0
    st.compute_force(p)
    assert np.allclose(st["force"], [0, np.float32(-9.8), 0])


def test_boundary_particle_count_of_shipped_box():
    sc = scenes.reference_scene(with_boundary=True)
    n_rigid = int((sc["type"] == scenes.TYPE_RIGID).sum())
    # OpenBox of config/big box.xml: 101 x 92 lattice on four walls (pairs), 101 x 101 floor
    ax = len(scenes.lattice_axis(-0.25, 0.25, 0.005, True))
    ay = len(scenes.lattice_axis(-0.006, 0.45, 0.005, True))
    assert n_rigid == 2 * ax * ay + 2 * ax * ay + ax * ax
    assert 40000 < n_rigid < 50000
