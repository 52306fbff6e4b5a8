"""fsa_step runs a step in seven launches (the integration computes the next hash on the way, both prefix sums
share two launches, the rank pass moves the particles and writes the row table).  It must leave every buffer as
the staged calls of the reference's launcher seam do, bit for bit, in any interleaving with them."""
import numpy as np
import pytest

from fluidsolver_amd import scenes
from tests.test_gpu_parity import SCENES
from tests.util import gpu_solver

pytestmark = pytest.mark.gpu

NAMES = ("pos", "vel", "force", "v_star", "density", "pressure", "uniqueId", "type", "mass", "restDensity",
         "particleHash", "cellStartDense")


def staged_step(s):
    s.calc_hash()
    s.sort_particle()
    s.reorder()
    s.compute_pressure()
    s.compute_force()
    s.advect()


def snapshot(s):
    return {k: s.get(k) for k in NAMES}


def same(a, b):
    for k in NAMES:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("scene", sorted(SCENES))
@pytest.mark.parametrize("precision,variant", [(0, 3), (0, 0), (1, 0)])
def test_fused_step_equals_staged_calls(scene, precision, variant):
    sc = SCENES[scene]()
    _, a = gpu_solver(sc, precision)
    _, b = gpu_solver(sc, precision)
    a.set_variant(variant)
    b.set_variant(variant)
    for _ in range(6):
        staged_step(a)
    b.step(6)
    same(snapshot(a), snapshot(b))
    # interleaved: the hash the last fused step computed ahead must serve the staged calls that follow, and a
    # position upload must discard it
    staged_step(b)
    staged_step(a)
    same(snapshot(a), snapshot(b))
    pos = a.get("pos")
    pos[: len(pos) // 3, 0] += np.float32(0.004)
    a.set("pos", pos)
    b.set("pos", pos)
    a.step(2)
    staged_step(b)
    staged_step(b)
    same(snapshot(a), snapshot(b))
    a.close()
    b.close()


def test_fused_step_row_table_large_rows():
    """a lattice lined up with the cell faces has rows beyond one tile (class 2): the row table the rank pass
    writes must drive the tiled passes to the same bits as the staged calls' table"""
    sc = scenes.dam_break(40, 12, 12, tank_x_factor=1.5)
    _, a = gpu_solver(sc)
    _, b = gpu_solver(sc)
    for _ in range(3):
        staged_step(a)
    b.step(3)
    same(snapshot(a), snapshot(b))
    a.close()
    b.close()
