"""The oracle's WCSPH pair terms against answers worked out by hand from the reference text
(solver/sph_kernel_impl.cuh:32-40 Kernel_Cubic, :246-266 ComputePressureKernel, :280-352 ForceCell, :372 gravity
literal; constants of MultiphaseSPHSolver.cpp:464-469).  Every expected number below is a literal with its
derivation beside it, written from the reference's formulas in exact arithmetic -- none of them is an output of the
oracle.  They anchor the restatement; they do not pin it to a run of the reference (it ships no fixtures).

Shipped parameters: smoothing radius sr = 0.01, kernel width h = sr / 2 = 0.005, spacing 0.005,
mass = spacing^3 * 1000 = 1.25e-4, pressureK = 1119.
  kernel_cubic = 1 / 3.141593 / h^3        = 1 / (3.141593 * 1.25e-7)  = 2546478.8087
  kspikydiff   = -45 / (3.141592 * sr^6)   = -45 / 3.141592e-12        = -1.4323947858e13
"""
import ctypes as C
import math
from fractions import Fraction

import numpy as np
import pytest

from fluidsolver_amd import scenes
from util import oracle_pair

F32 = np.float32
SR, H, M = 0.01, 0.005, 1.25e-4
KC = 1 / 3.141593 / H ** 3
KS = -45 / (3.141592 * SR ** 6)


def test_the_two_constants():
    assert abs(KC - 2546478.8087) < 1e-3
    assert abs(KS - -1.4323947858e13) < 1e4


def pair_scene(offset, type_j=0, vel_i=(0, 0, 0), vel_j=(0, 0, 0), mass_j=M, base=0.005, **params):
    """particle 0 (fluid) in the cell [0, 0.01)^3 (at its centre unless `base` says otherwise), particle 1 at
    `offset` from it"""
    base = np.array([base, base, base], np.float64)
    pos = np.array([base, base + np.asarray(offset, np.float64)], F32)
    n = 2
    sc = {"pos": pos, "vel": np.array([vel_i, vel_j], F32), "normal": np.zeros((n, 3), F32),
          "color": np.ones((n, 4), F32), "type": np.array([0, type_j], np.int32),
          "group": np.zeros(n, np.int32), "mass": np.array([M, mass_j], F32),
          "restDensity": np.full(n, 1000.0, F32), "uniqueId": np.arange(n, dtype=np.int32),
          "params": scenes._params((-0.05, -0.05, -0.05), (0.05, 0.05, 0.05), viscosity=0.01, **params)}
    return sc


def sorted_index(st, uid):
    return int(np.where(st["uniqueId"] == uid)[0][0])


# ---- Kernel_Cubic through the density sum (:32-40, :222-240) ---------------------------------------------------
# rho_0 = m * (W(0) + W(r)), W(r) = fq(q) * kernel_cubic, q = r / h
#   q = 0   : fq = 0.666667 * 1.5                         = 1.0000005
#   q = 0.5 : fq = (0.666667 - 0.25 + 0.5 * 0.125) * 1.5  = 0.7187505
#   q = 1   : fq = (2 - 1)^3 * 0.25                       = 0.25       (the q >= 1 branch)
#   q = 1.5 : fq = 0.5^3 * 0.25                           = 0.03125
#   q = 2   : outside (d2 >= sr2 is tested first, and q >= 2 returns 0)
FQ = {0.0: 1.0000005, 0.5: 0.7187505, 1.0: 0.25, 1.5: 0.03125, 2.0: 0.0}
# m * kernel_cubic = 1.25e-4 * 2546478.8087 = 318.3098511;   * fq:
W_TIMES_M = {0.0: 318.3100102, 0.5: 228.7853646, 1.0: 79.5774628, 1.5: 9.9471828, 2.0: 0.0}


@pytest.mark.parametrize("q", [0.5, 1.0, 1.5, 2.0])
def test_cubic_kernel_values(orc32, q):
    for k, fq in FQ.items():
        assert abs(W_TIMES_M[k] - M * KC * fq) < 2e-6, k
    sc = pair_scene((q * H, 0, 0))
    p, st = oracle_pair(orc32, sc)
    assert abs(p.kernel_cubic - KC) <= 1e-6 * KC
    st.sort_particles(p)
    st.compute_pressure(p)
    i = sorted_index(st, 0)
    want = W_TIMES_M[0.0] + W_TIMES_M[q]
    assert abs(float(st["density"][i]) - want) <= 1e-5 * want
    # a lone particle and its own term: W(0) * m
    one = pair_scene((0.03, 0.03, 0.03))      # the other particle three cells away
    p1, s1 = oracle_pair(orc32, one)
    s1.sort_particles(p1)
    s1.compute_pressure(p1)
    assert abs(float(s1["density"][sorted_index(s1, 0)]) - W_TIMES_M[0.0]) <= 2e-6 * W_TIMES_M[0.0]
    st.close()
    s1.close()


def test_kernel_is_continuous_at_q_1_and_vanishes_just_inside_2(orc32):
    for q, lo, hi in ((0.99999, 0.2499, 0.2502), (1.00001, 0.2499, 0.2501), (1.9999, 0.0, 1e-9)):
        sc = pair_scene((q * H, 0, 0))
        p, st = oracle_pair(orc32, sc)
        st.sort_particles(p)
        st.compute_pressure(p)
        w = float(st["density"][sorted_index(st, 0)]) - W_TIMES_M[0.0]
        assert lo * M * KC <= w + 1e-4 <= hi * M * KC + 2e-4, (q, w)
        st.close()


# ---- the pressure equation and its clamp (:263-265) ------------------------------------------------------------
#   rho < rho0 : K * ((rho / rho0)^7 - 1) < 0 -> 0.       A lone particle has rho = 318.31 -> p = 0.
#   rho = 1100 : 1119 * (1.1^7 - 1) = 1119 * 0.9487171 = 1061.6144
#   rho = 1001 : 1119 * (1.001^7 - 1) = 1119 * 0.007021035 = 7.856538
@pytest.mark.parametrize("rho,want", [(318.3100102, 0.0), (999.0, 0.0), (1100.0, 1061.6144), (1001.0, 7.856538)])
def test_pressure_equation_and_clamp(orc32, rho, want):
    sc = pair_scene((0.03, 0.03, 0.03))
    sc["mass"] = np.array([rho / (KC * FQ[0.0]), M], F32)   # the lone particle's own term gives exactly rho
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.compute_pressure(p)
    i = sorted_index(st, 0)
    got_rho = float(st["density"][i])
    assert abs(got_rho - rho) <= 2e-6 * rho
    got = float(st["pressure"][i])
    if want == 0.0:
        assert got == 0.0
    else:
        # the power is taken of the density the oracle found, whose last-bit error is amplified 7 K / p times
        exact = 1119.0 * ((got_rho / 1000.0) ** 7 - 1)
        assert abs(got - exact) <= 1e-6 * 1119.0 * 2
        assert abs(got - want) <= 1e-5 * want + 7 * 1119.0 * 2e-6
    st.close()


# ---- fluid-fluid pair force (:285-299) -------------------------------------------------------------------------
# i at the origin, j at (+0.006, 0, 0): x_ij = (-0.006, 0, 0), d = 0.006, c = sr - d = 0.004
#   nablaw = kspikydiff * c^2 / d = -1.4323947858e13 * 1.6e-5 / 0.006 = -3.8197194289e10
# pressure: rho = 1000 both, p_i = 100, p_j = 300:  p_j/rho_j^2 + p_i/rho_i^2 = 4e-4
#   pc = nablaw * m_j * 4e-4 = -3.8197194289e10 * 1.25e-4 * 4e-4 = -1909.859714
#   force_x = -x_ij.x * pc = -(-0.006) * (-1909.859714) = -11.45915829      (j pushes i away from itself)
# viscosity: v_i = (0, 1, 0), v_j = 0, viscosity 0.01:
#   vc = nablaw * d^2 / (d^2 + 0.01 sr^2) * m_j / rho_j * 2 * visc
#      = -3.8197194289e10 * (3.6e-5 / 3.7e-5) * 1.25e-7 * 0.02 = -92.91209422
#   force_y = 1 * vc - 9.8 = -102.71209422                                   (:372 adds (0, -9.8, 0))
def test_two_particle_pressure_and_viscosity_force(orc32):
    assert abs(KS * 1.6e-5 / 0.006 - -3.8197194289e10) < 1e2
    sc = pair_scene((0.006, 0, 0), vel_i=(0, 1, 0))
    p, st = oracle_pair(orc32, sc)
    assert abs(p.kspikydiff - KS) <= 1e-6 * abs(KS)
    st.sort_particles(p)
    i, j = sorted_index(st, 0), sorted_index(st, 1)
    st["density"][:] = 1000.0
    st["pressure"][i], st["pressure"][j] = 100.0, 300.0
    st.compute_force(p)
    f = st["force"][i].astype(np.float64)
    # the float32 coordinates 0.005 and 0.011 are 0.006 apart to 1e-7 relative, and sr - d amplifies that 2.5 times
    assert abs(f[0] - -11.45915829) <= 2e-5 * 11.46
    assert abs(f[1] - -102.71209422) <= 2e-5 * 102.7
    assert f[2] == 0.0
    # the partner: x_ji = +0.006, same pc -> +11.459; v_j - v_i = (0, -1, 0) -> +92.912 - 9.8
    g = st["force"][j].astype(np.float64)
    assert abs(g[0] - 11.45915829) <= 2e-5 * 11.46
    assert abs(g[1] - (92.91209422 - 9.8)) <= 2e-5 * 92.9
    st.close()


def test_pair_beyond_the_support_and_the_particle_itself_add_nothing(orc32):
    sc = pair_scene((0.0100001, 0, 0), vel_i=(0, 1, 0))
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st["density"][:] = 1000.0
    st["pressure"][:] = 500.0
    st.compute_force(p)
    assert st["force"][sorted_index(st, 0)].tolist() == [0.0, float(F32(-9.8)), 0.0]
    st.close()


# ---- wall penalty (:315-339) -----------------------------------------------------------------------------------
# j is TYPE_BOUNDARY with the fluid mass (m_j / (m_i + m_j) = 0.5), at distance d along +x; q = d / sr:
#   B(q) = 0.66666 (q < 0.66666);  2q - 1.5 q^2 (q < 1);  0.5 (2 - q)^2 (q < 2);  0
#   B *= 0.02 * 88.5^2 / d = 156.645 / d
#   force_x = x_ij.x * 0.5 * B = -d * 0.5 * B(q) * 156.645 / d = -78.3225 * B(q)
#   q = 0.5 : -78.3225 * 0.66666  = -52.2144779
#   q = 0.8 : -78.3225 * 0.64     = -50.1264
#   q = 1.5 : -78.3225 * 0.125    = -9.7903125      (no range test: a wall particle counts beyond the support)
#   q = 2.2 : 0                                     (along the cell diagonal: still one of the 27 cells)
# the fluid particle sits at 0.002 in its cell so that the wall particle is in the cell next to it for q = 1.5
@pytest.mark.parametrize("q,want", [(0.5, -52.2144779), (0.8, -50.1264), (1.5, -9.7903125), (2.2, 0.0)])
def test_wall_penalty_in_each_branch(orc32, q, want):
    assert abs(0.02 * 88.5 * 88.5 - 156.645) < 1e-9
    off = (q * SR, 0, 0) if q < 2 else (q * SR / math.sqrt(3),) * 3
    vel = (-1, 0, 0) if q < 2 else (-1, -1, -1)                   # receding: x.v > 0, no wall viscosity
    sc = pair_scene(off, type_j=2, vel_i=vel, base=0.002)
    p, st = oracle_pair(orc32, sc, boundary_visc=5.0)
    st.sort_particles(p)
    st["density"][:] = 1000.0
    st.compute_force(p)
    f = st["force"][sorted_index(st, 0)].astype(np.float64)
    assert abs(f[0] - want) <= 2e-5 * abs(want) + 1e-12
    assert f[1] == float(F32(-9.8)) and f[2] == 0.0
    st.close()


def test_wall_penalty_weighs_by_the_mass_ratio(orc32):
    # m_j = 3 m_i: m_j / (m_i + m_j) = 0.75 -> -156.645 * 0.66666 * 0.75 = -78.32172
    sc = pair_scene((0.5 * SR, 0, 0), type_j=2, mass_j=3 * M)
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st["density"][:] = 1000.0
    st.compute_force(p)
    f = st["force"][sorted_index(st, 0)].astype(np.float64)
    assert abs(f[0] - -78.32172) <= 2e-5 * 78.3
    st.close()


# ---- wall viscosity on approach (:342-350) ---------------------------------------------------------------------
# j (wall, at rest) at (+0.006, 0, 0), i moves towards it with v_i = (2, 0, 0), boundary_visc = 5:
#   x_ij = (-0.006, 0, 0), xv = v_ij . x_ij = -0.012 < 0
#   visc = 2 * 5 * 0.01 * 88.5 / 2000 = 0.004425
#   pi = -visc * xv / (d^2 + 0.01 sr^2) = 0.004425 * 0.012 / 3.7e-5 = 1.435135135
#   nablaw = -3.8197194289e10 (as above)
#   force_x += -x_ij.x * pi * nablaw * m_j = 0.006 * 1.435135135 * (-3.8197194289e10) * 1.25e-4 = -41113.6017
#   penalty (q = 0.6 < 0.66666): -78.3225 * 0.66666 = -52.2144779;  total force_x = -41165.8162
def test_wall_viscosity_on_approach_only(orc32):
    want_visc = 0.006 * (0.004425 * 0.012 / 3.7e-5) * (KS * 1.6e-5 / 0.006) * M
    assert abs(want_visc - -41113.6017) < 0.05
    for v, want in (((2, 0, 0), -41113.6017 - 52.2144779), ((-2, 0, 0), -52.2144779)):
        sc = pair_scene((0.006, 0, 0), type_j=2, vel_i=v)
        p, st = oracle_pair(orc32, sc, boundary_visc=5.0)
        st.sort_particles(p)
        st["density"][:] = 1000.0
        st.compute_force(p)
        f = st["force"][sorted_index(st, 0)].astype(np.float64)
        assert abs(f[0] - want) <= 2e-5 * abs(want), v
        st.close()


# ---- integration (:377-391) ------------------------------------------------------------------------------------
# force (0, -9.8, 0), dt 4e-4, from rest: v = (0, -0.00392, 0); y moves by -1.568e-6; with acceleration_limit = 0
# the drift acceleration is cut to nothing and v_star = gravity = (0, -10, 0)
def test_symplectic_euler_and_v_star(orc32):
    sc = pair_scene((0.03, 0.03, 0.03))
    p, st = oracle_pair(orc32, sc)
    st.sort_particles(p)
    st.compute_pressure(p)
    st.compute_force(p)
    i = sorted_index(st, 0)
    y0 = float(st["pos"][i][1])
    st.advect(p)
    assert abs(float(st["vel"][i][1]) - -0.00392) <= 1e-6 * 0.00392
    assert abs((float(st["pos"][i][1]) - y0) - -1.568e-6) <= 1e-9   # one rounding of a coordinate near 0.005
    assert st["v_star"][i].tolist() == [0.0, -10.0, 0.0]
    st.close()


# ---- the seventh power -----------------------------------------------------------------------------------------
def correctly_rounded_pow7(x):
    """x^7 of a float32 rounded to nearest-even float32, in exact rational arithmetic"""
    f = Fraction(float(x)) ** 7
    e = math.floor(math.log2(f))
    if Fraction(2) ** e > f:
        e -= 1
    elif Fraction(2) ** (e + 1) <= f:
        e += 1
    scaled = f / Fraction(2) ** (e - 23)          # in [2^23, 2^24)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return float(Fraction(n) * Fraction(2) ** (e - 23))


def ulps(a, b):
    ia = np.ascontiguousarray(a, F32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, F32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_seventh_power_is_within_one_unit_of_libm_powf_and_equal_where_powf_is_right(orc32):
    """the oracle (and the HIP kernels) take (rho / rho0)^7 as a product in double rounded once instead of calling
    powf (:263): over [0.5, 1.5] -- every density the pressure equation sees in these scenes -- it must be within one
    unit in the last place of this host's powf, equal to it wherever powf returns the correctly rounded value, and
    itself the correctly rounded value"""
    probe = orc32.lib.orc_pow7_probe
    probe.restype = None
    probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    # dense: every 4th float32 of [0.5, 1) and of [1, 1.5], and every float32 within 1e-3 of 1
    a = np.arange(np.float32(0.5).view(np.int32), np.float32(1.0).view(np.int32), 4, dtype=np.int32).view(F32)
    b = np.arange(np.float32(1.0).view(np.int32), np.float32(1.5).view(np.int32) + 1, 4, dtype=np.int32).view(F32)
    c = np.arange(np.float32(0.999).view(np.int32), np.float32(1.001).view(np.int32), 1, dtype=np.int32).view(F32)
    x = np.ascontiguousarray(np.concatenate([a, b, c]))
    prod = np.empty_like(x)
    libm = np.empty_like(x)
    probe(x.ctypes.data, len(x), prod.ctypes.data, libm.ctypes.data)
    d = ulps(prod, libm)
    assert len(x) > 3_000_000
    assert int(d.max()) <= 1
    assert float((d != 0).mean()) < 1e-3           # glibc's powf is nearly always the correctly rounded value
    # exact arithmetic on a sample, and on every argument where the two differ
    rng = np.random.Generator(np.random.PCG64(7))
    pick = np.unique(np.concatenate([rng.integers(0, len(x), 4000), np.flatnonzero(d != 0)[:2000]]))
    exact = np.array([correctly_rounded_pow7(v) for v in x[pick]], F32)
    assert np.array_equal(prod[pick], exact)       # the product rounded once IS the correctly rounded power here
    right = libm[pick] == exact
    assert np.array_equal(prod[pick][right], libm[pick][right])
    assert prod[np.flatnonzero(x == F32(1.0))[0]] == F32(1.0)
