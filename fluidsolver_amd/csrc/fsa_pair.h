// Pair interactions of the WCSPH passes, shared by every kernel variant so that all of them
// perform the same operations in the same order.  Restates the loop bodies of DensityCell and
// ForceCell (solver/sph_kernel_impl.cuh:222-241, 278-353 under /root/reference/particle_solver).
#pragma once

#include "fsa_device.h"

namespace fsa {

// One neighbour j of particle i in the density sum (DensityCell :225-239).  `restDensity` is only
// read for a TYPE_RIGID neighbour.
template <typename T>
__device__ __forceinline__ void density_pair(const DevParams<T> &P, T h, T sr2, T mass_i, T rd_i,
                                             const Vec4<T> &pi, const Vec4<T> &pj,
                                             const T *__restrict__ restDensity, int j, T &dens) {
    const T xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
    const T d2 = xx * xx + xy * xy + xz * xz;
    if (d2 >= sr2) return;
    const T wij = kernel_cubic(P, h, xx, xy, xz);
    const int tj = unpack_int(pj.w);
    if (tj == FSA_TYPE_FLUID) dens += mass_i * wij;
    if (tj == FSA_TYPE_RIGID) dens += rd_i * restDensity[j] * wij;
}

// Cubic spline of Kernel_Cubic (solver/sph_kernel_impl.cuh:32-40) in the fast fp32 flavour: hardware square root,
// fused multiply-adds, both branches evaluated and one selected.
__device__ __forceinline__ float cubic_fast(float d2, float inv_h, float kcubic) {
    const float q = __builtin_amdgcn_sqrtf(d2) * inv_h;
    // q < 1: (0.666667 - q^2 + 0.5 q^3) * 1.5     1 <= q < 2: 0.25 (2 - q)^3
    const float q2 = q * q;
    const float w1 = __builtin_fmaf(q2, __builtin_fmaf(0.75f, q, -1.5f), 1.0000005f);
    const float t = 2.0f - q;
    const float w2 = 0.25f * t * t * t;
    return (q < 1.0f ? w1 : w2) * kcubic;
}

// Per-particle state of the force pass.
template <typename T> struct ForceSelf {
    Vec4<T> p;  // position, type
    Vec4<T> v;  // velocity, mass
    T pterm;    // pressure / density / density of particle i
    T fx, fy, fz;
};

// Does neighbour j (type tj, squared distance d2) contribute anything?  Fluid neighbours only
// inside the support; boundary neighbours always (ForceCell has no range test on that branch).
template <typename T> __device__ __forceinline__ bool force_pair_active(int tj, bool self, T d2, T sr2) {
    return (tj == FSA_TYPE_FLUID && !self && d2 < sr2) || tj == FSA_TYPE_BOUNDARY;
}

// One neighbour j in the force sum (ForceCell :280-352).  pterm_j = pressure_j / density_j /
// density_j, evaluated in that order.
template <typename T>
__device__ __forceinline__ void force_pair(const DevParams<T> &P, T sr, T sr2, ForceSelf<T> &S, bool self,
                                           const Vec4<T> &pj, const Vec4<T> &vj, T pterm_j, T dens_j) {
    const T xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
    const T d2 = xx * xx + xy * xy + xz * xz;
    const T d = r_sqrt(d2);
    const int tj = unpack_int(pj.w);
    const T mass_j = vj.w;
    if (!self && tj == FSA_TYPE_FLUID) {
        if (d2 >= sr2) return;
        const T c = sr - d;
        const T nablaw = P.kspikydiff * c * c / d;
        const T pc = nablaw * mass_j * (pterm_j + S.pterm);
        S.fx += xx * pc * (T)(-1);
        S.fy += xy * pc * (T)(-1);
        S.fz += xz * pc * (T)(-1);
        // `0.01 * sr2` is a double: the quotient and everything after it are evaluated in double
        const T vc = (T)((double)(nablaw * d2) / ((double)d2 + 0.01 * (double)sr2) * (double)mass_j /
                         (double)dens_j * 2 * (double)P.viscosity);
        S.fx += (S.v.x - vj.x) * vc;
        S.fy += (S.v.y - vj.y) * vc;
        S.fz += (S.v.z - vj.z) * vc;
    }
    if (tj == FSA_TYPE_BOUNDARY) {
        T B = 1;
        const T q = d / sr;
        if ((double)q < 0.66666)
            B = (T)((double)B * 0.66666);
        else if (q < 1)
            B = (T)((double)B * ((double)(2 * q) - 1.5 * (double)q * (double)q));
        else if (q < 2)
            B = (T)((double)B * (0.5 * (double)(2 - q) * (double)(2 - q)));
        else
            B = 0;
        B = (T)((double)B * (0.02 * 88.5 * 88.5 / (double)d));
        const T magnitude = mass_j / (S.v.w + mass_j) * B;
        S.fx += xx * magnitude;
        S.fy += xy * magnitude;
        S.fz += xz * magnitude;
        const T vx = S.v.x - vj.x, vy = S.v.y - vj.y, vz = S.v.z - vj.z;
        const T xv = vx * xx + vy * xy + vz * xz;
        if (xv < 0) {
            const T c = sr - d;
            const T nablaw = P.kspikydiff * c * c / d;
            const T visc = (T)((double)(2 * P.boundary_visc * P.sr) * 88.5 / 2000);
            const T pi_ = (T)((double)(-visc * xv) / ((double)d2 + 0.01 * (double)sr2));
            S.fx += xx * pi_ * nablaw * mass_j * (T)(-1);
            S.fy += xy * pi_ * nablaw * mass_j * (T)(-1);
            S.fz += xz * pi_ * nablaw * mass_j * (T)(-1);
        }
    }
}

// The 3 x 3 rows of a particle's 27-cell neighbourhood as index ranges of the sorted arrays.
// The hash is x-fastest, so cells x-1, x, x+1 of one (y, z) row are adjacent: walking the rows in
// the order z, y and each range in ascending index visits the neighbours in the reference's order
// (z, y, x cell loops, ascending j inside a cell; solver/sph_kernel_impl.cuh:255-260, 222).
template <typename T> struct RowWalk {
    int cx, cy, cz, xlo, xhi;
    bool any;
    __device__ __forceinline__ RowWalk(const DevParams<T> &P, T x, T y, T z) {
        calc_grid_pos(P, x, y, z, cx, cy, cz);
        xlo = max(cx - 1, 0);
        xhi = min(cx + 1, P.gres[0] - 1);
        any = xlo <= xhi;
    }
    // first cell of the range in row (cy + dy, cz + dz), or -1 when that row is outside the held grid
    __device__ __forceinline__ int row_cell(const DevParams<T> &P, int dy, int dz) const {
        const int y = cy + dy, z = cz + dz, lz = z - P.z0;
        if (!any || y < 0 || y >= P.gres[1] || z < 0 || z >= P.gres[2] || lz < 0 || lz >= P.nz) return -1;
        return (y * P.nz + lz) * P.gres[0] + xlo;
    }
    __device__ __forceinline__ bool row(const DevParams<T> &P, const int *__restrict__ cellStart, int dy, int dz,
                                        int &start, int &end) const {
        const int c = row_cell(P, dy, dz);
        if (c < 0) return false;
        start = cellStart[c];
        end = cellStart[c + (xhi - xlo) + 1];
        return true;
    }
};

// Whole density walk of one particle from global memory (generic path).
template <typename T>
__device__ __forceinline__ T density_walk_global(const DevParams<T> &P, int index, const Vec4<T> &pi, T mass_i,
                                                 T rd_i, const Vec4<T> *__restrict__ posType,
                                                 const T *__restrict__ restDensity,
                                                 const int *__restrict__ cellStart) {
    const T sr = P.sr, sr2 = sr * sr;
    const T h = (T)((double)sr * 0.5);
    const RowWalk<T> walk(P, pi.x, pi.y, pi.z);
    T dens = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) density_pair(P, h, sr2, mass_i, rd_i, pi, posType[j], restDensity, j, dens);
        }
    return dens;
}

// Whole force walk of one particle from global memory (generic path).
template <typename T>
__device__ __forceinline__ void force_walk_global(const DevParams<T> &P, int index, ForceSelf<T> &S,
                                                  const Vec4<T> *__restrict__ posType,
                                                  const Vec4<T> *__restrict__ velMass,
                                                  const T *__restrict__ density, const T *__restrict__ pressure,
                                                  const int *__restrict__ cellStart) {
    const T sr = P.sr, sr2 = sr * sr;
    const RowWalk<T> walk(P, S.p.x, S.p.y, S.p.z);
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) {
                const Vec4<T> pj = posType[j];
                const int tj = unpack_int(pj.w);
                const T xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
                const T d2 = xx * xx + xy * xy + xz * xz;
                if (!force_pair_active(tj, j == index, d2, sr2)) continue;
                const T dj = density[j];
                force_pair(P, sr, sr2, S, j == index, pj, velMass[j], pressure[j] / dj / dj, dj);
            }
        }
}

}  // namespace fsa
