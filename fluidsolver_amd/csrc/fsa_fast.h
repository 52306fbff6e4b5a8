// Fast fp32 flavour of the WCSPH pair arithmetic (kernel variant 2), shared by the tiled kernels, the replay
// kernels and their per-particle fallback walks, so that a particle's result never depends on the path taken.
#pragma once

#include "fsa_pair.h"

namespace fsa {

// ---- fast fp32 pair arithmetic -----------------------------------------------------------------
// One fluid-fluid pair of the force pass in the fast flavour.  `ok` = the pair counts (inside the
// support, not the particle itself, a fluid neighbour); a pair that does not count adds exact zeros.
__device__ __forceinline__ void force_pair_fast(ForceSelf<float> &S, float sr, float kspiky, float visc_eps,
                                                float xx, float xy, float xz, float d2, bool ok, const float4 &vj,
                                                const float2 &aj) {
    const float rinv = __builtin_amdgcn_rsqf(d2);
    const float c = sr - d2 * rinv;
    const float nablaw = kspiky * c * c * rinv;
    const float pc = ok ? nablaw * vj.w * (aj.x + S.pterm) : 0.0f;
    const float vc = ok ? nablaw * d2 * __builtin_amdgcn_rcpf(d2 + visc_eps) * aj.y : 0.0f;
    S.fx = __builtin_fmaf(-xx, pc, S.fx);
    S.fy = __builtin_fmaf(-xy, pc, S.fy);
    S.fz = __builtin_fmaf(-xz, pc, S.fz);
    S.fx = __builtin_fmaf(S.v.x - vj.x, vc, S.fx);
    S.fy = __builtin_fmaf(S.v.y - vj.y, vc, S.fy);
    S.fz = __builtin_fmaf(S.v.z - vj.z, vc, S.fz);
}

// One TYPE_BOUNDARY neighbour of a fluid particle in the fast flavour (the penalty force and the wall viscosity of
// ForceCell, solver/sph_kernel_impl.cuh:315-352, which have no range test: every wall particle of the 27 cells
// counts): fp32 throughout, reciprocal square root and reciprocal instead of the divisions.  Fluid in contact with
// a wall has dozens of such neighbours per particle, so their cost decides the force pass of a developed flow.
struct WallConsts {
    float inv_sr, kb, visc;
    __device__ __forceinline__ explicit WallConsts(const DevParams<float> &P)
        : inv_sr(1.0f / P.sr), kb((float)(0.02 * 88.5 * 88.5)), visc((float)((double)(2 * P.boundary_visc * P.sr) * 88.5 / 2000)) {}
};
__device__ __forceinline__ void wall_pair_fast(ForceSelf<float> &Sb, const WallConsts &W, float sr, float kspiky,
                                               float visc_eps, float xx, float xy, float xz, float d2, bool ok,
                                               float mass_j, float vjx, float vjy, float vjz) {
    const float rinv = __builtin_amdgcn_rsqf(d2);
    const float d = d2 * rinv;
    const float q = d * W.inv_sr;
    const float t2 = 2.0f - q;
    float B = q < 0.66666f ? 0.66666f : (q < 1.0f ? q * __builtin_fmaf(-1.5f, q, 2.0f) : 0.5f * t2 * t2);
    B = q < 2.0f ? B * W.kb * rinv : 0.0f;
    const float mag = ok ? mass_j * __builtin_amdgcn_rcpf(Sb.v.w + mass_j) * B : 0.0f;
    const float vx = Sb.v.x - vjx, vy = Sb.v.y - vjy, vz = Sb.v.z - vjz;
    const float xv = __builtin_fmaf(vz, xz, __builtin_fmaf(vy, xy, vx * xx));
    const float c = sr - d;
    const float nablaw = kspiky * c * c * rinv;
    const float pi_ = -W.visc * xv * __builtin_amdgcn_rcpf(d2 + visc_eps);
    const float tv = (ok && xv < 0.0f) ? pi_ * nablaw * mass_j : 0.0f;
    const float k = mag - tv;
    Sb.fx = __builtin_fmaf(xx, k, Sb.fx);
    Sb.fy = __builtin_fmaf(xy, k, Sb.fy);
    Sb.fz = __builtin_fmaf(xz, k, Sb.fz);
}

// Per-particle walks in the fast flavour, for blocks that cannot use the row tiles.  They evaluate the
// same expressions in the same order as the tiled kernels, so a particle's result does not depend on
// which path its block took (nor, in a slab run, on how the particles are split over solvers).
__device__ __forceinline__ float density_walk_global_fast(const DevParams<float> &P, const float4 &pi, float mass_i,
                                                          const float4 *__restrict__ posType,
                                                          const int *__restrict__ cellStart) {
    const float sr = P.sr, sr2 = sr * sr;
    const float h = (float)((double)sr * 0.5);
    const float inv_h = 1.0f / h, kcubic = P.kernel_cubic;
    const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
    float dens = 0.f;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) {
                const float4 pj = posType[j];
                if (unpack_int(pj.w) != FSA_TYPE_FLUID) continue;
                const float xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
                const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                if (d2 < sr2) dens = __builtin_fmaf(mass_i, cubic_fast(d2, inv_h, kcubic), dens);
            }
        }
    return dens;
}

// S collects the fluid neighbours, Sb the TYPE_BOUNDARY ones (summed separately).
__device__ __forceinline__ void force_walk_global_fast(const DevParams<float> &P, int index, ForceSelf<float> &S,
                                                       ForceSelf<float> &Sb, const float4 *__restrict__ posType,
                                                       const float4 *__restrict__ velMass,
                                                       const float2 *__restrict__ pd,
                                                       const int *__restrict__ cellStart) {
    const float sr = P.sr, sr2 = sr * sr;
    const float kspiky = P.kspikydiff, visc_eps = 0.01f * sr2;
    const WallConsts W(P);
    const RowWalk<float> walk(P, S.p.x, S.p.y, S.p.z);
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) {
                const float4 pj = posType[j];
                const int tj = unpack_int(pj.w);
                const float xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
                if (tj == FSA_TYPE_FLUID) {
                    const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                    if (d2 < sr2 && j != index)
                        force_pair_fast(S, sr, kspiky, visc_eps, xx, xy, xz, d2, true, velMass[j], pd[j]);
                } else if (tj == FSA_TYPE_BOUNDARY) {
                    const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                    const float4 vj = velMass[j];
                    wall_pair_fast(Sb, W, sr, kspiky, visc_eps, xx, xy, xz, d2, true, vj.w, vj.x, vj.y, vj.z);
                }
            }
        }
}

}  // namespace fsa
