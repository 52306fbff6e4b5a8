// Single-phase DFSPH step for gfx950: one thread per particle, neighbours read from the sorted arrays
// through the 9-row walk (same visiting order as the reference's 27-cell loops).
//
// Replaces computeDensityAlpha_kernel, computeNPF_kernel, solveDensityStiff, solveDivergenceStiff,
// applyPStiff, updatePosition, UpdateVelocities (solver/sph_kernel_impl.cuh:582-1018 under
// /root/reference/particle_solver) and the host-side maximum of error[] that the reference's
// correctDensityError / correctDivergenceError take after copying the whole array to the host
// (solver/sph_solver.cu:175-179, 232-235): here a two-level device reduction returns one number.
// Arithmetic follows the reference text operation for operation (explicit casts where its double
// literals promote a float expression).
#include "fsa_fast.h"
#include <cstring>

#include "fsa_pass.h"

namespace fsa {

// ---- DensityAlphaCell :582-624 + computeDensityAlpha_kernel :626-660 -------------------------------
template <typename T> struct AlphaPolicy {
    static constexpr bool kDeepReplay = true;
    static constexpr bool kHitReplay = sizeof(T) == 4;
    // which neighbour types count: 1 = inside the support, 2 = anywhere in the 27 cells, 0 = never
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T dens, mwij, mx, my, mz, mass; };
    struct Nb { T mass; };
    static __device__ __forceinline__ bool wants(int) { return true; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.dens = S.mwij = S.mx = S.my = S.mz = 0;
        S.mass = A.velMass[index].w;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        return Nb{reinterpret_cast<const T *>(A.velMass)[4 * (size_t)j + 3]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool self, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (self) return;
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        const T d = r_sqrt(d2);
        const T c2 = sr2 - d2;
        const T c = sr - d;
        const T nablaw = P.kspikydiff * c * c / d;
        if (tj == FSA_TYPE_FLUID) {
            S.dens += c2 * c2 * c2 * nb.mass;
            const T ax = xx * nablaw * nb.mass, ay = xy * nablaw * nb.mass, az = xz * nablaw * nb.mass;
            S.mwij += ax * ax + ay * ay + az * az;
            S.mx += ax;
            S.my += ay;
            S.mz += az;
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        const T sr2 = P.sr * P.sr;
        T dens = S.dens + sr2 * sr2 * sr2 * S.mass;
        dens *= P.kpoly6;
        A.densityOut[index] = dens;
        T denom = S.mx * S.mx + S.my * S.my + S.mz * S.mz + S.mwij;
        if ((double)denom < 0.000001) denom = (T)0.000001;  // clamp for stability, :656-657
        A.dfFactorOut[index] = dens / denom;
    }
};

// ---- computeNPFCell :663-742 + computeNPF_kernel :744-773 ------------------------------------------
template <typename T> struct NpfPolicy {
    static __device__ __forceinline__ int reach(int tj) {
        return tj == FSA_TYPE_FLUID ? 1 : (tj == FSA_TYPE_BOUNDARY ? 2 : 0);
    }
    struct Self { ForceSelf<T> F; };
    struct Nb { Vec4<T> v; T dens; };
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &, int index) {
        A.vStarOut[index] = Vec3<T>{0, 0, 0};
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.F.p = A.posType[index];
        S.F.v = A.velMass[index];
        S.F.pterm = 0;
        S.F.fx = S.F.fy = S.F.fz = 0;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.velMass[j], A.density[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool self, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &pj, const Nb &nb) {
        const T sr = P.sr, sr2 = sr * sr;
        if (!self && tj == FSA_TYPE_FLUID) {
            if (d2 >= sr2) return;
            const T d = r_sqrt(d2);
            const T c = sr - d;
            const T nablaw = P.kspikydiff * c * c / d;
            // :703-704, `0.01 * sr2` is a double
            const T vc = (T)((double)(nablaw * d2) / ((double)d2 + 0.01 * (double)sr2) * (double)nb.v.w /
                             (double)nb.dens * 2 * (double)P.viscosity);
            S.F.fx += (S.F.v.x - nb.v.x) * vc;
            S.F.fy += (S.F.v.y - nb.v.y) * vc;
            S.F.fz += (S.F.v.z - nb.v.z) * vc;
        }
        // the wall branch is ForceCell's, text for text; force_pair does nothing else for this type
        if (tj == FSA_TYPE_BOUNDARY) force_pair(P, sr, sr2, S.F, self, pj, nb.v, (T)0, (T)1);
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        S.F.fx += (T)0;
        S.F.fy += (T)-9.8;  // :770
        S.F.fz += (T)0;
        Vec3<T> vs;
        vs.x = S.F.v.x + S.F.fx * P.dt;
        vs.y = S.F.v.y + S.F.fy * P.dt;
        vs.z = S.F.v.z + S.F.fz * P.dt;
        A.vStarOut[index] = vs;
    }
};

// FAST flavour of the same pass (fp32, variant >= 2): hardware reciprocal square root and reciprocal, fused
// multiply-adds, no double arithmetic per pair, the wall terms as in the WCSPH fast flavour (summed apart), and one
// packed record per neighbour: {velocity, mass / density * 2 * viscosity} of a fluid particle, {velocity, mass} of a
// wall particle.  About 1e-6 relative away from the exact flavour per term.
__global__ __launch_bounds__(kBlock) void pack_npf_kernel(int n, float two_visc, const float4 *__restrict__ posType,
                                                          const float4 *__restrict__ velMass,
                                                          const float *__restrict__ density,
                                                          float4 *__restrict__ rec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = velMass[i];
    if (unpack_int(posType[i].w) == FSA_TYPE_FLUID) v.w = v.w / density[i] * two_visc;
    rec[i] = v;
}

struct NpfFastPolicy {
    using T = float;
    static constexpr bool kHitReplay = true;
    static constexpr bool kWallWords = true;
    static __device__ __forceinline__ int reach(int tj) { return NpfPolicy<float>::reach(tj); }
    struct Self { ForceSelf<float> F, Fb; };
    struct Nb { float4 r; };
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &, int index) {
        A.vStarOut[index] = Vec3<T>{0, 0, 0};
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.F.p = A.posType[index];
        S.F.v = A.velMass[index];
        S.F.pterm = 0;
        S.F.fx = S.F.fy = S.F.fz = 0;
        S.Fb = S.F;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.aux4[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool self, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        const float sr = P.sr, sr2 = sr * sr, visc_eps = 0.01f * sr2;
        if (tj == FSA_TYPE_FLUID) {
            if (self || d2 >= sr2) return;
            const float rinv = __builtin_amdgcn_rsqf(d2);
            const float c = sr - d2 * rinv;
            const float vc = P.kspikydiff * c * c * rinv * d2 * __builtin_amdgcn_rcpf(d2 + visc_eps) * nb.r.w;
            S.F.fx = __builtin_fmaf(S.F.v.x - nb.r.x, vc, S.F.fx);
            S.F.fy = __builtin_fmaf(S.F.v.y - nb.r.y, vc, S.F.fy);
            S.F.fz = __builtin_fmaf(S.F.v.z - nb.r.z, vc, S.F.fz);
        } else if (tj == FSA_TYPE_BOUNDARY) {
            const WallConsts W(P);
            wall_pair_fast(S.Fb, W, sr, P.kspikydiff, visc_eps, xx, xy, xz, d2, true, nb.r.w, nb.r.x, nb.r.y, nb.r.z);
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        Vec3<T> vs;
        vs.x = __builtin_fmaf(S.F.fx + S.Fb.fx, P.dt, S.F.v.x);
        vs.y = __builtin_fmaf(S.F.fy + S.Fb.fy + (-9.8f), P.dt, S.F.v.y);  // :770
        vs.z = __builtin_fmaf(S.F.fz + S.Fb.fz, P.dt, S.F.v.z);
        A.vStarOut[index] = vs;
    }
};

// What applyPStiff reads of a neighbour when every fluid particle has the same mass (ApplyUniformPolicy below):
// {x, y, z, pstiff / density}; a particle that is not fluid lies far outside any support.
constexpr float kFarAway = 1e30f;
__device__ __forceinline__ float4 apply_record(const float4 &p, float w) {
    return unpack_int(p.w) == FSA_TYPE_FLUID ? make_float4(p.x, p.y, p.z, w)
                                             : make_float4(kFarAway, kFarAway, kFarAway, 0.0f);
}

// ---- DensityChangeCell :775-813 / PredictDensityCell :846-887 and the two kernels around them -------
// TIMES_DT = false: solveDivergenceStiff :818-842; true: solveDensityStiff :890-919
// FAST (fp32, variant 2): hardware reciprocal square root and fused multiply-adds, about 1e-7 relative per term
template <typename T, bool TIMES_DT, bool PACKED, bool FAST = false> struct StiffPolicy {
    static constexpr bool kDeepReplay = FAST;
    static constexpr bool kHitReplay = FAST && PACKED;  // (the row tiles of the list kernel bring 1 %: 10.8 -> 10.6 ms per step)
    // which neighbour types count: 1 = inside the support, 2 = anywhere in the 27 cells, 0 = never
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T acc; Vec3<T> vs; };
    struct Nb { T vx, vy, vz, mass; };
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    // what applyPStiff reads of this particle, as pack_pstiff_mass_kernel writes it
    static __device__ __forceinline__ void record(const PassArrays<T> &A, int index, T ps) {
        if constexpr (sizeof(T) == 4) {
            if (A.recOut) A.recOut[index] = apply_record(A.posType[index], ps / A.density[index]);
        }
        if (A.rec2Out) {
            const T mass = reinterpret_cast<const T *>(A.velMass)[4 * (size_t)index + 3];
            const T w = ps / A.density[index];
            if constexpr (sizeof(T) == 4) {
                reinterpret_cast<float2 *>(A.rec2Out)[index] = make_float2(w, mass);
            } else {
                A.rec2Out[2 * (size_t)index] = w;
                A.rec2Out[2 * (size_t)index + 1] = mass;
            }
        }
    }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &, int index) {
        A.errorOut[index] = 0;
        A.pstiffOut[index] = 0;
        record(A, index, (T)0);
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.acc = TIMES_DT ? A.density[index] : (T)0;
        S.vs = A.vStarIn[index];
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        if (PACKED) {
            const Vec4<T> a = A.aux4[j];
            return Nb{a.x, a.y, a.z, a.w};
        }
        const Vec3<T> v = A.vStarIn[j];
        return Nb{v.x, v.y, v.z, reinterpret_cast<const T *>(A.velMass)[4 * (size_t)j + 3]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool self, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (self) return;
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        if (FAST) {
            if (tj != FSA_TYPE_FLUID) return;
            const float rinv = __builtin_amdgcn_rsqf((float)d2);
            const float c = (float)sr - (float)d2 * rinv;
            const float w = (float)P.kspikydiff * c * c * rinv * (float)nb.mass;
            const float dot = __builtin_fmaf((float)(S.vs.x - nb.vx), (float)xx,
                                             __builtin_fmaf((float)(S.vs.y - nb.vy), (float)xy,
                                                            (float)(S.vs.z - nb.vz) * (float)xz)) * w;
            S.acc = (T)__builtin_fmaf(dot, TIMES_DT ? (float)P.dt : 1.0f, (float)S.acc);
            return;
        }
        const T d = r_sqrt(d2);
        const T c = sr - d;
        const T nablaw = P.kspikydiff * c * c / d;
        if (tj == FSA_TYPE_FLUID) {
            const T vx = S.vs.x - nb.vx, vy = S.vs.y - nb.vy, vz = S.vs.z - nb.vz;
            const T nx = xx * nablaw * nb.mass, ny = xy * nablaw * nb.mass, nz = xz * nablaw * nb.mass;
            const T dot = vx * nx + vy * ny + vz * nz;
            S.acc += TIMES_DT ? dot * P.dt : dot;
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        const T rd = A.restDensity[index];
        if (TIMES_DT) {
            T ps = (S.acc - rd) * A.dfFactor[index] / P.dt / P.dt;
            T err = S.acc - rd;
            if (ps < 0) {
                ps = 0;
                err = 0;
            }
            A.pstiffOut[index] = ps;
            A.errorOut[index] = err;
            record(A, index, ps);
        } else {
            T densChange = S.acc;
            if (densChange < 0) densChange = 0;
            A.errorOut[index] = densChange / rd;
            const T ps = densChange / P.dt * A.dfFactor[index];
            A.pstiffOut[index] = ps;
            record(A, index, ps);
        }
    }
};

// ---- applyPressureCell :921-961 + applyPStiff :963-984 ---------------------------------------------
// Neighbours are read through pos / pstiff / density only, v_star is written for the particle itself:
// no ordering hazard between threads.
template <typename T, bool PACKED, bool FAST = false> struct ApplyPolicy {
    static constexpr bool kDeepReplay = FAST;
    static constexpr bool kHitReplay = FAST && PACKED;
    // which neighbour types count: 1 = inside the support, 2 = anywhere in the 27 cells, 0 = never
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T wi, fx, fy, fz; };
    struct Nb { T w, mass; };  // w = pstiff_j / density_j
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.wi = A.pstiffIn[index] / A.density[index];
        S.fx = S.fy = S.fz = 0;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        if constexpr (PACKED) {  // one 8-byte load
            const float2 v = reinterpret_cast<const float2 *>(A.aux2)[j];
            return Nb{(T)v.x, (T)v.y};
        }
        return Nb{A.pstiffIn[j] / A.density[j], reinterpret_cast<const T *>(A.velMass)[4 * (size_t)j + 3]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool self, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (self || tj != FSA_TYPE_FLUID) return;
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        if (FAST) {
            const float rinv = __builtin_amdgcn_rsqf((float)d2);
            const float c = (float)sr - (float)d2 * rinv;
            const float f = (float)P.kspikydiff * c * c * rinv * (float)nb.mass * (float)(S.wi + nb.w);
            S.fx = (T)__builtin_fmaf((float)xx, f, (float)S.fx);
            S.fy = (T)__builtin_fmaf((float)xy, f, (float)S.fy);
            S.fz = (T)__builtin_fmaf((float)xz, f, (float)S.fz);
            return;
        }
        // KernelGradient_Spiky :24-30
        const T d = r_sqrt(xx * xx + xy * xy + xz * xz);
        const T c = sr - d;
        const T factor = P.kspikydiff * c * c / d;
        const T nx = xx * factor * nb.mass, ny = xy * factor * nb.mass, nz = xz * factor * nb.mass;
        const T w = S.wi + nb.w;
        S.fx += nx * w;
        S.fy += ny * w;
        S.fz += nz * w;
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        Vec3<T> vs = A.vStarIn[index];
        vs.x += S.fx * P.dt * (T)(-1);
        vs.y += S.fy * P.dt * (T)(-1);
        vs.z += S.fz * P.dt * (T)(-1);
        A.vStarOut[index] = vs;
        if (A.recOut)  // what the density-change passes read of this particle, as pack_vstar_mass_kernel writes it
            A.recOut[index] = Vec4<T>{vs.x, vs.y, vs.z, reinterpret_cast<const T *>(A.velMass)[4 * (size_t)index + 3]};
    }
};

// applyPStiff for scenes whose fluid particles all have one mass (Solver::fluid_mass_uniform), fast fp32 flavour: the
// neighbour's mass is the particle's own, so everything the pass reads of a neighbour fits one 16-byte record (see
// apply_record) and a neighbour costs one gather instead of two.  Same operations on the same values as
// ApplyPolicy<float, true, true>: same bits.
struct ApplyUniformPolicy {
    using T = float;
    static constexpr bool kPosInNb = true;
    static constexpr bool kDeepReplay = true;
    static constexpr bool kHitReplay = true;
#ifndef FSA_APPLY_PER
#define FSA_APPLY_PER 2
#endif
    static constexpr int kReplayPer = FSA_APPLY_PER;
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T wi, fx, fy, fz, mass; };
    struct Nb {
        float4 r;
        __device__ __forceinline__ float4 pos() const { return make_float4(r.x, r.y, r.z, 0.0f); }  // type bits: fluid
    };
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.wi = A.pstiffIn[index] / A.density[index];
        S.mass = A.velMass[index].w;
        S.fx = S.fy = S.fz = 0;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.aux4b[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool self, int,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (self) return;
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        const float rinv = __builtin_amdgcn_rsqf(d2);
        const float c = sr - d2 * rinv;
        const float f = P.kspikydiff * c * c * rinv * S.mass * (S.wi + nb.r.w);
        S.fx = __builtin_fmaf(xx, f, S.fx);
        S.fy = __builtin_fmaf(xy, f, S.fy);
        S.fz = __builtin_fmaf(xz, f, S.fz);
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        ApplyPolicy<float, true, true>::Self S0{S.wi, S.fx, S.fy, S.fz};
        ApplyPolicy<float, true, true>::finish(A, P, index, S0);
    }
};
__global__ __launch_bounds__(kBlock) void pack_apply_record_kernel(int n, const float4 *__restrict__ posType,
                                                                   const float *__restrict__ pstiff,
                                                                   const float *__restrict__ density,
                                                                   float4 *__restrict__ rec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    rec[i] = apply_record(posType[i], pstiff[i] / density[i]);
}

// ---- DensityCell :203-244 + ComputePressureKernel :246-266, TYPE_RIGID term included ---------------
// (the dedicated WCSPH density kernel of fsa_sph_tuned.hip leaves that term out and is used for scenes
// without rigid particles)
template <typename T> struct DensityRigidPolicy {
    static __device__ __forceinline__ int reach(int tj) {
        return (tj == FSA_TYPE_FLUID || tj == FSA_TYPE_RIGID) ? 1 : 0;
    }
    struct Self { T dens, mass, rd; };
    struct Nb { T rd; };
    static __device__ __forceinline__ bool wants(int) { return true; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.dens = 0;
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.restDensity[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj, int,
                                                T xx, T xy, T xz, T d2,
                                                const Vec4<T> &, const Nb &nb) {
        const T sr2 = P.sr * P.sr;
        if (d2 >= sr2) return;
        const T h = (T)((double)P.sr * 0.5);
        const T wij = kernel_cubic(P, h, xx, xy, xz);
        if (tj == FSA_TYPE_FLUID) S.dens += S.mass * wij;
        if (tj == FSA_TYPE_RIGID) S.dens += S.rd * nb.rd * wij;
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        A.densityOut[index] = S.dens;
        T pr = P.pressureK * (r_pow7(S.dens / S.rd) - 1);
        if (pr < 0) pr = 0;
        A.pstiffOut[index] = pr;  // the launcher points this at the pressure array
    }
};

// per-particle records the tuned passes gather instead of two or three separate arrays
__global__ __launch_bounds__(kBlock) void pack_vstar_mass_kernel(int n, const Vec3<float> *__restrict__ v_star,
                                                                 const float4 *__restrict__ velMass,
                                                                 float4 *__restrict__ aux4) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Vec3<float> v = v_star[i];
    aux4[i] = make_float4(v.x, v.y, v.z, velMass[i].w);
}
__global__ __launch_bounds__(kBlock) void pack_pstiff_mass_kernel(int n, const float *__restrict__ pstiff,
                                                                  const float *__restrict__ density,
                                                                  const float4 *__restrict__ velMass,
                                                                  float2 *__restrict__ aux2) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    aux2[i] = make_float2(pstiff[i] / density[i], velMass[i].w);
}

// updatePosition :986-992
template <typename T>
__global__ __launch_bounds__(kBlock) void update_position_kernel(DevParams<T> P, int n, Vec4<T> *__restrict__ posType,
                                                                 const Vec3<T> *__restrict__ v_star) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    Vec4<T> p = posType[index];
    if (unpack_int(p.w) == FSA_TYPE_RIGID) return;
    const Vec3<T> vs = v_star[index];
    p.x += vs.x * P.dt;
    p.y += vs.y * P.dt;
    p.z += vs.z * P.dt;
    posType[index] = p;
}

// UpdateVelocities :995-1018
template <typename T>
__global__ __launch_bounds__(kBlock) void update_velocities_kernel(DevParams<T> P, int n,
                                                                   const Vec4<T> *__restrict__ posType,
                                                                   Vec4<T> *__restrict__ velMass,
                                                                   Vec3<T> *__restrict__ v_star) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    if (unpack_int(posType[index].w) == FSA_TYPE_RIGID) return;
    Vec4<T> v = velMass[index];
    const Vec3<T> vs = v_star[index];
    T ax = vs.x - v.x, ay = vs.y - v.y, az = vs.z - v.z;
    ax /= P.dt;
    ay /= P.dt;
    az /= P.dt;
    const T magnitude = r_sqrt(ax * ax + ay * ay + az * az);
    if (magnitude > P.accel_limit) {
        ax = ax / magnitude * P.accel_limit;
        ay = ay / magnitude * P.accel_limit;
        az = az / magnitude * P.accel_limit;
    }
    Vec3<T> drift;
    drift.x = P.gravity[0] - ax;
    drift.y = P.gravity[1] - ay;
    drift.z = P.gravity[2] - az;
    v.x = vs.x;
    v.y = vs.y;
    v.z = vs.z;
    velMass[index] = v;
    v_star[index] = drift;
}

// max over error[]: per-block maxima, then one block over those (the comparison is `x > max`, so NaN
// entries are ignored exactly as in the reference's host loop)
template <typename T>
__global__ __launch_bounds__(kBlock) void block_max_kernel(int n, const T *__restrict__ in, T start,
                                                           T *__restrict__ out) {
    __shared__ T part[kBlock / 64];
    T m = start;
    strided_fold(in, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, n, m, [](T &a, T v) { a = v > a ? v : a; });
    for (int d = 32; d >= 1; d >>= 1) {
        const T o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) m = part[w] > m ? part[w] : m;
        out[blockIdx.x] = m;
    }
}

static bool use_fast(const Solver *s) { return use_tiles(s) && s->variant >= 2; }

static int pack_vstar_mass(Solver *s) {
    ScopedTimer t(s, "pack_vstar_mass");
    hipLaunchKernelGGL(pack_vstar_mass_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                       (const Vec3<float> *)s->b.v_star, (const float4 *)s->b.velMass[s->cur], (float4 *)s->b.aux4);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// WCSPH density for scenes with TYPE_RIGID particles, tiled (fp32, variant >= 1)
int launch_density_rigid_tuned(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "density");
    PassArrays<float> A = arrays_of<float>(s);
    A.pstiffOut = (float *)s->b.pressure;
    return run_tiled<DensityRigidPolicy<float>>(s, A, false);
}

int launch_density_alpha(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) return launch_tuned<AlphaPolicy<float>>(s, "density_alpha", true);
    DF_GENERIC(AlphaPolicy<T>, "density_alpha");
    return FSA_OK;
}

int launch_non_pressure_force(Solver *s) {
    if (s->n == 0) return FSA_OK;
    // (no lists: wall particles count beyond the support, solver/sph_kernel_impl.cuh:315-352)
    if (use_fast(s)) {
        {
            ScopedTimer t(s, "pack_npf");
            hipLaunchKernelGGL(pack_npf_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               2.0f * s->pf.viscosity, (const float4 *)s->b.posType[s->cur],
                               (const float4 *)s->b.velMass[s->cur], (const float *)s->b.density, (float4 *)s->b.aux4);
            FSA_HIP(hipGetLastError());
        }
        return launch_tuned<NpfFastPolicy>(s, "non_pressure_force");
    }
    if (use_tiles(s)) return launch_tuned<NpfPolicy<float>>(s, "non_pressure_force");
    DF_GENERIC(NpfPolicy<T>, "non_pressure_force");
    return FSA_OK;
}

static bool apply_uniform(const Solver *s) { return use_fast(s) && s->fluid_mass_uniform && s->b.hitWords; }
// Where the records of ApplyUniformPolicy live: in the buffer of the WCSPH hit record (which is stale after that)
// when this step does not replay it, else in a buffer of their own.
static float4 *apply_records(Solver *s) {
    if (replays_hits<ApplyUniformPolicy>(s) && s->b.applyRec) return (float4 *)s->b.applyRec;
    s->hits_valid = false;
    return (float4 *)s->b.hitWords;
}

// The tuned passes read packed records that a small kernel writes first.  Inside fsa_correct_density_error /
// fsa_correct_divergence_error (Solver::dfsph_chain) the passes hand the records on to each other instead: a
// density-change pass also writes what applyPStiff reads of each particle, applyPStiff what the next density-change
// pass reads, and only the first pass of a loop packs.  Same records, same bits.
template <typename Policy> static int run_stiff(Solver *s, const char *tag) {
    if (s->dfsph_chain != 2) {
        int rc = pack_vstar_mass(s);
        if (rc) return rc;
    }
    int rc = ensure_neighbours<Policy>(s);
    if (rc) return rc;
    PassArrays<float> A = arrays_of<float>(s);
    if (s->dfsph_chain) {
        if (apply_uniform(s)) {
            A.recOut = apply_records(s);
        } else {
            s->pd_valid = false;  // the buffer is shared with the WCSPH force pass
            A.rec2Out = (float *)s->b.pd;
        }
        s->dfsph_chain = 2;
    }
    ScopedTimer t(s, tag);
    return run_tiled<Policy>(s, A, true);
}

int launch_solve_density_stiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) {
        if (use_fast(s)) return run_stiff<StiffPolicy<float, true, true, true>>(s, "density_stiff");
        return run_stiff<StiffPolicy<float, true, true>>(s, "density_stiff");
    }
    // (these policy names carry commas, which a macro argument cannot: hence the alias macro)
#define FSA_POLICY StiffPolicy<T, true, false>
    DF_GENERIC(FSA_POLICY, "density_stiff");
#undef FSA_POLICY
    return FSA_OK;
}

int launch_solve_divergence_stiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) {
        if (use_fast(s)) return run_stiff<StiffPolicy<float, false, true, true>>(s, "divergence_stiff");
        return run_stiff<StiffPolicy<float, false, true>>(s, "divergence_stiff");
    }
#define FSA_POLICY StiffPolicy<T, false, false>
    DF_GENERIC(FSA_POLICY, "divergence_stiff");
#undef FSA_POLICY
    return FSA_OK;
}

int launch_apply_pstiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (apply_uniform(s)) {
        const bool chained = s->dfsph_chain == 2;  // the density-change pass before has written the records
        int rc = ensure_neighbours<ApplyUniformPolicy>(s);  // (before the records: the hit record may share their buffer)
        if (rc) return rc;
        float4 *const recs = apply_records(s);
        if (!chained) {
            ScopedTimer t(s, "pack_pstiff_mass");
            hipLaunchKernelGGL(pack_apply_record_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               (const float4 *)s->b.posType[s->cur], (const float *)s->b.pstiff,
                               (const float *)s->b.density, recs);
            FSA_HIP(hipGetLastError());
        }
        PassArrays<float> A = arrays_of<float>(s);
        A.aux4b = recs;
        if (chained) A.recOut = (float4 *)s->b.aux4;
        ScopedTimer t(s, "apply_pstiff");
        return run_tiled<ApplyUniformPolicy>(s, A, true);
    }
    if (use_tiles(s)) {
        const bool chained = s->dfsph_chain == 2;  // the density-change pass before has written the records
        if (!chained) {
            ScopedTimer t(s, "pack_pstiff_mass");
            s->pd_valid = false;  // the buffer is shared with the WCSPH force pass
            hipLaunchKernelGGL(pack_pstiff_mass_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               (const float *)s->b.pstiff, (const float *)s->b.density,
                               (const float4 *)s->b.velMass[s->cur], (float2 *)s->b.pd);
            FSA_HIP(hipGetLastError());
        }
        int rc = use_fast(s) ? ensure_neighbours<ApplyPolicy<float, true, true>>(s) : ensure_lists(s);
        if (rc) return rc;
        PassArrays<float> A = arrays_of<float>(s);
        if (chained) A.recOut = (float4 *)s->b.aux4;
        ScopedTimer t(s, "apply_pstiff");
        if (use_fast(s)) return run_tiled<ApplyPolicy<float, true, true>>(s, A, true);
        return run_tiled<ApplyPolicy<float, true>>(s, A, true);
    }
#define FSA_POLICY ApplyPolicy<T, false>
    DF_GENERIC(FSA_POLICY, "apply_pstiff");
#undef FSA_POLICY
    return FSA_OK;
}

int launch_update_position(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "update_position");
    s->lists_valid = false;  // positions move
    s->hits_valid = false;
    DF_DISPATCH(update_position_kernel, (Vec4<T> *)s->b.posType[s->cur], (const Vec3<T> *)s->b.v_star);
    return FSA_OK;
}

int launch_update_velocities(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "update_velocities");
    const int c = s->cur;
    DF_DISPATCH(update_velocities_kernel, (const Vec4<T> *)s->b.posType[c], (Vec4<T> *)s->b.velMass[c],
                (Vec3<T> *)s->b.v_star);
    return FSA_OK;
}

constexpr int kReduceBlocks = 1024;

int launch_max_error(Solver *s, double start, double *out) {
    *out = start;
    if (s->n == 0) return FSA_OK;
    {
        ScopedTimer t(s, "max_error");
        const int nb = std::min(kReduceBlocks, blocks_for(s->n));
        if (s->precision == FSA_PRECISION_F32) {
            hipLaunchKernelGGL(block_max_kernel<float>, dim3(nb), dim3(kBlock), 0, s->stream, s->n,
                               (const float *)s->b.error, (float)start, (float *)s->b.reduce_tmp);
            hipLaunchKernelGGL(block_max_kernel<float>, dim3(1), dim3(kBlock), 0, s->stream, nb,
                               (const float *)s->b.reduce_tmp, (float)start, (float *)s->b.reduce_tmp + kReduceBlocks);
        } else {
            hipLaunchKernelGGL(block_max_kernel<double>, dim3(nb), dim3(kBlock), 0, s->stream, s->n,
                               (const double *)s->b.error, start, (double *)s->b.reduce_tmp);
            hipLaunchKernelGGL(block_max_kernel<double>, dim3(1), dim3(kBlock), 0, s->stream, nb,
                               (const double *)s->b.reduce_tmp, start, (double *)s->b.reduce_tmp + kReduceBlocks);
        }
        FSA_HIP(hipGetLastError());
    }
    if (s->precision == FSA_PRECISION_F32) {
        float v = 0.f;
        FSA_HIP(hipMemcpyAsync(&v, (const float *)s->b.reduce_tmp + kReduceBlocks, sizeof(v), hipMemcpyDeviceToHost,
                               s->stream));
        FSA_HIP(counted_stream_sync(s->stream));
        *out = v;
    } else {
        double v = 0.0;
        FSA_HIP(hipMemcpyAsync(&v, (const double *)s->b.reduce_tmp + kReduceBlocks, sizeof(v), hipMemcpyDeviceToHost,
                               s->stream));
        FSA_HIP(counted_stream_sync(s->stream));
        *out = v;
    }
    return FSA_OK;
}


// ---- the two correction loops with the stop test on the device ---------------------------------------------
// The reference copies error[] to the host after every density-change pass and stops when its maximum is below the
// threshold (solver/sph_solver.cu:169-186, 226-243).  Here all rounds are queued at once: after each density-change pass
// two small kernels take the maximum and apply the test on the device, and once it has fired the pass kernels of
// the remaining rounds return at their first instruction (DevParams::halt).  State behind the block maxima in
// reduce_tmp: {last maximum} as a double, one spare, then {applyPStiff passes done, stop} as ints.
constexpr int kLoopState = kReduceBlocks + 8;

template <typename T>
__global__ __launch_bounds__(kBlock) void halted_block_max_kernel(int n, const T *__restrict__ in, T start,
                                                                  T *__restrict__ out, const int *__restrict__ halt) {
    if (*halt) return;
    __shared__ T part[kBlock / 64];
    T m = start;
    strided_fold(in, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, n, m, [](T &a, T v) { a = v > a ? v : a; });
    for (int d = 32; d >= 1; d >>= 1) {
        const T o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) m = part[w] > m ? part[w] : m;
        out[blockIdx.x] = m;
    }
}

// One block: the maximum of the block maxima (folded as block_max_kernel folds them) and the test `error < ethres`.
template <typename T>
__global__ __launch_bounds__(kBlock) void dfsph_round_end_kernel(int nb, T start, double ethres, const T *__restrict__ part_in,
                                                                 double *__restrict__ st) {
    int *ist = reinterpret_cast<int *>(st + 2);
    if (ist[1]) return;
    __shared__ T part[kBlock / 64];
    T m = start;
    for (int i = threadIdx.x; i < nb; i += kBlock) {
        const T v = part_in[i];
        m = v > m ? v : m;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const T o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) m = part[w] > m ? part[w] : m;
        st[0] = (double)m;
        if ((double)m < ethres)
            ist[1] = 1;
        else
            ist[0] += 1;  // applyPStiff follows
    }
}

int launch_dfsph_loop_begin(Solver *s) {
    double *st = (double *)s->b.reduce_tmp + kLoopState;
    FSA_HIP(hipMemsetAsync(st, 0, 4 * sizeof(double), s->stream));
    s->pf.halt = s->pd.halt = reinterpret_cast<const int *>(st + 2) + 1;
    return FSA_OK;
}

int launch_dfsph_round_end(Solver *s, double start, double ethres) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "max_error");
    const int nb = std::min(kReduceBlocks, blocks_for(s->n));
    double *st = (double *)s->b.reduce_tmp + kLoopState;
    if (s->precision == FSA_PRECISION_F32) {
        hipLaunchKernelGGL(halted_block_max_kernel<float>, dim3(nb), dim3(kBlock), 0, s->stream, s->n,
                           (const float *)s->b.error, (float)start, (float *)s->b.reduce_tmp, s->pf.halt);
        hipLaunchKernelGGL(dfsph_round_end_kernel<float>, dim3(1), dim3(kBlock), 0, s->stream, nb, (float)start, ethres,
                           (const float *)s->b.reduce_tmp, st);
    } else {
        hipLaunchKernelGGL(halted_block_max_kernel<double>, dim3(nb), dim3(kBlock), 0, s->stream, s->n,
                           (const double *)s->b.error, start, (double *)s->b.reduce_tmp, s->pf.halt);
        hipLaunchKernelGGL(dfsph_round_end_kernel<double>, dim3(1), dim3(kBlock), 0, s->stream, nb, start, ethres,
                           (const double *)s->b.reduce_tmp, st);
    }
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// The kernels queued after this run unconditionally again.
void dfsph_loop_end(Solver *s) { s->pf.halt = s->pd.halt = nullptr; }

// One read-back: how many applyPStiff passes ran and the last maximum.
int launch_dfsph_loop_result(Solver *s, double start, int *iterations, double *last_error) {
    double st[3] = {0, 0, 0};
    FSA_HIP(hipMemcpyAsync(st, (const double *)s->b.reduce_tmp + kLoopState, sizeof(st), hipMemcpyDeviceToHost, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    int ist[2];
    std::memcpy(ist, &st[2], sizeof(ist));
    if (iterations) *iterations = s->n == 0 ? 0 : ist[0];
    if (last_error) *last_error = s->n == 0 ? start : st[0];
    return FSA_OK;
}

}  // namespace fsa
