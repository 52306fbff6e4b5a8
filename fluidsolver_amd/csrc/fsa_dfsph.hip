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
#include <algorithm>

#include <type_traits>

#include "fsa_pair.h"
#include "fsa_solver.h"
#include "fsa_tile.h"

namespace fsa {

// Every neighbour pass of the step is written once, as a policy: which particles it covers, what it
// reads of a neighbour, the loop body of the reference's *Cell function (`pair`), and what it stores.
// Two kernels run a policy: generic_pass_kernel (one thread per particle, neighbours straight from
// global memory; any precision) and tuned_pass_kernel (fp32: the LDS row tiles and per-lane hit queues
// of the WCSPH force pass).  Both call the same `pair` in the same neighbour order, so they give the
// same bits.
template <typename T> struct DfArrays {
    const Vec4<T> *posType;
    const Vec4<T> *velMass;
    const T *restDensity;
    const T *density;
    const T *dfFactor;
    const T *pstiffIn;
    const Vec3<T> *vStarIn;
    const Vec4<T> *aux4;  // {v_star, mass} per particle (tuned density-change passes)
    const T *aux2;        // {pstiff / density, mass} pairs (tuned applyPStiff)
    T *densityOut;
    T *dfFactorOut;
    T *errorOut;
    T *pstiffOut;
    Vec3<T> *vStarOut;
};

// ---- DensityAlphaCell :582-624 + computeDensityAlpha_kernel :626-660 -------------------------------
template <typename T> struct AlphaPolicy {
    // which neighbour types count: 1 = inside the support, 2 = anywhere in the 27 cells, 0 = never
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T dens, mwij, mx, my, mz, mass; };
    struct Nb { T mass; };
    static __device__ __forceinline__ bool wants(int) { return true; }
    static __device__ __forceinline__ void skip(const DfArrays<T> &, int) {}
    static __device__ __forceinline__ void init(const DfArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.dens = S.mwij = S.mx = S.my = S.mz = 0;
        S.mass = A.velMass[index].w;
    }
    static __device__ __forceinline__ Nb load(const DfArrays<T> &A, int j) {
        return Nb{reinterpret_cast<const T *>(A.velMass)[4 * (size_t)j + 3]};
    }
    static __device__ __forceinline__ void pair(const DevParams<T> &P, Self &S, bool self, int tj, T xx, T xy, T xz,
                                                T d2, const Vec4<T> &, const Nb &nb) {
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
    static __device__ __forceinline__ void finish(const DfArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
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
    static __device__ __forceinline__ void skip(const DfArrays<T> &A, int index) {
        A.vStarOut[index] = Vec3<T>{0, 0, 0};
    }
    static __device__ __forceinline__ void init(const DfArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.F.p = A.posType[index];
        S.F.v = A.velMass[index];
        S.F.pterm = 0;
        S.F.fx = S.F.fy = S.F.fz = 0;
    }
    static __device__ __forceinline__ Nb load(const DfArrays<T> &A, int j) { return Nb{A.velMass[j], A.density[j]}; }
    static __device__ __forceinline__ void pair(const DevParams<T> &P, Self &S, bool self, int tj, T xx, T xy, T xz,
                                                T d2, const Vec4<T> &pj, const Nb &nb) {
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
    static __device__ __forceinline__ void finish(const DfArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
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

// ---- DensityChangeCell :775-813 / PredictDensityCell :846-887 and the two kernels around them -------
// TIMES_DT = false: solveDivergenceStiff :818-842; true: solveDensityStiff :890-919
template <typename T, bool TIMES_DT, bool PACKED> struct StiffPolicy {
    // which neighbour types count: 1 = inside the support, 2 = anywhere in the 27 cells, 0 = never
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T acc; Vec3<T> vs; };
    struct Nb { T vx, vy, vz, mass; };
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    static __device__ __forceinline__ void skip(const DfArrays<T> &A, int index) {
        A.errorOut[index] = 0;
        A.pstiffOut[index] = 0;
    }
    static __device__ __forceinline__ void init(const DfArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.acc = TIMES_DT ? A.density[index] : (T)0;
        S.vs = A.vStarIn[index];
    }
    static __device__ __forceinline__ Nb load(const DfArrays<T> &A, int j) {
        if (PACKED) {
            const Vec4<T> a = A.aux4[j];
            return Nb{a.x, a.y, a.z, a.w};
        }
        const Vec3<T> v = A.vStarIn[j];
        return Nb{v.x, v.y, v.z, reinterpret_cast<const T *>(A.velMass)[4 * (size_t)j + 3]};
    }
    static __device__ __forceinline__ void pair(const DevParams<T> &P, Self &S, bool self, int tj, T xx, T xy, T xz,
                                                T d2, const Vec4<T> &, const Nb &nb) {
        if (self) return;
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
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
    static __device__ __forceinline__ void finish(const DfArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
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
        } else {
            T densChange = S.acc;
            if (densChange < 0) densChange = 0;
            A.errorOut[index] = densChange / rd;
            A.pstiffOut[index] = densChange / P.dt * A.dfFactor[index];
        }
    }
};

// ---- applyPressureCell :921-961 + applyPStiff :963-984 ---------------------------------------------
// Neighbours are read through pos / pstiff / density only, v_star is written for the particle itself:
// no ordering hazard between threads.
template <typename T, bool PACKED> struct ApplyPolicy {
    // which neighbour types count: 1 = inside the support, 2 = anywhere in the 27 cells, 0 = never
    static __device__ __forceinline__ int reach(int tj) { return tj == FSA_TYPE_FLUID ? 1 : 0; }
    struct Self { T wi, fx, fy, fz; };
    struct Nb { T w, mass; };  // w = pstiff_j / density_j
    static __device__ __forceinline__ bool wants(int type) { return type == FSA_TYPE_FLUID; }
    static __device__ __forceinline__ void skip(const DfArrays<T> &, int) {}
    static __device__ __forceinline__ void init(const DfArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.wi = A.pstiffIn[index] / A.density[index];
        S.fx = S.fy = S.fz = 0;
    }
    static __device__ __forceinline__ Nb load(const DfArrays<T> &A, int j) {
        if (PACKED) return Nb{A.aux2[2 * (size_t)j], A.aux2[2 * (size_t)j + 1]};
        return Nb{A.pstiffIn[j] / A.density[j], reinterpret_cast<const T *>(A.velMass)[4 * (size_t)j + 3]};
    }
    static __device__ __forceinline__ void pair(const DevParams<T> &P, Self &S, bool self, int tj, T xx, T xy, T xz,
                                                T d2, const Vec4<T> &, const Nb &nb) {
        if (self || tj != FSA_TYPE_FLUID) return;
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
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
    static __device__ __forceinline__ void finish(const DfArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        Vec3<T> vs = A.vStarIn[index];
        vs.x += S.fx * P.dt * (T)(-1);
        vs.y += S.fy * P.dt * (T)(-1);
        vs.z += S.fz * P.dt * (T)(-1);
        A.vStarOut[index] = vs;
    }
};

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
    static __device__ __forceinline__ void skip(const DfArrays<T> &, int) {}
    static __device__ __forceinline__ void init(const DfArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.dens = 0;
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
    }
    static __device__ __forceinline__ Nb load(const DfArrays<T> &A, int j) { return Nb{A.restDensity[j]}; }
    static __device__ __forceinline__ void pair(const DevParams<T> &P, Self &S, bool, int tj, T xx, T xy, T xz, T d2,
                                                const Vec4<T> &, const Nb &nb) {
        const T sr2 = P.sr * P.sr;
        if (d2 >= sr2) return;
        const T h = (T)((double)P.sr * 0.5);
        const T wij = kernel_cubic(P, h, xx, xy, xz);
        if (tj == FSA_TYPE_FLUID) S.dens += S.mass * wij;
        if (tj == FSA_TYPE_RIGID) S.dens += S.rd * nb.rd * wij;
    }
    static __device__ __forceinline__ void finish(const DfArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        A.densityOut[index] = S.dens;
        T pr = P.pressureK * (r_pow(S.dens / S.rd, (T)7) - 1);
        if (pr < 0) pr = 0;
        A.pstiffOut[index] = pr;  // the launcher points this at the pressure array
    }
};

// One particle against its 27 cells in the reference's order, neighbours from global memory.
template <typename T, typename Policy>
__device__ __forceinline__ void generic_walk(const DevParams<T> &P, const DfArrays<T> &A, int index,
                                             const Vec4<T> &pi, const int *__restrict__ cellStart,
                                             typename Policy::Self &S) {
    const RowWalk<T> walk(P, pi.x, pi.y, pi.z);
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) {
                const Vec4<T> pj = A.posType[j];
                const int tj = unpack_int(pj.w);
                if (Policy::reach(tj) == 0) continue;
                const T xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
                const T d2 = xx * xx + xy * xy + xz * xz;
                Policy::pair(P, S, j == index, tj, xx, xy, xz, d2, pj, Policy::load(A, j));
            }
        }
}

template <typename T, typename Policy>
__global__ __launch_bounds__(kBlock) void generic_pass_kernel(DevParams<T> P, int n, DfArrays<T> A,
                                                              const int *__restrict__ cellStart) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = A.posType[index];
    if (!Policy::wants(unpack_int(pi.w))) {
        Policy::skip(A, index);
        return;
    }
    typename Policy::Self S;
    Policy::init(A, P, index, S);
    generic_walk<T, Policy>(P, A, index, pi, cellStart, S);
    Policy::finish(A, P, index, S);
}

// The tiled flavour (fp32): row tiles in LDS, each lane culls its range of every row into a queue of
// 16-bit (row, index) entries, the queue is drained in double-buffered batches of gathered neighbours.
// Candidates enter the queue on a fused squared distance with a little slack; `pair` then applies the
// reference's own tests, in queue (= reference) order.
template <typename Policy, int BS, int ROWCAP, int Q>
__global__ __launch_bounds__(BS) void tuned_pass_kernel(DevParams<float> P, int n, DfArrays<float> A,
                                                        const int *__restrict__ cellStart,
                                                        const int *__restrict__ sortedHash, int xcd_log) {
    using Shared = RowShared<BS, ROWCAP, true>;
    __shared__ Shared sh;
    __shared__ unsigned short queue[(Q + 2 * kDrainBatch) * BS];  // the drain reads ahead of the last entry
    const int tid = threadIdx.x;
    const int b0 = xcd_block(xcd_log) * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const unsigned hlo = (unsigned)sortedHash[b0], hhi = (unsigned)sortedHash[b0 + nb - 1];

    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    bool active = false;
    if (index < n) {
        pi = A.posType[index];
        active = Policy::wants(unpack_int(pi.w));
        if (!active) Policy::skip(A, index);
    }
    if (!__syncthreads_or(active ? 1 : 0)) return;  // e.g. a block of wall particles
    const bool staged = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, hlo, hhi);
    typename Policy::Self S;
    if (active) Policy::init(A, P, index, S);
    if (!staged) {
        if (active) {
            generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
            Policy::finish(A, P, index, S);
        }
        return;
    }
    const float sr = P.sr, sr2c = sr * sr * 1.000002f;
    const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
    unsigned short *const q = queue + tid;
    const int safe = index < n ? index : 0;
    int cnt = 0;

    struct Batch {
        int j[kDrainBatch];
        float4 pj[kDrainBatch];
        typename Policy::Nb nb[kDrainBatch];
    };
    auto batch_load = [&](Batch &B, int s) {
#pragma unroll
        for (int u = 0; u < kDrainBatch; u++) {
            const int e = q[(s + u) * BS];
            B.j[u] = s + u < cnt ? sh.rowStartG[e >> 9] + (e & 511) : safe;
        }
#pragma unroll
        for (int u = 0; u < kDrainBatch; u++) {
            B.pj[u] = A.posType[B.j[u]];
            B.nb[u] = Policy::load(A, B.j[u]);
        }
    };
    auto batch_eval = [&](const Batch &B, int s) {
#pragma unroll
        for (int u = 0; u < kDrainBatch; u++)
            if (s + u < cnt) {
                const float xx = pi.x - B.pj[u].x, xy = pi.y - B.pj[u].y, xz = pi.z - B.pj[u].z;
                const float d2 = xx * xx + xy * xy + xz * xz;
                Policy::pair(P, S, B.j[u] == index, unpack_int(B.pj[u].w), xx, xy, xz, d2, B.pj[u], B.nb[u]);
            }
    };
    auto drain = [&]() {
        Batch B0, B1;
        batch_load(B0, 0);
        for (int s = 0; wave_any(s < cnt); s += 2 * kDrainBatch) {
            batch_load(B1, s + kDrainBatch);
            batch_eval(B0, s);
            if (!wave_any(s + kDrainBatch < cnt)) break;
            batch_load(B0, s + 2 * kDrainBatch);
            batch_eval(B1, s + kDrainBatch);
        }
        cnt = 0;
    };

    float4 e0, e1;
    int lj, le, nlj = 0, nle = 0;
    row_fetch<BS, ROWCAP, false>(sh, 0, A.posType, e0, e1);
    lane_range(P, sh, walk, active, 0, cellStart, lj, le);
    row_store<BS, ROWCAP, true>(sh, 0, e0, e1);
    __syncthreads();
    for (int r = 0; r < 9; r++) {
        const int buf = r & 1;
        if (r + 1 < 9) {
            row_fetch<BS, ROWCAP, false>(sh, r + 1, A.posType, e0, e1);
            lane_range(P, sh, walk, active, r + 1, cellStart, nlj, nle);
        }
        const int tag = r << 9;
        auto cull_group = [&](int g, int lo, int hi, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            float d2[kChunk];
            group_d2<true>(sh.x[buf][g], sh.y[buf][g], sh.z[buf][g], pi.x, pi.y, pi.z, d2);
            const int4 T4 = sh.t[buf][g];
            const int tj[kChunk] = {T4.x, T4.y, T4.z, T4.w};
            const int etag = tag | (g * kChunk);
            if (!wave_any((T4.x | T4.y | T4.z | T4.w) != 0)) {
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    q[cnt * BS] = (unsigned short)(etag + u);  // a miss is overwritten by the next store
                    cnt += (d2[u] < sr2c && (!MASKED || (u >= lo && u < hi))) ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const bool mine = !MASKED || (u >= lo && u < hi);
                    const int reach = Policy::reach(tj[u]);
                    if (mine && ((reach == 1 && d2[u] < sr2c) || reach == 2)) {
                        q[cnt * BS] = (unsigned short)(etag + u);
                        cnt++;
                    }
                }
            }
        };
        const bool has = lj < le;
        int base = lj & ~(kChunk - 1);
        if (wave_any(has)) {
            cull_group(has ? base / kChunk : Shared::kPad, lj - base, le - base, std::true_type());
            if (wave_any(cnt > Q - kChunk)) drain();
        }
        base = has ? base + kChunk : le;
        while (wave_any(base + kChunk <= le)) {
            const bool go = base + kChunk <= le;
            cull_group(go ? base / kChunk : Shared::kPad, 0, kChunk, std::false_type());
            base += go ? kChunk : 0;
            if (wave_any(cnt > Q - kChunk)) drain();
        }
        if (wave_any(base < le)) {
            const int rem = le - base;
            cull_group(rem > 0 ? base / kChunk : Shared::kPad, 0, rem, std::true_type());
            if (wave_any(cnt > Q - kChunk)) drain();
        }
        if (r + 1 < 9) {
            row_store<BS, ROWCAP, true>(sh, buf ^ 1, e0, e1);
            lj = nlj;
            le = nle;
            __syncthreads();
        }
    }
    drain();
    if (active) Policy::finish(A, P, index, S);
}

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
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const T v = in[i];
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
        out[blockIdx.x] = m;
    }
}

static inline int blocks_for(long long n) { return (int)((n + kBlock - 1) / kBlock); }

#define DF_DISPATCH(kernel, ...)                                                                                  \
    do {                                                                                                          \
        if (s->precision == FSA_PRECISION_F32) {                                                                  \
            using T = float;                                                                                      \
            const DevParams<T> &P = s->pf;                                                                        \
            hipLaunchKernelGGL(kernel<T>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, P, s->n, __VA_ARGS__); \
        } else {                                                                                                  \
            using T = double;                                                                                     \
            const DevParams<T> &P = s->pd;                                                                        \
            hipLaunchKernelGGL(kernel<T>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, P, s->n, __VA_ARGS__); \
        }                                                                                                         \
        FSA_HIP(hipGetLastError());                                                                               \
    } while (0)

template <typename T> static DfArrays<T> arrays_of(Solver *s) {
    const int c = s->cur;
    DfArrays<T> A{};
    A.posType = (const Vec4<T> *)s->b.posType[c];
    A.velMass = (const Vec4<T> *)s->b.velMass[c];
    A.restDensity = (const T *)s->b.restDensity[c];
    A.density = (const T *)s->b.density;
    A.dfFactor = (const T *)s->b.DF_factor;
    A.pstiffIn = (const T *)s->b.pstiff;
    A.vStarIn = (const Vec3<T> *)s->b.v_star;
    A.aux4 = (const Vec4<T> *)s->b.aux4;
    A.aux2 = (const T *)s->b.pd;
    A.densityOut = (T *)s->b.density;
    A.dfFactorOut = (T *)s->b.DF_factor;
    A.errorOut = (T *)s->b.error;
    A.pstiffOut = (T *)s->b.pstiff;
    A.vStarOut = (Vec3<T> *)s->b.v_star;
    return A;
}

constexpr int kQueueDf = 48;  // 16-bit slots, as in the WCSPH force pass

static bool use_tiles(const Solver *s) { return s->precision == FSA_PRECISION_F32 && s->variant >= 1; }

// Policy<float> through the tiled kernel
template <typename Policy> static int launch_tuned(Solver *s, const char *tag) {
    ScopedTimer t(s, tag);
    const int unit = kXcds << xcd_run_log();
    const int nb = ((s->n + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
    hipLaunchKernelGGL((tuned_pass_kernel<Policy, kTunedBS, kRowCap, kQueueDf>), dim3(nb), dim3(kTunedBS), 0,
                       s->stream, s->pf, s->n, arrays_of<float>(s), s->b.cellStart, s->b.sortedHash, xcd_run_log());
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// Policy<T> through the one-thread-per-particle kernel, either precision
#define DF_GENERIC(PolicyT, tag)                                                                           \
    do {                                                                                                   \
        ScopedTimer t(s, tag);                                                                             \
        if (s->precision == FSA_PRECISION_F32) {                                                           \
            using T = float;                                                                               \
            hipLaunchKernelGGL((generic_pass_kernel<T, PolicyT>), dim3(blocks_for(s->n)), dim3(kBlock), 0, \
                               s->stream, s->pf, s->n, arrays_of<T>(s), s->b.cellStart);                   \
        } else {                                                                                           \
            using T = double;                                                                              \
            hipLaunchKernelGGL((generic_pass_kernel<T, PolicyT>), dim3(blocks_for(s->n)), dim3(kBlock), 0, \
                               s->stream, s->pd, s->n, arrays_of<T>(s), s->b.cellStart);                   \
        }                                                                                                  \
        FSA_HIP(hipGetLastError());                                                                        \
    } while (0)

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
    DfArrays<float> A = arrays_of<float>(s);
    A.pstiffOut = (float *)s->b.pressure;
    const int unit = kXcds << xcd_run_log();
    const int nb = ((s->n + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
    hipLaunchKernelGGL((tuned_pass_kernel<DensityRigidPolicy<float>, kTunedBS, kRowCap, kQueueDf>), dim3(nb),
                       dim3(kTunedBS), 0, s->stream, s->pf, s->n, A, s->b.cellStart, s->b.sortedHash, xcd_run_log());
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_density_alpha(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) return launch_tuned<AlphaPolicy<float>>(s, "density_alpha");
    DF_GENERIC(AlphaPolicy<T>, "density_alpha");
    return FSA_OK;
}

int launch_non_pressure_force(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) return launch_tuned<NpfPolicy<float>>(s, "non_pressure_force");
    DF_GENERIC(NpfPolicy<T>, "non_pressure_force");
    return FSA_OK;
}

int launch_solve_density_stiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) {
        int rc = pack_vstar_mass(s);
        if (rc) return rc;
        return launch_tuned<StiffPolicy<float, true, true>>(s, "density_stiff");
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
        int rc = pack_vstar_mass(s);
        if (rc) return rc;
        return launch_tuned<StiffPolicy<float, false, true>>(s, "divergence_stiff");
    }
#define FSA_POLICY StiffPolicy<T, false, false>
    DF_GENERIC(FSA_POLICY, "divergence_stiff");
#undef FSA_POLICY
    return FSA_OK;
}

int launch_apply_pstiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) {
        {
            ScopedTimer t(s, "pack_pstiff_mass");
            s->pd_valid = false;  // the buffer is shared with the WCSPH force pass
            hipLaunchKernelGGL(pack_pstiff_mass_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               (const float *)s->b.pstiff, (const float *)s->b.density,
                               (const float4 *)s->b.velMass[s->cur], (float2 *)s->b.pd);
            FSA_HIP(hipGetLastError());
        }
        return launch_tuned<ApplyPolicy<float, true>>(s, "apply_pstiff");
    }
#define FSA_POLICY ApplyPolicy<T, false>
    DF_GENERIC(FSA_POLICY, "apply_pstiff");
#undef FSA_POLICY
    return FSA_OK;
}

int launch_update_position(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "update_position");
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
        FSA_HIP(hipStreamSynchronize(s->stream));
        *out = v;
    } else {
        double v = 0.0;
        FSA_HIP(hipMemcpyAsync(&v, (const double *)s->b.reduce_tmp + kReduceBlocks, sizeof(v), hipMemcpyDeviceToHost,
                               s->stream));
        FSA_HIP(hipStreamSynchronize(s->stream));
        *out = v;
    }
    return FSA_OK;
}

}  // namespace fsa
