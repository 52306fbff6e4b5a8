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

#include "fsa_pair.h"
#include "fsa_solver.h"

namespace fsa {

// Visits every particle j of the 27 cells around particle `index` in the reference's order.
template <typename T, typename F>
__device__ __forceinline__ void for_neighbours(const DevParams<T> &P, const Vec4<T> &pi,
                                               const int *__restrict__ cellStart, F &&body) {
    const RowWalk<T> walk(P, pi.x, pi.y, pi.z);
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            int start, end;
            if (!walk.row(P, cellStart, dy, dz, start, end)) continue;
            for (int j = start; j < end; j++) body(j);
        }
}

// DensityAlphaCell :582-624 + computeDensityAlpha_kernel :626-660
template <typename T>
__global__ __launch_bounds__(kBlock) void density_alpha_kernel(DevParams<T> P, int n,
                                                               const Vec4<T> *__restrict__ posType,
                                                               const Vec4<T> *__restrict__ velMass,
                                                               const int *__restrict__ cellStart,
                                                               T *__restrict__ density, T *__restrict__ dfFactor) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = posType[index];
    const T sr = P.sr, sr2 = sr * sr;
    T dens = 0, mwij = 0, mx = 0, my = 0, mz = 0;
    for_neighbours(P, pi, cellStart, [&](int j) {
        if (j == index) return;
        const Vec4<T> pj = posType[j];
        const T xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
        const T d2 = xx * xx + xy * xy + xz * xz;
        if (d2 >= sr2) return;
        const T d = r_sqrt(d2);
        const T c2 = sr2 - d2;
        const T c = sr - d;
        const T nablaw = P.kspikydiff * c * c / d;
        if (unpack_int(pj.w) == FSA_TYPE_FLUID) {
            const T mj = velMass[j].w;
            dens += c2 * c2 * c2 * mj;
            const T ax = xx * nablaw * mj, ay = xy * nablaw * mj, az = xz * nablaw * mj;
            mwij += ax * ax + ay * ay + az * az;
            mx += ax;
            my += ay;
            mz += az;
        }
    });
    dens += sr2 * sr2 * sr2 * velMass[index].w;
    dens *= P.kpoly6;
    density[index] = dens;
    T denom = mx * mx + my * my + mz * mz + mwij;
    if ((double)denom < 0.000001) denom = (T)0.000001;  // clamp for stability, :656-657
    dfFactor[index] = dens / denom;
}

// computeNPFCell :663-742 + computeNPF_kernel :744-773: viscosity, wall penalty, gravity -> v_star
template <typename T>
__global__ __launch_bounds__(kBlock) void non_pressure_force_kernel(DevParams<T> P, int n,
                                                                    const Vec4<T> *__restrict__ posType,
                                                                    const Vec4<T> *__restrict__ velMass,
                                                                    const T *__restrict__ density,
                                                                    const int *__restrict__ cellStart,
                                                                    Vec3<T> *__restrict__ v_star) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    ForceSelf<T> S;
    S.p = posType[index];
    if (unpack_int(S.p.w) != FSA_TYPE_FLUID) {
        v_star[index] = Vec3<T>{0, 0, 0};
        return;
    }
    S.v = velMass[index];
    S.pterm = 0;
    S.fx = S.fy = S.fz = 0;
    const T sr = P.sr, sr2 = sr * sr;
    for_neighbours(P, S.p, cellStart, [&](int j) {
        const Vec4<T> pj = posType[j];
        const int tj = unpack_int(pj.w);
        const T xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
        const T d2 = xx * xx + xy * xy + xz * xz;
        if (j != index && tj == FSA_TYPE_FLUID) {
            if (d2 >= sr2) return;
            const T d = r_sqrt(d2);
            const T c = sr - d;
            const T nablaw = P.kspikydiff * c * c / d;
            const Vec4<T> vj = velMass[j];
            const T vc = (T)((double)(nablaw * d2) / ((double)d2 + 0.01 * (double)sr2) * (double)vj.w /
                             (double)density[j] * 2 * (double)P.viscosity);
            S.fx += (S.v.x - vj.x) * vc;
            S.fy += (S.v.y - vj.y) * vc;
            S.fz += (S.v.z - vj.z) * vc;
        }
        // the wall branch is ForceCell's, text for text; force_pair does nothing else for this type
        if (tj == FSA_TYPE_BOUNDARY) force_pair(P, sr, sr2, S, j == index, pj, velMass[j], (T)0, (T)1);
    });
    S.fx += (T)0;
    S.fy += (T)-9.8;  // :770
    S.fz += (T)0;
    Vec3<T> vs;
    vs.x = S.v.x + S.fx * P.dt;
    vs.y = S.v.y + S.fy * P.dt;
    vs.z = S.v.z + S.fz * P.dt;
    v_star[index] = vs;
}

// sum over fluid neighbours of dot(v*_i - v*_j, grad W_ij m_j): DensityChangeCell :775-813 and, with
// every term times dt, PredictDensityCell :846-887
template <typename T, bool TIMES_DT>
__device__ __forceinline__ T density_change(const DevParams<T> &P, int index, const Vec4<T> &pi, T start,
                                            const Vec4<T> *__restrict__ posType,
                                            const Vec4<T> *__restrict__ velMass,
                                            const Vec3<T> *__restrict__ v_star,
                                            const int *__restrict__ cellStart) {
    const T sr = P.sr, sr2 = sr * sr;
    const Vec3<T> vi = v_star[index];
    T acc = start;
    for_neighbours(P, pi, cellStart, [&](int j) {
        if (j == index) return;
        const Vec4<T> pj = posType[j];
        const T xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
        const T d2 = xx * xx + xy * xy + xz * xz;
        if (d2 >= sr2) return;
        const T d = r_sqrt(d2);
        const T c = sr - d;
        const T nablaw = P.kspikydiff * c * c / d;
        if (unpack_int(pj.w) == FSA_TYPE_FLUID) {
            const Vec3<T> vj = v_star[j];
            const T vx = vi.x - vj.x, vy = vi.y - vj.y, vz = vi.z - vj.z;
            const T mj = velMass[j].w;
            const T nx = xx * nablaw * mj, ny = xy * nablaw * mj, nz = xz * nablaw * mj;
            const T dot = vx * nx + vy * ny + vz * nz;
            acc += TIMES_DT ? dot * P.dt : dot;
        }
    });
    return acc;
}

// solveDivergenceStiff :818-842
template <typename T>
__global__ __launch_bounds__(kBlock) void divergence_stiff_kernel(DevParams<T> P, int n,
                                                                  const Vec4<T> *__restrict__ posType,
                                                                  const Vec4<T> *__restrict__ velMass,
                                                                  const T *__restrict__ restDensity,
                                                                  const Vec3<T> *__restrict__ v_star,
                                                                  const T *__restrict__ dfFactor,
                                                                  const int *__restrict__ cellStart,
                                                                  T *__restrict__ error, T *__restrict__ pstiff) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = posType[index];
    if (unpack_int(pi.w) != FSA_TYPE_FLUID) {
        error[index] = 0;
        pstiff[index] = 0;
        return;
    }
    T densChange = density_change<T, false>(P, index, pi, (T)0, posType, velMass, v_star, cellStart);
    if (densChange < 0) densChange = 0;
    error[index] = densChange / restDensity[index];
    pstiff[index] = densChange / P.dt * dfFactor[index];
}

// solveDensityStiff :890-919
template <typename T>
__global__ __launch_bounds__(kBlock) void density_stiff_kernel(DevParams<T> P, int n,
                                                               const Vec4<T> *__restrict__ posType,
                                                               const Vec4<T> *__restrict__ velMass,
                                                               const T *__restrict__ restDensity,
                                                               const Vec3<T> *__restrict__ v_star,
                                                               const T *__restrict__ density,
                                                               const T *__restrict__ dfFactor,
                                                               const int *__restrict__ cellStart,
                                                               T *__restrict__ error, T *__restrict__ pstiff) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = posType[index];
    if (unpack_int(pi.w) != FSA_TYPE_FLUID) {
        error[index] = 0;
        pstiff[index] = 0;
        return;
    }
    const T dens = density_change<T, true>(P, index, pi, density[index], posType, velMass, v_star, cellStart);
    const T rd = restDensity[index];
    T ps = (dens - rd) * dfFactor[index] / P.dt / P.dt;
    T err = dens - rd;
    if (ps < 0) {
        ps = 0;
        err = 0;
    }
    pstiff[index] = ps;
    error[index] = err;
}

// applyPressureCell :921-961 + applyPStiff :963-984.  Neighbours are read through pos / pstiff /
// density only, v_star is written for the particle itself: no ordering hazard between threads.
template <typename T>
__global__ __launch_bounds__(kBlock) void apply_pstiff_kernel(DevParams<T> P, int n,
                                                              const Vec4<T> *__restrict__ posType,
                                                              const Vec4<T> *__restrict__ velMass,
                                                              const T *__restrict__ density,
                                                              const T *__restrict__ pstiff,
                                                              const int *__restrict__ cellStart,
                                                              Vec3<T> *__restrict__ v_star) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = posType[index];
    if (unpack_int(pi.w) != FSA_TYPE_FLUID) return;
    const T sr = P.sr, sr2 = sr * sr;
    const T wi = pstiff[index] / density[index];
    T fx = 0, fy = 0, fz = 0;
    for_neighbours(P, pi, cellStart, [&](int j) {
        if (j == index) return;
        const Vec4<T> pj = posType[j];
        if (unpack_int(pj.w) != FSA_TYPE_FLUID) return;
        const T xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
        const T d2 = xx * xx + xy * xy + xz * xz;
        if (d2 >= sr2) return;
        // KernelGradient_Spiky :24-30
        const T d = r_sqrt(xx * xx + xy * xy + xz * xz);
        const T c = sr - d;
        const T factor = P.kspikydiff * c * c / d;
        const T mj = velMass[j].w;
        const T nx = xx * factor * mj, ny = xy * factor * mj, nz = xz * factor * mj;
        const T w = wi + pstiff[j] / density[j];
        fx += nx * w;
        fy += ny * w;
        fz += nz * w;
    });
    Vec3<T> vs = v_star[index];
    vs.x += fx * P.dt * (T)(-1);
    vs.y += fy * P.dt * (T)(-1);
    vs.z += fz * P.dt * (T)(-1);
    v_star[index] = vs;
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

int launch_density_alpha(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "density_alpha");
    const int c = s->cur;
    DF_DISPATCH(density_alpha_kernel, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                s->b.cellStart, (T *)s->b.density, (T *)s->b.DF_factor);
    return FSA_OK;
}

int launch_non_pressure_force(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "non_pressure_force");
    const int c = s->cur;
    DF_DISPATCH(non_pressure_force_kernel, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                (const T *)s->b.density, s->b.cellStart, (Vec3<T> *)s->b.v_star);
    return FSA_OK;
}

int launch_solve_density_stiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "density_stiff");
    const int c = s->cur;
    DF_DISPATCH(density_stiff_kernel, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                (const T *)s->b.restDensity[c], (const Vec3<T> *)s->b.v_star, (const T *)s->b.density,
                (const T *)s->b.DF_factor, s->b.cellStart, (T *)s->b.error, (T *)s->b.pstiff);
    return FSA_OK;
}

int launch_solve_divergence_stiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "divergence_stiff");
    const int c = s->cur;
    DF_DISPATCH(divergence_stiff_kernel, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                (const T *)s->b.restDensity[c], (const Vec3<T> *)s->b.v_star, (const T *)s->b.DF_factor,
                s->b.cellStart, (T *)s->b.error, (T *)s->b.pstiff);
    return FSA_OK;
}

int launch_apply_pstiff(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "apply_pstiff");
    const int c = s->cur;
    DF_DISPATCH(apply_pstiff_kernel, (const Vec4<T> *)s->b.posType[c], (const Vec4<T> *)s->b.velMass[c],
                (const T *)s->b.density, (const T *)s->b.pstiff, s->b.cellStart, (Vec3<T> *)s->b.v_star);
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
