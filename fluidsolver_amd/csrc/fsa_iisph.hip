// IISPH step for gfx950, fluid and rigid particles: the step the reference's MultiphaseSPHSolver runs by
// default (SolveIISPH, solver/MultiphaseSPHSolver.cpp:173-188 under /root/reference/particle_solver).
//
// Replaces NonPressureForceKernel_Multiphase (solver/sph_kernel_impl.cuh:1126-1306), IISPHFactorKernel,
// IISPHPredictDensityKernel, CalcDIJPJLKernel, CalcNewPressureKernel, CalcPressureForceKernel,
// IISPHUpdateKernel (:3663-4076) and the host-side mean of error[] that IISPHSolvePressure takes after
// copying the whole array to the host every round (solver/sph_solver.cu:1428-1436): here a two-level
// device reduction returns one number.  Deformable and cloth particles are not covered: the kernels of
// the step that act on them alone (ComputeTensionWithP_Kernel, UpdateSolidStateF_Kernel) are absent, and
// the TYPE_DEFORMABLE pair term of the non-pressure pass (:1234-1237) is left out.
//
// Every pass is a policy run by the kernels of fsa_pass.h.  Arithmetic follows the reference text
// operation for operation (explicit casts where its double literals promote a float expression).
#include <cstdlib>
#include <cstring>

#include "fsa_pass.h"

namespace fsa {

// KernelGradient_Cubic :42-54; `q > EPSILON` compares in double, both polynomial branches are double
// expressions rounded to T once.  r = sqrt(d2) with d2 = x*x + y*y + z*z, as xij.Norm() evaluates it.
template <typename T>
__device__ __forceinline__ void kernel_gradient_cubic(const DevParams<T> &P, T h, T x, T y, T z, T d2, T &gx, T &gy,
                                                      T &gz) {
    const T r = r_sqrt(d2);
    const T q = r / h;
    if (!(q < 2 && (double)q > FSA_EPSILON)) {
        gx = gy = gz = 0;
        return;
    }
    T df;
    if (q >= 1)
        df = (T)(0.5 * (double)(2 - q) * (double)(2 - q) * (-1));
    else
        df = (T)((double)(-2 * q) + 1.5 * (double)q * (double)q);
    gx = x * df * P.kernel_cubic_gradient / r;
    gy = y * df * P.kernel_cubic_gradient / r;
    gz = z * df * P.kernel_cubic_gradient / r;
}

// fast fp32 flavour of the same gradient, as a factor of xij
__device__ __forceinline__ float grad_scale_fast(float d2, float inv_h, float kgrad) {
    // gradW = xij * (df * kernel_cubic_gradient / r), KernelGradient_Cubic :42-54; 0 for q = 0 and q >= 2
    const float rinv = __builtin_amdgcn_rsqf(d2);
    const float q = d2 * rinv * inv_h;
    const float two_q = 2.0f - q;
    const float df = q >= 1.0f ? -0.5f * two_q * two_q : q * __builtin_fmaf(1.5f, q, -2.0f);
    return (d2 > 0.0f && q < 2.0f) ? df * kgrad * rinv : 0.0f;
}

// the three `case` labels that share one body in every IISPH cell function
__device__ __forceinline__ bool fluid_like(int t) {
    return t == FSA_TYPE_FLUID || t == FSA_TYPE_DEFORMABLE || t == FSA_TYPE_CLOTH;
}
// neighbour types with a `case` of their own: 1 = inside the support
__device__ __forceinline__ int iisph_reach(int tj) { return (fluid_like(tj) || tj == FSA_TYPE_RIGID) ? 1 : 0; }

template <typename T> __device__ __forceinline__ T dot3(T ax, T ay, T az, T bx, T by, T bz) {
    return ax * bx + ay * by + az * bz;
}

// d_ji of :3784 / :3925: gradW * -1 * mass_j / density_i / density_i * dt * dt * -1, per component
template <typename T> __device__ __forceinline__ T dji_term(const DevParams<T> &P, T g, T mass_j, T dens_i) {
    return g * (T)(-1) * mass_j / dens_i / dens_i * P.dt * P.dt * (T)(-1);
}

// ---- NonPressureForceCell_Multiphase :1126-1262 + NonPressureForceKernel_Multiphase :1264-1306 -------
template <typename T> struct NpfMultiPolicy {
    static __device__ __forceinline__ int reach(int) { return 1; }  // surface tension couples any two types
    struct Self {
        Vec4<T> v;  // velocity, mass
        T fx, fy, fz, dens, rd;
        int type, group;
        T h, h2, support, fac, sr6;
    };
    struct Nb { Vec4<T> v; T dens; int group; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID && type != FSA_TYPE_CLOTH; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &P, int index) {
        Vec3<T> vs{0, 0, 0};  // TYPE_RIGID, :1274-1277
        if (unpack_int(A.posType[index].w) == FSA_TYPE_CLOTH) {  // :1279-1282
            const Vec4<T> v = A.velMass[index];
            vs.x = v.x + P.gravity[0] * P.dt;
            vs.y = v.y + P.gravity[1] * P.dt;
            vs.z = v.z + P.gravity[2] * P.dt;
        }
        A.vStarOut[index] = vs;
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        S.v = A.velMass[index];
        S.fx = S.fy = S.fz = 0;
        S.dens = A.density[index];
        S.rd = A.restDensity[index];
        S.type = unpack_int(A.posType[index].w);
        S.group = A.groupSorted[index];
        S.h = (T)((double)P.sr * 0.5);  // :1142
        S.h2 = S.h * S.h;
        S.support = S.h * 2;
        // :1157, :1160; pow(float, int) taken as the double overload
        S.fac = (T)(32.0f / 3.141593 / pow((double)S.support, 9.0));
        S.sr6 = (T)(pow((double)P.sr, 6.0) / 64.0);
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        return Nb{A.velMass[j], A.density[j], A.groupSorted[j]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &A, const DevParams<T> &P, Self &S, bool, int tj,
                                                int j, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (d2 >= S.h2 * 4 || (double)d2 < FSA_EPSILON) return;  // :1167 (leaves the particle itself out)
        const T d = r_sqrt(d2);
        T gx, gy, gz;
        kernel_gradient_cubic(P, S.h, xx, xy, xz, d2, gx, gy, gz);
        const T vx = S.v.x - nb.v.x, vy = S.v.y - nb.v.y, vz = S.v.z - nb.v.z;
        const T volj = nb.v.w / nb.dens;
        if (S.type == FSA_TYPE_FLUID && tj == FSA_TYPE_FLUID) {  // xsph artificial viscosity, :1205-1208
            const T w = kernel_cubic_d2(P, S.h, d2);
            T c;
            if (S.group == 0 || nb.group == 0)
                c = P.viscosity * volj * (T)(-1) * w / P.dt;
            else
                c = (T)(0.5 * (double)volj * (-1) * (double)w / (double)P.dt);
            S.fx += vx * c;
            S.fy += vy * c;
            S.fz += vz * c;
        }
        if (S.group == nb.group) {  // surface tension, :1214-1232
            const T c = S.support - d;
            T sf_kernel;
            if (d < S.h)
                sf_kernel = 2 * c * c * c * d * d2 - S.sr6;
            else if (d < S.support)
                sf_kernel = c * c * c * d * d2;
            else
                sf_kernel = 0;
            sf_kernel *= S.fac;
            T tmp = P.surface_tension;
            if (S.group == 4) tmp *= 10;
            const T k = tmp * volj * sf_kernel / d * (T)(-1);
            S.fx += xx * k;
            S.fy += xy * k;
            S.fz += xz * k;
        }
        if (tj == FSA_TYPE_RIGID && nb.group == 0) {  // wall viscosity and friction, :1239-1259
            const T xv = dot3(vx, vy, vz, xx, xy, xz);
            if (xv < 0) {
                const Vec3<T> nj = A.normalById[A.uniqueId[j]];
                const T rd_j = A.restDensity[j];
                const T ndv = dot3(nj.x, nj.y, nj.z, vx, vy, vz);
                const T tx = vx - nj.x * ndv, ty = vy - nj.y * ndv, tz = vz - nj.z * ndv;
                const T visc = (T)((double)(P.boundary_visc * P.sr) * 88.5 / (double)S.dens * 0.25);
                const T pi = (T)((double)(visc * xv) / ((double)d2 + 0.01 * (double)S.h2));
                const T f0 = gx * pi * S.rd * rd_j, f1 = gy * pi * S.rd * rd_j, f2 = gz * pi * S.rd * rd_j;
                const T fdn = dot3(f0, f1, f2, nj.x, nj.y, nj.z);
                const T nx = nj.x * fdn, ny = nj.y * fdn, nz = nj.z * fdn;
                const T k = r_sqrt(nx * nx + ny * ny + nz * nz) * (T)(-1) * P.boundary_friction;
                S.fx += nx;
                S.fy += ny;
                S.fz += nz;
                S.fx += tx * k;
                S.fy += ty * k;
                S.fz += tz * k;
            }
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        S.fx += P.gravity[0];
        S.fy += P.gravity[1];
        S.fz += P.gravity[2];
        Vec3<T> vs;
        vs.x = S.v.x + S.fx * P.dt;
        vs.y = S.v.y + S.fy * P.dt;
        vs.z = S.v.z + S.fz * P.dt;
        A.vStarOut[index] = vs;
    }
};

// Fast fp32 flavour of the same pass (kernel variants 2 and 3): hardware reciprocal square root / reciprocal, fused
// multiply-adds, no double arithmetic per pair (about 1e-6 relative away from the exact flavour per term), and two
// packed 16-byte records per neighbour, side by side, instead of four gathers:
//   {x, y, z, type | group << 8}   {vx, vy, vz, mass / density}
__global__ __launch_bounds__(kBlock) void iisph_pack_npf_kernel(int n, const float4 *__restrict__ posType,
                                                                const float4 *__restrict__ velMass,
                                                                const float *__restrict__ density,
                                                                const int *__restrict__ groupSorted,
                                                                float4 *__restrict__ rec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = posType[i], v = velMass[i];
    rec[2 * (size_t)i] = make_float4(p.x, p.y, p.z, __int_as_float(unpack_int(p.w) | (groupSorted[i] << 8)));
    rec[2 * (size_t)i + 1] = make_float4(v.x, v.y, v.z, v.w / density[i]);
}

struct NpfMultiFastPolicy {
    using T = float;
    static constexpr bool kPosInNb = true;
    static constexpr bool kDeepReplay = true;  // (the row tiles are no faster here: 2.15 against 2.22 ms)
    static __device__ __forceinline__ int reach(int) { return 1; }  // surface tension couples any two types
    struct Self {
        float4 v;  // velocity, mass
        float fx, fy, fz, dens, rd;
        int type, group;
        float h, inv_h, support, fac, sr6, visc_dt, half_dt, tension;
    };
    struct Nb {
        float4 a, b;
        __device__ __forceinline__ float4 pos() const {
            return make_float4(a.x, a.y, a.z, __int_as_float(__float_as_int(a.w) & 255));
        }
    };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID && type != FSA_TYPE_CLOTH; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &P, int index) {
        NpfMultiPolicy<float>::skip(A, P, index);
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        S.v = A.velMass[index];
        S.fx = S.fy = S.fz = 0;
        S.dens = A.density[index];
        S.rd = A.restDensity[index];
        S.type = unpack_int(A.posType[index].w);
        S.group = A.groupSorted[index];
        S.h = (float)((double)P.sr * 0.5);
        S.inv_h = 1.0f / S.h;
        S.support = S.h * 2;
        S.fac = (float)(32.0f / 3.141593 / pow((double)S.support, 9.0));
        S.sr6 = (float)(pow((double)P.sr, 6.0) / 64.0);
        S.visc_dt = P.viscosity / P.dt;
        S.half_dt = 0.5f / P.dt;
        S.tension = S.group == 4 ? P.surface_tension * 10 : P.surface_tension;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        const float4 *r = A.aux4b + 2 * (size_t)j;
        return Nb{r[0], r[1]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &A, const DevParams<T> &P, Self &S, bool, int tj,
                                                int j, T xx, T xy, T xz, T d2, const float4 &, const Nb &nb) {
        if (d2 >= S.support * S.support || d2 < 1e-10f) return;  // (leaves the particle itself out)
        const float rinv = __builtin_amdgcn_rsqf(d2), d = d2 * rinv;
        const int gj = __float_as_int(nb.a.w) >> 8;
        const float vx = S.v.x - nb.b.x, vy = S.v.y - nb.b.y, vz = S.v.z - nb.b.z;
        const float volj = nb.b.w;
        if (S.type == FSA_TYPE_FLUID && tj == FSA_TYPE_FLUID) {  // xsph artificial viscosity
            const float w = cubic_fast(d2, S.inv_h, P.kernel_cubic);
            const float c = -((S.group == 0 || gj == 0) ? S.visc_dt : S.half_dt) * volj * w;
            S.fx = __builtin_fmaf(vx, c, S.fx);
            S.fy = __builtin_fmaf(vy, c, S.fy);
            S.fz = __builtin_fmaf(vz, c, S.fz);
        }
        if (S.group == gj) {  // surface tension
            const float c = S.support - d, c3 = c * c * c, d3 = d * d2;
            const float sf = d < S.h ? __builtin_fmaf(2.0f * c3, d3, -S.sr6) : c3 * d3;  // (d < support here)
            const float k = -(S.tension * volj) * (sf * S.fac) * rinv;
            S.fx = __builtin_fmaf(xx, k, S.fx);
            S.fy = __builtin_fmaf(xy, k, S.fy);
            S.fz = __builtin_fmaf(xz, k, S.fz);
        }
        if (tj == FSA_TYPE_RIGID && gj == 0) {  // wall viscosity and friction
            const float xv = __builtin_fmaf(vz, xz, __builtin_fmaf(vy, xy, vx * xx));
            if (xv < 0) {
                const float gs = grad_scale_fast(d2, S.inv_h, P.kernel_cubic_gradient);
                const float gx = xx * gs, gy = xy * gs, gz = xz * gs;
                const Vec3<float> nj = A.normalById[A.uniqueId[j]];
                const float rd_j = A.restDensity[j];
                const float ndv = __builtin_fmaf(nj.z, vz, __builtin_fmaf(nj.y, vy, nj.x * vx));
                const float tx = vx - nj.x * ndv, ty = vy - nj.y * ndv, tz = vz - nj.z * ndv;
                const float visc = P.boundary_visc * P.sr * 88.5f * 0.25f * __builtin_amdgcn_rcpf(S.dens);
                const float pi = visc * xv * __builtin_amdgcn_rcpf(__builtin_fmaf(0.01f, S.h * S.h, d2));
                const float m = pi * S.rd * rd_j;
                const float f0 = gx * m, f1 = gy * m, f2 = gz * m;
                const float fdn = __builtin_fmaf(f2, nj.z, __builtin_fmaf(f1, nj.y, f0 * nj.x));
                const float nx = nj.x * fdn, ny = nj.y * fdn, nz = nj.z * fdn;
                const float k = -__builtin_amdgcn_sqrtf(__builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx))) *
                                P.boundary_friction;
                S.fx += nx;
                S.fy += ny;
                S.fz += nz;
                S.fx = __builtin_fmaf(tx, k, S.fx);
                S.fy = __builtin_fmaf(ty, k, S.fy);
                S.fz = __builtin_fmaf(tz, k, S.fz);
            }
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        Vec3<float> vs;
        vs.x = __builtin_fmaf(S.fx + P.gravity[0], P.dt, S.v.x);
        vs.y = __builtin_fmaf(S.fy + P.gravity[1], P.dt, S.v.y);
        vs.z = __builtin_fmaf(S.fz + P.gravity[2], P.dt, S.v.z);
        A.vStarOut[index] = vs;
    }
};

// ---- IISPHFactorCell :3665-3715 + IISPHFactorKernel :3717-3749 ----------------------------------------
// FAST (fp32, variant 2): hardware square root / reciprocal square root and fused multiply-adds, no double
// arithmetic per pair; about 1e-6 relative away from the exact flavour per term
template <typename T, bool FAST = false> struct IisphFactorPolicy {
    static __device__ __forceinline__ int reach(int tj) { return iisph_reach(tj); }
    struct Self { T dens, sx, sy, sz, mass, rd; };
    struct Nb { T rd; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.dens = S.sx = S.sy = S.sz = 0;
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.restDensity[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        if constexpr (FAST) {
            const float inv_h = 2.0f / (float)sr;
            const float mj = fluid_like(tj) ? (float)S.mass : (tj == FSA_TYPE_RIGID ? (float)(nb.rd * S.rd) : 0.0f);
            const float gm = grad_scale_fast((float)d2, inv_h, (float)P.kernel_cubic_gradient) * mj;
            S.sx = (T)__builtin_fmaf((float)xx, gm, (float)S.sx);
            S.sy = (T)__builtin_fmaf((float)xy, gm, (float)S.sy);
            S.sz = (T)__builtin_fmaf((float)xz, gm, (float)S.sz);
            S.dens = (T)__builtin_fmaf(cubic_fast((float)d2, inv_h, (float)P.kernel_cubic), mj, (float)S.dens);
            return;
        }
        T gx, gy, gz;
        kernel_gradient_cubic(P, sr / 2, xx, xy, xz, d2, gx, gy, gz);
        T mass_j;
        if (fluid_like(tj))
            mass_j = S.mass;
        else if (tj == FSA_TYPE_RIGID)
            mass_j = nb.rd * S.rd;
        else
            return;
        S.sx += gx * mass_j;
        S.sy += gy * mass_j;
        S.sz += gz * mass_j;
        S.dens += kernel_cubic_d2(P, sr / 2, d2) * mass_j;
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        A.densityOut[index] = S.dens;
        const T k = P.dt * P.dt / S.dens / S.dens * (T)(-1);
        A.dii[index] = Vec3<T>{S.sx * k, S.sy * k, S.sz * k};
    }
};

// ---- IISPHPredictDensityCell :3751-3805 + IISPHPredictDensityKernel :3807-3833 ------------------------
template <typename T, bool FAST = false> struct IisphPredictPolicy {
    static constexpr bool kDeepReplay = FAST;
    static constexpr bool kRowReplay = FAST;  // (1.82 -> 1.69 ms at 16 M; the factor, force and non-pressure passes lose)
    static __device__ __forceinline__ int reach(int tj) { return iisph_reach(tj); }
    struct Self { T dens, aii, dens_i, mass, rd; Vec3<T> vs, dii; };
    struct Nb { T vx, vy, vz, rd; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.dens_i = A.density[index];
        S.dens = S.dens_i;
        S.aii = 0;
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
        S.vs = A.vStarIn[index];
        S.dii = A.dii[index];
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        if constexpr (FAST) {  // {v_star, restDensity} packed by iisph_pack_vstar_rd_kernel: one gather
            const Vec4<T> a = A.aux4[j];
            return Nb{a.x, a.y, a.z, a.w};
        }
        const Vec3<T> v = A.vStarIn[j];
        return Nb{v.x, v.y, v.z, A.restDensity[j]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        if constexpr (FAST) {
            const bool fl = fluid_like(tj);
            if (!fl && tj != FSA_TYPE_RIGID) return;
            const float gs = grad_scale_fast((float)d2, 2.0f / (float)sr, (float)P.kernel_cubic_gradient);
            const float gx = (float)xx * gs, gy = (float)xy * gs, gz = (float)xz * gs;
            const float mj = fl ? (float)S.mass : (float)(nb.rd * S.rd);
            // a rigid neighbour does not move: its v_star counts as zero (:3790)
            const float ux = (float)S.vs.x - (fl ? (float)nb.vx : 0.0f), uy = (float)S.vs.y - (fl ? (float)nb.vy : 0.0f),
                        uz = (float)S.vs.z - (fl ? (float)nb.vz : 0.0f);
            const float dv = __builtin_fmaf(ux, gx, __builtin_fmaf(uy, gy, uz * gz));
            S.dens = (T)__builtin_fmaf((float)P.dt * mj, dv, (float)S.dens);
            const float k = mj * (float)P.dt * (float)P.dt / ((float)S.dens_i * (float)S.dens_i);  // d_ji = gradW * k
            const float lx = (float)S.dii.x - gx * k, ly = (float)S.dii.y - gy * k, lz = (float)S.dii.z - gz * k;
            S.aii = (T)__builtin_fmaf(mj, __builtin_fmaf(lx, gx, __builtin_fmaf(ly, gy, lz * gz)), (float)S.aii);
            return;
        }
        T gx, gy, gz;
        kernel_gradient_cubic(P, sr / 2, xx, xy, xz, d2, gx, gy, gz);
        T mass_j;
        if (fluid_like(tj)) {
            mass_j = S.mass;
            S.dens += P.dt * mass_j * dot3(S.vs.x - nb.vx, S.vs.y - nb.vy, S.vs.z - nb.vz, gx, gy, gz);
        } else if (tj == FSA_TYPE_RIGID) {
            mass_j = nb.rd * S.rd;
            S.dens += P.dt * mass_j * dot3(S.vs.x, S.vs.y, S.vs.z, gx, gy, gz);
        } else {
            return;
        }
        const T jx = dji_term(P, gx, mass_j, S.dens_i), jy = dji_term(P, gy, mass_j, S.dens_i),
                jz = dji_term(P, gz, mass_j, S.dens_i);
        S.aii += mass_j * dot3(S.dii.x - jx, S.dii.y - jy, S.dii.z - jz, gx, gy, gz);
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        A.pressureOut[index] = A.pressure[index] / 2;  // :3813
        A.densityStar[index] = S.dens;
        A.aii[index] = S.aii;
    }
};

// ---- CalcDIJPJLCell :3835-3880 + CalcDIJPJLKernel :3882-3902 -----------------------------------------
template <typename T> struct IisphDijpjlPolicy {
    static __device__ __forceinline__ int reach(int tj) { return fluid_like(tj) ? 1 : 0; }  // `case TYPE_RIGID: break;`
    struct Self { T ax, ay, az, mass; };
    struct Nb { T dens, pressure; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.ax = S.ay = S.az = 0;
        S.mass = A.velMass[index].w;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.density[j], A.pressure[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2 || !fluid_like(tj)) return;
        T gx, gy, gz;
        kernel_gradient_cubic(P, sr / 2, xx, xy, xz, d2, gx, gy, gz);
        S.ax += gx * S.mass / nb.dens / nb.dens * nb.pressure * (T)(-1);
        S.ay += gy * S.mass / nb.dens / nb.dens * nb.pressure * (T)(-1);
        S.az += gz * S.mass / nb.dens / nb.dens * nb.pressure * (T)(-1);
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        const T k = P.dt * P.dt;
        A.dijpjl[index] = Vec3<T>{S.ax * k, S.ay * k, S.az * k};
    }
};

// ---- CalcNewPressureCell :3904-3959 + CalcNewPressureKernel :3961-3996 -------------------------------
// The reference kernel overwrites pressure[index] while other threads still read pressure[j] from the
// same array, so what a neighbour sees there depends on scheduling.  Here every particle reads the
// previous round's pressures (Jacobi): the new values go to a second array that the launcher swaps in.
template <typename T> struct IisphNewPressurePolicy {
    static __device__ __forceinline__ int reach(int tj) { return iisph_reach(tj); }
    struct Self { T lll, p_i, dens_i, mass, rd; Vec3<T> dijpjl; };
    struct Nb { Vec3<T> dii, dijpjl; T pressure, rd; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &, int index) {
        A.errorOut[index] = 0;  // :3965-3969
        A.pressureOut[index] = A.pressure[index];
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.lll = 0;
        S.p_i = A.pressure[index];
        S.dens_i = A.density[index];
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
        S.dijpjl = A.dijpjl[index];
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        return Nb{A.dii[j], A.dijpjl[j], A.pressure[j], A.restDensity[j]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        T gx, gy, gz;
        kernel_gradient_cubic(P, sr / 2, xx, xy, xz, d2, gx, gy, gz);
        if (fluid_like(tj)) {
            const T mass_j = S.mass;
            const T jx = dji_term(P, gx, mass_j, S.dens_i), jy = dji_term(P, gy, mass_j, S.dens_i),
                    jz = dji_term(P, gz, mass_j, S.dens_i);
            const T lx = S.dijpjl.x - nb.dii.x * nb.pressure - nb.dijpjl.x + jx * S.p_i;
            const T ly = S.dijpjl.y - nb.dii.y * nb.pressure - nb.dijpjl.y + jy * S.p_i;
            const T lz = S.dijpjl.z - nb.dii.z * nb.pressure - nb.dijpjl.z + jz * S.p_i;
            S.lll += mass_j * dot3(lx, ly, lz, gx, gy, gz);
        } else if (tj == FSA_TYPE_RIGID) {
            const T mass_j = nb.rd * S.rd;
            S.lll += mass_j * dot3(S.dijpjl.x, S.dijpjl.y, S.dijpjl.z, gx, gy, gz);
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        const T prev = S.p_i;
        T density_corr = A.densityStar[index] + S.lll;
        const T aii = A.aii[index];
        T np;
        if ((double)(aii < 0 ? -aii : aii) > 1e-8) {
            np = (T)(0.5 * (double)prev + 0.5 * (double)(S.rd - density_corr) / (double)aii);
            if (np < 0) np = 0;
        } else {
            np = 0;
        }
        A.pressureOut[index] = np;
        density_corr += prev * aii;
        A.errorOut[index] = density_corr;
    }
};

// ---- CalcPressureForceCell :3998-4046 + CalcPressureForceKernel :4048-4065 ---------------------------
template <typename T, bool FAST = false> struct IisphPressureForcePolicy {
    static constexpr bool kDeepReplay = FAST;
    static __device__ __forceinline__ int reach(int tj) { return iisph_reach(tj); }
    struct Self { T fx, fy, fz, p_i, dens_i, mass, rd; };
    struct Nb { T pressure, dens, rd; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        S.fx = S.fy = S.fz = 0;
        S.p_i = A.pressure[index];
        S.dens_i = A.density[index];
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) {
        if constexpr (FAST) {  // {p / rho^2 (0 for a rigid particle), restDensity} packed by iisph_pack_pterm_rd_kernel
            const float2 v = reinterpret_cast<const float2 *>(A.aux2)[j];
            return Nb{(T)v.x, (T)1, (T)v.y};
        }
        return Nb{A.pressure[j], A.density[j], A.restDensity[j]};
    }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        const T sr = P.sr, sr2 = sr * sr;
        if (d2 >= sr2) return;
        if constexpr (FAST) {
            const bool fl = fluid_like(tj);
            if (!fl && tj != FSA_TYPE_RIGID) return;
            const float gs = grad_scale_fast((float)d2, 2.0f / (float)sr, (float)P.kernel_cubic_gradient);
            const float wi = (float)S.p_i / ((float)S.dens_i * (float)S.dens_i);
            const float wj = fl ? (float)nb.pressure : 0.0f;  // p_j / rho_j^2, packed
            const float mj = fl ? (float)S.mass : (float)(nb.rd * S.rd);
            const float f = -gs * (float)S.mass * mj * (wi + wj);
            S.fx = (T)__builtin_fmaf((float)xx, f, (float)S.fx);
            S.fy = (T)__builtin_fmaf((float)xy, f, (float)S.fy);
            S.fz = (T)__builtin_fmaf((float)xz, f, (float)S.fz);
            return;
        }
        T gx, gy, gz;
        kernel_gradient_cubic(P, sr / 2, xx, xy, xz, d2, gx, gy, gz);
        if (fluid_like(tj)) {
            const T mass_j = S.mass;
            const T w = S.p_i / S.dens_i / S.dens_i + nb.pressure / nb.dens / nb.dens;
            S.fx += gx * (T)(-1) * mass_j * mass_j * w;
            S.fy += gy * (T)(-1) * mass_j * mass_j * w;
            S.fz += gz * (T)(-1) * mass_j * mass_j * w;
        } else if (tj == FSA_TYPE_RIGID) {
            const T mass_j = nb.rd * S.rd;
            const T w = S.p_i / S.dens_i / S.dens_i;
            S.fx += gx * (T)(-1) * S.mass * mass_j * w;
            S.fy += gy * (T)(-1) * S.mass * mass_j * w;
            S.fz += gz * (T)(-1) * S.mass * mass_j * w;
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        A.pressureForce[index] = Vec3<T>{S.fx, S.fy, S.fz};
    }
};

// EffectiveMassKernel :1722-1755: mass, rest density and colour of a non-rigid particle from its volume
// fractions (kept in uniqueId order, like colour; the reference carries the table through the sort).  The
// printf on a vanishing beta is left out; effective_mass[] of the reference equals mass[].
template <typename T>
__global__ __launch_bounds__(kBlock) void effective_mass_kernel(DevParams<T> P, int n, T spacing, int typenum,
                                                                const T *__restrict__ densArr,
                                                                const T *__restrict__ vFrac,
                                                                const int *__restrict__ uniqueId,
                                                                const Vec4<T> *__restrict__ posType,
                                                                Vec4<T> *__restrict__ velMass,
                                                                T *__restrict__ restDensity, T *__restrict__ colorById) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    if (unpack_int(posType[index].w) == FSA_TYPE_RIGID) return;
    const int uid = uniqueId[index];
    const T *vol_frac = vFrac + (size_t)uid * (size_t)typenum;
    const T rd = restDensity[index];
    T rest_density = 0;
    for (int k = 0; k < typenum; k++) {
        const T mass_frac = vol_frac[k] * densArr[k] / rd;
        if ((double)mass_frac < FSA_EPSILON) continue;
        rest_density += vol_frac[k] * densArr[k];
    }
    Vec4<T> v = velMass[index];
    v.w = rest_density * spacing * spacing * spacing;
    velMass[index] = v;
    restDensity[index] = rest_density;
    T *c = colorById + 4 * (size_t)uid;
    for (int k = 0; k < 3; k++) c[k] = k < typenum ? vol_frac[k] : (T)0;
    c[3] = 1;
    (void)P;
}

// IISPHUpdateKernel :4067-4076
template <typename T>
__global__ __launch_bounds__(kBlock) void iisph_integrate_kernel(DevParams<T> P, int n, Vec4<T> *__restrict__ posType,
                                                                 Vec4<T> *__restrict__ velMass,
                                                                 const Vec3<T> *__restrict__ v_star,
                                                                 const Vec3<T> *__restrict__ pressureForce) {
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    Vec4<T> p = posType[index];
    if (unpack_int(p.w) == FSA_TYPE_RIGID) return;
    Vec4<T> v = velMass[index];
    const Vec3<T> vs = v_star[index], pf = pressureForce[index];
    v.x = vs.x + pf.x * P.dt / v.w;
    v.y = vs.y + pf.y * P.dt / v.w;
    v.z = vs.z + pf.z * P.dt / v.w;
    p.x += v.x * P.dt;
    p.y += v.y * P.dt;
    p.z += v.z * P.dt;
    velMass[index] = v;
    posType[index] = p;
}

// The sort of the IISPH step also carries pressure[] (sorted_pressure, solver/sph_kernel_impl.cuh:175-198
// and MultiphaseSPHSolver.cpp:99); group[] is kept in uniqueId order and gathered for the sorted positions.
template <typename T>
__global__ __launch_bounds__(kBlock) void iisph_carry_kernel(int n, const int *__restrict__ sortedIndex,
                                                             const int *__restrict__ uniqueIdSorted,
                                                             const int *__restrict__ groupById,
                                                             const T *__restrict__ pressureIn,
                                                             T *__restrict__ pressureOut,
                                                             int *__restrict__ groupSorted) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    pressureOut[p] = pressureIn[sortedIndex[p]];
    groupSorted[p] = groupById[uniqueIdSorted[p]];
}

// sum over error[] in double: per-block sums, then one block over those
template <typename T>
__global__ __launch_bounds__(kBlock) void block_sum_kernel(int n, const T *__restrict__ in, double *__restrict__ out) {
    __shared__ double part[kBlock / 64];
    double acc = 0.0;
    strided_fold(in, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, n, acc, [](double &a, T v) { a += (double)v; });
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) acc += part[w];
        out[blockIdx.x] = acc;
    }
}

// ---- fast fp32 flavour of the two passes of a relaxation round (variant 2) --------------------------------
// Same sums over the same neighbours, arranged for speed: hardware reciprocal square root and fused
// multiply-adds, no double arithmetic per pair, and everything a pass needs of a neighbour gathered from one
// packed record that a small kernel writes first:
//   CalcDIJPJL        reads  {x_j, p_j / rho_j^2}
//   CalcNewPressure   reads  {d_jj p_j + dijpjl_j, restDensity_j}
// About 1e-6 relative away from the exact flavour per term.
// {x, y, z, p / rho^2}: all CalcDIJPJL reads of a neighbour.  The last entry is 0 for a particle that does not
// count (rigid ones, whose density is never set): its term then adds an exact zero.
__global__ __launch_bounds__(kBlock) void iisph_pack_pterm_kernel(int n, const float4 *__restrict__ posType,
                                                                  const float *__restrict__ pressure,
                                                                  const float *__restrict__ density,
                                                                  float4 *__restrict__ rec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = posType[i];
    const float d = density[i];
    rec[i] = make_float4(p.x, p.y, p.z, fluid_like(unpack_int(p.w)) ? pressure[i] / d / d : 0.0f);
}

__global__ __launch_bounds__(kBlock) void iisph_pack_cvec_kernel(int n, const Vec3<float> *__restrict__ dii,
                                                                 const Vec3<float> *__restrict__ dijpjl,
                                                                 const float *__restrict__ pressure,
                                                                 const float *__restrict__ restDensity,
                                                                 float4 *__restrict__ cvec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Vec3<float> a = dii[i], b = dijpjl[i];
    const float p = pressure[i];
    cvec[i] = make_float4(__builtin_fmaf(a.x, p, b.x), __builtin_fmaf(a.y, p, b.y), __builtin_fmaf(a.z, p, b.z),
                          restDensity[i]);
}

struct IisphDijpjlFastPolicy {
    using T = float;
    // (what the search admits; a record that does not count carries a zero instead of being skipped)
    static __device__ __forceinline__ int reach(int tj) { return fluid_like(tj) ? 1 : 0; }
    struct Self { T ax, ay, az, mass, inv_h; };
    static constexpr bool kPosInNb = true;
    static constexpr bool kDeepReplay = true;  // (one gather per neighbour: the LDS row replay is slower here, 0.96 -> 1.07 ms)
    struct Nb {
        Vec4<T> r;  // {x, y, z, pterm or 0}
        __device__ __forceinline__ Vec4<T> pos() const { return make_float4(r.x, r.y, r.z, 0.0f); }  // type bits: fluid
    };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &, const DevParams<T> &, int) {}
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        S.ax = S.ay = S.az = 0;
        S.mass = A.velMass[index].w;
        S.inv_h = 2.0f / P.sr;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.aux4[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (d2 >= P.sr * P.sr) return;
        const T w = grad_scale_fast(d2, S.inv_h, P.kernel_cubic_gradient) * nb.r.w;
        S.ax = __builtin_fmaf(xx, w, S.ax);
        S.ay = __builtin_fmaf(xy, w, S.ay);
        S.az = __builtin_fmaf(xz, w, S.az);
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        const T k = -(S.mass * P.dt * P.dt);
        const Vec3<T> d{S.ax * k, S.ay * k, S.az * k};
        A.dijpjl[index] = d;
        if (A.recOut) {  // what CalcNewPressure reads of this particle (as iisph_pack_cvec_kernel writes it)
            const Vec3<T> a = A.dii[index];
            const T p = A.pressure[index];
            A.recOut[index] = make_float4(__builtin_fmaf(a.x, p, d.x), __builtin_fmaf(a.y, p, d.y),
                                          __builtin_fmaf(a.z, p, d.z), A.restDensity[index]);
        }
    }
};

struct IisphNewPressureFastPolicy {
    using T = float;
    static constexpr bool kDeepReplay = true;
    static constexpr bool kRowReplay = true;  // two gathers per neighbour otherwise: 1.96 -> 1.77 ms from the row tiles
    static __device__ __forceinline__ int reach(int tj) { return iisph_reach(tj); }
    struct Self { T lll, p_i, mass, rd, k1, inv_h, dens; Vec3<T> dijpjl; };
    struct Nb { Vec4<T> c; };
    static __device__ __forceinline__ bool wants(int type) { return type != FSA_TYPE_RIGID; }
    static __device__ __forceinline__ void skip(const PassArrays<T> &A, const DevParams<T> &, int index) {
        A.errorOut[index] = 0;
        A.pressureOut[index] = A.pressure[index];
    }
    static __device__ __forceinline__ void init(const PassArrays<T> &A, const DevParams<T> &P, int index, Self &S) {
        S.lll = 0;
        S.p_i = A.pressure[index];
        const T dens = A.density[index];
        S.dens = dens;
        S.mass = A.velMass[index].w;
        S.rd = A.restDensity[index];
        S.dijpjl = A.dijpjl[index];
        S.k1 = S.mass * P.dt * P.dt / (dens * dens) * S.p_i;  // d_ji p_i = gradW * k1 for a fluid neighbour
        S.inv_h = 2.0f / P.sr;
    }
    static __device__ __forceinline__ Nb load(const PassArrays<T> &A, int j) { return Nb{A.aux4b[j]}; }
    static __device__ __forceinline__ void pair(const PassArrays<T> &, const DevParams<T> &P, Self &S, bool, int tj,
                                                int, T xx, T xy, T xz, T d2, const Vec4<T> &, const Nb &nb) {
        if (d2 >= P.sr * P.sr) return;
        const T gs = grad_scale_fast(d2, S.inv_h, P.kernel_cubic_gradient);
        const T gx = xx * gs, gy = xy * gs, gz = xz * gs;
        if (fluid_like(tj)) {
            const T lx = __builtin_fmaf(gx, S.k1, S.dijpjl.x - nb.c.x);
            const T ly = __builtin_fmaf(gy, S.k1, S.dijpjl.y - nb.c.y);
            const T lz = __builtin_fmaf(gz, S.k1, S.dijpjl.z - nb.c.z);
            S.lll = __builtin_fmaf(S.mass, __builtin_fmaf(lx, gx, __builtin_fmaf(ly, gy, lz * gz)), S.lll);
        } else if (tj == FSA_TYPE_RIGID) {
            const T d = __builtin_fmaf(S.dijpjl.x, gx, __builtin_fmaf(S.dijpjl.y, gy, S.dijpjl.z * gz));
            S.lll = __builtin_fmaf(nb.c.w * S.rd, d, S.lll);
        }
    }
    static __device__ __forceinline__ void finish(const PassArrays<T> &A, const DevParams<T> &, int index, Self &S) {
        const T prev = S.p_i;
        T density_corr = A.densityStar[index] + S.lll;
        const T aii = A.aii[index];
        T np = 0;
        if (fabsf(aii) > 1e-8f) {
            np = 0.5f * prev + 0.5f * (S.rd - density_corr) / aii;
            if (np < 0) np = 0;
        }
        A.pressureOut[index] = np;
        density_corr += prev * aii;
        A.errorOut[index] = density_corr;
        if (A.recOut) {  // what CalcDIJPJL reads of this particle in the next round (as iisph_pack_pterm_kernel writes it)
            const Vec4<T> p = A.posType[index];
            A.recOut[index] = make_float4(p.x, p.y, p.z, fluid_like(unpack_int(p.w)) ? np / S.dens / S.dens : 0.0f);
        }
    }
};

// packed neighbour records of the fast predict and pressure-force passes
__global__ __launch_bounds__(kBlock) void iisph_pack_vstar_rd_kernel(int n, const Vec3<float> *__restrict__ v_star,
                                                                     const float *__restrict__ restDensity,
                                                                     float4 *__restrict__ rec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Vec3<float> v = v_star[i];
    rec[i] = make_float4(v.x, v.y, v.z, restDensity[i]);
}
__global__ __launch_bounds__(kBlock) void iisph_pack_pterm_rd_kernel(int n, const float4 *__restrict__ posType,
                                                                     const float *__restrict__ pressure,
                                                                     const float *__restrict__ density,
                                                                     const float *__restrict__ restDensity,
                                                                     float2 *__restrict__ rec) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float d = density[i];
    rec[i] = make_float2(fluid_like(unpack_int(posType[i].w)) ? pressure[i] / d / d : 0.0f, restDensity[i]);
}

static bool use_fast(const Solver *s) { return use_tiles(s) && s->variant >= 2; }

#define IISPH_PASS(PolicyName, tag)                                              \
    do {                                                                         \
        if (s->n == 0) return FSA_OK;                                            \
        if (use_tiles(s)) return launch_tuned<PolicyName<float>>(s, tag, true);  \
        DF_GENERIC(PolicyName<T>, tag);                                          \
        return FSA_OK;                                                           \
    } while (0)

int launch_effective_mass(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "effective_mass");
    const int c = s->cur;
    DF_DISPATCH(effective_mass_kernel, (T)s->hp.spacing, s->typenum, (const T *)s->b.densArr, (const T *)s->b.vFrac,
                s->b.uniqueId[c], (const Vec4<T> *)s->b.posType[c], (Vec4<T> *)s->b.velMass[c],
                (T *)s->b.restDensity[c], (T *)s->b.color);
    return FSA_OK;
}

int launch_npf_multiphase(Solver *s) {
    if (s->n > 0 && use_fast(s)) {
        const int rc = ensure_lists(s);
        if (rc) return rc;
        ScopedTimer t(s, "npf_multiphase");
        s->hits_valid = false;  // the records live in the buffer of the WCSPH hit record, which this step does not use
        hipLaunchKernelGGL(iisph_pack_npf_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                           (const float4 *)s->b.posType[s->cur], (const float4 *)s->b.velMass[s->cur],
                           (const float *)s->b.density, s->b.groupSorted, (float4 *)s->b.hitWords);
        PassArrays<float> A = arrays_of<float>(s);
        A.aux4b = (const float4 *)s->b.hitWords;
        return run_tiled<NpfMultiFastPolicy>(s, A, true);
    }
    IISPH_PASS(NpfMultiPolicy, "npf_multiphase");
}
int launch_iisph_factor(Solver *s) {
    if (s->n > 0 && use_fast(s)) return launch_tuned<IisphFactorPolicy<float, true>>(s, "iisph_factor", true);
    IISPH_PASS(IisphFactorPolicy, "iisph_factor");
}
int launch_iisph_predict_density(Solver *s) {
    if (s->n > 0 && use_fast(s)) {
        {
            ScopedTimer t(s, "iisph_pack_vstar_rd");
            hipLaunchKernelGGL(iisph_pack_vstar_rd_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               (const Vec3<float> *)s->b.v_star, (const float *)s->b.restDensity[s->cur],
                               (float4 *)s->b.aux4);
            FSA_HIP(hipGetLastError());
        }
        return launch_tuned<IisphPredictPolicy<float, true>>(s, "iisph_predict", true);
    }
    IISPH_PASS(IisphPredictPolicy, "iisph_predict");
}
// The two passes of a pressure round in the fast flavour read one packed record per neighbour:
//   CalcDIJPJL       {x, y, z, p / rho^2}                   in aux4
//   CalcNewPressure  {d_jj p_j + dijpjl_j, restDensity_j}   in the second record buffer
// Called on their own they pack the record first.  Inside fsa_iisph_solve_pressure (Solver::iisph_chain) each pass
// writes, for its own particle, the record the next pass reads, so the two packing kernels run once per solve
// instead of once per round (the particles the passes skip keep the records of that first packing: nothing about
// them changes during the rounds).
static void *second_records(Solver *s) { return s->b.hitWords; }  // (the WCSPH hit record is idle in this step)

int launch_iisph_dijpjl(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_fast(s)) {
        if (s->iisph_chain != 2) {
            ScopedTimer t(s, "iisph_pack_pterm");
            hipLaunchKernelGGL(iisph_pack_pterm_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               (const float4 *)s->b.posType[s->cur], (const float *)s->b.pressure,
                               (const float *)s->b.density, (float4 *)s->b.aux4);
            FSA_HIP(hipGetLastError());
        }
        const int rc = ensure_lists(s);
        if (rc) return rc;
        ScopedTimer t(s, "iisph_dijpjl");
        PassArrays<float> A = arrays_of<float>(s);
        if (s->iisph_chain) {
            if (s->iisph_chain == 1) {  // first round: the records of the particles the pass skips
                s->hits_valid = false;
                hipLaunchKernelGGL(iisph_pack_cvec_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                                   (const Vec3<float> *)s->b.dii, (const Vec3<float> *)s->b.dijpjl,
                                   (const float *)s->b.pressure, (const float *)s->b.restDensity[s->cur],
                                   (float4 *)second_records(s));
            }
            A.recOut = (float4 *)second_records(s);
        }
        return run_tiled<IisphDijpjlFastPolicy>(s, A, true);
    }
    IISPH_PASS(IisphDijpjlPolicy, "iisph_dijpjl");
}
int launch_iisph_pressure_force(Solver *s) {
    if (s->n > 0 && use_fast(s)) {
        {
            ScopedTimer t(s, "iisph_pack_pterm_rd");
            s->pd_valid = false;  // the buffer is shared with the WCSPH force pass
            hipLaunchKernelGGL(iisph_pack_pterm_rd_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               (const float4 *)s->b.posType[s->cur], (const float *)s->b.pressure,
                               (const float *)s->b.density, (const float *)s->b.restDensity[s->cur],
                               (float2 *)s->b.pd);
            FSA_HIP(hipGetLastError());
        }
        return launch_tuned<IisphPressureForcePolicy<float, true>>(s, "iisph_pforce", true);
    }
    IISPH_PASS(IisphPressureForcePolicy, "iisph_pforce");
}

int launch_iisph_new_pressure(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (use_tiles(s)) {
        const int rc = ensure_lists(s);
        if (rc) return rc;
    }
    {
        ScopedTimer t(s, "iisph_new_pressure");
        int rc = FSA_OK;
        if (use_fast(s)) {
            PassArrays<float> A = arrays_of<float>(s);
            A.pressureOut = (float *)s->b.pressure_alt;
            if (s->iisph_chain) {  // the dijpjl pass before has written the records; this one writes the next round's
                A.aux4b = (const float4 *)second_records(s);
                A.recOut = (float4 *)s->b.aux4;
            } else {
                hipLaunchKernelGGL(iisph_pack_cvec_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                                   (const Vec3<float> *)s->b.dii, (const Vec3<float> *)s->b.dijpjl,
                                   (const float *)s->b.pressure, (const float *)s->b.restDensity[s->cur],
                                   (float4 *)s->b.aux4);
                A.aux4b = (const float4 *)s->b.aux4;
            }
            rc = run_tiled<IisphNewPressureFastPolicy>(s, A, true);
        } else if (use_tiles(s)) {
            PassArrays<float> A = arrays_of<float>(s);
            A.pressureOut = (float *)s->b.pressure_alt;
            rc = run_tiled<IisphNewPressurePolicy<float>>(s, A, true);
        } else if (s->precision == FSA_PRECISION_F32) {
            PassArrays<float> A = arrays_of<float>(s);
            A.pressureOut = (float *)s->b.pressure_alt;
            hipLaunchKernelGGL((generic_pass_kernel<float, IisphNewPressurePolicy<float>>), dim3(blocks_for(s->n)),
                               dim3(kBlock), 0, s->stream, s->pf, s->n, A, s->b.cellStart);
        } else {
            PassArrays<double> A = arrays_of<double>(s);
            A.pressureOut = (double *)s->b.pressure_alt;
            hipLaunchKernelGGL((generic_pass_kernel<double, IisphNewPressurePolicy<double>>), dim3(blocks_for(s->n)),
                               dim3(kBlock), 0, s->stream, s->pd, s->n, A, s->b.cellStart);
        }
        if (rc) return rc;
        FSA_HIP(hipGetLastError());
    }
    std::swap(s->b.pressure, s->b.pressure_alt);
    return FSA_OK;
}

int launch_iisph_integrate(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "iisph_integrate");
    s->lists_valid = false;  // positions move
    s->hits_valid = false;
    const int c = s->cur;
    DF_DISPATCH(iisph_integrate_kernel, (Vec4<T> *)s->b.posType[c], (Vec4<T> *)s->b.velMass[c],
                (const Vec3<T> *)s->b.v_star, (const Vec3<T> *)s->b.pressureForce);
    return FSA_OK;
}

// `uid_buffer`: which copy of uniqueId holds the sorted ids (the reorder calls this before the flip)
int launch_iisph_carry(Solver *s, int uid_buffer) {
    if (s->n == 0) return FSA_OK;
    {
        ScopedTimer t(s, "iisph_carry");
        if (s->precision == FSA_PRECISION_F32)
            hipLaunchKernelGGL(iisph_carry_kernel<float>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               s->b.sortedIndex, s->b.uniqueId[uid_buffer], s->b.group, (const float *)s->b.pressure,
                               (float *)s->b.pressure_alt, s->b.groupSorted);
        else
            hipLaunchKernelGGL(iisph_carry_kernel<double>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                               s->b.sortedIndex, s->b.uniqueId[uid_buffer], s->b.group, (const double *)s->b.pressure,
                               (double *)s->b.pressure_alt, s->b.groupSorted);
        FSA_HIP(hipGetLastError());
    }
    std::swap(s->b.pressure, s->b.pressure_alt);
    return FSA_OK;
}

constexpr int kSumBlocks = 1024;

// Mean of error[] over all particles (IISPHSolvePressure, solver/sph_solver.cu:1428-1436, where the array is
// copied to the host and added up in one float in index order): reduced on the device in double over a fixed
// tree, the mean rounded once to the solver's precision.
int launch_mean_error(Solver *s, double *out) {
    *out = 0.0;
    if (s->n == 0) return FSA_OK;
    const int n = s->n;
    {
        ScopedTimer t(s, "mean_error");
        const int nb = std::min(kSumBlocks, blocks_for(n));
        double *tmp = (double *)s->b.reduce_tmp;
        if (s->precision == FSA_PRECISION_F32)
            hipLaunchKernelGGL(block_sum_kernel<float>, dim3(nb), dim3(kBlock), 0, s->stream, n,
                               (const float *)s->b.error, tmp);
        else
            hipLaunchKernelGGL(block_sum_kernel<double>, dim3(nb), dim3(kBlock), 0, s->stream, n,
                               (const double *)s->b.error, tmp);
        hipLaunchKernelGGL(block_sum_kernel<double>, dim3(1), dim3(kBlock), 0, s->stream, nb, (const double *)tmp,
                           tmp + kSumBlocks);
        FSA_HIP(hipGetLastError());
    }
    double sum = 0.0;
    FSA_HIP(hipMemcpyAsync(&sum, (const double *)s->b.reduce_tmp + kSumBlocks, sizeof(sum), hipMemcpyDeviceToHost,
                           s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    const double mean = sum / (double)n;
    *out = s->precision == FSA_PRECISION_F32 ? (double)(float)mean : mean;
    return FSA_OK;
}

// ---- the pressure rounds with the stop test on the device -------------------------------------------------
// State behind the partial sums in reduce_tmp: {last mean, mean} as doubles, then {rounds done, stop} as ints.
constexpr int kRoundState = kSumBlocks + 8;

template <typename T>
__global__ __launch_bounds__(kBlock) void halted_block_sum_kernel(int n, const T *__restrict__ in, double *__restrict__ out,
                                                                  const int *__restrict__ halt) {
    if (*halt) return;
    __shared__ double part[kBlock / 64];
    double acc = 0.0;
    strided_fold(in, blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, n, acc, [](double &a, T v) { a += (double)v; });
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) acc += part[w];
        out[blockIdx.x] = acc;
    }
}

// One block: folds the partial sums in the order of block_sum_kernel, takes the mean, applies the two tests of
// solver/sph_solver.cu:1440 (|mean| < 1e-6, |mean - last mean| < 1e-6, on values of the solver's precision).
template <typename T>
__global__ __launch_bounds__(kBlock) void iisph_round_end_kernel(int n, int nb, double *__restrict__ tmp) {
    double *st = tmp + kRoundState;
    int *ist = reinterpret_cast<int *>(st + 2);
    if (ist[1]) return;
    __shared__ double part[kBlock / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += kBlock) acc += tmp[i];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) acc += part[w];
        const double m = acc / (double)n;
        const double mean = sizeof(T) == 4 ? (double)(float)m : m, last = st[1];
        const double a = fabs(mean);
        const double b = sizeof(T) == 4 ? (double)fabsf((float)mean - (float)last) : fabs(mean - last);
        st[0] = last;
        st[1] = mean;
        ist[0] += 1;
        if (a < 1e-6 || b < 1e-6) ist[1] = 1;
    }
}

int launch_iisph_rounds_begin(Solver *s) {
    double *tmp = (double *)s->b.reduce_tmp;
    FSA_HIP(hipMemsetAsync(tmp + kRoundState, 0, 4 * sizeof(double), s->stream));
    s->pf.halt = s->pd.halt = reinterpret_cast<const int *>(tmp + kRoundState + 2) + 1;
    return FSA_OK;
}

int launch_iisph_round_end(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "mean_error");
    const int n = s->n, nb = std::min(kSumBlocks, blocks_for(n));
    double *tmp = (double *)s->b.reduce_tmp;
    if (s->precision == FSA_PRECISION_F32) {
        hipLaunchKernelGGL(halted_block_sum_kernel<float>, dim3(nb), dim3(kBlock), 0, s->stream, n,
                           (const float *)s->b.error, tmp, s->pf.halt);
        hipLaunchKernelGGL(iisph_round_end_kernel<float>, dim3(1), dim3(kBlock), 0, s->stream, n, nb, tmp);
    } else {
        hipLaunchKernelGGL(halted_block_sum_kernel<double>, dim3(nb), dim3(kBlock), 0, s->stream, n,
                           (const double *)s->b.error, tmp, s->pf.halt);
        hipLaunchKernelGGL(iisph_round_end_kernel<double>, dim3(1), dim3(kBlock), 0, s->stream, n, nb, tmp);
    }
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_iisph_rounds_finish(Solver *s, int launched, int *done, double *mean) {
    s->pf.halt = s->pd.halt = nullptr;
    double st[3] = {0, 0, 0};
    FSA_HIP(hipMemcpyAsync(st, (const double *)s->b.reduce_tmp + kRoundState, sizeof(st), hipMemcpyDeviceToHost, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    int ist[2];
    std::memcpy(ist, &st[2], sizeof(ist));
    *done = s->n == 0 ? 0 : ist[0];
    *mean = st[1];
    // every queued round swapped the two pressure arrays on the host; the ones that did not run must not count
    if (s->n > 0 && ((launched - ist[0]) & 1)) std::swap(s->b.pressure, s->b.pressure_alt);
    return FSA_OK;
}

}  // namespace fsa
