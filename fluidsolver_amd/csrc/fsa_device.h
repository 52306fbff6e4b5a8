// Shared device-side definitions of the gfx950 particle-solver kernels.
//
// Arithmetic rule for every kernel in this directory: one IEEE rounding per
// operation, in the operation order of the reference text (the translation
// units are built with -ffp-contract=off, correctly rounded division and
// sqrt).  Where the reference's unsuffixed literals promote a float
// expression to double, the promotion is written out with explicit casts; for
// T = double the casts are no-ops.  Citations are relative to
// /root/reference/particle_solver.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/fluidsolver_amd.h"  // FSA_TYPE_*, FSA_GRID_UNDEF, FSA_BOUNDARY_*

#define FSA_EPSILON 1e-10  // solver/cuda_common.cuh:10 (a double literal)

namespace fsa {

constexpr int kBlock = 256;  // 4 wavefronts of 64

template <typename T> struct Vec4T;
template <> struct Vec4T<float> { using type = float4; };
template <> struct Vec4T<double> { using type = double4; };
template <typename T> using Vec4 = typename Vec4T<T>::type;

// tight 3-vector, the reference's cfloat3 layout (catpaw/geometry.h:58-60)
template <typename T> struct Vec3 { T x, y, z; };

// Particle type travels in the w lane of the position vector as raw bits.
__device__ __forceinline__ float pack_int(int v, float) { return __int_as_float(v); }
__device__ __forceinline__ double pack_int(int v, double) { return __longlong_as_double((long long)v); }
__device__ __forceinline__ int unpack_int(float w) { return __float_as_int(w); }
__device__ __forceinline__ int unpack_int(double w) { return (int)__double_as_longlong(w); }

// Hot-path fields of SimParam_SPH (solver/sph_solver.cuh:10-77) at the solver's precision.
template <typename T> struct DevParams {
    T dx;
    T gmin[3];
    T gmax[3];
    int gres[3];   // global grid resolution
    T gravity[3];
    T dt;
    T viscosity;
    T sr;          // smoothradius
    T kspikydiff;
    T kernel_cubic;
    T kernel_cubic_gradient;
    T kpoly6;
    T surface_tension;
    T boundary_friction;
    T pressureK;
    T boundary_visc;
    T accel_limit;
    int boundary_mode;
    int z0;        // first global cell layer held by this solver
    int nz;        // number of layers held (ghost layers included)
    int num_cells; // gres[0] * gres[1] * nz
    // z-slab runs: global layers [own_lo, own_hi) whose particles need forces (density: one layer more on
    // each side); what else the solver holds is halo, whose results nobody reads.  Default: everything.
    int own_lo, own_hi;
    // z-slab groups: the global layers next to the cuts, [edge[0], edge[1]) on the left and [edge[2], edge[3]) on
    // the right (empty where there is no neighbour): a particle that may have to go out to a neighbour after the
    // integration was in one of them at the last sort.  Their forces are computed and their records sent first.
    int edge[4];
    // fp32 solvers: the row table of the tiled passes for the current sorted order (see fsa_tile.h), else null
    const int *rows;
    const int *big;   // [0] = number of blocks with large tiles, then their ids (fsa_tile.h); rebuilt with `rows`
    // z-slab groups that run without host round trips: the number of particles lives in device memory (it changes
    // with every exchange) and the kernels of the step read it from here instead of taking their argument, which is
    // then only an upper bound the grid was sized for.  Null otherwise.
    const int *count;
    // Iterations whose stop test lives on the device (the IISPH pressure rounds): the rounds are all queued at
    // once, and a pass kernel returns at the start when this word is set (null: never).
    const int *halt;
};

template <typename T> __device__ __forceinline__ int live_count(const DevParams<T> &P, int n) {
    return P.count ? *P.count : n;
}

__device__ __forceinline__ float r_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double r_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float r_floor(float x) { return floorf(x); }
__device__ __forceinline__ double r_floor(double x) { return floor(x); }
__device__ __forceinline__ float r_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double r_pow(double x, double y) { return pow(x, y); }
// x^7 of the pressure equation (powf(density / restDensity, 7), solver/sph_kernel_impl.cuh:263).  No powf is correctly
// rounded everywhere and no two libraries agree on every argument (ROCm's differs from the correctly rounded value
// in the last bit for about one argument in a thousand); the fp32 solvers take the product in double, four roundings
// of 2^-53, and round once to float: the correctly rounded power but for ties closer than that, and cheaper than powf.
// Same special cases (odd power keeps the sign, NaN and infinities pass through, overflow gives infinity).
__device__ __forceinline__ float r_pow7(float x) {
    const double d = (double)x, d2 = d * d, d4 = d2 * d2;
    return (float)(d4 * d2 * d);
}
// (fp64 solvers, which the reference does not have: the same product, so that a run can be compared with the fp64
// oracle's bit for bit; within four roundings of 2^-53 of pow(x, 7.0))
__device__ __forceinline__ double r_pow7(double x) {
    const double d2 = x * x, d4 = d2 * d2;
    return d4 * d2 * x;
}

// float -> int, saturating, NaN -> 0 (what cvt.rzi.s32.f32 gives the reference)
template <typename T> __device__ __forceinline__ int to_int_sat(T v) {
    if (v != v) return 0;
    if (v >= (T)2147483647.0) return 2147483647;
    if (v <= (T)-2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

// calcGridPos, solver/sph_kernel_impl.cuh:74-80: global cell coordinates.
template <typename T>
__device__ __forceinline__ void calc_grid_pos(const DevParams<T> &P, T x, T y, T z, int &cx, int &cy, int &cz) {
    cx = to_int_sat(r_floor((x - P.gmin[0]) / P.dx));
    cy = to_int_sat(r_floor((y - P.gmin[1]) / P.dx));
    cz = to_int_sat(r_floor((z - P.gmin[2]) / P.dx));
}

// calcGridHash, solver/sph_kernel_impl.cuh:58-72, on the layers this solver holds:
// hash = y * gx * nz + (z - z0) * gx + x, GRID_UNDEF outside.
template <typename T>
__device__ __forceinline__ unsigned calc_grid_hash(const DevParams<T> &P, int cx, int cy, int cz) {
    int lz = cz - P.z0;
    if (cx < 0 || cx >= P.gres[0] || cy < 0 || cy >= P.gres[1] || cz < 0 || cz >= P.gres[2] || lz < 0 ||
        lz >= P.nz)
        return FSA_GRID_UNDEF;
    return (unsigned)(cy * P.gres[0] * P.nz + lz * P.gres[0] + cx);
}

// Domain test of calcHashD, solver/sph_kernel_impl.cuh:95-100 (strict, compared in double).
template <typename T> __device__ __forceinline__ bool in_domain(const DevParams<T> &P, T x, T y, T z) {
    return (double)x < (double)P.gmax[0] - FSA_EPSILON && (double)x > (double)P.gmin[0] + FSA_EPSILON &&
           (double)y < (double)P.gmax[1] - FSA_EPSILON && (double)y > (double)P.gmin[1] + FSA_EPSILON &&
           (double)z < (double)P.gmax[2] - FSA_EPSILON && (double)z > (double)P.gmin[2] + FSA_EPSILON;
}

// Is a particle in global cell layer gz needed (`margin` = 0 for the force pass, 1 for density)?
template <typename T> __device__ __forceinline__ bool layer_needed(const DevParams<T> &P, int gz, int margin) {
    return gz >= P.own_lo - margin && gz < P.own_hi + margin;
}
// A run of consecutive sorted particles from cell hlo to cell hhi (both in the grid): true when none of
// its layers is needed.  The sorted order is (y, layer, x), so the run covers the layers from hlo's to
// hhi's inside one y, or the tail of one y and the head of the next.
template <typename T>
__device__ __forceinline__ bool run_unneeded(const DevParams<T> &P, unsigned hlo, unsigned hhi, int margin) {
    const unsigned gx = (unsigned)P.gres[0], nz = (unsigned)P.nz;
    const int ylo = (int)(hlo / (gx * nz)), yhi = (int)(hhi / (gx * nz));
    const int llo = (int)((hlo / gx) % nz), lhi = (int)((hhi / gx) % nz);
    const int need_lo = P.own_lo - margin - P.z0, need_hi = P.own_hi + margin - P.z0;  // local layers
    if (ylo == yhi) return lhi < need_lo || llo >= need_hi;
    if (yhi == ylo + 1) return llo >= need_hi && lhi < need_lo;
    return false;
}

template <typename T> __device__ __forceinline__ bool layer_in_edge(const DevParams<T> &P, int gz) {
    return (gz >= P.edge[0] && gz < P.edge[1]) || (gz >= P.edge[2] && gz < P.edge[3]);
}
// Does a run of consecutive sorted particles from cell hlo to cell hhi (both in the grid) hold a layer of an edge?
template <typename T> __device__ __forceinline__ bool run_touches_edge(const DevParams<T> &P, unsigned hlo, unsigned hhi) {
    const unsigned gx = (unsigned)P.gres[0], nz = (unsigned)P.nz;
    const int ylo = (int)(hlo / (gx * nz)), yhi = (int)(hhi / (gx * nz));
    const int llo = (int)((hlo / gx) % nz), lhi = (int)((hhi / gx) % nz);
    if (yhi > ylo + 1) return P.edge[0] < P.edge[1] || P.edge[2] < P.edge[3];
    bool touch = false;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const int lo = P.edge[e] - P.z0, hi = P.edge[e + 1] - P.z0;  // local layers
        if (lo >= hi) continue;
        if (ylo == yhi)
            touch |= lhi >= lo && llo < hi;
        else
            touch |= llo < hi || lhi >= lo;  // the tail of one y and the head of the next
    }
    return touch;
}

// Kernel_Cubic, solver/sph_kernel_impl.cuh:32-40, from the squared norm d2 = x*x + y*y + z*z
// (xij.Norm() evaluates exactly that sum before the square root).
template <typename T> __device__ __forceinline__ T kernel_cubic_d2(const DevParams<T> &P, T h, T d2) {
    T q = r_sqrt(d2) / h;
    T fq;
    if (q >= 2) return 0;
    if (q >= 1)
        fq = (T)((double)((2 - q) * (2 - q) * (2 - q)) * 0.25);
    else
        fq = (T)((0.666667 - (double)(q * q) + 0.5 * (double)q * (double)q * (double)q) * 1.5);
    return fq * P.kernel_cubic;
}
template <typename T> __device__ __forceinline__ T kernel_cubic(const DevParams<T> &P, T h, T x, T y, T z) {
    return kernel_cubic_d2(P, h, x * x + y * y + z * z);
}

}  // namespace fsa
