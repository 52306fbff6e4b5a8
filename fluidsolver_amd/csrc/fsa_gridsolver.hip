// Eulerian grid solver for gfx950: semi-Lagrangian advection on a staggered grid, gravity, pressure
// projection by conjugate gradients, tracer particles.  Replaces the CPU loops of the reference's
// GridSolver (solver/grid_solver.cpp under /root/reference/particle_solver; every kernel cites the lines
// it restates) and is reached through the fsa_grid_* entries of include/fluidsolver_amd.h.
//
// Layout: face arrays are padded exactly as Grid::uId / vId / wId index them (solver/grid_solver.h:39-49),
// x fastest, so a wavefront reads 64 consecutive faces; cell arrays are dim.x * dim.y * dim.z, x fastest.
// Arithmetic follows the reference text operation for operation (explicit casts where its double
// literals promote a float expression); the only exception is the three sums of the CG loop (two dot
// products and the mean residual), which the reference accumulates in one float in index order and this
// file reduces in double over a fixed tree, the same for every run.
//
// The CG loop runs without host round trips: its scalars (alpha, beta, the stop decision) live in device
// memory, every kernel of a round returns at once when the loop has ended, and the host launches rounds in
// batches and looks at the state between batches.
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "fsa_solver.h"

namespace fsa {

constexpr int kPartials = 2048;  // block partial sums per reduction (grid-stride loops above that)

// A solver works on the z planes [kz0, kz0 + nz) of a grid of gnz planes: all of them (kz0 = 0, gnz = nz, padz = pad),
// or one z-slab of a team (further down).  nz, zlen, wlen and dimSize describe the planes held; plane indices in the
// arrays are local (0 = plane kz0), positions and the tests for the faces of the domain use the global index.
// Face arrays carry padz planes of padding on either side in z: beyond the domain they stay zero, as the
// reference's padding does; inside it they hold copies of the neighbour slabs' planes.  Cell arrays have one plane
// before and one after their nz planes for the same purpose.
template <typename T> struct GridDev {
    int nx, ny, nz, pad;
    int xlen, ylen, zlen, ulen, vlen, wlen;
    int dimSize;
    int kz0, gnz, padz;
    int *err;  // slabs only: set when a sample needs a plane the slab does not hold
    T xmin[3];
    T h, dt, rho;
};

// device-resident state of the CG loop
struct CgState {
    double rr, qAq, rr_new, rabs_sum;  // raw sums
    double alpha, beta, rabs;          // as the loop uses them (rounded to the solver's precision)
    int finished;                      // 1: loop ended (break at the top of a round)
    int pending;                       // 1: loop ends after this round's direction update
    int iter;                          // the reference's `iter`
    int maxiter;                       // < 0: none
    int rounds;                        // rounds completed
};

struct GridSolver {
    int precision = FSA_PRECISION_F32;
    int device = 0;
    GridDev<float> gf{};
    GridDev<double> gd{};
    int uSize = 0, vSize = 0, wSize = 0, dimSize = 0;
    void *u = nullptr, *v = nullptr, *w = nullptr, *uadv = nullptr, *vadv = nullptr, *wadv = nullptr;
    void *p = nullptr, *divU = nullptr, *b = nullptr, *Aq = nullptr, *r = nullptr, *q = nullptr;
    double *partials = nullptr;  // 4 * kPartials
    CgState *cg = nullptr;
    double *divsum = nullptr;
    void *cell_base[6] = {};     // what was allocated for p, divU, b, Aq, r, q (they point one plane in)
    void *ppos = nullptr, *pcolor = nullptr;
    int n_particles = 0, particle_cap = 0;
    int frame = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;      // false: a team's stream
    // z-slab of a team (kz0 = 0, gnz = nz otherwise)
    int kz0 = 0, gnz = 0, padz = 0;
    size_t real_size() const { return precision == FSA_PRECISION_F64 ? sizeof(double) : sizeof(float); }
};

__device__ __forceinline__ float g_floor(float x) { return floorf(x); }
__device__ __forceinline__ double g_floor(double x) { return floor(x); }
__device__ __forceinline__ float g_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double g_abs(double x) { return fabs(x); }

// Grid::cellId / uId / vId / wId, solver/grid_solver.h:36-49
template <typename T> __device__ __forceinline__ int cell_id(const GridDev<T> &G, int x, int y, int z) {
    return (z * G.ny + y) * G.nx + x;
}
template <typename T> __device__ __forceinline__ int u_id(const GridDev<T> &G, int x, int y, int z) {
    return ((z + G.padz) * G.ylen + y + G.pad) * G.ulen + x + G.pad;
}
template <typename T> __device__ __forceinline__ int v_id(const GridDev<T> &G, int x, int y, int z) {
    return ((z + G.padz) * G.vlen + y + G.pad) * G.xlen + x + G.pad;
}
template <typename T> __device__ __forceinline__ int w_id(const GridDev<T> &G, int x, int y, int z) {
    return ((z + G.padz) * G.ylen + y + G.pad) * G.xlen + x + G.pad;
}

// ---- sampling (sampleU :301-350, sampleV :352-401, sampleW :403-452) -------------------------------
// COMP = 0, 1, 2 selects the component; the component's own axis is sampled on faces, the other two on
// cell centres shifted by half a cell.
constexpr double kSampleEps = 0.000000001;  // the EPSILON in force from :299 on

template <typename T, int COMP>
__device__ __forceinline__ T sample_component(const GridDev<T> &G, const T *__restrict__ f, T px, T py, T pz) {
    const T p[3] = {px, py, pz};
    const int dim[3] = {G.nx, G.ny, G.gnz};
    T x[3];
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        x[a] = (p[a] - G.xmin[a]) / G.h;
        c[a] = (int)g_floor(x[a]);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if ((double)(x[a] - (T)c[a]) < kSampleEps && x[a] == (T)dim[a]) c[a] -= 1;
        if ((double)x[a] > -kSampleEps && x[a] == (T)-1) c[a] += 1;
    }
    if (c[0] < 0 || c[0] > G.nx - 1 || c[1] < 0 || c[1] > G.ny - 1 || c[2] < 0 || c[2] > G.gnz - 1) return 0;
    T wt[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (a == COMP) {
            wt[a] = (p[a] - (G.xmin[a] + (T)c[a] * G.h)) / G.h;
        } else {
            if ((double)(x[a] - (T)c[a]) < 0.5) c[a]--;
            wt[a] = (T)(((double)p[a] - ((double)G.xmin[a] + ((double)c[a] + 0.5) * (double)G.h)) / (double)G.h);
        }
    }
    const int zl = c[2] - G.kz0;  // local plane; zl and zl + 1 are read
    if (G.err && (zl < -G.padz || zl + 1 > G.nz + G.padz - (COMP == 2 ? 0 : 1))) {
        atomicOr(G.err, 1);
        return 0;
    }
    auto at = [&](int dx, int dy, int dz) -> T {
        const int X = c[0] + dx, Y = c[1] + dy, Z = zl + dz;
        return f[COMP == 0 ? u_id(G, X, Y, Z) : COMP == 1 ? v_id(G, X, Y, Z) : w_id(G, X, Y, Z)];
    };
    const T w1 = wt[0], w2 = wt[1], w3 = wt[2];
    return (1 - w1) * ((1 - w2) * ((1 - w3) * at(0, 0, 0) + w3 * at(0, 0, 1)) +
                       w2 * ((1 - w3) * at(0, 1, 0) + w3 * at(0, 1, 1))) +
           w1 * ((1 - w2) * ((1 - w3) * at(1, 0, 0) + w3 * at(1, 0, 1)) +
                 w2 * ((1 - w3) * at(1, 1, 0) + w3 * at(1, 1, 1)));
}

template <typename T> __device__ __forceinline__ T at_face(const GridDev<T> &G, int a, int i) {
    return G.xmin[a] + G.h * (T)i;
}
template <typename T> __device__ __forceinline__ T at_centre(const GridDev<T> &G, int a, int i) {
    return (T)((double)G.xmin[a] + (double)G.h * ((double)i + 0.5));
}
template <typename T> __device__ __forceinline__ T quarter(T a, T b, T c, T d) {
    return (T)((double)(a + b + c + d) * 0.25);
}

// advect :456-565, one thread per face of component COMP.  u is traced back one step along x only
// (:485-489); v and w take a midpoint step (:526-534, :552-560).
template <typename T, int COMP>
__global__ __launch_bounds__(kBlock) void grid_advect_kernel(GridDev<T> G, const T *__restrict__ u,
                                                             const T *__restrict__ v, const T *__restrict__ w,
                                                             T *__restrict__ out) {
    // (the face on top of the last plane belongs to the slab that ends the domain)
    const int ex = G.nx + (COMP == 0), ey = G.ny + (COMP == 1), ez = G.nz + (COMP == 2 && G.kz0 + G.nz == G.gnz);
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)ex * ey * ez) return;
    const int i = (int)(t % ex), j = (int)((t / ex) % ey), k = (int)(t / ((long long)ex * ey));
    T vel[3], xt[3];
    if (COMP == 0) {
        vel[0] = u[u_id(G, i, j, k)];
        vel[1] = 0;
        vel[2] = 0;
        xt[0] = at_face(G, 0, i);
        xt[1] = at_centre(G, 1, j);
        xt[2] = at_centre(G, 2, k + G.kz0);
        const T s = (T)-1;
        out[u_id(G, i, j, k)] = sample_component<T, 0>(G, u, xt[0] + vel[0] * G.dt * s, xt[1] + vel[1] * G.dt * s,
                                                       xt[2] + vel[2] * G.dt * s);
        return;
    }
    if (COMP == 1) {
        vel[0] = quarter(u[u_id(G, i, j, k)], u[u_id(G, i + 1, j, k)], u[u_id(G, i, j - 1, k)],
                         u[u_id(G, i + 1, j - 1, k)]);
        vel[1] = v[v_id(G, i, j, k)];
        vel[2] = quarter(w[w_id(G, i, j, k)], w[w_id(G, i, j, k + 1)], w[w_id(G, i, j - 1, k)],
                         w[w_id(G, i, j - 1, k + 1)]);
        xt[0] = at_centre(G, 0, i);
        xt[1] = at_face(G, 1, j);
        xt[2] = at_centre(G, 2, k + G.kz0);
    } else {
        vel[0] = quarter(u[u_id(G, i, j, k)], u[u_id(G, i, j, k - 1)], u[u_id(G, i + 1, j, k)],
                         u[u_id(G, i + 1, j, k - 1)]);
        vel[1] = quarter(v[v_id(G, i, j, k)], v[v_id(G, i, j, k - 1)], v[v_id(G, i, j + 1, k)],
                         v[v_id(G, i, j + 1, k - 1)]);
        vel[2] = w[w_id(G, i, j, k)];
        xt[0] = at_centre(G, 0, i);
        xt[1] = at_centre(G, 1, j);
        xt[2] = at_face(G, 2, k + G.kz0);
    }
    const T half = (T)-0.5, full = (T)-1;
    const T mx = xt[0] + vel[0] * G.dt * half, my = xt[1] + vel[1] * G.dt * half, mz = xt[2] + vel[2] * G.dt * half;
    const T su = sample_component<T, 0>(G, u, mx, my, mz);
    const T sv = sample_component<T, 1>(G, v, mx, my, mz);
    const T sw = sample_component<T, 2>(G, w, mx, my, mz);
    const T bx = xt[0] + su * G.dt * full, by = xt[1] + sv * G.dt * full, bz = xt[2] + sw * G.dt * full;
    if (COMP == 1)
        out[v_id(G, i, j, k)] = sample_component<T, 1>(G, v, bx, by, bz);
    else
        out[w_id(G, i, j, k)] = sample_component<T, 2>(G, w, bx, by, bz);
}

// advectParticles :639-651
// A slab moves the tracers that lie in its own planes (those below or above the domain go with the first and the
// last slab) and writes them to a second pair of arrays: another slab must not move a tracer again that has just
// arrived in its planes.  The single solver works in place (src == pos).
template <typename T>
__global__ __launch_bounds__(kBlock) void grid_advect_particles_kernel(GridDev<T> G, int n, const T *__restrict__ u,
                                                                       const T *__restrict__ v,
                                                                       const T *__restrict__ w, const T *src, T *pos,
                                                                       const T *csrc, T *color) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    T x = src[3 * (size_t)i], y = src[3 * (size_t)i + 1], z = src[3 * (size_t)i + 2];
    if (G.err) {
        int c = (int)g_floor((z - G.xmin[2]) / G.h);
        c = c < 0 ? 0 : c > G.gnz - 1 ? G.gnz - 1 : c;
        if (!(c >= G.kz0 && c < G.kz0 + G.nz)) return;  // (a position that is not a number goes nowhere)
    }
    const T vx = sample_component<T, 0>(G, u, x, y, z);
    const T vy = sample_component<T, 1>(G, v, x, y, z);
    const T vz = sample_component<T, 2>(G, w, x, y, z);
    pos[3 * (size_t)i] = x + vx * G.dt;
    pos[3 * (size_t)i + 1] = y + vy * G.dt;
    pos[3 * (size_t)i + 2] = z + vz * G.dt;
    const T l = 1;
    color[4 * (size_t)i] = (T)((double)g_abs(vx) / 0.01);
    color[4 * (size_t)i + 1] = 0;
    color[4 * (size_t)i + 2] = g_abs(vz) / l;
    if (G.err) color[4 * (size_t)i + 3] = csrc[4 * (size_t)i + 3];  // (a slab writes into another array)
}

// bodyForce :579-588
// (planes [k0, k0 + nk): a slab does its copies of the neighbours' planes as well, with the same operation)
template <typename T>
__global__ __launch_bounds__(kBlock) void grid_body_force_kernel(GridDev<T> G, T *__restrict__ v, int k0, int nk) {
    const int ey = G.ny + 1;
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)G.nx * ey * nk) return;
    const int i = (int)(t % G.nx), j = (int)((t / G.nx) % ey), k = k0 + (int)(t / ((long long)G.nx * ey));
    v[v_id(G, i, j, k)] += 10 * G.dt;
}

// testcase :590-606
template <typename T>
__global__ __launch_bounds__(kBlock) void grid_testcase_kernel(GridDev<T> G, T *__restrict__ u, T *__restrict__ v,
                                                               T *__restrict__ w) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)G.dimSize) return;
    const int i = (int)(t % G.nx), j = (int)((t / G.nx) % G.ny), k = (int)(t / ((long long)G.nx * G.ny));
    u[u_id(G, i, j, k)] = (T)(i * 2);
    v[v_id(G, i, j, k)] = (T)(j * 3);
    w[w_id(G, i, j, k)] = (T)(k * 4);
    if (i == G.nx - 1) u[u_id(G, i + 1, j, k)] = (T)((i + 1) * 2);
    if (j == G.ny - 1) v[v_id(G, i, j + 1, k)] = (T)((j + 1) * 3);
    if (k == G.nz - 1) w[w_id(G, i, j, k + 1)] = (T)((k + 1) * 4);
}

// Work split of the per-cell kernels.  The hardware deals consecutive block ids to the 8 XCDs in turn, each
// with an L2 of its own.  The cells are cut into 8 contiguous slabs, one per XCD, and the blocks of an XCD sweep
// their slab front to back, side by side: at any moment the chip streams through 8 compact windows (kind to
// DRAM), and the +-x, +-y, +-z neighbours a stencil reads were fetched by the same XCD a moment ago or will be
// used by it a moment later (grids launched here have a multiple of 8 blocks).
constexpr unsigned kXcdCount = 8;
struct CellSpan {
    unsigned begin, end, stride;  // this thread's cells: begin, begin + stride, ... < end
};
__device__ __forceinline__ CellSpan cell_span(unsigned n) {
    const unsigned per_xcd = gridDim.x / kXcdCount;
    const unsigned xcd = blockIdx.x % kXcdCount, local = blockIdx.x / kXcdCount;
    const unsigned sweep = per_xcd * kBlock;                                   // cells one XCD covers per pass
    const unsigned slab = ((n + kXcdCount - 1) / kXcdCount + sweep - 1) / sweep * sweep;
    const unsigned long long lo = (unsigned long long)xcd * slab, hi = lo + slab;
    CellSpan sp;
    sp.stride = sweep;
    sp.end = hi < n ? (unsigned)hi : n;
    const unsigned long long b = lo + (unsigned long long)local * kBlock + threadIdx.x;
    sp.begin = b < sp.end ? (unsigned)b : sp.end;
    return sp;
}

// Walk over a thread's cells that keeps (i, j, k) by carrying instead of dividing per element.  `nx`, `ny`
// and `n` describe the array walked: the cells, or the groups of four cells along x.
struct CellWalk {
    unsigned t, end, stride;
    int nx, ny, i, j, k, di, dj, dk;
    __device__ __forceinline__ CellWalk(int nx_, int ny_, unsigned n) : nx(nx_), ny(ny_) {
        const CellSpan sp = cell_span(n);
        t = sp.begin;
        end = sp.end;
        stride = sp.stride;
        const unsigned plane = (unsigned)nx * (unsigned)ny;
        i = (int)(t % (unsigned)nx);
        j = (int)((t / (unsigned)nx) % (unsigned)ny);
        k = (int)(t / plane);
        di = (int)(stride % (unsigned)nx);
        dj = (int)((stride / (unsigned)nx) % (unsigned)ny);
        dk = (int)(stride / plane);
    }
    template <typename T>
    __device__ __forceinline__ explicit CellWalk(const GridDev<T> &G) : CellWalk(G.nx, G.ny, (unsigned)G.dimSize) {}
    __device__ __forceinline__ bool valid() const { return t < end; }
    __device__ __forceinline__ void next() {
        // `end - t` cannot wrap: t < end here
        t = end - t > stride ? t + stride : end;
        i += di;
        int c = i >= nx ? 1 : 0;
        i -= c ? nx : 0;
        j += dj + c;
        c = j >= ny ? 1 : 0;
        j -= c ? ny : 0;
        k += dk + c;
    }
};

// block-level sum of up to four doubles per thread into partials[which * kPartials + blockIdx.x]
template <int NV>
__device__ __forceinline__ void block_sums(double (&acc)[NV], double *__restrict__ partials) {
    __shared__ double part[NV][kBlock / 64];
#pragma unroll
    for (int s = 0; s < NV; s++) {
        double a = acc[s];
        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
        if ((threadIdx.x & 63) == 0) part[s][threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int s = 0; s < NV; s++) {
            double a = part[s][0];
            for (int wv = 1; wv < kBlock / 64; wv++) a += part[s][wv];
            partials[s * kPartials + blockIdx.x] = a;
        }
    }
}

// sum of the `nb` block partials of slot `which`, by one block, fixed order
__device__ __forceinline__ double sum_partials(const double *partials, int which, int nb) {
    __shared__ double part[kBlock / 64];
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += kBlock) a += partials[which * kPartials + i];
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
    __syncthreads();  // `part` may still be read from a previous call
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    double s = part[0];
    for (int wv = 1; wv < kBlock / 64; wv++) s += part[wv];
    return s;
}

// divVelocity :124-141 (+ makeRHS :143-160 when WITH_RHS): per cell, grid-stride
template <typename T, bool WITH_RHS>
__global__ __launch_bounds__(kBlock) void grid_div_kernel(GridDev<T> G, const T *__restrict__ u,
                                                          const T *__restrict__ v, const T *__restrict__ w,
                                                          T *__restrict__ divU, T *__restrict__ b,
                                                          double *__restrict__ partials) {
    double acc[1] = {0.0};
    for (CellWalk cw(G); cw.valid(); cw.next()) {
        const int i = cw.i, j = cw.j, k = cw.k;
        const unsigned t = cw.t;
        T div = 0;
        div += u[u_id(G, i + 1, j, k)] - u[u_id(G, i, j, k)];
        div += v[v_id(G, i, j + 1, k)] - v[v_id(G, i, j, k)];
        div += w[w_id(G, i, j, k + 1)] - w[w_id(G, i, j, k)];
        div /= G.h;
        divU[t] = div;
        acc[0] += (double)(div * div);
        if (WITH_RHS) {
            T rhs = -div;
            if (j == 0) rhs += (0 - v[v_id(G, i, j, k)]) / G.h;
            b[t] = rhs * G.rho * G.h * G.h / G.dt;
        }
    }
    block_sums<1>(acc, partials);
}

__global__ __launch_bounds__(kBlock) void grid_divsum_kernel(const double *__restrict__ partials, int nb,
                                                             double *__restrict__ out) {
    const double s = sum_partials(partials, 0, nb);
    if (threadIdx.x == 0) *out = s;
}

// mvproduct :72-115 for one cell: the six terms in the order x-, x+, y-, y+, z-, z+; the floor (j = 0) is
// a wall, the other five sides see p = 0 outside
template <typename T>
__device__ __forceinline__ T laplace_cell(const GridDev<T> &G, const T *__restrict__ x, const CellWalk &cw) {
    const int i = cw.i, j = cw.j, k = cw.k;
    const size_t t = cw.t, sy = (size_t)G.nx, sz = (size_t)G.nx * (size_t)G.ny;
    const T c = x[t];
    T res = 0;
    if (i == 0) res += c; else res += c - x[t - 1];
    if (i == G.nx - 1) res += c; else res += c - x[t + 1];
    if (j == 0) res += 0; else res += c - x[t - sy];
    if (j == G.ny - 1) res += c; else res += c - x[t + sy];
    if (k + G.kz0 == 0) res += c; else res += c - x[t - sz];
    if (k + G.kz0 == G.gnz - 1) res += c; else res += c - x[t + sz];
    return res;
}

// Four cells along x at once (rows are a multiple of four long, so a group never straddles two rows and its
// loads are aligned): the same six terms per cell, in the same order.
template <typename T> struct alignas(4 * sizeof(T)) Quad { T v[4]; };

// the six-term sums of four cells (i0 .. i0 + 3, j, k) from their values and their neighbours'
template <typename T>
__device__ __forceinline__ Quad<T> laplace_quad_terms(const GridDev<T> &G, int i0, int j, int k, const Quad<T> &c, T xm,
                                                      T xp, const Quad<T> &ym, const Quad<T> &yp, const Quad<T> &zm,
                                                      const Quad<T> &zp) {
    Quad<T> out;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const T cc = c.v[e];
        const T left = e == 0 ? xm : c.v[e > 0 ? e - 1 : 0], right = e == 3 ? xp : c.v[e < 3 ? e + 1 : 3];
        T res = 0;
        if (i0 + e == 0) res += cc; else res += cc - left;
        if (i0 + e == G.nx - 1) res += cc; else res += cc - right;
        if (j == 0) res += 0; else res += cc - ym.v[e];
        if (j == G.ny - 1) res += cc; else res += cc - yp.v[e];
        if (k + G.kz0 == 0) res += cc; else res += cc - zm.v[e];
        if (k + G.kz0 == G.gnz - 1) res += cc; else res += cc - zp.v[e];
        out.v[e] = res;
    }
    return out;
}

template <typename T>
__device__ __forceinline__ Quad<T> laplace_quad(const GridDev<T> &G, const T *__restrict__ x, const CellWalk &qw,
                                                Quad<T> &centre) {
    const int i0 = qw.i * 4, j = qw.j, k = qw.k;
    const size_t t = (size_t)qw.t * 4, sy = (size_t)G.nx, sz = (size_t)G.nx * (size_t)G.ny;
    const Quad<T> zero{{0, 0, 0, 0}};
    const Quad<T> c = *reinterpret_cast<const Quad<T> *>(x + t);
    const Quad<T> ym = j > 0 ? *reinterpret_cast<const Quad<T> *>(x + t - sy) : zero;
    const Quad<T> yp = j < G.ny - 1 ? *reinterpret_cast<const Quad<T> *>(x + t + sy) : zero;
    const Quad<T> zm = k + G.kz0 > 0 ? *reinterpret_cast<const Quad<T> *>(x + t - sz) : zero;
    const Quad<T> zp = k + G.kz0 < G.gnz - 1 ? *reinterpret_cast<const Quad<T> *>(x + t + sz) : zero;
    const T xm = i0 > 0 ? x[t - 1] : (T)0, xp = i0 + 4 < G.nx ? x[t + 4] : (T)0;
    centre = c;
    return laplace_quad_terms(G, i0, j, k, c, xm, xp, ym, yp, zm, zp);
}

template <typename T, bool QUAD>
__global__ __launch_bounds__(kBlock) void grid_mvproduct_kernel(GridDev<T> G, const T *__restrict__ x, T *__restrict__ dst) {
    if (QUAD) {
        for (CellWalk qw(G.nx / 4, G.ny, (unsigned)G.dimSize / 4); qw.valid(); qw.next()) {
            Quad<T> c;
            reinterpret_cast<Quad<T> *>(dst)[qw.t] = laplace_quad(G, x, qw, c);
        }
        return;
    }
    for (CellWalk cw(G); cw.valid(); cw.next()) dst[cw.t] = laplace_cell(G, x, cw);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void grid_dot_kernel(int n, const T *__restrict__ a, const T *__restrict__ b,
                                                          double *__restrict__ partials) {
    double acc[1] = {0.0};
    const CellSpan sp = cell_span((unsigned)n);
    for (unsigned t = sp.begin; t < sp.end; t = sp.end - t > sp.stride ? t + sp.stride : sp.end)
        acc[0] += (double)(a[t] * b[t]);
    block_sums<1>(acc, partials);
}

// ---- solve :162-214 -----------------------------------------------------------------------------------
// :170-176: Aq = A p; r = b - Aq; q = r
template <typename T, bool QUAD>
__global__ __launch_bounds__(kBlock) void cg_init_kernel(GridDev<T> G, const T *__restrict__ p, const T *__restrict__ b,
                                                         T *__restrict__ Aq, T *__restrict__ r, T *__restrict__ q) {
    if (QUAD) {
        for (CellWalk qw(G.nx / 4, G.ny, (unsigned)G.dimSize / 4); qw.valid(); qw.next()) {
            Quad<T> c;
            const Quad<T> a = laplace_quad(G, p, qw, c);
            const Quad<T> bb = reinterpret_cast<const Quad<T> *>(b)[qw.t];
            Quad<T> res;
#pragma unroll
            for (int e = 0; e < 4; e++) res.v[e] = bb.v[e] - a.v[e];
            reinterpret_cast<Quad<T> *>(Aq)[qw.t] = a;
            reinterpret_cast<Quad<T> *>(r)[qw.t] = res;
            reinterpret_cast<Quad<T> *>(q)[qw.t] = res;
        }
        return;
    }
    for (CellWalk cw(G); cw.valid(); cw.next()) {
        const unsigned t = cw.t;
        const T a = laplace_cell(G, p, cw);
        Aq[t] = a;
        const T res = b[t] - a;
        r[t] = res;
        q[t] = res;
    }
}

__global__ void cg_reset_kernel(CgState *cg, int maxiter) {
    cg->rr = cg->qAq = cg->rr_new = cg->rabs_sum = 0.0;
    cg->alpha = cg->beta = cg->rabs = 0.0;
    cg->finished = cg->pending = 0;
    cg->iter = 0;
    cg->maxiter = maxiter;
    cg->rounds = 0;
}

__device__ __forceinline__ bool cg_over(const CgState *cg) { return cg->finished != 0 || cg->pending != 0; }

// :179-182: Aq = A q, and the partial sums of r.r and q.Aq
template <typename T, bool QUAD>
__global__ __launch_bounds__(kBlock) void cg_matvec_kernel(GridDev<T> G, const CgState *__restrict__ cg,
                                                           const T *__restrict__ q, const T *__restrict__ r,
                                                           T *__restrict__ Aq, double *__restrict__ partials) {
    if (cg_over(cg)) return;
    double acc[2] = {0.0, 0.0};
    if (QUAD) {
        for (CellWalk qw(G.nx / 4, G.ny, (unsigned)G.dimSize / 4); qw.valid(); qw.next()) {
            Quad<T> c;
            const Quad<T> a = laplace_quad(G, q, qw, c);
            const Quad<T> rv = reinterpret_cast<const Quad<T> *>(r)[qw.t];
            reinterpret_cast<Quad<T> *>(Aq)[qw.t] = a;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[0] += (double)(rv.v[e] * rv.v[e]);
                acc[1] += (double)(c.v[e] * a.v[e]);
            }
        }
    } else {
        for (CellWalk cw(G); cw.valid(); cw.next()) {
            const unsigned t = cw.t;
            const T a = laplace_cell(G, q, cw);
            Aq[t] = a;
            const T rv = r[t];
            acc[0] += (double)(rv * rv);
            acc[1] += (double)(q[t] * a);
        }
    }
    block_sums<2>(acc, partials);
}

// The same pass, marching along z.  A block owns a tile of whole rows (256 threads = 256 groups of four cells
// = kBlock / (nx / 4) rows) and a run of z planes inside its XCD's slab; every thread keeps its column's
// previous, current and next plane values in registers, so per plane it reads one new group of q (plus, for
// the tile's first and last row, the neighbouring row) instead of five.  +-x and +-y neighbours inside the
// tile come from LDS.  Needs nx / 4 to be a power of two <= 256; same six-term sums, same bits.
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_matvec_march_kernel(GridDev<T> G, const CgState *__restrict__ cg,
                                                                 const T *__restrict__ q, const T *__restrict__ r,
                                                                 T *__restrict__ Aq, double *__restrict__ partials,
                                                                 int tiles, int zsegs) {
    if (cg_over(cg)) return;
    __shared__ Quad<T> tile[2][kBlock];
    const int qx = G.nx >> 2, rows = kBlock / qx;
    const int xcd = blockIdx.x % kXcdCount, local = blockIdx.x / kXcdCount;
    const int tile_id = local % tiles, zseg = local / tiles;
    const int slab = (G.nz + (int)kXcdCount - 1) / (int)kXcdCount, seglen = (slab + zsegs - 1) / zsegs;
    const int kbeg = xcd * slab + zseg * seglen;
    const int kend = min(min(kbeg + seglen, (xcd + 1) * slab), G.nz);
    const int row = threadIdx.x / qx, qc = threadIdx.x % qx;
    const int j = tile_id * rows + row;
    const bool active = j < G.ny;
    const bool first_row = row == 0, last_row = row == rows - 1;
    const Quad<T> *q4 = reinterpret_cast<const Quad<T> *>(q);
    const Quad<T> *r4 = reinterpret_cast<const Quad<T> *>(r);
    Quad<T> *Aq4 = reinterpret_cast<Quad<T> *>(Aq);
    const Quad<T> zero{{0, 0, 0, 0}};
    double acc[2] = {0.0, 0.0};
    if (kbeg < kend) {  // the same for the whole block
        const size_t rowq = (size_t)qx, planeq = rowq * (size_t)G.ny;
        size_t Q = (size_t)kbeg * planeq + (size_t)(active ? j : 0) * rowq + (size_t)qc;
        // (plain ifs: `cond ? q4[..] : zero` selects between a global and a private address when the compiler
        // keeps `zero` in memory, and every load of the loop becomes a flat load beside a scratch store)
        Quad<T> zm = zero, c = zero;
        if (active && kbeg + G.kz0 > 0) zm = q4[Q - planeq];
        if (active) c = q4[Q];
        for (int k = kbeg; k < kend; k++, Q += planeq) {
            const int buf = (k - kbeg) & 1;
            Quad<T> zp = zero, rv = zero;
            if (active && k + 1 + G.kz0 < G.gnz) zp = q4[Q + planeq];
            if (active) rv = r4[Q];
            Quad<T> ym = zero, yp = zero;
            if (active && first_row && j > 0) ym = q4[Q - rowq];
            if (active && last_row && j < G.ny - 1) yp = q4[Q + rowq];
            tile[buf][threadIdx.x] = c;
            __syncthreads();
            if (active) {
                if (!first_row) ym = tile[buf][threadIdx.x - qx];
                if (!last_row && j < G.ny - 1) yp = tile[buf][threadIdx.x + qx];
                const T xm = qc > 0 ? tile[buf][threadIdx.x - 1].v[3] : (T)0;
                const T xp = qc < qx - 1 ? tile[buf][threadIdx.x + 1].v[0] : (T)0;
                const Quad<T> a = laplace_quad_terms(G, qc * 4, j, k, c, xm, xp, ym, yp, zm, zp);
                Aq4[Q] = a;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    acc[0] += (double)(rv.v[e] * rv.v[e]);
                    acc[1] += (double)(c.v[e] * a.v[e]);
                }
            }
            zm = c;
            c = zp;
        }
    }
    block_sums<2>(acc, partials);
}

// :183-188: the "trivial all-zero" break and alpha = rr / qAq, in the solver's precision
template <typename T> __device__ __forceinline__ void cg_alpha_apply(CgState *cg, double rr, double qAq) {
    const T rrT = (T)rr, qAqT = (T)qAq;
    cg->rr = (double)rrT;
    cg->qAq = (double)qAqT;
    if (rrT == 0 || qAqT == 0) {
        cg->finished = 1;
        return;
    }
    cg->alpha = (double)(T)(rrT / qAqT);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void cg_alpha_kernel(CgState *cg, const double *__restrict__ partials, int nb) {
    if (cg_over(cg)) return;
    const double rr = sum_partials(partials, 0, nb);
    const double qAq = sum_partials(partials, 1, nb);
    if (threadIdx.x != 0) return;
    cg_alpha_apply<T>(cg, rr, qAq);
}

// :190-193 and the sums behind :195 and :201: p += alpha q; r -= alpha Aq; partial r.r and |r|
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_update_kernel(GridDev<T> G, const CgState *__restrict__ cg,
                                                           const T *__restrict__ q, const T *__restrict__ Aq,
                                                           T *__restrict__ p, T *__restrict__ r,
                                                           double *__restrict__ partials) {
    if (cg_over(cg)) return;
    const T alpha = (T)cg->alpha;
    double acc[2] = {0.0, 0.0};
    const CellSpan sp = cell_span((unsigned)G.dimSize);
    for (unsigned t = sp.begin; t < sp.end; t = sp.end - t > sp.stride ? t + sp.stride : sp.end) {
        p[t] += alpha * q[t];
        const T rv = r[t] - alpha * Aq[t];
        r[t] = rv;
        acc[0] += (double)(rv * rv);
        acc[1] += (double)g_abs(rv);
    }
    block_sums<2>(acc, partials + 2 * kPartials);
}

// :195, :203-207: beta, the mean residual, the stop test, iter++ (and this build's optional cap)
template <typename T> __device__ __forceinline__ void cg_beta_apply(CgState *cg, double rr_new, double rabs_sum, int n) {
    const T rrT = (T)cg->rr;
    cg->rr_new = (double)(T)rr_new;
    cg->beta = (double)(T)((T)rr_new / rrT);
    const T rabs = (T)rabs_sum / (T)n;
    cg->rabs_sum = rabs_sum;
    cg->rabs = (double)rabs;
    cg->rounds++;
    if (!(rabs == rabs)) {  // not a number: no test of the reference's loop would ever end it
        cg->finished = 2;
        return;
    }
    if ((double)rabs < 0.000001) {
        cg->pending = 1;  // the direction update of this round still runs (:199-200 precede the test)
        return;
    }
    cg->iter++;
    if (cg->maxiter >= 0 && cg->iter >= cg->maxiter) cg->pending = 1;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void cg_beta_kernel(CgState *cg, const double *__restrict__ partials, int nb, int n) {
    if (cg_over(cg)) return;
    const double rr_new = sum_partials(partials + 2 * kPartials, 0, nb);
    const double rabs_sum = sum_partials(partials + 2 * kPartials, 1, nb);
    if (threadIdx.x != 0) return;
    cg_beta_apply<T>(cg, rr_new, rabs_sum, n);
}

// ---- the sums of a team of z-slabs --------------------------------------------------------------------------
// Every slab's kernels leave block partials in the slab's own array; one block adds them up, slab after slab in
// the order of the planes, and the same total goes to every slab's loop state, which therefore takes the same
// decisions as its neighbours'.  Between processes the per-process totals are added by an all-reduce between the
// two kernels.
constexpr int kMaxTeam = 16;
struct TeamParts {
    const double *partials[kMaxTeam];
    CgState *cg[kMaxTeam];
    int nb[kMaxTeam];
    int count;
};

// out[0], out[1] = totals of the partial-sum slots 0 and (if `two`) 1 at `offset` in every slab's array
__global__ __launch_bounds__(kBlock) void team_sum_kernel(TeamParts tp, int offset, int two, double *__restrict__ out) {
    double a = 0.0, b = 0.0;
    for (int m = 0; m < tp.count; m++) {
        const double sa = sum_partials(tp.partials[m] + offset, 0, tp.nb[m]);
        const double sb = two ? sum_partials(tp.partials[m] + offset, 1, tp.nb[m]) : 0.0;
        a = m == 0 ? sa : a + sa;
        b = m == 0 ? sb : b + sb;
    }
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
    }
}

template <typename T> __global__ void team_alpha_kernel(TeamParts tp, const double *__restrict__ sums) {
    const int m = threadIdx.x;
    if (m >= tp.count || cg_over(tp.cg[m])) return;
    cg_alpha_apply<T>(tp.cg[m], sums[0], sums[1]);
}

template <typename T> __global__ void team_beta_kernel(TeamParts tp, const double *__restrict__ sums, int n) {
    const int m = threadIdx.x;
    if (m >= tp.count || cg_over(tp.cg[m])) return;
    cg_beta_apply<T>(tp.cg[m], sums[0], sums[1], n);
}

// :199-200: q = r + beta q, for round number `round` (0-based, counted by the host as it launches): runs
// only if that round's update ran, which also covers the round that ends the loop.
template <typename T>
__global__ __launch_bounds__(kBlock) void cg_direction_kernel(GridDev<T> G, const CgState *__restrict__ cg, int round,
                                                              const T *__restrict__ r, T *__restrict__ q) {
    if (cg->finished != 0 || cg->rounds != round + 1) return;
    const T beta = (T)cg->beta;
    const CellSpan sp = cell_span((unsigned)G.dimSize);
    for (unsigned t = sp.begin; t < sp.end; t = sp.end - t > sp.stride ? t + sp.stride : sp.end)
        q[t] = r[t] + beta * q[t];
}

// updateU :216-266, one thread per face and component (every face is updated by exactly one cell of the
// reference's loop).  The maximum-velocity estimate of :251-256 only feeds a commented-out statement.
template <typename T>
__global__ __launch_bounds__(kBlock) void grid_update_u_kernel(GridDev<T> G, const T *__restrict__ p, T *__restrict__ u,
                                                               T *__restrict__ v, T *__restrict__ w) {
    const T c = G.dt / G.rho / G.h;
    const long long nu = (long long)(G.nx + 1) * G.ny * G.nz, nv = (long long)G.nx * (G.ny + 1) * G.nz,
                    nw = (long long)G.nx * G.ny * (G.nz + (G.kz0 + G.nz == G.gnz));
    long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t < nu) {
        const int ex = G.nx + 1;
        const int i = (int)(t % ex), j = (int)((t / ex) % G.ny), k = (int)(t / ((long long)ex * G.ny));
        T *f = &u[u_id(G, i, j, k)];
        if (i == 0) *f -= (p[cell_id(G, 0, j, k)] - 0) * c;
        else if (i < G.nx) *f -= (p[cell_id(G, i, j, k)] - p[cell_id(G, i - 1, j, k)]) * c;
        else *f -= (0 - p[cell_id(G, G.nx - 1, j, k)]) * c;
        return;
    }
    t -= nu;
    if (t < nv) {
        const int ey = G.ny + 1;
        const int i = (int)(t % G.nx), j = (int)((t / G.nx) % ey), k = (int)(t / ((long long)G.nx * ey));
        T *f = &v[v_id(G, i, j, k)];
        if (j == 0) *f = 0;  // near solid
        else if (j < G.ny) *f -= (p[cell_id(G, i, j, k)] - p[cell_id(G, i, j - 1, k)]) * c;
        else *f -= (0 - p[cell_id(G, i, G.ny - 1, k)]) * c;
        return;
    }
    t -= nv;
    if (t < nw) {
        const int i = (int)(t % G.nx), j = (int)((t / G.nx) % G.ny), k = (int)(t / ((long long)G.nx * G.ny));
        T *f = &w[w_id(G, i, j, k)];
        if (k + G.kz0 == 0) *f -= (p[cell_id(G, i, j, 0)] - 0) * c;
        else if (k + G.kz0 < G.gnz) *f -= (p[cell_id(G, i, j, k)] - p[cell_id(G, i, j, k - 1)]) * c;
        else *f -= (0 - p[cell_id(G, i, j, G.nz - 1)]) * c;
    }
}

// ---- host side -----------------------------------------------------------------------------------------
static inline int gblocks(long long n) { return (int)((n + kBlock - 1) / kBlock); }
// blocks of a per-cell kernel: a multiple of the XCD count, at most one partial-sum slot each
static inline int rblocks(long long n) {
    const long long want = std::min<long long>(kPartials, (n + kBlock - 1) / kBlock);
    return (int)((want + kXcdCount - 1) / kXcdCount * kXcdCount);
}

// `dim` is the whole grid; the solver holds its planes [kz0, kz0 + nz) with padz planes of padding in z
template <typename T> static void fill_dev(GridDev<T> &G, const int dim[3], int pad, double h, double dt, double rho,
                                           int kz0, int nz, int padz, int *err) {
    G.nx = dim[0]; G.ny = dim[1]; G.nz = nz; G.pad = pad;
    G.kz0 = kz0; G.gnz = dim[2]; G.padz = padz; G.err = err;
    G.xlen = dim[0] + 2 * pad; G.ylen = dim[1] + 2 * pad; G.zlen = nz + 2 * padz;
    G.ulen = G.xlen + 1; G.vlen = G.ylen + 1; G.wlen = G.zlen + 1;
    G.dimSize = dim[0] * dim[1] * nz;
    G.h = (T)h; G.dt = (T)dt; G.rho = (T)rho;
    // loadConfig :26-28: `-grid.dim.x/2*grid.h` negates, divides as integers, then multiplies
    G.xmin[0] = (T)(-dim[0] / 2) * G.h;
    G.xmin[1] = 0;
    G.xmin[2] = (T)(-dim[2] / 2) * G.h;
}

#define GRID_LAUNCH(kernel, grid, ...)                                                                      \
    do {                                                                                                    \
        if (g->precision == FSA_PRECISION_F32) {                                                            \
            using T = float;                                                                                \
            const GridDev<T> &G = g->gf;                                                                    \
            (void)G;                                                                                        \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, g->stream, __VA_ARGS__);                \
        } else {                                                                                            \
            using T = double;                                                                               \
            const GridDev<T> &G = g->gd;                                                                    \
            (void)G;                                                                                        \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, g->stream, __VA_ARGS__);                \
        }                                                                                                   \
        FSA_HIP(hipGetLastError());                                                                         \
    } while (0)

static void grid_free(GridSolver *g) {
    void *ptrs[] = {g->u, g->v, g->w, g->uadv, g->vadv, g->wadv, g->cell_base[0], g->cell_base[1], g->cell_base[2],
                    g->cell_base[3], g->cell_base[4], g->cell_base[5], g->partials, g->cg, g->divsum, g->ppos, g->pcolor};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (g->stream && g->own_stream) (void)hipStreamDestroy(g->stream);
}

static int galloc(void **out, size_t bytes) {
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) return fail(FSA_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    e = hipMemset(*out, 0, bytes);
    if (e != hipSuccess) return fail(FSA_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    return FSA_OK;
}

static int grid_advect(GridSolver *g) {
    const long long nx = g->gf.nx, ny = g->gf.ny, nz = g->gf.nz;
    GRID_LAUNCH((grid_advect_kernel<T, 0>), gblocks((nx + 1) * ny * nz), G, (const T *)g->u, (const T *)g->v,
                (const T *)g->w, (T *)g->uadv);
    GRID_LAUNCH((grid_advect_kernel<T, 1>), gblocks(nx * (ny + 1) * nz), G, (const T *)g->u, (const T *)g->v,
                (const T *)g->w, (T *)g->vadv);
    GRID_LAUNCH((grid_advect_kernel<T, 2>), gblocks(nx * ny * (nz + (g->kz0 + nz == g->gnz))), G, (const T *)g->u, (const T *)g->v,
                (const T *)g->w, (T *)g->wadv);
    std::swap(g->u, g->uadv);  // swapBuffer :573-577
    std::swap(g->v, g->vadv);
    std::swap(g->w, g->wadv);
    return FSA_OK;
}

static int grid_advect_particles(GridSolver *g) {
    if (g->n_particles == 0) return FSA_OK;
    GRID_LAUNCH(grid_advect_particles_kernel<T>, gblocks(g->n_particles), G, g->n_particles, (const T *)g->u,
                (const T *)g->v, (const T *)g->w, (const T *)g->ppos, (T *)g->ppos, (const T *)g->pcolor, (T *)g->pcolor);
    return FSA_OK;
}

static int grid_body_force(GridSolver *g) {
    // a slab: its copies of the neighbours' planes too (not the padding beyond the domain, which stays zero)
    const int below = g->kz0 > 0 ? g->padz : 0, above = g->kz0 + g->gf.nz < g->gnz ? g->padz : 0;
    const int nk = below + g->gf.nz + above;
    GRID_LAUNCH(grid_body_force_kernel<T>, gblocks((long long)g->gf.nx * (g->gf.ny + 1) * nk), G, (T *)g->v, -below, nk);
    return FSA_OK;
}

static int grid_div(GridSolver *g, bool with_rhs) {
    const int nb = rblocks(g->dimSize);
    if (with_rhs)
        GRID_LAUNCH((grid_div_kernel<T, true>), nb, G, (const T *)g->u, (const T *)g->v, (const T *)g->w, (T *)g->divU,
                    (T *)g->b, g->partials);
    else
        GRID_LAUNCH((grid_div_kernel<T, false>), nb, G, (const T *)g->u, (const T *)g->v, (const T *)g->w,
                    (T *)g->divU, (T *)g->b, g->partials);
    hipLaunchKernelGGL(grid_divsum_kernel, dim3(1), dim3(kBlock), 0, g->stream, g->partials, nb, g->divsum);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

static int grid_update_u(GridSolver *g) {
    const long long nx = g->gf.nx, ny = g->gf.ny, nz = g->gf.nz;
    const long long faces = (nx + 1) * ny * nz + nx * (ny + 1) * nz + nx * ny * (nz + (g->kz0 + nz == g->gnz));
    GRID_LAUNCH(grid_update_u_kernel<T>, gblocks(faces), G, (const T *)g->p, (T *)g->u, (T *)g->v, (T *)g->w);
    return FSA_OK;
}

constexpr int kCgBatch = 16;  // rounds launched between two looks at the loop state

// which stencil kernel a solver's planes take, and with how many blocks
struct SolvePlan {
    bool quad = false, march = false;
    int nb = 0, nbq = 0, tiles = 0, zsegs = 0, nbm = 0, nb_dots = 0;
};
static SolvePlan solve_plan(const GridSolver *g) {
    SolvePlan pl;
    pl.nb = rblocks(g->dimSize);
    pl.quad = g->gf.nx % 4 == 0;
    pl.nbq = pl.quad ? rblocks(g->dimSize / 4) : pl.nb;  // blocks of the stencil kernels
    // the z-marching stencil: rows of a power-of-two number of four-cell groups, and few enough tiles
    const int qx = g->gf.nx / 4;
    pl.march = pl.quad && qx <= kBlock && (qx & (qx - 1)) == 0 && !std::getenv("FSA_GRID_NO_MARCH");
    if (pl.march) {
        const int rows = kBlock / qx, slab = (g->gf.nz + (int)kXcdCount - 1) / (int)kXcdCount;
        pl.tiles = (g->gf.ny + rows - 1) / rows;
        pl.zsegs = std::max(1, std::min(slab, (kPartials / (int)kXcdCount) / std::max(pl.tiles, 1)));
        pl.nbm = (int)kXcdCount * pl.tiles * pl.zsegs;
        if (pl.tiles > kPartials / (int)kXcdCount) pl.march = false;
    }
    pl.nb_dots = pl.march ? pl.nbm : pl.nbq;
    return pl;
}

static int launch_cg_init(GridSolver *g, const SolvePlan &pl) {
    if (pl.quad)
        GRID_LAUNCH((cg_init_kernel<T, true>), pl.nbq, G, (const T *)g->p, (const T *)g->b, (T *)g->Aq, (T *)g->r,
                    (T *)g->q);
    else
        GRID_LAUNCH((cg_init_kernel<T, false>), pl.nb, G, (const T *)g->p, (const T *)g->b, (T *)g->Aq, (T *)g->r,
                    (T *)g->q);
    return FSA_OK;
}

static int launch_cg_matvec(GridSolver *g, const SolvePlan &pl) {
    if (pl.march)
        GRID_LAUNCH(cg_matvec_march_kernel<T>, pl.nbm, G, g->cg, (const T *)g->q, (const T *)g->r, (T *)g->Aq,
                    g->partials, pl.tiles, pl.zsegs);
    else if (pl.quad)
        GRID_LAUNCH((cg_matvec_kernel<T, true>), pl.nbq, G, g->cg, (const T *)g->q, (const T *)g->r, (T *)g->Aq,
                    g->partials);
    else
        GRID_LAUNCH((cg_matvec_kernel<T, false>), pl.nb, G, g->cg, (const T *)g->q, (const T *)g->r, (T *)g->Aq,
                    g->partials);
    return FSA_OK;
}

static int launch_mvproduct_p(GridSolver *g, const SolvePlan &pl) {  // :210
    if (pl.quad)
        GRID_LAUNCH((grid_mvproduct_kernel<T, true>), pl.nbq, G, (const T *)g->p, (T *)g->Aq);
    else
        GRID_LAUNCH((grid_mvproduct_kernel<T, false>), pl.nb, G, (const T *)g->p, (T *)g->Aq);
    return FSA_OK;
}

static int grid_solve(GridSolver *g, int maxiter, int *iters, double *residual, int *rounds) {
    const SolvePlan pl = solve_plan(g);
    hipLaunchKernelGGL(cg_reset_kernel, dim3(1), dim3(1), 0, g->stream, g->cg, maxiter);
    int rc = launch_cg_init(g, pl);
    if (rc) return rc;
    CgState st{};
    int round = 0;
    for (;;) {
        for (int k = 0; k < kCgBatch; k++, round++) {
            rc = launch_cg_matvec(g, pl);
            if (rc) return rc;
            GRID_LAUNCH(cg_alpha_kernel<T>, 1, g->cg, g->partials, pl.nb_dots);
            GRID_LAUNCH(cg_update_kernel<T>, pl.nb, G, g->cg, (const T *)g->q, (const T *)g->Aq, (T *)g->p, (T *)g->r,
                        g->partials);
            GRID_LAUNCH(cg_beta_kernel<T>, 1, g->cg, g->partials, pl.nb, g->dimSize);
            GRID_LAUNCH(cg_direction_kernel<T>, pl.nb, G, g->cg, round, (const T *)g->r, (T *)g->q);
        }
        FSA_HIP(hipMemcpyAsync(&st, g->cg, sizeof(st), hipMemcpyDeviceToHost, g->stream));
        FSA_HIP(counted_stream_sync(g->stream));
        if (st.finished || st.pending) break;
    }
    if (st.finished == 2)
        return fail(FSA_ERR_INVALID, "conjugate gradients: the residual is not a number after %d rounds", st.rounds);
    rc = launch_mvproduct_p(g, pl);
    if (rc) return rc;
    if (iters) *iters = st.iter;
    if (residual) *residual = st.rabs;
    if (rounds) *rounds = st.rounds;
    return FSA_OK;
}

static int grid_check_args(const int dim[3], int padding, double h, double dt, double rho, int precision, int device) {
    if (!dim || dim[0] <= 0 || dim[1] <= 0 || dim[2] <= 0) return fail(FSA_ERR_INVALID, "dim must be positive");
    // the advection reads one face outside the effective grid on every side
    if (padding < 1) return fail(FSA_ERR_INVALID, "padding %d: at least 1 is needed", padding);
    if (!(h > 0) || !(dt > 0) || !(rho > 0)) return fail(FSA_ERR_INVALID, "h, dt and rho must be positive");
    if (precision != FSA_PRECISION_F32 && precision != FSA_PRECISION_F64)
        return fail(FSA_ERR_INVALID, "unknown precision %d", precision);
    const double faces = ((double)dim[0] + 2 * padding + 1) * ((double)dim[1] + 2 * padding + 1) *
                         ((double)dim[2] + 2 * padding + 1);
    if (faces >= 2147483647.0) return fail(FSA_ERR_INVALID, "grid too large for 32-bit face indices");
    int ndev = 0;
    FSA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(FSA_ERR_INVALID, "device %d of %d", device, ndev);
    return FSA_OK;
}

// A solver for the planes [kz0, kz0 + nz) of the grid `dim` (all of them: kz0 = 0, nz = dim[2], padz = padding).
// `stream` != nullptr: work goes there (a team's stream) instead of a stream of the solver's own.
static int grid_make(const int dim[3], int padding, double h, double dt, double rho, int precision, int device, int kz0,
                     int nz, int padz, int *err, hipStream_t stream, GridSolver **out) {
    GridSolver *g = new (std::nothrow) GridSolver();
    if (!g) return fail(FSA_ERR_NOMEM, "out of host memory");
    g->precision = precision;
    g->device = device;
    g->kz0 = kz0;
    g->gnz = dim[2];
    g->padz = padz;
    fill_dev(g->gf, dim, padding, h, dt, rho, kz0, nz, padz, err);
    fill_dev(g->gd, dim, padding, h, dt, rho, kz0, nz, padz, err);
    g->uSize = g->gf.ulen * g->gf.ylen * g->gf.zlen;
    g->vSize = g->gf.xlen * g->gf.vlen * g->gf.zlen;
    g->wSize = g->gf.xlen * g->gf.ylen * g->gf.wlen;
    g->dimSize = g->gf.dimSize;
    int rc = FSA_OK;
    if (hipSetDevice(device) != hipSuccess) rc = fail(FSA_ERR_HIP, "hipSetDevice failed");
    if (stream) {
        g->stream = stream;
        g->own_stream = false;
    } else if (rc == FSA_OK && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
        rc = fail(FSA_ERR_HIP, "hipStreamCreate failed");
    }
    const size_t R = g->real_size();
    void **face[] = {&g->u, &g->v, &g->w, &g->uadv, &g->vadv, &g->wadv};
    const size_t face_n[] = {(size_t)g->uSize, (size_t)g->vSize, (size_t)g->wSize,
                             (size_t)g->uSize, (size_t)g->vSize, (size_t)g->wSize};
    for (int k = 0; k < 6 && rc == FSA_OK; k++) rc = galloc(face[k], face_n[k] * R);
    // cell arrays; a slab's have one plane before and one after for the copies of its neighbours' planes
    void **cell[] = {&g->p, &g->divU, &g->b, &g->Aq, &g->r, &g->q};
    const size_t plane = nz == dim[2] && kz0 == 0 && !err ? 0 : (size_t)dim[0] * dim[1];
    for (int k = 0; k < 6 && rc == FSA_OK; k++) {
        rc = galloc(&g->cell_base[k], ((size_t)g->dimSize + 2 * plane) * R);
        if (rc == FSA_OK) *cell[k] = (char *)g->cell_base[k] + plane * R;
    }
    if (rc == FSA_OK) rc = galloc((void **)&g->partials, 4 * kPartials * sizeof(double));
    if (rc == FSA_OK) rc = galloc((void **)&g->cg, sizeof(CgState));
    if (rc == FSA_OK) rc = galloc((void **)&g->divsum, 2 * sizeof(double));
    if (rc != FSA_OK) {
        grid_free(g);
        delete g;
        return rc;
    }
    *out = g;
    return FSA_OK;
}

static void *grid_field(GridSolver *g, const std::string &k, size_t *count) {
    const size_t U = (size_t)g->uSize, V = (size_t)g->vSize, W = (size_t)g->wSize, C = (size_t)g->dimSize;
    if (k == "u") { *count = U; return g->u; }
    if (k == "v") { *count = V; return g->v; }
    if (k == "w") { *count = W; return g->w; }
    if (k == "uadv") { *count = U; return g->uadv; }
    if (k == "vadv") { *count = V; return g->vadv; }
    if (k == "wadv") { *count = W; return g->wadv; }
    if (k == "p") { *count = C; return g->p; }
    if (k == "divU") { *count = C; return g->divU; }
    if (k == "b") { *count = C; return g->b; }
    if (k == "Aq") { *count = C; return g->Aq; }
    if (k == "r") { *count = C; return g->r; }
    if (k == "q") { *count = C; return g->q; }
    return nullptr;
}


// ---- a team of z-slabs ----------------------------------------------------------------------------------
// The grid cut into slabs of whole z planes, one solver each (GridDev::kz0 ..).  The reference has one grid on one
// host (solver/grid_solver.cpp); what the cut adds is
//   * copies of the neighbours' planes: `padz` planes of the face arrays (the back-trace of the advection may leave
//     a slab by padz - 1 planes; a sample that needs more raises the team's error word), refreshed after the
//     advection and after the velocity update; one plane of p before the first and after the last stencil of a
//     solve, one plane of q before every stencil in between;
//   * sums over all slabs for the two dot products and the residual of every round and for the divergence sum.
// LOCAL teams hold all slabs in this process on one GPU and one stream (a rehearsal, and the reference for the
// other kinds): refreshes are device copies.  RCCL teams hold one slab per process: refreshes are ncclSend /
// ncclRecv pairs with the z-neighbours, the sums an ncclAllReduce of two doubles, all queued on the stream; the
// loop state is the same on every rank, so every rank queues the same rounds.  HOSTED teams are RCCL teams whose
// bytes travel through functions the caller supplies (fsa_host_transport: MPI, sockets, torch.distributed ...):
// every exchange is staged through pinned host memory and waits for the stream, which is slow and meant for hosts
// without RCCL between their processes -- and it lets several processes share one GPU, which RCCL refuses.
struct GridTeam {
    int kind = 0;  // 0 LOCAL, 1 RCCL, 2 HOSTED
    fsa_host_transport via{};      // HOSTED
    char *stage = nullptr;         // HOSTED: pinned host memory, send runs then receive runs
    size_t stage_bytes = 0;
    int world = 1, rank = 0;
    int dim[3] = {0, 0, 0}, pad = 1, padz = 2, precision = FSA_PRECISION_F32, device = 0;
    std::vector<int> cuts;
    std::vector<GridSolver *> m;  // this process's slabs, bottom to top
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    int *err = nullptr;      // device
    double *sums = nullptr;  // device, two doubles
    void *ppos = nullptr, *pcolor = nullptr, *pnew = nullptr, *pcnew = nullptr;
    int n_particles = 0, particle_cap = 0;
    int frame = 0;
    size_t real_size() const { return precision == FSA_PRECISION_F64 ? sizeof(double) : sizeof(float); }
    bool first_slab(const GridSolver *g) const { return g->kz0 == 0; }
    bool last_slab(const GridSolver *g) const { return g->kz0 + g->gf.nz == g->gnz; }
};

static void team_free(GridTeam *t) {
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    for (GridSolver *g : t->m) {
        grid_free(g);
        delete g;
    }
    if (t->comm) (void)ncclCommDestroy(t->comm);
    if (t->stage) (void)hipHostFree(t->stage);
    void *ptrs[] = {t->err, t->sums, t->ppos, t->pcolor, t->pnew, t->pcnew};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (t->stream) (void)hipStreamDestroy(t->stream);
}

enum { kFaceU = 0, kFaceV = 1, kFaceW = 2 };
static void *face_ptr(GridSolver *g, int f) { return f == kFaceU ? g->u : f == kFaceV ? g->v : g->w; }
static size_t face_plane(const GridSolver *g, int f) {
    const GridDev<float> &G = g->gf;
    return f == kFaceU ? (size_t)G.ylen * G.ulen : f == kFaceV ? (size_t)G.vlen * G.xlen : (size_t)G.ylen * G.xlen;
}
enum { kCellP = 0, kCellDiv, kCellB, kCellAq, kCellR, kCellQ };
static void *cell_ptr(GridSolver *g, int c) {
    void *v[] = {g->p, g->divU, g->b, g->Aq, g->r, g->q};
    return v[c];
}

#define TEAM_NCCL(expr)                                                                                \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess) return fail(FSA_ERR_HIP, "%s failed: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)

// One run of bytes to the slab below / above and one from it: [0] = with the slab below, [1] = with the slab above.
struct HaloRuns {
    char *send[2] = {nullptr, nullptr}, *recv[2] = {nullptr, nullptr};
    size_t send_bytes[2] = {0, 0}, recv_bytes[2] = {0, 0};
};

// what slab g exchanges for face array f (padz planes up, padz (+ 1 for w: the shared face belongs to the upper
// slab) planes down) or, planes == 0, for cell array f (one plane each way)
static HaloRuns halo_runs(const GridTeam *t, GridSolver *g, bool faces, int f) {
    HaloRuns h;
    const size_t R = t->real_size();
    const int nz = g->gf.nz;
    if (faces) {
        const size_t st = face_plane(g, f) * R;
        char *base = (char *)face_ptr(g, f);
        const int pz = t->padz, extra = f == kFaceW ? 1 : 0;
        if (!t->first_slab(g)) {  // with the slab below: my first planes go down, its last planes come up
            h.send[0] = base + (size_t)pz * st;
            h.send_bytes[0] = (size_t)(pz + extra) * st;
            h.recv[0] = base;
            h.recv_bytes[0] = (size_t)pz * st;
        }
        if (!t->last_slab(g)) {
            h.send[1] = base + (size_t)nz * st;  // local planes [nz - pz, nz)
            h.send_bytes[1] = (size_t)pz * st;
            h.recv[1] = base + (size_t)(nz + pz) * st;
            h.recv_bytes[1] = (size_t)(pz + extra) * st;
        }
    } else {
        const size_t st = (size_t)g->gf.nx * g->gf.ny * R;
        char *base = (char *)cell_ptr(g, f);
        if (!t->first_slab(g)) {
            h.send[0] = base;
            h.recv[0] = base - st;
            h.send_bytes[0] = h.recv_bytes[0] = st;
        }
        if (!t->last_slab(g)) {
            h.send[1] = base + (size_t)(nz - 1) * st;
            h.recv[1] = base + (size_t)nz * st;
            h.send_bytes[1] = h.recv_bytes[1] = st;
        }
    }
    return h;
}

static int team_stage(GridTeam *t, size_t bytes) {
    if (bytes <= t->stage_bytes) return FSA_OK;
    if (t->stage) (void)hipHostFree(t->stage);
    t->stage = nullptr;
    t->stage_bytes = 0;
    FSA_HIP(hipHostMalloc((void **)&t->stage, bytes, hipHostMallocDefault));
    t->stage_bytes = bytes;
    return FSA_OK;
}

// HOSTED: an element-wise sum (kind 0 doubles, 1 floats) or maximum (2 ints) of a device array over all ranks
static int team_hosted_allreduce(GridTeam *t, void *dev, size_t count, int what) {
    const size_t bytes = count * (what == 0 ? sizeof(double) : 4);
    int rc = team_stage(t, bytes);
    if (rc) return rc;
    FSA_HIP(hipMemcpyAsync(t->stage, dev, bytes, hipMemcpyDeviceToHost, t->stream));
    FSA_HIP(counted_stream_sync(t->stream));
    if (t->via.allreduce(t->via.ctx, t->stage, count, what) != 0)
        return fail(FSA_ERR_HIP, "grid team: the caller's allreduce failed");
    FSA_HIP(hipMemcpyAsync(dev, t->stage, bytes, hipMemcpyHostToDevice, t->stream));
    FSA_HIP(counted_stream_sync(t->stream));  // (the staging memory is used again by the next exchange)
    return FSA_OK;
}

static int team_refresh(GridTeam *t, bool faces, int f) {
    if (t->kind == 2) {
        const HaloRuns h = halo_runs(t, t->m[0], faces, f);
        if (!h.send[0] && !h.send[1]) return FSA_OK;
        const size_t need = h.send_bytes[0] + h.send_bytes[1] + h.recv_bytes[0] + h.recv_bytes[1];
        int rc = team_stage(t, need);
        if (rc) return rc;
        char *out[2] = {t->stage, t->stage + h.send_bytes[0]};
        char *in[2] = {out[1] + h.send_bytes[1], out[1] + h.send_bytes[1] + h.recv_bytes[0]};
        for (int d = 0; d < 2; d++)
            if (h.send[d]) FSA_HIP(hipMemcpyAsync(out[d], h.send[d], h.send_bytes[d], hipMemcpyDeviceToHost, t->stream));
        FSA_HIP(counted_stream_sync(t->stream));
        // the slab below first on every rank: rank r's call meets rank r - 1's second one, no two wait for each other
        for (int d = 0; d < 2; d++)
            if (h.send[d] &&
                t->via.sendrecv(t->via.ctx, t->rank + (d == 0 ? -1 : 1), out[d], h.send_bytes[d], in[d], h.recv_bytes[d]) != 0)
                return fail(FSA_ERR_HIP, "grid team: the caller's sendrecv failed");
        for (int d = 0; d < 2; d++)
            if (h.send[d]) FSA_HIP(hipMemcpyAsync(h.recv[d], in[d], h.recv_bytes[d], hipMemcpyHostToDevice, t->stream));
        FSA_HIP(counted_stream_sync(t->stream));
        return FSA_OK;
    }
    if (t->kind == 0) {
        for (size_t i = 0; i + 1 < t->m.size(); i++) {
            const HaloRuns lo = halo_runs(t, t->m[i], faces, f), hi = halo_runs(t, t->m[i + 1], faces, f);
            if (lo.send_bytes[1] != hi.recv_bytes[0] || hi.send_bytes[0] != lo.recv_bytes[1])
                return fail(FSA_ERR_INVALID, "grid team: slabs %zu and %zu disagree about their exchange", i, i + 1);
            FSA_HIP(hipMemcpyAsync(hi.recv[0], lo.send[1], lo.send_bytes[1], hipMemcpyDeviceToDevice, t->stream));
            FSA_HIP(hipMemcpyAsync(lo.recv[1], hi.send[0], hi.send_bytes[0], hipMemcpyDeviceToDevice, t->stream));
        }
        return FSA_OK;
    }
    const HaloRuns h = halo_runs(t, t->m[0], faces, f);
    if (!h.send[0] && !h.send[1]) return FSA_OK;
    TEAM_NCCL(ncclGroupStart());
    ncclResult_t nr = ncclSuccess;
    for (int d = 0; d < 2 && nr == ncclSuccess; d++) {
        if (!h.send[d]) continue;
        const int peer = t->rank + (d == 0 ? -1 : 1);
        nr = ncclSend(h.send[d], h.send_bytes[d], ncclChar, peer, t->comm, t->stream);
        if (nr == ncclSuccess) nr = ncclRecv(h.recv[d], h.recv_bytes[d], ncclChar, peer, t->comm, t->stream);
    }
    const ncclResult_t ne = ncclGroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return fail(FSA_ERR_HIP, "grid team: RCCL send / receive failed: %s", ncclGetErrorString(nr));
    return FSA_OK;
}

static int team_refresh_velocity(GridTeam *t) {
    for (int f = 0; f < 3; f++) {
        int rc = team_refresh(t, true, f);
        if (rc) return rc;
    }
    return FSA_OK;
}

// totals of every slab's block partials at `offset` (slot 0, and slot 1 if `two`) into t->sums, over all processes
static int team_sums(GridTeam *t, int offset, bool two, const std::vector<int> &nb) {
    TeamParts tp{};
    tp.count = (int)t->m.size();
    for (int i = 0; i < tp.count; i++) {
        tp.partials[i] = t->m[i]->partials;
        tp.cg[i] = t->m[i]->cg;
        tp.nb[i] = nb[i];
    }
    hipLaunchKernelGGL(team_sum_kernel, dim3(1), dim3(kBlock), 0, t->stream, tp, offset, two ? 1 : 0, t->sums);
    FSA_HIP(hipGetLastError());
    if (t->kind == 1) TEAM_NCCL(ncclAllReduce(t->sums, t->sums, 2, ncclDouble, ncclSum, t->comm, t->stream));
    if (t->kind == 2) return team_hosted_allreduce(t, t->sums, 2, 0);
    return FSA_OK;
}

static TeamParts team_states(GridTeam *t) {
    TeamParts tp{};
    tp.count = (int)t->m.size();
    for (int i = 0; i < tp.count; i++) tp.cg[i] = t->m[i]->cg;
    return tp;
}

#define TEAM_EACH(call)                   \
    do {                                  \
        for (GridSolver * g : t->m) {     \
            int _rc = (call);             \
            if (_rc) return _rc;          \
        }                                 \
    } while (0)

static int team_advect(GridTeam *t) {
    TEAM_EACH(grid_advect(g));
    return team_refresh_velocity(t);
}

static int team_advect_particles(GridTeam *t) {
    if (t->n_particles == 0) return FSA_OK;
    const size_t R = t->real_size(), n = (size_t)t->n_particles;
    if (t->kind == 0) {  // (a tracer nobody claims -- its position is not a number -- stays as it is)
        FSA_HIP(hipMemcpyAsync(t->pnew, t->ppos, n * 3 * R, hipMemcpyDeviceToDevice, t->stream));
        FSA_HIP(hipMemcpyAsync(t->pcnew, t->pcolor, n * 4 * R, hipMemcpyDeviceToDevice, t->stream));
    } else {  // the other ranks' tracers arrive by a sum with zeros
        FSA_HIP(hipMemsetAsync(t->pnew, 0, n * 3 * R, t->stream));
        FSA_HIP(hipMemsetAsync(t->pcnew, 0, n * 4 * R, t->stream));
    }
    for (GridSolver *g : t->m)
        GRID_LAUNCH(grid_advect_particles_kernel<T>, gblocks(t->n_particles), G, t->n_particles, (const T *)g->u,
                    (const T *)g->v, (const T *)g->w, (const T *)t->ppos, (T *)t->pnew, (const T *)t->pcolor, (T *)t->pcnew);
    if (t->kind == 1) {
        const ncclDataType_t dt = t->precision == FSA_PRECISION_F64 ? ncclDouble : ncclFloat;
        TEAM_NCCL(ncclAllReduce(t->pnew, t->pnew, n * 3, dt, ncclSum, t->comm, t->stream));
        TEAM_NCCL(ncclAllReduce(t->pcnew, t->pcnew, n * 4, dt, ncclSum, t->comm, t->stream));
    }
    if (t->kind == 2) {
        const int what = t->precision == FSA_PRECISION_F64 ? 0 : 1;
        int rc = team_hosted_allreduce(t, t->pnew, n * 3, what);
        if (!rc) rc = team_hosted_allreduce(t, t->pcnew, n * 4, what);
        if (rc) return rc;
    }
    std::swap(t->ppos, t->pnew);
    std::swap(t->pcolor, t->pcnew);
    return FSA_OK;
}

static int team_div(GridTeam *t, bool with_rhs) {
    std::vector<int> nb;
    for (GridSolver *g : t->m) {
        const int n = rblocks(g->dimSize);
        nb.push_back(n);
        if (with_rhs)
            GRID_LAUNCH((grid_div_kernel<T, true>), n, G, (const T *)g->u, (const T *)g->v, (const T *)g->w, (T *)g->divU,
                        (T *)g->b, g->partials);
        else
            GRID_LAUNCH((grid_div_kernel<T, false>), n, G, (const T *)g->u, (const T *)g->v, (const T *)g->w,
                        (T *)g->divU, (T *)g->b, g->partials);
    }
    return team_sums(t, 0, false, nb);
}

static int team_update_u(GridTeam *t) {
    TEAM_EACH(grid_update_u(g));
    return team_refresh_velocity(t);
}

static int team_solve(GridTeam *t, int maxiter, int *iters, double *residual, int *rounds) {
    std::vector<SolvePlan> pl;
    std::vector<int> nb_dots, nb;
    for (GridSolver *g : t->m) {
        pl.push_back(solve_plan(g));
        nb_dots.push_back(pl.back().nb_dots);
        nb.push_back(pl.back().nb);
        hipLaunchKernelGGL(cg_reset_kernel, dim3(1), dim3(1), 0, t->stream, g->cg, maxiter);
    }
    const long long cells = (long long)t->dim[0] * t->dim[1] * t->dim[2];
    const TeamParts states = team_states(t);
    const int nm = (int)t->m.size();
    int rc = team_refresh(t, false, kCellP);
    if (rc) return rc;
    for (int i = 0; i < nm; i++)
        if ((rc = launch_cg_init(t->m[i], pl[i]))) return rc;
    CgState st{};
    int round = 0;
    for (;;) {
        for (int k = 0; k < kCgBatch; k++, round++) {
            if ((rc = team_refresh(t, false, kCellQ))) return rc;
            for (int i = 0; i < nm; i++)
                if ((rc = launch_cg_matvec(t->m[i], pl[i]))) return rc;
            if ((rc = team_sums(t, 0, true, nb_dots))) return rc;
            if (t->precision == FSA_PRECISION_F32)
                hipLaunchKernelGGL(team_alpha_kernel<float>, dim3(1), dim3(64), 0, t->stream, states, t->sums);
            else
                hipLaunchKernelGGL(team_alpha_kernel<double>, dim3(1), dim3(64), 0, t->stream, states, t->sums);
            for (GridSolver *g : t->m)
                GRID_LAUNCH(cg_update_kernel<T>, rblocks(g->dimSize), G, g->cg, (const T *)g->q, (const T *)g->Aq,
                            (T *)g->p, (T *)g->r, g->partials);
            if ((rc = team_sums(t, 2 * kPartials, true, nb))) return rc;
            if (t->precision == FSA_PRECISION_F32)
                hipLaunchKernelGGL(team_beta_kernel<float>, dim3(1), dim3(64), 0, t->stream, states, t->sums, (int)cells);
            else
                hipLaunchKernelGGL(team_beta_kernel<double>, dim3(1), dim3(64), 0, t->stream, states, t->sums, (int)cells);
            for (GridSolver *g : t->m)
                GRID_LAUNCH(cg_direction_kernel<T>, rblocks(g->dimSize), G, g->cg, round, (const T *)g->r, (T *)g->q);
        }
        FSA_HIP(hipMemcpyAsync(&st, t->m[0]->cg, sizeof(st), hipMemcpyDeviceToHost, t->stream));
        FSA_HIP(counted_stream_sync(t->stream));
        if (st.finished || st.pending) break;
    }
    if (st.finished == 2)
        return fail(FSA_ERR_INVALID, "conjugate gradients: the residual is not a number after %d rounds", st.rounds);
    if ((rc = team_refresh(t, false, kCellP))) return rc;
    for (int i = 0; i < nm; i++)
        if ((rc = launch_mvproduct_p(t->m[i], pl[i]))) return rc;
    if (iters) *iters = st.iter;
    if (residual) *residual = st.rabs;
    if (rounds) *rounds = st.rounds;
    return FSA_OK;
}

// A field by name: which of the face / cell arrays
static bool team_field(const std::string &k, bool *faces, int *f, bool *adv) {
    *adv = false;
    const char *fn[] = {"u", "v", "w"};
    for (int i = 0; i < 3; i++) {
        if (k == fn[i]) { *faces = true; *f = i; return true; }
        if (k == std::string(fn[i]) + "adv") { *faces = true; *f = i; *adv = true; return true; }
    }
    const char *cn[] = {"p", "divU", "b", "Aq", "r", "q"};
    for (int i = 0; i < 6; i++)
        if (k == cn[i]) { *faces = false; *f = i; return true; }
    return false;
}

// Copies between a field in the layout of the single solver (the whole grid, `pad` planes of padding in z) and
// the slabs.  to_host: every slab writes the planes it owns (the first and the last also the padding below and
// above the domain); else every slab takes all the planes it holds, copies of the neighbours' included.
static int team_copy_field(GridTeam *t, const char *name, void *host, size_t bytes, bool to_host) {
    if (!name || (!host && bytes)) return fail(FSA_ERR_INVALID, "NULL argument");
    bool faces = false, adv = false;
    int f = 0;
    if (!team_field(name, &faces, &f, &adv)) return fail(FSA_ERR_UNKNOWN_NAME, "unknown grid field '%s'", name);
    const size_t R = t->real_size();
    const int gnz = t->dim[2], pad = t->pad;
    const int extra = (faces && f == kFaceW) ? 1 : 0;
    size_t plane = 0, total = 0;
    if (faces) {
        plane = face_plane(t->m[0], f) * R;
        total = plane * (size_t)(gnz + 2 * pad + extra);
    } else {
        plane = (size_t)t->dim[0] * t->dim[1] * R;
        total = plane * (size_t)gnz;
    }
    if (bytes != total) return fail(FSA_ERR_INVALID, "field '%s' is %zu bytes, caller passed %zu", name, total, bytes);
    FSA_HIP(counted_stream_sync(t->stream));
    for (GridSolver *g : t->m) {
        const int nz = g->gf.nz;
        // local planes [lo, hi) and the array's plane index of local plane 0; global plane = kz0 + gpad + local
        int lo, hi, base0;
        const int gpad = faces ? pad : 0, gtop = gnz + 2 * gpad + extra;  // planes of the whole-grid array
        char *dev;
        if (faces) {
            void *arr[] = {adv ? g->uadv : g->u, adv ? g->vadv : g->v, adv ? g->wadv : g->w};
            dev = (char *)arr[f];
            base0 = t->padz;
            if (to_host) {
                lo = t->first_slab(g) ? -pad : 0;
                hi = nz + (t->last_slab(g) ? pad + extra : 0);
            } else {
                lo = -t->padz;
                hi = nz + t->padz + extra;
            }
        } else {
            dev = (char *)cell_ptr(g, f);
            base0 = 0;
            lo = to_host ? 0 : -1;
            hi = to_host ? nz : nz + 1;
        }
        lo = std::max(lo, -(g->kz0 + gpad));
        hi = std::min(hi, gtop - (g->kz0 + gpad));
        if (hi <= lo) continue;
        char *d = dev + (ptrdiff_t)(lo + base0) * (ptrdiff_t)plane;
        char *h = (char *)host + (size_t)(g->kz0 + gpad + lo) * plane;
        const size_t nbytes = (size_t)(hi - lo) * plane;
        if (to_host) FSA_HIP(hipMemcpy(h, d, nbytes, hipMemcpyDeviceToHost));
        else FSA_HIP(hipMemcpy(d, h, nbytes, hipMemcpyHostToDevice));
    }
    return FSA_OK;
}

static int team_make(const int dim[3], int padding, double h, double dt, double rho, int precision, int device, int kind,
                     int rank, int world, const int *cuts, int reach, GridTeam **out) {
    int rc = grid_check_args(dim, padding, h, dt, rho, precision, device);
    if (rc) return rc;
    if (world < 1 || (kind == 0 && world > kMaxTeam) || !cuts || reach < 1 || rank < 0 || rank >= world)
        return fail(FSA_ERR_INVALID, "grid team: at least one slab (at most %d in one process), a reach of at least one plane",
                    kMaxTeam);
    const int padz = std::max(padding, reach + 1);
    if (cuts[0] != 0 || cuts[world] != dim[2]) return fail(FSA_ERR_INVALID, "grid team: the cuts must span the planes 0 .. %d", dim[2]);
    for (int r = 0; r < world; r++)
        if (world > 1 && cuts[r + 1] - cuts[r] < padz + 1)
            return fail(FSA_ERR_INVALID, "grid team: slab %d has %d planes, fewer than reach + 2 = %d", r, cuts[r + 1] - cuts[r],
                        padz + 1);
    GridTeam *t = new (std::nothrow) GridTeam();
    if (!t) return fail(FSA_ERR_NOMEM, "out of host memory");
    t->kind = kind;
    t->world = world;
    t->rank = rank;
    for (int a = 0; a < 3; a++) t->dim[a] = dim[a];
    t->pad = padding;
    t->padz = padz;
    t->precision = precision;
    t->device = device;
    t->cuts.assign(cuts, cuts + world + 1);
    rc = FSA_OK;
    if (hipSetDevice(device) != hipSuccess) rc = fail(FSA_ERR_HIP, "hipSetDevice failed");
    if (rc == FSA_OK && hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(FSA_ERR_HIP, "hipStreamCreate failed");
    if (rc == FSA_OK) rc = galloc((void **)&t->err, 4 * sizeof(int));
    if (rc == FSA_OK) rc = galloc((void **)&t->sums, 2 * sizeof(double));
    const int r0 = kind == 0 ? 0 : rank, r1 = kind == 0 ? world : rank + 1;  // (RCCL, HOSTED: this process's slab)
    for (int r = r0; r < r1 && rc == FSA_OK; r++) {
        GridSolver *g = nullptr;
        rc = grid_make(dim, padding, h, dt, rho, precision, device, cuts[r], cuts[r + 1] - cuts[r], padz, t->err, t->stream,
                       &g);
        if (rc == FSA_OK) t->m.push_back(g);
    }
    if (rc != FSA_OK) {
        team_free(t);
        delete t;
        return rc;
    }
    *out = t;
    return FSA_OK;
}

}  // namespace fsa

using namespace fsa;

#define GNEED(g)                                                        \
    do {                                                                \
        if (!(g)) return fail(FSA_ERR_INVALID, "grid solver is NULL");  \
        FSA_HIP(hipSetDevice((g)->device));                             \
    } while (0)
#define GTRY(expr)           \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

extern "C" {

int fsa_grid_create(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                    fsa_grid **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = grid_check_args(dim, padding, h, dt, rho, precision, device);
    if (rc) return rc;
    GridSolver *g = nullptr;
    rc = grid_make(dim, padding, h, dt, rho, precision, device, 0, dim[2], padding, nullptr, nullptr, &g);
    if (rc) return rc;
    *out = reinterpret_cast<fsa_grid *>(g);
    return FSA_OK;
}

int fsa_grid_destroy(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    if (!g) return FSA_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)counted_stream_sync(g->stream);
    grid_free(g);
    delete g;
    return FSA_OK;
}

int fsa_grid_sizes(fsa_grid *hg, int sizes[4], double xmin[3]) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    if (!g) return fail(FSA_ERR_INVALID, "grid solver is NULL");
    if (sizes) {
        sizes[0] = g->uSize;
        sizes[1] = g->vSize;
        sizes[2] = g->wSize;
        sizes[3] = g->dimSize;
    }
    if (xmin)
        for (int a = 0; a < 3; a++) xmin[a] = g->precision == FSA_PRECISION_F32 ? (double)g->gf.xmin[a] : g->gd.xmin[a];
    return FSA_OK;
}

int fsa_grid_synchronize(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    FSA_HIP(counted_stream_sync(g->stream));
    return FSA_OK;
}

int fsa_grid_get_field(fsa_grid *hg, const char *name, void *dst, size_t bytes) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (!name || (!dst && bytes)) return fail(FSA_ERR_INVALID, "NULL argument");
    size_t count = 0;
    void *p = grid_field(g, name, &count);
    if (!p) return fail(FSA_ERR_UNKNOWN_NAME, "unknown grid field '%s'", name);
    if (bytes != count * g->real_size())
        return fail(FSA_ERR_INVALID, "field '%s' is %zu bytes, caller passed %zu", name, count * g->real_size(), bytes);
    FSA_HIP(counted_stream_sync(g->stream));
    FSA_HIP(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return FSA_OK;
}

int fsa_grid_set_field(fsa_grid *hg, const char *name, const void *src, size_t bytes) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (!name || (!src && bytes)) return fail(FSA_ERR_INVALID, "NULL argument");
    size_t count = 0;
    void *p = grid_field(g, name, &count);
    if (!p) return fail(FSA_ERR_UNKNOWN_NAME, "unknown grid field '%s'", name);
    if (bytes != count * g->real_size())
        return fail(FSA_ERR_INVALID, "field '%s' is %zu bytes, caller passed %zu", name, count * g->real_size(), bytes);
    FSA_HIP(counted_stream_sync(g->stream));
    FSA_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    return FSA_OK;
}

int fsa_grid_set_particles(fsa_grid *hg, int n, const void *pos3) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (n < 0 || (n > 0 && !pos3)) return fail(FSA_ERR_INVALID, "bad particle array");
    FSA_HIP(counted_stream_sync(g->stream));
    const size_t R = g->real_size();
    if (n > g->particle_cap) {
        if (g->ppos) (void)hipFree(g->ppos);
        if (g->pcolor) (void)hipFree(g->pcolor);
        g->ppos = g->pcolor = nullptr;
        g->particle_cap = 0;
        GTRY(galloc(&g->ppos, (size_t)n * 3 * R));
        GTRY(galloc(&g->pcolor, (size_t)n * 4 * R));
        g->particle_cap = n;
    }
    g->n_particles = n;
    if (n == 0) return FSA_OK;
    FSA_HIP(hipMemcpy(g->ppos, pos3, (size_t)n * 3 * R, hipMemcpyHostToDevice));
    // addParticles :635: colour (1, 1, 1, 1)
    if (g->precision == FSA_PRECISION_F32) {
        std::vector<float> one((size_t)n * 4, 1.f);
        FSA_HIP(hipMemcpy(g->pcolor, one.data(), one.size() * sizeof(float), hipMemcpyHostToDevice));
    } else {
        std::vector<double> one((size_t)n * 4, 1.0);
        FSA_HIP(hipMemcpy(g->pcolor, one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return FSA_OK;
}

int fsa_grid_get_particles(fsa_grid *hg, void *pos3, void *color4, int *n) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (n) *n = g->n_particles;
    FSA_HIP(counted_stream_sync(g->stream));
    const size_t R = g->real_size(), N = (size_t)g->n_particles;
    if (N == 0) return FSA_OK;
    if (pos3) FSA_HIP(hipMemcpy(pos3, g->ppos, N * 3 * R, hipMemcpyDeviceToHost));
    if (color4) FSA_HIP(hipMemcpy(color4, g->pcolor, N * 4 * R, hipMemcpyDeviceToHost));
    return FSA_OK;
}

int fsa_grid_advect(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    return grid_advect(g);
}

int fsa_grid_advect_particles(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    return grid_advect_particles(g);
}

int fsa_grid_body_force(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    return grid_body_force(g);
}

int fsa_grid_testcase(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    GRID_LAUNCH(grid_testcase_kernel<T>, gblocks(g->dimSize), G, (T *)g->u, (T *)g->v, (T *)g->w);
    return FSA_OK;
}

int fsa_grid_div_velocity(fsa_grid *hg, double *div_u_sum) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    GTRY(grid_div(g, false));
    if (div_u_sum) {
        FSA_HIP(hipMemcpyAsync(div_u_sum, g->divsum, sizeof(double), hipMemcpyDeviceToHost, g->stream));
        FSA_HIP(counted_stream_sync(g->stream));
    }
    return FSA_OK;
}

int fsa_grid_make_rhs(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    return grid_div(g, true);
}

int fsa_grid_solve(fsa_grid *hg, int maxiter, int *iterations, double *residual) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    return grid_solve(g, maxiter, iterations, residual, nullptr);
}

int fsa_grid_update_u(fsa_grid *hg) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    return grid_update_u(g);
}

int fsa_grid_mvproduct(fsa_grid *hg, const char *src, const char *dst) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (!src || !dst) return fail(FSA_ERR_INVALID, "NULL argument");
    size_t ca = 0, cb = 0;
    void *a = grid_field(g, src, &ca), *b = grid_field(g, dst, &cb);
    if (!a || !b || ca != (size_t)g->dimSize || cb != (size_t)g->dimSize || a == b)
        return fail(FSA_ERR_INVALID, "mvproduct needs two different cell-centred fields");
    if (g->gf.nx % 4 == 0)
        GRID_LAUNCH((grid_mvproduct_kernel<T, true>), rblocks(g->dimSize / 4), G, (const T *)a, (T *)b);
    else
        GRID_LAUNCH((grid_mvproduct_kernel<T, false>), rblocks(g->dimSize), G, (const T *)a, (T *)b);
    return FSA_OK;
}

int fsa_grid_dotproduct(fsa_grid *hg, const char *a_name, const char *b_name, double *out) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (!a_name || !b_name || !out) return fail(FSA_ERR_INVALID, "NULL argument");
    size_t ca = 0, cb = 0;
    void *a = grid_field(g, a_name, &ca), *b = grid_field(g, b_name, &cb);
    if (!a || !b || ca != (size_t)g->dimSize || cb != (size_t)g->dimSize)
        return fail(FSA_ERR_INVALID, "dotproduct needs two cell-centred fields");
    const int nb = rblocks(g->dimSize);
    GRID_LAUNCH(grid_dot_kernel<T>, nb, g->dimSize, (const T *)a, (const T *)b, g->partials);
    hipLaunchKernelGGL(grid_divsum_kernel, dim3(1), dim3(kBlock), 0, g->stream, g->partials, nb, g->divsum);
    FSA_HIP(hipGetLastError());
    double s = 0.0;
    FSA_HIP(hipMemcpyAsync(&s, g->divsum, sizeof(double), hipMemcpyDeviceToHost, g->stream));
    FSA_HIP(counted_stream_sync(g->stream));
    *out = g->precision == FSA_PRECISION_F32 ? (double)(float)s : s;
    return FSA_OK;
}

int fsa_grid_step(fsa_grid *hg, int nsteps, int maxiter) {
    GridSolver *g = reinterpret_cast<GridSolver *>(hg);
    GNEED(g);
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    for (int k = 0; k < nsteps; k++) {  // Step :608-619
        GTRY(grid_advect(g));
        GTRY(grid_advect_particles(g));
        GTRY(grid_body_force(g));
        GTRY(grid_div(g, true));
        GTRY(grid_solve(g, maxiter, nullptr, nullptr, nullptr));
        GTRY(grid_update_u(g));
        GTRY(grid_div(g, false));
        g->frame++;
    }
    return FSA_OK;
}

// ---- grid team ------------------------------------------------------------------------------------------
#define TNEED(t)                                                      \
    do {                                                              \
        if (!(t)) return fail(FSA_ERR_INVALID, "grid team is NULL");  \
        FSA_HIP(hipSetDevice((t)->device));                           \
    } while (0)

int fsa_grid_team_create_local(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                               int world, const int *cuts, int reach, fsa_grid_team **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    GridTeam *t = nullptr;
    int rc = team_make(dim, padding, h, dt, rho, precision, device, 0, 0, world, cuts, reach, &t);
    if (rc) return rc;
    *out = reinterpret_cast<fsa_grid_team *>(t);
    return FSA_OK;
}

int fsa_grid_team_create_rccl(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                              int rank, int world, const char id[FSA_SLAB_GROUP_ID_BYTES], const int *cuts, int reach,
                              fsa_grid_team **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!id) return fail(FSA_ERR_INVALID, "id is NULL");
    GridTeam *t = nullptr;
    int rc = team_make(dim, padding, h, dt, rho, precision, device, 1, rank, world, cuts, reach, &t);
    if (rc) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t r = ncclCommInitRank(&t->comm, world, u, rank);
    if (r != ncclSuccess) {
        t->comm = nullptr;
        team_free(t);
        delete t;
        return fail(FSA_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    *out = reinterpret_cast<fsa_grid_team *>(t);
    return FSA_OK;
}

int fsa_grid_team_create_hosted(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                                int rank, int world, const fsa_host_transport *via, const int *cuts, int reach,
                                fsa_grid_team **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!via || !via->sendrecv || !via->allreduce) return fail(FSA_ERR_INVALID, "the transport needs both functions");
    GridTeam *t = nullptr;
    int rc = team_make(dim, padding, h, dt, rho, precision, device, 2, rank, world, cuts, reach, &t);
    if (rc) return rc;
    t->via = *via;
    *out = reinterpret_cast<fsa_grid_team *>(t);
    return FSA_OK;
}

int fsa_grid_team_destroy(fsa_grid_team *ht) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    if (!t) return FSA_OK;
    team_free(t);
    delete t;
    return FSA_OK;
}

int fsa_grid_team_get_field(fsa_grid_team *ht, const char *name, void *dst, size_t bytes) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_copy_field(t, name, dst, bytes, true);
}

int fsa_grid_team_set_field(fsa_grid_team *ht, const char *name, const void *src, size_t bytes) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_copy_field(t, name, const_cast<void *>(src), bytes, false);
}

int fsa_grid_team_set_particles(fsa_grid_team *ht, int n, const void *pos3) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    if (n < 0 || (n > 0 && !pos3)) return fail(FSA_ERR_INVALID, "bad particle array");
    FSA_HIP(counted_stream_sync(t->stream));
    const size_t R = t->real_size();
    if (n > t->particle_cap) {
        void **arr[] = {&t->ppos, &t->pcolor, &t->pnew, &t->pcnew};
        for (void **a : arr) {
            if (*a) (void)hipFree(*a);
            *a = nullptr;
        }
        t->particle_cap = 0;
        GTRY(galloc(&t->ppos, (size_t)n * 3 * R));
        GTRY(galloc(&t->pnew, (size_t)n * 3 * R));
        GTRY(galloc(&t->pcolor, (size_t)n * 4 * R));
        GTRY(galloc(&t->pcnew, (size_t)n * 4 * R));
        t->particle_cap = n;
    }
    t->n_particles = n;
    if (n == 0) return FSA_OK;
    FSA_HIP(hipMemcpy(t->ppos, pos3, (size_t)n * 3 * R, hipMemcpyHostToDevice));
    if (t->precision == FSA_PRECISION_F32) {  // addParticles :635: colour (1, 1, 1, 1)
        std::vector<float> one((size_t)n * 4, 1.f);
        FSA_HIP(hipMemcpy(t->pcolor, one.data(), one.size() * sizeof(float), hipMemcpyHostToDevice));
    } else {
        std::vector<double> one((size_t)n * 4, 1.0);
        FSA_HIP(hipMemcpy(t->pcolor, one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return FSA_OK;
}

int fsa_grid_team_get_particles(fsa_grid_team *ht, void *pos3, void *color4, int *n) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    if (n) *n = t->n_particles;
    FSA_HIP(counted_stream_sync(t->stream));
    const size_t R = t->real_size(), N = (size_t)t->n_particles;
    if (N == 0) return FSA_OK;
    if (pos3) FSA_HIP(hipMemcpy(pos3, t->ppos, N * 3 * R, hipMemcpyDeviceToHost));
    if (color4) FSA_HIP(hipMemcpy(color4, t->pcolor, N * 4 * R, hipMemcpyDeviceToHost));
    return FSA_OK;
}

int fsa_grid_team_advect(fsa_grid_team *ht) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_advect(t);
}

int fsa_grid_team_advect_particles(fsa_grid_team *ht) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_advect_particles(t);
}

int fsa_grid_team_body_force(fsa_grid_team *ht) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    TEAM_EACH(grid_body_force(g));
    return FSA_OK;
}

int fsa_grid_team_make_rhs(fsa_grid_team *ht) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_div(t, true);
}

int fsa_grid_team_div_velocity(fsa_grid_team *ht, double *div_u_sum) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    GTRY(team_div(t, false));
    if (div_u_sum) {
        FSA_HIP(hipMemcpyAsync(div_u_sum, t->sums, sizeof(double), hipMemcpyDeviceToHost, t->stream));
        FSA_HIP(counted_stream_sync(t->stream));
    }
    return FSA_OK;
}

int fsa_grid_team_solve(fsa_grid_team *ht, int maxiter, int *iterations, double *residual) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_solve(t, maxiter, iterations, residual, nullptr);
}

int fsa_grid_team_update_u(fsa_grid_team *ht) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    return team_update_u(t);
}

int fsa_grid_team_step(fsa_grid_team *ht, int nsteps, int maxiter) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    for (int k = 0; k < nsteps; k++) {  // Step :608-619
        GTRY(team_advect(t));
        GTRY(team_advect_particles(t));
        TEAM_EACH(grid_body_force(g));
        GTRY(team_div(t, true));
        GTRY(team_solve(t, maxiter, nullptr, nullptr, nullptr));
        GTRY(team_update_u(t));
        GTRY(team_div(t, false));
        t->frame++;
    }
    return FSA_OK;
}

int fsa_grid_team_sync(fsa_grid_team *ht, int *error_bits) {
    GridTeam *t = reinterpret_cast<GridTeam *>(ht);
    TNEED(t);
    // (between processes every rank learns of any rank's error: all of them call this)
    if (t->kind == 1) TEAM_NCCL(ncclAllReduce(t->err, t->err, 1, ncclInt, ncclMax, t->comm, t->stream));
    if (t->kind == 2) GTRY(team_hosted_allreduce(t, t->err, 1, 2));
    FSA_HIP(counted_stream_sync(t->stream));
    int err = 0;
    FSA_HIP(hipMemcpy(&err, t->err, sizeof(int), hipMemcpyDeviceToHost));
    if (error_bits) *error_bits = err;
    if (err)
        return fail(FSA_ERR_CAPACITY, "grid team: a sample of the velocity needed a plane that its slab does not hold (a "
                                      "back-trace crossed more planes than the team's reach); the state is invalid");
    return FSA_OK;
}

}  // extern "C"
