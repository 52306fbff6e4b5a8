// Tuned fp32 density and force passes for gfx950: LDS-staged cell-row tiles + per-lane hit queues.
//
// A workgroup owns BS consecutive particles of the sorted arrays, i.e. a run of cells [hlo, hhi]
// in hash order.  For each of the 9 (dy, dz) row offsets the neighbour cells of all of them form
// ONE contiguous cell range [hlo + off - 1, hhi + off + 1], hence one contiguous particle range
// ("row").  The kernel walks the 9 rows in the reference's order (z outer, y inner):
//
//   * the row's {x, y, z, type} are copied into a small LDS buffer with coalesced 16-byte loads;
//     two buffers alternate, the global loads of row r + 1 are issued before row r is examined
//     and land in LDS after it, so their latency is hidden behind the arithmetic;
//   * phase 1 (cull): every lane walks its own index range of the row, four ds_read_b128 in
//     flight, and appends the candidates that pass the distance test to its private queue in LDS
//     (queue[slot][lane]: conflict-free; 16-bit entries = row number and index inside the row);
//   * phase 2 (evaluate): when a queue nears its capacity, and after the last row, the wavefront
//     drains the queues: every lane evaluates its queued neighbours, so the expensive arithmetic
//     runs with most lanes busy.  Queues are first-in first-out, so each particle's sum is
//     accumulated in the reference's neighbour order.
//
// LDS per workgroup is about 32 KiB, which keeps five workgroups (20 wavefronts) on a CU: the
// VALU of gfx950 needs at least four wavefronts per SIMD to issue at its full rate.
//
// Two arithmetic flavours share this structure:
//   EXACT: the pair routines of fsa_pair.h (operation for operation the reference's expressions).
//   FAST : fp32 only, v_sqrt/v_rsq/v_rcp and fused multiply-adds; differs from EXACT by a few
//          units in the last place per pair (about 1e-7 relative in density and force).
//
// Workgroups whose rows do not fit the buffers (very sparse or very crowded regions) or that hold
// out-of-grid particles take the generic global-memory walk with the exact routines.
#include <cstdlib>

#include "fsa_pair.h"
#include "fsa_solver.h"

namespace fsa {

constexpr int kChunk = 4;     // candidates fetched per lane and loop trip
constexpr int kCellCap = 96;  // cell-start entries staged per row
constexpr float kFar = 3.0e18f;

template <int BS, int ROWCAP> struct RowShared {
    float4 row[2][ROWCAP + kChunk];  // + far-away padding read by the last chunk of a lane
    int cellL[9 * kCellCap];         // cell starts of the 9 rows, relative to the row's first particle
    int rowStartG[9];                // first particle (global index) of each row
    int rowLen[9];
    int rowCell0[9];                 // first cell of each row
    int staged;
};

// Block-uniform set-up.  Returns false when this block must take the generic path.
template <int BS, int ROWCAP>
__device__ __forceinline__ bool setup_rows(const DevParams<float> &P, int b0, int nb,
                                           const int *__restrict__ cellStart,
                                           const int *__restrict__ sortedHash, RowShared<BS, ROWCAP> &sh,
                                           int ablate = 0) {
    const int tid = threadIdx.x;
    const unsigned hlo = (unsigned)sortedHash[b0], hhi = (unsigned)sortedHash[b0 + nb - 1];
    const int C = P.num_cells;
    // gx >= 4: a candidate that lies in the same hash row but outside a lane's three cells is then at
    // least one cell (= one smoothing radius) away in x, so it can never pass the distance test
    const bool grid_ok = hhi != (unsigned)FSA_GRID_UNDEF && (long long)hhi - (long long)hlo + 4 <= kCellCap &&
                         P.gres[0] >= 4;
    if (tid < 9) {
        int s = 0, len = -1, c0 = 0;
        if (grid_ok) {
            const int dz = tid / 3 - 1, dy = tid % 3 - 1;
            const long long off = (long long)dy * P.gres[0] * P.nz + (long long)dz * P.gres[0];
            long long a = (long long)hlo + off - 1, b = (long long)hhi + off + 2;
            a = a < 0 ? 0 : (a > C ? C : a);
            b = b < 0 ? 0 : (b > C ? C : b);
            c0 = (int)a;
            s = cellStart[a];
            len = cellStart[b] - s;
        }
        sh.rowStartG[tid] = s;
        sh.rowLen[tid] = len;
        sh.rowCell0[tid] = c0;
    }
    __syncthreads();
    if (tid == 0) {
        bool ok = grid_ok;
        for (int r = 0; r < 9; r++)
            if (sh.rowLen[r] < 0 || sh.rowLen[r] > ROWCAP) ok = false;
        sh.staged = ok ? 1 : 0;
    }
    __syncthreads();
    if (!sh.staged) return false;
    const int ncell = (int)(hhi - hlo) + 4;  // entries per row: cells hlo-1 .. hhi+2
    for (int t = tid; t < 9 * ncell && !(ablate & 16); t += BS) {
        const int r = t / ncell, k = t - r * ncell;
        long long c = (long long)sh.rowCell0[r] + k;
        if (c > C) c = C;
        sh.cellL[r * kCellCap + k] = cellStart[c] - sh.rowStartG[r];
    }
    return true;
}

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// Row r of the tile: every thread fetches its (at most two) entries into registers ...
template <int BS, int ROWCAP>
__device__ __forceinline__ void row_fetch(const RowShared<BS, ROWCAP> &sh, int r,
                                          const float4 *__restrict__ posType, float4 &e0, float4 &e1) {
    static_assert(ROWCAP + kChunk <= 2 * BS, "two entries per thread must cover a row");
    const int tid = threadIdx.x;
    const int s = sh.rowStartG[r], len = sh.rowLen[r];
    // entries past the row's end read some valid address and are then replaced by far-away padding
    const int last = len > 0 ? len - 1 : 0;
    e0 = posType[s + min(tid, last)];
    e1 = posType[s + min(tid + BS, last)];
    if (tid >= len) e0 = make_float4(kFar, kFar, kFar, __int_as_float(-1));
    if (tid + BS >= len) e1 = make_float4(kFar, kFar, kFar, __int_as_float(-1));
}
// ... and stores them later, after the previous row has been examined.
template <int BS, int ROWCAP>
__device__ __forceinline__ void row_store(RowShared<BS, ROWCAP> &sh, int buf, const float4 &e0, const float4 &e1) {
    const int tid = threadIdx.x;
    sh.row[buf][tid] = e0;
    if (tid + BS < ROWCAP + kChunk) sh.row[buf][tid + BS] = e1;
}

// ---- fast fp32 pair arithmetic -----------------------------------------------------------------
struct FastConst {
    float sr, sr2, inv_h, kcubic, kspiky, visc_eps;  // visc_eps = 0.01 * sr^2
};

__device__ __forceinline__ float cubic_fast(const FastConst &K, float d2) {
    const float q = __builtin_amdgcn_sqrtf(d2) * K.inv_h;
    // q < 1: (0.666667 - q^2 + 0.5 q^3) * 1.5     1 <= q < 2: 0.25 (2 - q)^3
    const float q2 = q * q;
    const float w1 = __builtin_fmaf(q2, __builtin_fmaf(0.75f, q, -1.5f), 1.0000005f);
    const float t = 2.0f - q;
    const float w2 = 0.25f * t * t * t;
    return (q < 1.0f ? w1 : w2) * K.kcubic;
}

// ------------------------------------------------------------------------------------------------
// density + pressure
// ------------------------------------------------------------------------------------------------
template <int BS, int ROWCAP, int Q, bool FAST>
__global__ __launch_bounds__(BS) void density_tuned_kernel(DevParams<float> P, int n,
                                                           const float4 *__restrict__ posType,
                                                           const float4 *__restrict__ velMass,
                                                           const float *__restrict__ restDensity,
                                                           const int *__restrict__ cellStart,
                                                           const int *__restrict__ sortedHash,
                                                           float *__restrict__ density,
                                                           float *__restrict__ pressure, int ablate) {
    __shared__ RowShared<BS, ROWCAP> sh;
    __shared__ float queue[Q * BS];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * BS;
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const bool active = index < n;
    const bool staged = setup_rows<BS, ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, ablate);

    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    float mass_i = 0.f, rd_i = 1.f;
    if (active) {
        pi = posType[index];
        mass_i = velMass[index].w;
        rd_i = restDensity[index];
    }
    float dens = 0.f;
    if (!staged) {
        if (active && !(ablate & 32))
            dens = density_walk_global(P, index, pi, mass_i, rd_i, posType, restDensity, cellStart);
    } else {
        const float sr = P.sr, sr2 = sr * sr;
        const float h = (float)((double)sr * 0.5);
        FastConst K;
        K.sr = sr; K.sr2 = sr2; K.inv_h = 1.0f / h; K.kcubic = P.kernel_cubic; K.kspiky = P.kspikydiff;
        K.visc_eps = 0.01f * sr2;
        const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
        int cnt = 0;
        // Queue entries are the squared distances of the fluid neighbours found: the density
        // contribution depends on nothing else.  A TYPE_RIGID neighbour also needs its own
        // restDensity, so it takes two slots: -(global index + 1) as an integer, then d2.
        float *const q = queue + tid;
        auto drain = [&]() {
            if (ablate & 1) { cnt = 0; return; }
            for (int s = 0; wave_any(s < cnt); s++) {
                if (s < cnt) {
                    float d2 = q[s * BS];
                    float coef = mass_i;
                    if (__float_as_int(d2) < 0) {  // rigid marker (a squared distance is never negative)
                        coef = rd_i * restDensity[-__float_as_int(d2) - 1];
                        s++;
                        d2 = q[s * BS];
                    }
                    if (FAST)
                        dens = __builtin_fmaf(coef, cubic_fast(K, d2), dens);
                    else
                        dens += coef * kernel_cubic_d2(P, h, d2);
                }
            }
            cnt = 0;
        };

        float4 e0, e1;
        row_fetch(sh, 0, posType, e0, e1);
        row_store(sh, 0, e0, e1);
        __syncthreads();  // row 0 and cellL are in LDS
        for (int r = 0; r < ((ablate & 8) ? 0 : 9); r++) {
            const int buf = r & 1;
            if (r + 1 < 9) row_fetch(sh, r + 1, posType, e0, e1);  // in flight during the cull below
            int lj = 0, le = 0;
            if (active) {
                const int c = walk.row_cell(P, r % 3 - 1, r / 3 - 1);
                if (c >= 0) {
                    const int rel = r * kCellCap + (c - sh.rowCell0[r]);
                    lj = sh.cellL[rel];
                    le = sh.cellL[rel + (walk.xhi - walk.xlo) + 1];
                }
            }
            const float4 *__restrict__ row = sh.row[buf];
            const int gofs = sh.rowStartG[r];
            if (ablate & 4) lj = le;
            while (wave_any(lj < le)) {
                const int base = lj < le ? lj : ROWCAP;  // finished lanes read the far-away padding
                float4 c4[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) c4[u] = row[base + u];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const float xx = pi.x - c4[u].x, xy = pi.y - c4[u].y, xz = pi.z - c4[u].z;
                    const float d2 = FAST ? __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx))
                                          : xx * xx + xy * xy + xz * xz;
                    const int tj = unpack_int(c4[u].w);
                    if (d2 < sr2 && !(ablate & 2)) {
                        if (tj == FSA_TYPE_FLUID) {
                            q[cnt * BS] = d2;
                            cnt++;
                        } else if (tj == FSA_TYPE_RIGID) {
                            q[cnt * BS] = __int_as_float(-(gofs + base + u) - 1);
                            q[(cnt + 1) * BS] = d2;
                            cnt += 2;
                        }
                    }
                }
                lj += kChunk;
                if (wave_any(cnt > Q - 2 * kChunk)) drain();
            }
            if (r + 1 < 9) {
                row_store(sh, buf ^ 1, e0, e1);
                __syncthreads();  // row r + 1 visible; everyone is done with row r - 1's buffer
            }
        }
        drain();
    }
    if (active) {
        density[index] = dens;
        float pr = P.pressureK * (r_pow(dens / rd_i, 7.0f) - 1);
        if (pr < 0) pr = 0;
        pressure[index] = pr;
    }
}

// ------------------------------------------------------------------------------------------------
// force
// ------------------------------------------------------------------------------------------------
// Per-particle terms of the force pass, computed once instead of once per pair:
//   x = pressure / density / density (the reference's two divisions, in that order)
//   y = density (EXACT)   or   mass / density * 2 * viscosity (FAST)
template <bool FAST>
__global__ __launch_bounds__(kBlock) void force_prep_kernel(int n, float viscosity,
                                                            const float4 *__restrict__ velMass,
                                                            const float *__restrict__ density,
                                                            const float *__restrict__ pressure,
                                                            float2 *__restrict__ pd) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float d = density[i];
    const float a = pressure[i] / d / d;
    pd[i] = make_float2(a, FAST ? velMass[i].w / d * 2.0f * viscosity : d);
}

template <int BS, int ROWCAP, int Q, bool FAST>
__global__ __launch_bounds__(BS) void force_tuned_kernel(DevParams<float> P, int n,
                                                         const float4 *__restrict__ posType,
                                                         const float4 *__restrict__ velMass,
                                                         const float *__restrict__ density,
                                                         const float *__restrict__ pressure,
                                                         const float2 *__restrict__ pd,
                                                         const int *__restrict__ cellStart,
                                                         const int *__restrict__ sortedHash,
                                                         Vec3<float> *__restrict__ force, int ablate) {
    __shared__ RowShared<BS, ROWCAP> sh;
    __shared__ unsigned short queue[Q * BS];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * BS;
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;

    ForceSelf<float> S;
    S.p = make_float4(0.f, 0.f, 0.f, 0.f);
    S.v = S.p;
    S.pterm = 0.f;
    S.fx = S.fy = S.fz = 0.f;
    bool active = index < n;
    if (active) {
        S.p = posType[index];
        active = unpack_int(S.p.w) == FSA_TYPE_FLUID;  // other types keep force[index] as it is
    }
    // a block without fluid particles (wall layers) has nothing to do
    if (!__syncthreads_or(active ? 1 : 0)) return;
    const bool staged = setup_rows<BS, ROWCAP>(P, b0, nb, cellStart, sortedHash, sh);
    if (active) {
        S.v = velMass[index];
        S.pterm = pd[index].x;
    }
    const float sr = P.sr, sr2 = sr * sr;
    if (!staged) {
        if (active) force_walk_global(P, index, S, posType, velMass, density, pressure, cellStart);
    } else {
        FastConst K;
        K.sr = sr; K.sr2 = sr2; K.inv_h = 0.f; K.kcubic = 0.f; K.kspiky = P.kspikydiff;
        K.visc_eps = 0.01f * sr2;
        const float sr2c = FAST ? sr2 : sr2 * 1.000002f;
        const RowWalk<float> walk(P, S.p.x, S.p.y, S.p.z);
        int cnt = 0;

        auto evaluate = [&](int j, const float4 &pj, const float4 &vj, const float2 &aj) {
            const int tj = unpack_int(pj.w);
            if (FAST && tj == FSA_TYPE_FLUID) {
                const float xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
                const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                if (d2 < sr2 && j != index) {
                    const float rinv = __builtin_amdgcn_rsqf(d2);
                    const float c = sr - d2 * rinv;
                    const float nablaw = K.kspiky * c * c * rinv;
                    const float pc = nablaw * vj.w * (aj.x + S.pterm);
                    S.fx = __builtin_fmaf(-xx, pc, S.fx);
                    S.fy = __builtin_fmaf(-xy, pc, S.fy);
                    S.fz = __builtin_fmaf(-xz, pc, S.fz);
                    const float vc = nablaw * d2 * __builtin_amdgcn_rcpf(d2 + K.visc_eps) * aj.y;
                    S.fx = __builtin_fmaf(S.v.x - vj.x, vc, S.fx);
                    S.fy = __builtin_fmaf(S.v.y - vj.y, vc, S.fy);
                    S.fz = __builtin_fmaf(S.v.z - vj.z, vc, S.fz);
                }
            } else {
                const float xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
                const float d2 = xx * xx + xy * xy + xz * xz;
                if (force_pair_active(tj, j == index, d2, sr2)) {
                    // the boundary branch needs no per-particle term; the fluid branch only runs here for EXACT
                    force_pair(P, sr, sr2, S, j == index, pj, vj, aj.x, FAST ? 1.0f : aj.y);
                }
            }
        };
        const int safe = index < n ? index : 0;
        auto drain = [&]() {
            if (ablate & 1) { cnt = 0; return; }
            // four neighbours per trip: their 12 loads are in flight together before any is evaluated
            for (int s = 0; wave_any(s < cnt); s += kChunk) {
                int j[kChunk];
                float4 pj[kChunk], vj[kChunk];
                float2 aj[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const int e = queue[(s + u) * BS + tid];
                    j[u] = s + u < cnt ? sh.rowStartG[e >> 9] + (e & 511) : safe;
                }
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    pj[u] = posType[j[u]];
                    vj[u] = velMass[j[u]];
                    aj[u] = pd[j[u]];
                }
#pragma unroll
                for (int u = 0; u < kChunk; u++)
                    if (s + u < cnt) evaluate(j[u], pj[u], vj[u], aj[u]);
            }
            cnt = 0;
        };

        float4 e0, e1;
        row_fetch(sh, 0, posType, e0, e1);
        row_store(sh, 0, e0, e1);
        __syncthreads();
        for (int r = 0; r < 9; r++) {
            const int buf = r & 1;
            if (r + 1 < 9) row_fetch(sh, r + 1, posType, e0, e1);
            int lj = 0, le = 0;
            if (active) {
                const int c = walk.row_cell(P, r % 3 - 1, r / 3 - 1);
                if (c >= 0) {
                    const int rel = r * kCellCap + (c - sh.rowCell0[r]);
                    lj = sh.cellL[rel];
                    le = sh.cellL[rel + (walk.xhi - walk.xlo) + 1];
                }
            }
            const float4 *__restrict__ row = sh.row[buf];
            const int tag = r << 9;
            if (ablate & 4) lj = le;
            while (wave_any(lj < le)) {
                const int base = lj < le ? lj : ROWCAP;
                float4 c4[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) c4[u] = row[base + u];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const float xx = S.p.x - c4[u].x, xy = S.p.y - c4[u].y, xz = S.p.z - c4[u].z;
                    const float d2c = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                    const int tj = unpack_int(c4[u].w);
                    // fluid neighbours inside the support; boundary neighbours of the lane's own cells always
                    if (((d2c < sr2c && tj == FSA_TYPE_FLUID) || (tj == FSA_TYPE_BOUNDARY && base + u < le)) &&
                        !(ablate & 2)) {
                        queue[cnt * BS + tid] = (unsigned short)(tag | (base + u));
                        cnt++;
                    }
                }
                lj += kChunk;
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            if (r + 1 < 9) {
                row_store(sh, buf ^ 1, e0, e1);
                __syncthreads();
            }
        }
        drain();
    }
    if (active) {
        Vec3<float> f;
        f.x = S.fx + 0.0f;
        f.y = S.fy + (float)-9.8;  // the literal of solver/sph_kernel_impl.cuh:372
        f.z = S.fz + 0.0f;
        force[index] = f;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
constexpr int kTunedBS = 256;
constexpr int kRowCap = 384;  // typical row: 34 cells x 8 particles = 272
constexpr int kQueueD = 24;  // density: 4-byte slots
constexpr int kQueueF = 32;  // force: 2-byte slots

static int g_ablate = 0;         // timing experiments only (fsa_debug_time_pass)
static void *g_out0 = nullptr;   // when set, results go to these scratch arrays instead
static void *g_out1 = nullptr;

int launch_density_tuned(Solver *s) {
    if (s->n == 0) return FSA_OK;
    const int ablate = g_ablate;
    ScopedTimer t(s, g_out0 ? "density_dbg" : "density");
    const int c = s->cur;
    const int nb = (s->n + kTunedBS - 1) / kTunedBS;
    if (s->variant == 2)
        hipLaunchKernelGGL((density_tuned_kernel<kTunedBS, kRowCap, kQueueD, true>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.restDensity[c], s->b.cellStart, s->b.sortedHash,
                           (float *)(g_out0 ? g_out0 : s->b.density), (float *)(g_out1 ? g_out1 : s->b.pressure),
                           ablate);
    else
        hipLaunchKernelGGL((density_tuned_kernel<kTunedBS, kRowCap, kQueueD, false>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.restDensity[c], s->b.cellStart, s->b.sortedHash,
                           (float *)(g_out0 ? g_out0 : s->b.density), (float *)(g_out1 ? g_out1 : s->b.pressure),
                           ablate);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_force_tuned(Solver *s) {
    if (s->n == 0) return FSA_OK;
    const int c = s->cur;
    const bool fast = s->variant == 2;
    {
        ScopedTimer t(s, "force_prep");
        const dim3 g((s->n + kBlock - 1) / kBlock), b(kBlock);
        if (fast)
            hipLaunchKernelGGL(force_prep_kernel<true>, g, b, 0, s->stream, s->n, s->pf.viscosity,
                               (const float4 *)s->b.velMass[c], (const float *)s->b.density,
                               (const float *)s->b.pressure, (float2 *)s->b.pd);
        else
            hipLaunchKernelGGL(force_prep_kernel<false>, g, b, 0, s->stream, s->n, s->pf.viscosity,
                               (const float4 *)s->b.velMass[c], (const float *)s->b.density,
                               (const float *)s->b.pressure, (float2 *)s->b.pd);
    }
    ScopedTimer t(s, g_out0 ? "force_dbg" : "force");
    const int nb = (s->n + kTunedBS - 1) / kTunedBS;
    const int ablate = g_ablate;
    if (fast)
        hipLaunchKernelGGL((force_tuned_kernel<kTunedBS, kRowCap, kQueueF, true>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.density, (const float *)s->b.pressure, (const float2 *)s->b.pd,
                           s->b.cellStart, s->b.sortedHash, (Vec3<float> *)(g_out0 ? g_out0 : s->b.force), ablate);
    else
        hipLaunchKernelGGL((force_tuned_kernel<kTunedBS, kRowCap, kQueueF, false>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.density, (const float *)s->b.pressure, (const float2 *)s->b.pd,
                           s->b.cellStart, s->b.sortedHash, (Vec3<float> *)(g_out0 ? g_out0 : s->b.force), ablate);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// Launches one pass `reps` times with parts of the kernel switched off and the results sent to
// scratch arrays, so that the solver state is untouched.  Timing experiments only.
int debug_time_pass(Solver *s, int which, int ablate, int reps) {
    g_ablate = ablate;
    int rc = FSA_OK;
    for (int k = 0; k < reps && rc == FSA_OK; k++) {
        if (which == 0) {
            g_out0 = s->b.tmpIndex;
            g_out1 = s->b.tmpHash;
            rc = launch_density_tuned(s);
        } else {
            g_out0 = s->b.v_star;
            rc = launch_force_tuned(s);
        }
    }
    g_ablate = 0;
    g_out0 = g_out1 = nullptr;
    return rc;
}

}  // namespace fsa
