// Tuned fp32 density and force passes for gfx950: LDS-staged cell tiles + per-lane hit queues.
//
// Same pair arithmetic, in the same order, as the generic kernels (fsa_pair.h); only the way the
// neighbours are found and fetched differs:
//
//  * A workgroup owns BS consecutive particles of the sorted arrays, i.e. a run of cells
//    [hlo, hhi] in hash order.  For each of the 9 (dy, dz) row offsets the neighbour cells of all
//    of them form ONE contiguous cell range [hlo + off - 1, hhi + off + 1], hence one contiguous
//    particle range.  Those 9 ranges of {x, y, z, type} are copied into LDS once per workgroup
//    with coalesced 16-byte loads (the "tile"), together with the matching slices of the cell
//    start table rebased to tile indices.
//  * Phase 1 (cull): every lane walks its own 9 index ranges over the tile, four ds_read_b128 in
//    flight, and appends the neighbours that pass a slightly widened fused-multiply-add distance
//    test to its private queue in LDS (queue[slot][lane]: conflict-free).
//  * Phase 2 (evaluate): when a queue nears its capacity, and at the end, the wavefront drains
//    the queues: every lane evaluates its queued neighbours with the exact pair routine (which
//    repeats the exact, unfused range test), so the expensive arithmetic runs with most lanes
//    busy instead of one lane in seven.  Queues are first-in first-out: each particle's sum is
//    accumulated in the reference's neighbour order.
//
// Workgroups whose rows do not fit the tile (very sparse or very crowded regions) or that hold
// out-of-grid particles take the generic global-memory walk; results are identical either way.
#include "fsa_pair.h"
#include "fsa_solver.h"

namespace fsa {

constexpr int kChunk = 4;        // candidates fetched per lane and loop trip
constexpr int kCellCap = 96;     // cell-start entries staged per row

template <int BS, int TILE> struct TileShared {
    float4 tile[TILE];
    int cellL[9 * kCellCap];  // cell starts of the 9 rows as tile indices
    int rowStartG[9];         // first particle (global index) of each staged row
    int rowLen[9];
    int rowBaseL[10];         // first tile index of each row
    int rowCell0[9];          // first cell of each staged row
    int staged;
};

// Fills the tile.  Returns false (block-uniform) when this block must take the generic path.
template <int BS, int TILE>
__device__ __forceinline__ bool stage_tile(const DevParams<float> &P, int b0, int nb,
                                           const int *__restrict__ cellStart,
                                           const int *__restrict__ sortedHash,
                                           const float4 *__restrict__ posType, TileShared<BS, TILE> &sh) {
    const int tid = threadIdx.x;
    const unsigned hlo = (unsigned)sortedHash[b0], hhi = (unsigned)sortedHash[b0 + nb - 1];
    const int C = P.num_cells;
    const bool grid_ok = hhi != (unsigned)FSA_GRID_UNDEF && (long long)hhi - (long long)hlo + 4 <= kCellCap;
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
        int base = 0;
        bool ok = grid_ok;
        for (int r = 0; r < 9; r++) {
            sh.rowBaseL[r] = base;
            if (sh.rowLen[r] < 0) ok = false;
            else base += sh.rowLen[r];
        }
        sh.rowBaseL[9] = base;
        sh.staged = (ok && base <= TILE) ? 1 : 0;
    }
    __syncthreads();
    if (!sh.staged) return false;
    // cell-start slices, rebased to tile indices
    const int ncell = (int)(hhi - hlo) + 4;  // entries per row: cells hlo-1 .. hhi+2
    for (int t = tid; t < 9 * ncell; t += BS) {
        const int r = t / ncell, k = t - r * ncell;
        long long c = (long long)sh.rowCell0[r] + k;
        if (c > C) c = C;
        sh.cellL[r * kCellCap + k] = cellStart[c] - sh.rowStartG[r] + sh.rowBaseL[r];
    }
    // particle tile
    const int total = sh.rowBaseL[9];
    for (int t = tid; t < total; t += BS) {
        int r = 0;
#pragma unroll
        for (int q = 1; q < 9; q++) r += (t >= sh.rowBaseL[q]) ? 1 : 0;
        sh.tile[t] = posType[sh.rowStartG[r] + (t - sh.rowBaseL[r])];
    }
    __syncthreads();
    return true;
}

// tile index -> global particle index (only needed on rare branches)
template <int BS, int TILE> __device__ __forceinline__ int tile_to_global(const TileShared<BS, TILE> &sh, int li) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < 9; q++) r += (li >= sh.rowBaseL[q]) ? 1 : 0;
    return sh.rowStartG[r] + (li - sh.rowBaseL[r]);
}

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// widened cull threshold: the fused distance differs from the unfused one by a few ulp at most
__device__ __forceinline__ float cull_radius2(float sr2) { return sr2 * 1.000002f; }

// ------------------------------------------------------------------------------------------------
// density + pressure
// ------------------------------------------------------------------------------------------------
template <int BS, int TILE, int Q>
__global__ __launch_bounds__(BS) void density_tuned_kernel(DevParams<float> P, int n,
                                                           const float4 *__restrict__ posType,
                                                           const float4 *__restrict__ velMass,
                                                           const float *__restrict__ restDensity,
                                                           const int *__restrict__ cellStart,
                                                           const int *__restrict__ sortedHash,
                                                           float *__restrict__ density,
                                                           float *__restrict__ pressure) {
    __shared__ TileShared<BS, TILE> sh;
    __shared__ unsigned short queue[Q * BS];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * BS;
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const bool active = index < n;
    const bool staged = stage_tile<BS, TILE>(P, b0, nb, cellStart, sortedHash, posType, sh);

    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    float mass_i = 0.f, rd_i = 1.f;
    if (active) {
        pi = posType[index];
        mass_i = velMass[index].w;
        rd_i = restDensity[index];
    }
    float dens = 0.f;
    if (!staged) {
        if (active) dens = density_walk_global(P, index, pi, mass_i, rd_i, posType, restDensity, cellStart);
    } else {
        const float sr = P.sr, sr2 = sr * sr, sr2c = cull_radius2(sr2);
        const float h = (float)((double)sr * 0.5);
        const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
        int cnt = 0;
        auto drain = [&]() {
            for (int s = 0; wave_any(s < cnt); s++) {
                if (s < cnt) {
                    const int li = queue[s * BS + tid];
                    const float4 pj = sh.tile[li];
                    const float xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
                    const float d2 = xx * xx + xy * xy + xz * xz;
                    if (d2 < sr2) {
                        const float wij = kernel_cubic(P, h, xx, xy, xz);
                        const int tj = unpack_int(pj.w);
                        if (tj == FSA_TYPE_FLUID) dens += mass_i * wij;
                        if (tj == FSA_TYPE_RIGID) dens += rd_i * restDensity[tile_to_global(sh, li)] * wij;
                    }
                }
            }
            cnt = 0;
        };
        for (int r = 0; r < 9; r++) {
            int lj = 0, le = 0;
            if (active) {
                const int c = walk.row_cell(P, r % 3 - 1, r / 3 - 1);
                if (c >= 0) {
                    const int rel = r * kCellCap + (c - sh.rowCell0[r]);
                    lj = sh.cellL[rel];
                    le = sh.cellL[rel + (walk.xhi - walk.xlo) + 1];
                }
            }
            while (wave_any(lj < le)) {
                float4 c4[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) c4[u] = sh.tile[(lj + u < le) ? lj + u : 0];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    if (lj + u < le) {
                        const float xx = pi.x - c4[u].x, xy = pi.y - c4[u].y, xz = pi.z - c4[u].z;
                        const float d2c = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                        if (d2c < sr2c) {
                            queue[cnt * BS + tid] = (unsigned short)(lj + u);
                            cnt++;
                        }
                    }
                }
                lj += kChunk;
                if (wave_any(cnt > Q - kChunk)) drain();
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
// {pressure / density / density, density} per particle, so that the force pass reads 8 bytes per
// neighbour instead of dividing twice per pair (same two divisions, done once per particle).
__global__ __launch_bounds__(kBlock) void force_prep_kernel(int n, const float *__restrict__ density,
                                                            const float *__restrict__ pressure,
                                                            float2 *__restrict__ pd) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float d = density[i];
    pd[i] = make_float2(pressure[i] / d / d, d);
}

template <int BS, int TILE, int Q>
__global__ __launch_bounds__(BS) void force_tuned_kernel(DevParams<float> P, int n,
                                                         const float4 *__restrict__ posType,
                                                         const float4 *__restrict__ velMass,
                                                         const float *__restrict__ density,
                                                         const float *__restrict__ pressure,
                                                         const float2 *__restrict__ pd,
                                                         const int *__restrict__ cellStart,
                                                         const int *__restrict__ sortedHash,
                                                         Vec3<float> *__restrict__ force) {
    __shared__ TileShared<BS, TILE> sh;
    __shared__ int queue[Q * BS];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * BS;
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const bool staged = stage_tile<BS, TILE>(P, b0, nb, cellStart, sortedHash, posType, sh);

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
    if (active) {
        S.v = velMass[index];
        S.pterm = pd[index].x;
    }
    const float sr = P.sr, sr2 = sr * sr;
    if (!staged) {
        if (active) force_walk_global(P, index, S, posType, velMass, density, pressure, cellStart);
    } else {
        const float sr2c = cull_radius2(sr2);
        const RowWalk<float> walk(P, S.p.x, S.p.y, S.p.z);
        const int safe = index < n ? index : 0;
        int cnt = 0;
        auto drain = [&]() {
            // software-pipelined: the loads of entry s + 1 are issued before entry s is evaluated
            bool has = cnt > 0;
            int j = has ? queue[tid] : safe;
            float4 pj = posType[j], vj = velMass[j];
            float2 aj = pd[j];
            for (int s = 0; wave_any(has); s++) {
                const bool hasn = s + 1 < cnt;
                const int jn = hasn ? queue[(s + 1) * BS + tid] : safe;
                const float4 pjn = posType[jn], vjn = velMass[jn];
                const float2 ajn = pd[jn];
                if (has) {
                    const float xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
                    const float d2 = xx * xx + xy * xy + xz * xz;
                    if (force_pair_active(unpack_int(pj.w), j == index, d2, sr2))
                        force_pair(P, sr, sr2, S, j == index, pj, vj, aj.x, aj.y);
                }
                has = hasn;
                j = jn;
                pj = pjn;
                vj = vjn;
                aj = ajn;
            }
            cnt = 0;
        };
        for (int r = 0; r < 9; r++) {
            int lj = 0, le = 0;
            if (active) {
                const int c = walk.row_cell(P, r % 3 - 1, r / 3 - 1);
                if (c >= 0) {
                    const int rel = r * kCellCap + (c - sh.rowCell0[r]);
                    lj = sh.cellL[rel];
                    le = sh.cellL[rel + (walk.xhi - walk.xlo) + 1];
                }
            }
            const int gofs = sh.rowStartG[r] - sh.rowBaseL[r];  // tile index -> global index in this row
            while (wave_any(lj < le)) {
                float4 c4[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) c4[u] = sh.tile[(lj + u < le) ? lj + u : 0];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    if (lj + u < le) {
                        const float xx = S.p.x - c4[u].x, xy = S.p.y - c4[u].y, xz = S.p.z - c4[u].z;
                        const float d2c = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                        const int tj = unpack_int(c4[u].w);
                        const int j = lj + u + gofs;
                        if ((tj == FSA_TYPE_FLUID && d2c < sr2c && j != index) || tj == FSA_TYPE_BOUNDARY) {
                            queue[cnt * BS + tid] = j;
                            cnt++;
                        }
                    }
                }
                lj += kChunk;
                if (wave_any(cnt > Q - kChunk)) drain();
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
constexpr int kTunedTile = 2560;  // 40 KiB of float4
constexpr int kQDensity = 48;
constexpr int kQForce = 32;

int launch_density_tuned(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "density");
    const int c = s->cur;
    const int nb = (s->n + kTunedBS - 1) / kTunedBS;
    hipLaunchKernelGGL((density_tuned_kernel<kTunedBS, kTunedTile, kQDensity>), dim3(nb), dim3(kTunedBS), 0, s->stream,
                       s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                       (const float *)s->b.restDensity[c], s->b.cellStart, s->b.sortedHash, (float *)s->b.density,
                       (float *)s->b.pressure);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_force_tuned(Solver *s) {
    if (s->n == 0) return FSA_OK;
    const int c = s->cur;
    {
        ScopedTimer t(s, "force_prep");
        hipLaunchKernelGGL(force_prep_kernel, dim3((s->n + kBlock - 1) / kBlock), dim3(kBlock), 0, s->stream, s->n,
                           (const float *)s->b.density, (const float *)s->b.pressure, (float2 *)s->b.pd);
    }
    ScopedTimer t(s, "force");
    const int nb = (s->n + kTunedBS - 1) / kTunedBS;
    hipLaunchKernelGGL((force_tuned_kernel<kTunedBS, kTunedTile, kQForce>), dim3(nb), dim3(kTunedBS), 0, s->stream, s->pf,
                       s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                       (const float *)s->b.density, (const float *)s->b.pressure, (const float2 *)s->b.pd,
                       s->b.cellStart, s->b.sortedHash, (Vec3<float> *)s->b.force);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

}  // namespace fsa
