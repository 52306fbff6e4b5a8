// Tuned fp32 density and force passes for gfx950: LDS-staged cell-row tiles + per-lane hit queues.
//
// A workgroup owns BS consecutive particles of the sorted arrays, i.e. a run of cells [hlo, hhi]
// in hash order.  For each of the 9 (dy, dz) row offsets the neighbour cells of all of them form
// ONE contiguous cell range [hlo + off - 1, hhi + off + 1], hence one contiguous particle range
// ("row").  The kernel walks the 9 rows in the reference's order (z outer, y inner):
//
//   * the row's {x, y, z, type} are copied into a small LDS buffer with coalesced 16-byte loads and
//     stored there as separate x / y / z / type arrays; two buffers alternate, the global loads of
//     row r + 1 (and of the lane's cell range in it) are issued before row r is examined and land
//     after it, so their latency is hidden;
//   * phase 1 (cull): every lane walks its own index range of the row in aligned groups of four
//     candidates (one ds_read_b128 per coordinate gives four candidates, straight-line code, no
//     divergent branches; the first and the last group of a range are masked) and appends those
//     that pass the distance test to its private queue in LDS (queue[slot][lane]: conflict-free).
//     The append is branch-free: all four candidates are stored, each at the slot counted up by
//     the hits before it, so a miss is simply overwritten by the next store;
//   * phase 2 (evaluate): when a queue nears its capacity, and after the last row, the wavefront
//     drains the queues four entries per trip, so the expensive arithmetic runs with most lanes
//     busy.  Queues are first-in first-out: each particle's sum is accumulated in the reference's
//     neighbour order.
//
// LDS per workgroup is 20 KiB (density, eight workgroups per CU) and 39 KiB (force, four): the VALU
// of gfx950 needs about four wavefronts per SIMD to issue at its full rate.
//
// Two arithmetic flavours share this structure:
//   EXACT: the expressions of fsa_pair.h / fsa_device.h, operation for operation the reference's.
//   FAST : fp32 only, v_sqrt/v_rsq/v_rcp and fused multiply-adds; differs from EXACT by a few
//          units in the last place per pair (about 1e-7 relative in density and force).
//
// Workgroups whose rows do not fit the buffers (very crowded regions) or that hold out-of-grid
// particles take the generic global-memory walk with the exact routines.
#include <cstdlib>
#include <type_traits>

#include "fsa_fast.h"
#include "fsa_pair.h"
#include "fsa_solver.h"
#include "fsa_tile.h"

namespace fsa {


// ------------------------------------------------------------------------------------------------
// density + pressure.  Scenes with TYPE_RIGID particles use the generic kernel (the rigid term
// needs the neighbour's restDensity); here only fluid neighbours count, so the queue holds nothing
// but their squared distances.
// ------------------------------------------------------------------------------------------------
template <int BS, int ROWCAP, int Q, bool FAST>
__global__ __launch_bounds__(BS) void density_tuned_kernel(DevParams<float> P, int n,
                                                           const float4 *__restrict__ posType,
                                                           const float4 *__restrict__ velMass,
                                                           const float *__restrict__ restDensity,
                                                           const int *__restrict__ cellStart,
                                                           const int *__restrict__ sortedHash,
                                                           float *__restrict__ density,
                                                           float *__restrict__ pressure,
                                                           float2 *__restrict__ pd, int ablate) {
    using Shared = RowShared<BS, ROWCAP, false>;
    __shared__ Shared sh;
    __shared__ Segments sg;
    __shared__ float queue[Q * BS];
    const int tid = threadIdx.x;
    const int b0 = xcd_block(ablate >> 8) * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const bool active = index < n;
    // the cells of the block's first and last particle: only needed without a row table, and in slab runs,
    // where a tile that lies entirely in the outer halo layer feeds nobody
    unsigned hlo = 0, hhi = 0;
    if (!P.rows || P.nz != P.gres[2]) {
        hlo = (unsigned)sortedHash[b0];
        hhi = (unsigned)sortedHash[b0 + nb - 1];
        if (hhi != (unsigned)FSA_GRID_UNDEF && run_unneeded(P, hlo, hhi, 1)) return;
    }
    // class 2 (rows beyond a tile, up to kRowCapBig): the long rows are walked as two tiles each
    const int cls = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, hlo, hhi, P.rows);
    const bool staged = cls != 0 && !(ablate & 128);
    if (staged) setup_segments<ROWCAP>(sh, cls, sg);

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
            dens = FAST ? density_walk_global_fast(P, pi, mass_i, posType, cellStart)
                        : density_walk_global(P, index, pi, mass_i, rd_i, posType, restDensity, cellStart);
    } else {
        const float sr = P.sr, sr2 = sr * sr;
        const float h = (float)((double)sr * 0.5);
        const float inv_h = 1.0f / h, kcubic = P.kernel_cubic;
        const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
        float *const q = queue + tid;
        int cnt = 0;
        auto drain = [&]() {
            if (ablate & 1) { cnt = 0; return; }
            for (int s = 0; wave_any(s < cnt); s += kChunk) {
                float d2[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) d2[u] = q[(s + u) * BS];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    if (FAST) {
                        const float w = s + u < cnt ? cubic_fast(d2[u], inv_h, kcubic) : 0.0f;
                        dens = __builtin_fmaf(mass_i, w, dens);
                    } else if (s + u < cnt) {
                        dens += mass_i * kernel_cubic_d2(P, h, d2[u]);
                    }
                }
            }
            cnt = 0;
        };

        // the lane's index range inside a segment: its range in the row, cut to the segment
        auto seg_range = [&](int row, int off, int len, int &a, int &b) {
            lane_range(P, sh, walk, active, row, cellStart, a, b);
            a = min(max(a - off, 0), len);
            b = min(max(b - off, 0), len);
        };
        float4 e[row_regs<BS, ROWCAP>()];
        int lj, le, nlj = 0, nle = 0;
        const int len0 = min(ROWCAP, sh.rowLen[0]);  // the first segment is the head of row 0
        tile_fetch<BS, ROWCAP, true>(sh.rowStartG[0], len0, posType, e);
        seg_range(0, 0, len0, lj, le);
        row_store<BS, ROWCAP, false>(sh, 0, e);
        __syncthreads();
        const int nseg = (ablate & 8) ? 0 : sg.count;
        for (int r = 0; r < nseg; r++) {
            const int buf = r & 1;
            if (r + 1 < nseg) {  // in flight during the cull below
                const int row = sg.row[r + 1], st = sg.start[r + 1], len = sg.len[r + 1];
                tile_fetch<BS, ROWCAP, true>(st, len, posType, e);
                seg_range(row, st - sh.rowStartG[row], len, nlj, nle);
            }
            if (ablate & 4) lj = le;
            // One aligned group of four candidates; those with lo <= u < hi belong to the lane.
            auto cull_group = [&](int g, int lo, int hi, auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                float d2[kChunk];
                group_d2<FAST>(sh.x[buf][g], sh.y[buf][g], sh.z[buf][g], pi.x, pi.y, pi.z, d2);
                if (ablate & 2) return;
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    q[cnt * BS] = d2[u];  // a miss is overwritten by the next store
                    cnt += (d2[u] < sr2 && (!MASKED || (u >= lo && u < hi))) ? 1 : 0;
                }
            };
            const bool has = lj < le;
            int base = lj & ~(kChunk - 1);
            if (wave_any(has)) {  // first group of the range: candidates below lj belong to other cells
                cull_group(has ? base / kChunk : Shared::kPad, lj - base, le - base, std::true_type());
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            base = has ? base + kChunk : le;
            while (wave_any(base + kChunk <= le)) {  // whole groups
                const bool go = base + kChunk <= le;
                cull_group(go ? base / kChunk : Shared::kPad, 0, kChunk, std::false_type());
                base += go ? kChunk : 0;
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            if (wave_any(base < le)) {  // last group: what follows the range in it belongs to other cells
                const int rem = le - base;
                cull_group(rem > 0 ? base / kChunk : Shared::kPad, 0, rem, std::true_type());
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            if (r + 1 < nseg) {
                row_store<BS, ROWCAP, false>(sh, buf ^ 1, e);
                lj = nlj;
                le = nle;
                __syncthreads();  // segment r + 1 visible; everyone is done with the buffer of segment r - 1
            }
        }
        drain();
    }
    if (active) {
        density[index] = dens;
        float pr = P.pressureK * (r_pow7(dens / rd_i) - 1);
        if (pr < 0) pr = 0;
        pressure[index] = pr;
        // the per-particle terms of the force pass (what force_prep_kernel computes), while they are at hand
        if (pd) pd[index] = make_float2(pr / dens / dens, FAST ? mass_i / dens * 2.0f * P.viscosity : dens);
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
    using Shared = RowShared<BS, ROWCAP, true>;
    __shared__ Shared sh;
    __shared__ Segments sg;
    __shared__ unsigned short queue[(Q + 2 * kDrainBatch) * BS];  // the drain reads ahead of the last entry
    const int tid = threadIdx.x;
    const int b0 = xcd_block(ablate >> 8) * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    // the cells of the block's first and last particle: only needed without a row table, and in slab runs,
    // where forces are only needed for the layers the solver owns
    unsigned hlo = 0, hhi = 0;
    if (!P.rows || P.nz != P.gres[2]) {
        hlo = (unsigned)sortedHash[b0];
        hhi = (unsigned)sortedHash[b0 + nb - 1];
        if (hhi != (unsigned)FSA_GRID_UNDEF && run_unneeded(P, hlo, hhi, 0)) return;
    }

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
    const int cls = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, hlo, hhi, P.rows);
    const bool staged = cls != 0 && !(ablate & 128);
    if (staged) setup_segments<ROWCAP>(sh, cls, sg);  // long rows as two tiles each, see the density kernel
    if (active) {
        S.v = velMass[index];
        S.pterm = pd[index].x;
    }
    const float sr = P.sr, sr2 = sr * sr;
    // FAST: wall neighbours are summed on their own (Sb) and added at the end, so the result does not
    // depend on how the queue happened to be cut into batches
    ForceSelf<float> Sb = S;
    if (!staged) {
        if (active && !(ablate & 32)) {
            if (FAST)
                force_walk_global_fast(P, index, S, Sb, posType, velMass, pd, cellStart);
            else
                force_walk_global(P, index, S, posType, velMass, density, pressure, cellStart);
        }
    } else {
        const float kspiky = P.kspikydiff, visc_eps = 0.01f * sr2;
        const WallConsts W(P);
        const float sr2c = FAST ? sr2 : sr2 * 1.000002f;  // the unfused test of EXACT comes later
        const RowWalk<float> walk(P, S.p.x, S.p.y, S.p.z);
        unsigned short *const q = queue + tid;
        const int safe = index < n ? index : 0;
        int cnt = 0;

        // exact routine: TYPE_BOUNDARY neighbours always, fluid neighbours in the EXACT flavour
        auto evaluate_exact = [&](int j, const float4 &pj, const float4 &vj, const float2 &aj) {
            const float xx = S.p.x - pj.x, xy = S.p.y - pj.y, xz = S.p.z - pj.z;
            const float d2 = xx * xx + xy * xy + xz * xz;
            if (force_pair_active(unpack_int(pj.w), j == index, d2, sr2))
                force_pair(P, sr, sr2, FAST ? Sb : S, j == index, pj, vj, aj.x, FAST ? 1.0f : aj.y);
        };
        // One batch of the drain: DB neighbours whose loads are issued together.
        struct Batch {
            int j[kDrainBatch];
            float4 pj[kDrainBatch], vj[kDrainBatch];
            float2 aj[kDrainBatch];
        };
        auto batch_load = [&](Batch &B, int s) {
#pragma unroll
            for (int u = 0; u < kDrainBatch; u++) {
                const int e = q[(s + u) * BS];
                B.j[u] = (s + u < cnt && !(ablate & 64)) ? sg.start[e >> 9] + (e & 511) : safe;
            }
#pragma unroll
            for (int u = 0; u < kDrainBatch; u++) {
                B.pj[u] = posType[B.j[u]];
                B.vj[u] = velMass[B.j[u]];
                B.aj[u] = pd[B.j[u]];
            }
        };
        auto batch_eval = [&](const Batch &B, int s) {
            if (FAST) {
                bool other = false;
#pragma unroll
                for (int u = 0; u < kDrainBatch; u++) {
                    // straight-line: an entry that does not count (slot past the end, self, out of
                    // range, not a fluid particle) is evaluated anyway and then zeroed by selects
                    const float xx = S.p.x - B.pj[u].x, xy = S.p.y - B.pj[u].y, xz = S.p.z - B.pj[u].z;
                    const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                    const int tj = unpack_int(B.pj[u].w);
                    const bool ok = d2 < sr2 && B.j[u] != index && tj == FSA_TYPE_FLUID;
                    other |= tj != FSA_TYPE_FLUID && s + u < cnt;
                    force_pair_fast(S, sr, kspiky, visc_eps, xx, xy, xz, d2, ok, B.vj[u], B.aj[u]);
                }
                if (wave_any(other)) {  // wall particles among them: summed apart, see wall_pair_fast
#pragma unroll
                    for (int u = 0; u < kDrainBatch; u++) {
                        const float xx = S.p.x - B.pj[u].x, xy = S.p.y - B.pj[u].y, xz = S.p.z - B.pj[u].z;
                        const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                        const bool wl = s + u < cnt && unpack_int(B.pj[u].w) == FSA_TYPE_BOUNDARY;
                        wall_pair_fast(Sb, W, sr, kspiky, visc_eps, xx, xy, xz, d2, wl, B.vj[u].w, B.vj[u].x, B.vj[u].y,
                                       B.vj[u].z);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kDrainBatch; u++)
                    if (s + u < cnt) evaluate_exact(B.j[u], B.pj[u], B.vj[u], B.aj[u]);
            }
        };
        auto drain = [&]() {
            if (ablate & 1) { cnt = 0; return; }
            // double-buffered: the loads of batch k + 1 are in flight while batch k is evaluated
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

        auto seg_range = [&](int row, int off, int len, int &a, int &b) {
            lane_range(P, sh, walk, active, row, cellStart, a, b);
            a = min(max(a - off, 0), len);
            b = min(max(b - off, 0), len);
        };
        float4 e[row_regs<BS, ROWCAP>()];
        int lj, le, nlj = 0, nle = 0;
        const int len0 = min(ROWCAP, sh.rowLen[0]);  // the first segment is the head of row 0
        tile_fetch<BS, ROWCAP, false>(sh.rowStartG[0], len0, posType, e);
        seg_range(0, 0, len0, lj, le);
        row_store<BS, ROWCAP, true>(sh, 0, e);
        __syncthreads();
        const int nseg = (ablate & 8) ? 0 : sg.count;
        for (int r = 0; r < nseg; r++) {
            const int buf = r & 1;
            if (r + 1 < nseg) {
                const int row = sg.row[r + 1], st = sg.start[r + 1], len = sg.len[r + 1];
                tile_fetch<BS, ROWCAP, false>(st, len, posType, e);
                seg_range(row, st - sh.rowStartG[row], len, nlj, nle);
            }
            const int tag = r << 9;  // queue entries: segment << 9 | index in the segment
            if (ablate & 4) lj = le;
            // One aligned group of four candidates; those with lo <= u < hi belong to the lane.
            auto cull_group = [&](int g, int lo, int hi, auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                float d2[kChunk];
                // the queue only has to hold a superset of the neighbours: the fused sum decides here,
                // the evaluation repeats the test in its own arithmetic
                group_d2<true>(sh.x[buf][g], sh.y[buf][g], sh.z[buf][g], S.p.x, S.p.y, S.p.z, d2);
                const int4 T = sh.t[buf][g];
                const int tj[kChunk] = {T.x, T.y, T.z, T.w};
                const int etag = tag | (g * kChunk);
                if (ablate & 2) return;
                if (!wave_any((T.x | T.y | T.z | T.w) != 0)) {
                    // only fluid candidates (the common case): branch-free append
#pragma unroll
                    for (int u = 0; u < kChunk; u++) {
                        q[cnt * BS] = (unsigned short)(etag + u);  // a miss is overwritten by the next store
                        cnt += (d2[u] < sr2c && (!MASKED || (u >= lo && u < hi))) ? 1 : 0;
                    }
                } else {
                    // wall or other particles among the candidates: boundary neighbours always count,
                    // fluid ones inside the support, everything else never
#pragma unroll
                    for (int u = 0; u < kChunk; u++) {
                        const bool mine = !MASKED || (u >= lo && u < hi);
                        if (mine && ((tj[u] == FSA_TYPE_FLUID && d2[u] < sr2c) || tj[u] == FSA_TYPE_BOUNDARY)) {
                            q[cnt * BS] = (unsigned short)(etag + u);
                            cnt++;
                        }
                    }
                }
            };
            const bool has = lj < le;
            int base = lj & ~(kChunk - 1);
            if (wave_any(has)) {  // first group of the range
                cull_group(has ? base / kChunk : Shared::kPad, lj - base, le - base, std::true_type());
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            base = has ? base + kChunk : le;
            while (wave_any(base + kChunk <= le)) {  // whole groups
                const bool go = base + kChunk <= le;
                cull_group(go ? base / kChunk : Shared::kPad, 0, kChunk, std::false_type());
                base += go ? kChunk : 0;
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            if (wave_any(base < le)) {  // last group of the range
                const int rem = le - base;
                cull_group(rem > 0 ? base / kChunk : Shared::kPad, 0, rem, std::true_type());
                if (wave_any(cnt > Q - kChunk)) drain();
            }
            if (r + 1 < nseg) {
                row_store<BS, ROWCAP, true>(sh, buf ^ 1, e);
                lj = nlj;
                le = nle;
                __syncthreads();
            }
        }
        drain();
    }
    if (active) {
        Vec3<float> f;
        if (FAST) {
            S.fx += Sb.fx;
            S.fy += Sb.fy;
            S.fz += Sb.fz;
        }
        f.x = S.fx + 0.0f;
        f.y = S.fy + (float)-9.8;  // the literal of solver/sph_kernel_impl.cuh:372
        f.z = S.fz + 0.0f;
        force[index] = f;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
constexpr int kQueueD = 10;   // density: 4-byte slots (with the tile: 19.9 KB, eight blocks per CU)
constexpr int kQueueF = 48;   // force: 2-byte slots

int launch_density_tuned(Solver *s) {
    if (s->grid_n() == 0) return FSA_OK;
    if (s->variant == 3 && !s->dbg_out0) return launch_density_fast(s);
    ScopedTimer t(s, s->dbg_out0 ? "density_dbg" : "density");
    const int c = s->cur;
    const int unit = kXcds << s->xcd_run_log;  // the grid is rounded up to whole runs
    const int nb = ((s->n + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
    const int ablate = (s->dbg_ablate & 255) | (s->xcd_run_log << 8);
    float *out0 = (float *)(s->dbg_out0 ? s->dbg_out0 : s->b.density), *out1 = (float *)(s->dbg_out1 ? s->dbg_out1 : s->b.pressure);
    float2 *pd_out = s->dbg_out0 ? nullptr : (float2 *)s->b.pd;  // not in timing experiments
    if (s->variant >= 2)
        hipLaunchKernelGGL((density_tuned_kernel<kTunedBS, kRowCap, kQueueD, true>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.restDensity[c], s->b.cellStart, s->b.sortedHash, out0, out1, pd_out,
                           ablate);
    else
        hipLaunchKernelGGL((density_tuned_kernel<kTunedBS, kRowCap, kQueueD, false>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.restDensity[c], s->b.cellStart, s->b.sortedHash, out0, out1, pd_out,
                           ablate);
    FSA_HIP(hipGetLastError());
    if (pd_out) {
        s->pd_valid = true;
        s->pd_fast = s->variant >= 2;
    }
    return FSA_OK;
}

int launch_force_tuned(Solver *s) {
    if (s->grid_n() == 0) return FSA_OK;
    const int c = s->cur;
    const bool fast = s->variant >= 2;
    if (!(s->pd_valid && s->pd_fast == fast)) {
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
    // scenes with TYPE_RIGID particles run another density kernel, which records nothing: they keep the search
    if (s->variant == 3 && !s->dbg_out0 && !s->has_rigid) return launch_force_fast(s);
    ScopedTimer t(s, s->dbg_out0 ? "force_dbg" : "force");
    const int unit = kXcds << s->xcd_run_log;  // the grid is rounded up to whole runs
    const int nb = ((s->n + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
    const int ablate = (s->dbg_ablate & 255) | (s->xcd_run_log << 8);
    Vec3<float> *out = (Vec3<float> *)(s->dbg_out0 ? s->dbg_out0 : s->b.force);
    if (fast)
        hipLaunchKernelGGL((force_tuned_kernel<kTunedBS, kRowCap, kQueueF, true>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.density, (const float *)s->b.pressure, (const float2 *)s->b.pd,
                           s->b.cellStart, s->b.sortedHash, out, ablate);
    else
        hipLaunchKernelGGL((force_tuned_kernel<kTunedBS, kRowCap, kQueueF, false>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const float *)s->b.density, (const float *)s->b.pressure, (const float2 *)s->b.pd,
                           s->b.cellStart, s->b.sortedHash, out, ablate);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// Launches one pass `reps` times with parts of the kernel switched off and the results sent to
// scratch arrays, so that the solver state is untouched.  Timing experiments only.
int debug_time_pass(Solver *s, int which, int ablate, int reps) {
    s->dbg_ablate = ablate;
    int rc = FSA_OK;
    for (int k = 0; k < reps && rc == FSA_OK; k++) {
        if (which == 0) {
            s->dbg_out0 = s->b.tmpIndex;
            s->dbg_out1 = s->b.tmpHash;
            rc = launch_density_tuned(s);
        } else {
            s->dbg_out0 = s->b.v_star_alt;
            rc = launch_force_tuned(s);
        }
    }
    s->dbg_ablate = 0;
    s->dbg_out0 = s->dbg_out1 = nullptr;
    return rc;
}

}  // namespace fsa
