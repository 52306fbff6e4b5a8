// Fast fp32 density and force passes for gfx950 (kernel variant 3): one neighbour search per step.
//
// The two passes of a WCSPH step look at the same positions, so the search runs once:
//
//   density: a workgroup owns 256 consecutive particles of the sorted arrays; the nine neighbour rows stream
//            through double-buffered LDS tiles kept as separate x / y / z arrays (as in fsa_sph_tuned.hip);
//            every lane walks its own index range of a row in aligned groups of four candidates, appends the
//            squared distances of the hits to its LDS queue for the kernel evaluation, and also keeps ONE BIT
//            per candidate in a register (one add-with-carry per candidate).  At the end of a row the bits go to
//            global memory: the row's main word (57 candidates from the lane's first group on, and that group's
//            index), extra words for what a longer range holds beyond, and words of their own for the
//            TYPE_BOUNDARY neighbours (layout: fsa_tile.h);
//   walls  : a fluid particle at a wall has dozens of wall neighbours, the particles around it none.  They are
//            summed one particle per lane, either in the tail of the density pass or in a pass of their own over
//            the particles the density pass listed (launch_density_record says which, and why);
//   force  : no search.  The rows stream through LDS again, this time with everything a pair needs
//            ({x, y, z, m}, {v, p/rho^2}, viscosity term: 36 bytes per particle, coalesced loads), and every
//            lane walks the set bits of its recorded words for the row at hand and reads its neighbours from
//            the tile.  Gathering neighbours from global memory instead costs a wavefront one cache-line
//            look-up per distinct line and instruction (~30 per gather), several times the LDS reads.
//
// Words are stored per workgroup as [slot][lane] (a wavefront writes and reads 512 contiguous bytes per
// slot), at most kHitSlots per particle; a particle that needs more, or whose block cannot be tiled, is
// marked and walks the cells in the replay kernels with the same pair arithmetic in the same order.
// Sums run in the reference's neighbour order (rows z, y; ascending index; TYPE_BOUNDARY neighbours of the
// force pass summed apart), so a particle's result depends neither on the tile alignment nor on which
// particles share its workgroup, and equals that of the searching kernels of variant 2 bit for bit.
// Arithmetic of the pairs: cubic_fast / force_pair_fast / wall_pair_fast (hardware sqrt / rsq / rcp, fused
// multiply-adds), about 1e-7 relative per pair term away from the exact flavour.
#include <type_traits>

#include "fsa_fast.h"
#include "fsa_pair.h"
#include "fsa_solver.h"
#include "fsa_tile.h"

namespace fsa {

// Workgroups per CU the register budgets of the two passes are set for.  The density pass runs eight whatever the
// budget allows (six with 12 more registers to fetch a trip's candidates during the trip before was tried: 1.68 ->
// 2.06 ms: the pass is bound by instruction issue, LDS instructions included, not by the latency of its LDS reads).
// The force replay is bound by latency: six workgroups instead of five, 1.43 -> 1.27 ms.
#ifndef FSA_DENSITY_BLOCKS
#define FSA_DENSITY_BLOCKS 8
#endif
#ifndef FSA_FORCE_BLOCKS
#define FSA_FORCE_BLOCKS 6
#endif
constexpr int kForceBlocks = FSA_FORCE_BLOCKS;
constexpr int kDensityBlocks = FSA_DENSITY_BLOCKS;

// The wall words of particle i (at `wp`, stride BS), summed in neighbour order.
template <int BS>
__device__ __forceinline__ float4 wall_sum(const DevParams<float> &P, int i, const uint2 *__restrict__ wp, int words,
                                           const float4 *__restrict__ posType, const float4 *__restrict__ velMass) {
    const float sr = P.sr, sr2 = sr * sr;
    const float kspiky = P.kspikydiff, visc_eps = 0.01f * sr2;
    const WallConsts W(P);
    ForceSelf<float> Sb;
    Sb.p = posType[i];
    Sb.v = velMass[i];
    Sb.pterm = 0.f;
    Sb.fx = Sb.fy = Sb.fz = 0.f;
    for (int k = 0; k < words; k++) {
        const uint2 w = wp[(size_t)k * BS];
        unsigned m = w.x;
        while (m) {  // four neighbours' loads in flight at a time; the sums stay in neighbour order
            constexpr int kAhead = 4;
            int js[kAhead];
            bool has[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; u++) {
                has[u] = m != 0;
                const int b = __builtin_clz(m | 1u);
                if (has[u]) m &= ~(0x80000000u >> b);
                js[u] = has[u] ? (int)w.y + b : i;
            }
            float4 pj[kAhead], vj[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; u++) {
                pj[u] = posType[js[u]];
                vj[u] = velMass[js[u]];
            }
#pragma unroll
            for (int u = 0; u < kAhead; u++) {
                if (!has[u]) continue;
                const float xx = Sb.p.x - pj[u].x, xy = Sb.p.y - pj[u].y, xz = Sb.p.z - pj[u].z;
                const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                wall_pair_fast(Sb, W, sr, kspiky, visc_eps, xx, xy, xz, d2, true, vj[u].w, vj[u].x, vj[u].y, vj[u].z);
            }
        }
    }
    return make_float4(Sb.fx, Sb.fy, Sb.fz, 0.f);
}

// ------------------------------------------------------------------------------------------------
// density + pressure + hit record (scenes without TYPE_RIGID particles: only fluid neighbours count)
// ------------------------------------------------------------------------------------------------
// EVAL = false: only the record (for a force pass that is not preceded by this solver's density pass).
// the pass's LDS, in one piece so that another pass's can share the bytes (the fused probe at the end of the file)
template <int BS, int ROWCAP, int Q, bool EVAL> struct DensityLds {
    static constexpr int kWallWords = row_regs<BS, ROWCAP>() * BS / 32 + 1;
    RowShared<BS, ROWCAP, false> sh;
    Segments sg;
    int wallSeg[kMaxSegs];  // does the segment hold a TYPE_BOUNDARY particle?
    // which entries of a tile are TYPE_BOUNDARY particles, one bit each (entry 32 w + b at bit b of word w)
    unsigned wallBits[2][kWallWords];
    float queue[EVAL ? Q * BS : 1];
};

// `vb`: the workgroup's index in a grid of this pass alone (blockIdx.x when it is launched on its own)
template <int BS, int ROWCAP, int Q, bool EVAL>
__device__ __forceinline__ void density_record_body(RowShared<BS, ROWCAP, false> &sh, Segments &sg,
                                                            int (&wallSeg)[kMaxSegs],
                                                            unsigned (&wallBits)[2][DensityLds<BS, ROWCAP, Q, EVAL>::kWallWords],
                                                            float *const queue, const int vb,
                                                            const DevParams<float> P, int n,
                                                            const float4 *__restrict__ posType,
                                                            const float4 *__restrict__ velMass,
                                                            const float *__restrict__ restDensity,
                                                            const int *__restrict__ cellStart,
                                                            const int *__restrict__ sortedHash,
                                                            uint2 *__restrict__ hitWords,
                                                            unsigned char *__restrict__ hitCount,
                                                            unsigned char *__restrict__ wallCount,
                                                            int *__restrict__ wallList,
                                                            float4 *__restrict__ wallForce,
                                                            float *__restrict__ density,
                                                            float *__restrict__ pressure,
                                                            float2 *__restrict__ pd, int run_log) {
    using Shared = RowShared<BS, ROWCAP, false>;
    const int tid = threadIdx.x;
#ifdef FSA_ABLATE
    const int dbg = run_log >> 8;  // experiment switches of a measurement build (results are then wrong)
#else
    constexpr int dbg = 0;
#endif
    run_log &= 255;
    n = live_count(P, n);
    const int blk = xcd_block_of(vb, run_log);
    const int b0 = blk * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const bool active = index < n;
    unsigned hlo = 0, hhi = 0;
    if (!P.rows || P.nz != P.gres[2]) {  // slab runs: a tile that lies entirely in the outer halo layer feeds nobody
        hlo = (unsigned)sortedHash[b0];
        hhi = (unsigned)sortedHash[b0 + nb - 1];
        if (hhi != (unsigned)FSA_GRID_UNDEF && run_unneeded(P, hlo, hhi, 1)) return;
    }
    if (tid < kMaxSegs) wallSeg[tid] = 0;
    const int cls = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, hlo, hhi, P.rows);
    bool staged = cls != 0, sparse = false;
    if (staged) setup_segments<ROWCAP>(sh, cls, sg);
    else if (P.rows && P.rows[(size_t)blk * kRowTableStride + 19])  // rows too long as one range each (the tiles are free)
        staged = sparse = sparse_segments<ROWCAP>(P, b0, nb, sortedHash, cellStart, reinterpret_cast<int *>(sh.x), sg);

    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    float mass_i = 0.f, rd_i = 1.f;
    if (active) {
        pi = posType[index];
        if (EVAL) {
            mass_i = velMass[index].w;
            rd_i = restDensity[index];
        }
    }
    const float sr = P.sr, sr2 = sr * sr;
    const float h = (float)((double)sr * 0.5);
    const float inv_h = 1.0f / h, kcubic = P.kernel_cubic;
    float dens = 0.f;
    int words = kHitOverflow;  // an untiled block: the force pass walks the cells for its particles, too
    int wwords = 0;            // wall words of the record
    if (!staged) {
        if (EVAL && active) dens = density_walk_global_fast(P, pi, mass_i, posType, cellStart);
    } else {
        const bool fluid_i = active && unpack_int(pi.w) == FSA_TYPE_FLUID;
        const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
        uint2 *const out = hitWords + (size_t)blk * kHitSlots * BS + tid;
        float *const q = queue + (EVAL ? tid : 0);
        int cnt = 0;  // entries in the queue
        words = 0;
        wwords = 0;
        auto drain = [&]() {
            if (dbg & 2) { cnt = 0; return; }
            for (int s = 0; wave_any(s < cnt); s += kChunk) {
                float d2[kChunk];
#pragma unroll
                for (int u = 0; u < kChunk; u++) d2[u] = q[(s + u) * BS];
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    const float w = s + u < cnt ? cubic_fast(d2[u], inv_h, kcubic) : 0.0f;
                    dens = __builtin_fmaf(mass_i, w, dens);
                }
            }
            cnt = 0;
        };
        // One tile, in two steps so that the loads stay in flight during the cull of the tile before: issue ...
        auto fetch = [&](int s, int len, float4(&e)[row_regs<BS, ROWCAP>()]) {
            const int last = len > 0 ? len - 1 : 0;
#pragma unroll
            for (int k = 0; k < row_regs<BS, ROWCAP>(); k++) e[k] = posType[s + min(tid + k * BS, last)];
        };
        // ... and finish: everything that is not a fluid particle is moved far away, so the cull needs no type
        // test; the TYPE_BOUNDARY entries are noted in wallBits[buf].  True if there is one.
        auto finish = [&](int buf, int len, float4(&e)[row_regs<BS, ROWCAP>()]) -> bool {
            bool wall = false;
#pragma unroll
            for (int k = 0; k < row_regs<BS, ROWCAP>(); k++) {
                const int t = tid + k * BS < len ? unpack_int(e[k].w) : -1;
                const unsigned long long wb = __ballot(t == FSA_TYPE_BOUNDARY);
                if ((tid & 63) == 0) {
                    wallBits[buf][(tid + k * BS) >> 5] = (unsigned)wb;
                    wallBits[buf][((tid + k * BS) >> 5) + 1] = (unsigned)(wb >> 32);
                }
                wall |= wb != 0ull;
                if (t != FSA_TYPE_FLUID) e[k] = make_float4(kFar, kFar, kFar, 0.0f);
            }
            return wall;
        };
        // the lane's index range inside a segment, likewise: the two cell-table loads now, the arithmetic later
        auto range_issue = [&](int row, int &c0, int &c1) {
            c0 = c1 = 0;
            if (active) {
                const int c = walk.row_cell(P, row % 3 - 1, row / 3 - 1);
                if (c >= 0) {
                    c0 = cellStart[c];
                    c1 = cellStart[c + (walk.xhi - walk.xlo) + 1];
                }
            }
        };
        auto range_finish = [&](int c0, int c1, int seg_first, int len, int &a, int &b) {
            a = min(max(c0 - seg_first, 0), len);
            b = min(max(c1 - seg_first, 0), len);
            if (c1 <= c0) a = b = 0;
        };
        // A word of the record: 32 candidates of segment `seg` from candidate `first` on.
        auto put_word = [&](unsigned m, unsigned meta) {
            if (m) {
                if (words < kHitExtra && !(dbg & 1)) out[(size_t)(kMaxSegs + words) * BS] = make_uint2(m, meta);
                words++;
            }
        };

        float4 e[row_regs<BS, ROWCAP>()];
        int lj, le, nlj = 0, nle = 0;
        int rc0, rc1;
        // the first segment is the head of row 0 (known before the segment table is visible), except in a block
        // with segments of its own, whose table is
        const int start0 = sparse ? sg.start[0] : sh.rowStartG[0];
        const int len0 = sparse ? sg.len[0] : min(ROWCAP, sh.rowLen[0]);
        fetch(start0, len0, e);
        range_issue(sparse ? sg.row[0] : 0, rc0, rc1);
        bool wall = finish(0, len0, e);
        range_finish(rc0, rc1, start0, len0, lj, le);
        row_store<BS, ROWCAP, false>(sh, 0, e);
        __syncthreads();  // wallSeg cleared, segment table written
        if (wall) wallSeg[0] = 1;
        __syncthreads();
        const int nseg = sg.count;
        for (int r = 0; r < nseg; r++) {
            const int buf = r & 1;
            if (r + 1 < nseg) {  // in flight during the cull below
                fetch(sg.start[r + 1], sg.len[r + 1], e);
                range_issue(sg.row[r + 1], rc0, rc1);
            }
            // The aligned groups of four that cover [lj, le).  Candidates of the first and the last group outside
            // the range belong to other cells and must not count (with empty cells in between they can be the
            // lane's neighbours of ANOTHER row), so those two groups test the range per candidate.  All lanes take
            // their first group in the same trip and the following ones in step; a lane that runs out examines
            // far-away padding until the others are done, and the last (partial) groups come in one trip of
            // their own at the end.
            //
            // Record: one bit per candidate is shifted into `m` (an add with carry), and every eight trips -- the
            // same trips in every lane -- the full word moves to c2, c1, c0 in turn (96 bits: nothing is written
            // inside the cull -- a store issued there would make the wait for the next tile a wait for the store
            // as well), so bit k from the top belongs to candidate 4 * (lj >> 2) + k; the bits of the last partial
            // group are kept apart and put in their place afterwards.
            unsigned c0 = 0, c1 = 0, c2 = 0, m = 0, hl = 0;
            int trips = 0;  // the same in every lane
            auto cull_vals = [&](const float4 &GX, const float4 &GY, const float4 &GZ, int lo, int hi, unsigned &bits,
                                 auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                float d2[kChunk];
                group_d2<true>(GX, GY, GZ, pi.x, pi.y, pi.z, d2);
#pragma unroll
                for (int u = 0; u < kChunk; u++) {
                    // (the first and the last group of a range: candidates outside it are moved out of reach)
                    if constexpr (MASKED && EVAL) d2[u] = (unsigned)(u - lo) < (unsigned)(hi - lo) ? d2[u] : kFar;
                    if (EVAL) q[cnt * BS] = d2[u];  // a miss is overwritten by the next store
                    if constexpr (EVAL && (MASKED || !MASKED)) {  // (a condition that depends on the lambda's own parameter)
                        // one compare, then the hit goes into the queue count and into the bit mask as a carry
                        unsigned long long carry_out;
                        asm("v_cmp_gt_f32 vcc, %[sr2], %[d2]\n\t"
                            "v_addc_co_u32_e64 %[cnt], %[co], 0, %[cnt], vcc\n\t"
                            "v_addc_co_u32_e64 %[bits], %[co], %[bits], %[bits], vcc"
                            : [cnt] "+v"(cnt), [bits] "+v"(bits), [co] "=&s"(carry_out)
                            : [sr2] "v"(sr2), [d2] "v"(d2[u])
                            : "vcc");
                    } else {
                        const unsigned hit = (d2[u] < sr2 && (!MASKED || (u >= lo && u < hi))) ? 1u : 0u;
                        if (EVAL) cnt += (int)hit;
                        bits = bits + bits + hit;
                    }
                }
            };
            auto cull_group = [&](bool go, int g, int lo, int hi, unsigned &bits, auto masked_tag) {
                const int gg = go ? g : Shared::kPad;
                cull_vals(sh.x[buf][gg], sh.y[buf][gg], sh.z[buf][gg], lo, hi, bits, masked_tag);
            };
            auto trip_done = [&]() {
                trips++;
                if ((trips & 7) == 0) {
                    c2 = trips == 8 ? m : c2;
                    c1 = trips == 16 ? m : c1;
                    c0 = trips == 24 ? m : c0;
                    m = 0;
                }
            };
            const bool has = lj < le;
            const int g0 = lj >> 2;
            int base = lj & ~(kChunk - 1);
            if (wave_any(has)) {  // first group
                cull_group(has, g0, lj - base, le - base, m, std::true_type());
                trip_done();
                if (EVAL && wave_any(cnt > Q - kChunk)) drain();
            }
            base = has ? base + kChunk : le;
            for (;;) {  // whole groups
                // the inner loop leaves for everything that is not a plain trip: a full queue, a full word
                bool more;
                while ((more = wave_any(base + kChunk <= le))) {
                    const bool go = base + kChunk <= le;
                    cull_group(go, base >> 2, 0, kChunk, m, std::false_type());
                    base += go ? kChunk : 0;
                    trips++;
                    if ((trips & 7) == 0 || (EVAL && wave_any(cnt > Q - kChunk))) break;
                }
                if (!more) break;
                if ((trips & 7) == 0) {
                    c2 = trips == 8 ? m : c2;
                    c1 = trips == 16 ? m : c1;
                    c0 = trips == 24 ? m : c0;
                    m = 0;
                }
                if (EVAL && wave_any(cnt > Q - kChunk)) drain();
            }
            if (wave_any(base < le)) {  // last group
                cull_group(base < le, base >> 2, 0, le - base, hl, std::true_type());
                if (EVAL && wave_any(cnt > Q - kChunk)) drain();
            }
            const bool walls_here = wallSeg[r] != 0;
            if (r + 1 < nseg) {
                wall = finish(buf ^ 1, sg.len[r + 1], e);
                range_finish(rc0, rc1, sg.start[r + 1], sg.len[r + 1], nlj, nle);
                row_store<BS, ROWCAP, false>(sh, buf ^ 1, e);
                if (wall) wallSeg[r + 1] = 1;
            }
            {
                // the started word, left-aligned; a segment of more than 24 groups per lane has lost bits and the
                // lane gives up its record
                if (trips & 7) {
                    m <<= 32 - 4 * (trips & 7);
                    c2 = trips < 8 ? m : c2;
                    c1 = (trips > 8 && trips < 16) ? m : c1;
                    c0 = (trips > 16 && trips < 24) ? m : c0;
                }
                // the last partial group, (base >> 2) - g0 groups from the start
                const int off = 4 * ((base >> 2) - g0);
                const unsigned hv = off < 96 ? hl << (28 - (off & 31)) : 0u;
                c2 |= off < 32 ? hv : 0u;
                c1 |= (off >= 32 && off < 64) ? hv : 0u;
                c0 |= off >= 64 ? hv : 0u;
                const unsigned meta = ((unsigned)r << kHitSegShift) | (unsigned)(4 * g0);
                // the particle itself counts in the density sum but not in the force sum: its bit is dropped
                if (sg.row[r] == 4) {
                    const int own = index - sg.start[r], o = own - 4 * g0;
                    if (active && own >= 0 && own < sg.len[r] && o >= 0 && o < 96) {
                        const unsigned bit = ~(0x80000000u >> (o & 31));
                        c2 &= o < 32 ? bit : ~0u;
                        c1 &= (o >= 32 && o < 64) ? bit : ~0u;
                        c0 &= o >= 64 ? bit : ~0u;
                    }
                }
                // the segment's main word: candidates 0..56 from the group g0 on, g0 itself in the low seven bits
                static_assert((ROWCAP + kChunk) / kChunk < (1 << kHitMainShift), "group index does not fit");
                if (!(dbg & 1)) out[(size_t)r * BS] = make_uint2(c2, (c1 & ~kHitMainMask) | (unsigned)g0);
                // what lies beyond (ranges of more than 57 candidates: rare) goes to the extra words
                put_word((c1 << (32 - kHitMainShift)) | (c0 >> kHitMainShift), meta + 64 - kHitMainShift);
                put_word(c0 << (32 - kHitMainShift), meta + 96 - kHitMainShift);
                if (trips > 24 || (hl && off >= 96)) words = kHitExtra + 1;  // no record for this particle
            }
            // TYPE_BOUNDARY neighbours count for fluid particles wherever they are in the 27 cells (ForceCell has
            // no range test on that branch): words of their own, flagged, after the segment's fluid words
            if (walls_here) {
                for (int c0 = lj; wave_any(fluid_i && c0 < le); c0 += 32) {
                    unsigned m = 0;
                    if (fluid_i && c0 < le) {
                        const int wi = c0 >> 5, sh = c0 & 31;
                        unsigned bits = wallBits[buf][wi] >> sh;  // candidate c0 + b at bit b
                        if (sh) bits |= wallBits[buf][wi + 1] << (32 - sh);
                        if (le - c0 < 32) bits &= (1u << (le - c0)) - 1u;
                        m = __brev(bits);  // first candidate highest
                    }
                    if (m) {  // a wall word: where its first candidate is in the sorted arrays
                        if (wwords < kHitWall && !(dbg & 1))
                            out[(size_t)(kMaxSegs + kHitExtra + wwords) * BS] = make_uint2(m, (unsigned)(sg.start[r] + c0));
                        wwords++;
                    }
                }
            }
            if (r + 1 < nseg) {
                lj = nlj;
                le = nle;
                __syncthreads();  // segment r + 1 visible; everyone is done with the buffer of segment r - 1
            }
        }
        if (EVAL) drain();
        if (words > kHitExtra || wwords > kHitWall) words = kHitOverflow;
        if (words == kHitOverflow) wwords = 0;
    }
    // Wall neighbours.  wallList == nullptr: summed right here -- the wavefronts of a block go their own ways from
    // now on, and one that holds particles at a wall simply ends later; else the particles are listed for a pass
    // of their own (any order: they are independent).
    if (!wallList) {
        if (active && wwords > 0) {
            const uint2 *wp = hitWords + (size_t)blk * kHitSlots * BS + (size_t)(kMaxSegs + kHitExtra) * BS + tid;
            wallForce[index] = wall_sum<BS>(P, index, wp, wwords, posType, velMass);
        }
    } else {
        const bool listed = active && wwords > 0;
        const unsigned long long m = __ballot(listed);
        if (m) {
            const int lane = tid & 63, leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(wallList, __popcll(m));
            base = __shfl(base, leader, 64);
            if (listed) wallList[1 + base + __popcll(m & ((1ull << lane) - 1ull))] = index;
        }
    }
    if (active) {
        hitCount[index] = (unsigned char)words;
        wallCount[index] = (unsigned char)wwords;
        if (EVAL) {
            density[index] = dens;
            float pr = P.pressureK * (r_pow7(dens / rd_i) - 1);
            if (pr < 0) pr = 0;
            pressure[index] = pr;
            // the per-particle terms of the force pass, while they are at hand
            pd[index] = make_float2(pr / dens / dens, mass_i / dens * 2.0f * P.viscosity);
        }
    }
}

template <int BS, int ROWCAP, int Q, bool EVAL>
__global__ __launch_bounds__(BS, kDensityBlocks) void density_record_kernel(DevParams<float> P, int n,
                                                            const float4 *__restrict__ posType,
                                                            const float4 *__restrict__ velMass,
                                                            const float *__restrict__ restDensity,
                                                            const int *__restrict__ cellStart,
                                                            const int *__restrict__ sortedHash,
                                                            uint2 *__restrict__ hitWords,
                                                            unsigned char *__restrict__ hitCount,
                                                            unsigned char *__restrict__ wallCount,
                                                            int *__restrict__ wallList,
                                                            float4 *__restrict__ wallForce,
                                                            float *__restrict__ density,
                                                            float *__restrict__ pressure,
                                                            float2 *__restrict__ pd, int run_log) {
    __shared__ RowShared<BS, ROWCAP, false> sh;
    __shared__ Segments sg;
    __shared__ int wallSeg[kMaxSegs];
    __shared__ unsigned wallBits[2][DensityLds<BS, ROWCAP, Q, EVAL>::kWallWords];
    __shared__ float queue[EVAL ? Q * BS : 1];
    density_record_body<BS, ROWCAP, Q, EVAL>(sh, sg, wallSeg, wallBits, queue, (int)blockIdx.x, P, n, posType, velMass, restDensity, cellStart, sortedHash,
                                             hitWords, hitCount, wallCount, wallList, wallForce, density, pressure, pd,
                                             run_log);
}

// ------------------------------------------------------------------------------------------------
// wall pass: the TYPE_BOUNDARY neighbours of the listed particles, one particle per lane
// ------------------------------------------------------------------------------------------------
// A fluid particle in contact with a wall has dozens of wall neighbours (every wall particle of its 27 cells counts)
// while the particles around it have none; replayed inside the force pass they made their whole wavefront wait.
// Here every lane is such a particle.  Sums in neighbour order, into wallForce[i], which the force pass adds.
template <int BS>
__global__ __launch_bounds__(256) void wall_force_kernel(DevParams<float> P, const int *__restrict__ wallList,
                                                         const float4 *__restrict__ posType,
                                                         const float4 *__restrict__ velMass,
                                                         const uint2 *__restrict__ hitWords,
                                                         const unsigned char *__restrict__ wallCount,
                                                         float4 *__restrict__ wallForce) {
    const int count = wallList[0];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < count; e += gridDim.x * 256) {
        const int i = wallList[1 + e];
        const uint2 *wp = hitWords + (size_t)(i / BS) * kHitSlots * BS + (size_t)(kMaxSegs + kHitExtra) * BS + (i % BS);
        wallForce[i] = wall_sum<BS>(P, i, wp, wallCount[i], posType, velMass);
    }
}

// ------------------------------------------------------------------------------------------------
// force: replay of the recorded hits from full-data row tiles
// ------------------------------------------------------------------------------------------------
template <int ROWCAP> struct PairTile {
    float4 A[2 * ROWCAP];  // x, y, z, mass          (two tiles: segment s lives in tile s & 1)
    float4 B[2 * ROWCAP];  // velocity, pressure / density^2
    float C[2 * ROWCAP];   // mass / density * 2 * viscosity
};

// A lane works through its segments in order.  During segment r the tiles of segments r - 1 and r are both in LDS,
// so a lane may still be busy with segment r - 1 while others are ahead.  The loop of segment r runs in two parts:
// the first ends when no lane of the wavefront has bits of segment r - 1 left, then (workgroup barrier) the tile of
// segment r + 1 takes the place of that of r - 1, and the second part goes on with segment r until fewer than
// `quorum` lanes have bits left -- those carry them over into the next segment's loop.  That keeps most lanes
// busy in every trip although a lane's hits per row vary a lot (on a lattice half the lanes have no neighbour at
// all in a given row).
template <int ROWCAP> struct ForceLds {
    RowShared<1, 4, false> sh;  // only its row table is used here
    Segments sg;
    PairTile<ROWCAP> tile;
};

template <int BS, int ROWCAP>
__device__ __forceinline__ void force_rows_body(ForceLds<ROWCAP> &lds, const int vb, const DevParams<float> P, int n,
                                                        const float4 *__restrict__ posType,
                                                        const float4 *__restrict__ velMass,
                                                        const float2 *__restrict__ pd,
                                                        const int *__restrict__ cellStart,
                                                        const int *__restrict__ sortedHash,
                                                        const uint2 *__restrict__ hitWords,
                                                        const unsigned char *__restrict__ hitCount,
                                                        const unsigned char *__restrict__ wallCount,
                                                        const float4 *__restrict__ wallForce,
                                                        int *__restrict__ wallList,
                                                        Vec3<float> *__restrict__ force, int run_log, int zone) {
    using Shared = RowShared<1, 4, false>;
    Shared &sh = lds.sh;
    Segments &sg = lds.sg;
    PairTile<ROWCAP> &tile = lds.tile;
    const int tid = threadIdx.x;
#ifdef FSA_ABLATE
    const int dbg = run_log >> 8;  // experiment switches of a measurement build (results are then wrong)
#else
    constexpr int dbg = 0;
#endif
    run_log &= 255;
    const int quorum = (dbg >> 4) ? (dbg >> 4) : 40;
    // the wall pass has run: its list starts empty at the next density pass (Solver::wall_list_clean)
    if (wallList && vb == 0 && threadIdx.x == 0) wallList[0] = 0;
    n = live_count(P, n);
    const int blk = xcd_block_of(vb, run_log);
    const int b0 = blk * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    unsigned hlo = 0, hhi = 0;
    if (!P.rows || P.nz != P.gres[2] || zone) {  // slab runs: forces are only needed for the layers the solver owns
        hlo = (unsigned)sortedHash[b0];
        hhi = (unsigned)sortedHash[b0 + nb - 1];
        if (hhi != (unsigned)FSA_GRID_UNDEF && run_unneeded(P, hlo, hhi, 0)) return;
        if (zone) {  // slab groups: the blocks next to the cuts in one launch, the others in a later one
            const bool edge = hhi == (unsigned)FSA_GRID_UNDEF || run_touches_edge(P, hlo, hhi);
            if ((zone == 1) != edge) return;
        }
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
    bool staged = cls != 0, sparse = false;
    if (staged) setup_segments<ROWCAP>(sh, cls, sg);
    else if (P.rows && P.rows[(size_t)blk * kRowTableStride + 19])  // the segments the density pass recorded with
        staged = sparse = sparse_segments<ROWCAP>(P, b0, nb, sortedHash, cellStart, reinterpret_cast<int *>(tile.A), sg);
    int cnt = 0;       // extra words of the lane's record
    bool rec = false;  // the lane has a record
    if (active) {
        S.v = velMass[index];
        S.pterm = pd[index].x;
        cnt = hitCount[index];
        if (dbg & 4) cnt = kHitOverflow;
        rec = cnt != kHitOverflow;
    }
    // wall neighbours are summed on their own and added at the end
    ForceSelf<float> Sb = S;
    const float sr = P.sr, sr2 = sr * sr;
    const float kspiky = P.kspikydiff, visc_eps = 0.01f * sr2;
    const WallConsts W(P);
    if (cnt == kHitOverflow && !(dbg & 8)) force_walk_global_fast(P, index, S, Sb, posType, velMass, pd, cellStart);
    if (!rec) cnt = 0;
    if (staged) {
        const uint2 *const wp = hitWords + (size_t)blk * kHitSlots * BS + tid;
        constexpr unsigned kNoSeg = 0xffffu;
        auto extra_at = [&](int k) { return k < cnt ? wp[(size_t)(kMaxSegs + k) * BS] : make_uint2(0u, kNoSeg << kHitSegShift); };
        int xk = 0;                 // extra words taken so far
        uint2 xw = extra_at(0);     // the next one
        unsigned cur = 0, pend = 0; // bits at hand (32 candidates from `cbase` on), and the 25 after them
        int cseg = -1;              // the segment they belong to
        int cbase = 0;              // tile entry of bit 31 of `cur`, the two tiles counted through
        struct Stage {
            float4 p, v;
            float2 a;
        };
        auto fetch = [&](int s, int len, Stage(&e)[row_regs<BS, ROWCAP - kChunk>()]) {
            const int last = len > 0 ? len - 1 : 0;
#pragma unroll
            for (int k = 0; k < row_regs<BS, ROWCAP - kChunk>(); k++) {
                const int j = s + min(tid + k * BS, last);
                e[k].p = posType[j];
                e[k].v = velMass[j];
                e[k].a = pd[j];
            }
        };
        auto store = [&](int buf, int len, const Stage(&e)[row_regs<BS, ROWCAP - kChunk>()]) {
#pragma unroll
            for (int k = 0; k < row_regs<BS, ROWCAP - kChunk>(); k++) {
                const int i = tid + k * BS;
                if (i < len) {
                    tile.A[buf * ROWCAP + i] = make_float4(e[k].p.x, e[k].p.y, e[k].p.z, e[k].v.w);
                    tile.B[buf * ROWCAP + i] = make_float4(e[k].v.x, e[k].v.y, e[k].v.z, e[k].a.x);
                    tile.C[buf * ROWCAP + i] = e[k].a.y;
                }
            }
        };
        Stage e[row_regs<BS, ROWCAP - kChunk>()];
        if (tid < 2) {  // entry 0 of both tiles is what a lane without a neighbour reads: never uninitialised
            tile.A[tid * ROWCAP] = make_float4(0.f, 0.f, 0.f, 0.f);
            tile.B[tid * ROWCAP] = make_float4(0.f, 0.f, 0.f, 0.f);
            tile.C[tid * ROWCAP] = 0.f;
        }
        __syncthreads();
        // the first segment is the head of row 0, or the first of the block's own segments
        const int len0 = sparse ? sg.len[0] : min(ROWCAP, sh.rowLen[0]);
        fetch(sparse ? sg.start[0] : sh.rowStartG[0], len0, e);
        uint2 fw = rec ? wp[0] : make_uint2(0u, 0u);  // main word of the segment whose loop runs
        store(0, len0, e);
        __syncthreads();
        const int nseg = sg.count;
        if (nseg > 1) fetch(sg.start[1], sg.len[1], e);
        for (int r = 0; r < nseg; r++) {
            uint2 fwn = make_uint2(0u, 0u);
            if (r + 1 < nseg && rec) fwn = wp[(size_t)(r + 1) * BS];  // in flight during this segment's loop
            struct Nb {
                bool has;
                int t;
                float4 a, b;
                float c;
            };
            // Moves the lane on to its next bits when those at hand are used up.
            auto refill = [&]() {
                if (wave_any(cur == 0)) {
                    if (cur == 0) {
                        if (pend != 0) {
                            cur = pend;
                            pend = 0;
                            cbase += 32;
                        } else if ((xw.y >> kHitSegShift) == (unsigned)cseg) {  // an extra word of this segment
                            cur = xw.x;
                            cbase = (cseg & 1) * ROWCAP + (int)(xw.y & ((1u << kHitSegShift) - 1u));
                            xw = extra_at(++xk);
                        } else if (cseg < r) {  // on to segment r, whose main word is `fw`
                            cseg = r;
                            cur = fw.x;
                            pend = fw.y & ~kHitMainMask;
                            cbase = (r & 1) * ROWCAP + 4 * (int)(fw.y & kHitMainMask);
                        }
                    }
                }
            };
            // Takes the lane's next neighbour and starts the tile reads.  `part` 0: until every lane has reached
            // segment r; 1: until fewer than `quorum` lanes have bits left (all of them in the last segment).
            auto next = [&](Nb &N, int part) -> bool {
                refill();
                const bool has = cur != 0;
                if (part == 0) {
                    if (!wave_any(cseg < r)) return false;  // a lane moves on to r when it is through with r - 1
                } else {
                    const unsigned long long busy = __ballot(has);
                    if (busy == 0ull || (r + 1 < nseg && __popcll(busy) < quorum)) return false;
                }
                N.has = has;
                const int k = __builtin_clz(cur | 1u);  // cur == 0: nothing to clear
                cur &= ~(0x80000000u >> k);
                N.t = has ? cbase + k : 0;
                N.a = tile.A[N.t];
                N.b = tile.B[N.t];
                N.c = tile.C[N.t];
                return true;
            };
            auto eval = [&](const Nb &N) {
                if (dbg & 1) { S.fx += N.a.x + N.b.x + N.c; return; }
                const float xx = S.p.x - N.a.x, xy = S.p.y - N.a.y, xz = S.p.z - N.a.z;
                const float d2 = __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx));
                // straight-line: a lane without a neighbour in this trip evaluates entry 0 and adds exact zeros
                force_pair_fast(S, sr, kspiky, visc_eps, xx, xy, xz, d2, N.has, make_float4(N.b.x, N.b.y, N.b.z, N.a.w),
                                make_float2(N.b.w, N.c));
            };
            // two stages, roles rotating: the next neighbour's tile reads are in flight while one is evaluated
            auto run = [&](int part) {
                Nb N0, N1;
                bool go = next(N0, part);
                while (go) {
                    const bool go1 = next(N1, part);
                    eval(N0);
                    if (!go1) break;
                    go = next(N0, part);
                    eval(N1);
                }
            };
            run(0);
            if (r + 1 < nseg) {
                __syncthreads();  // everyone is done with the tile of segment r - 1
                store((r + 1) & 1, sg.len[r + 1], e);
                if (r + 2 < nseg && !(dbg & 2)) fetch(sg.start[r + 2], sg.len[r + 2], e);
            }
            run(1);
            if (r + 1 < nseg) {
                fw = fwn;
                __syncthreads();  // the tile of segment r + 1 is visible
            }
        }
    }
    if (active) {
        if (rec && wallCount[index] > 0) {  // the wall pass's sum for this particle
            const float4 wf = wallForce[index];
            Sb.fx = wf.x;
            Sb.fy = wf.y;
            Sb.fz = wf.z;
        }
        Vec3<float> f;
        S.fx += Sb.fx;
        S.fy += Sb.fy;
        S.fz += Sb.fz;
        f.x = S.fx + 0.0f;
        f.y = S.fy + (float)-9.8;  // the literal of solver/sph_kernel_impl.cuh:372
        f.z = S.fz + 0.0f;
        force[index] = f;
    }
}

template <int BS, int ROWCAP>
__global__ __launch_bounds__(BS, kForceBlocks) void force_rows_kernel(DevParams<float> P, int n,
                                                        const float4 *__restrict__ posType,
                                                        const float4 *__restrict__ velMass,
                                                        const float2 *__restrict__ pd,
                                                        const int *__restrict__ cellStart,
                                                        const int *__restrict__ sortedHash,
                                                        const uint2 *__restrict__ hitWords,
                                                        const unsigned char *__restrict__ hitCount,
                                                        const unsigned char *__restrict__ wallCount,
                                                        const float4 *__restrict__ wallForce,
                                                        int *__restrict__ wallList,
                                                        Vec3<float> *__restrict__ force, int run_log, int zone) {
    __shared__ ForceLds<ROWCAP> lds;
    force_rows_body<BS, ROWCAP>(lds, (int)blockIdx.x, P, n, posType, velMass, pd, cellStart, sortedHash, hitWords, hitCount,
                                wallCount, wallForce, wallList, force, run_log, zone);
}

#ifdef FSA_ABLATE
// ------------------------------------------------------------------------------------------------
// measurement only: both passes in one launch, every other workgroup of an XCD a density workgroup
// ------------------------------------------------------------------------------------------------
struct FusedProbeArgs {
    const float4 *posType, *velMass;
    const float *restDensity;
    const int *cellStart, *sortedHash;
    uint2 *hitWords;
    unsigned char *hitCount, *wallCount;
    int *wallList;
    float4 *wallForce;
    float *density, *pressure;
    float2 *pd;
    Vec3<float> *force;
};

template <int BS, int ROWCAP, int Q>
__global__ __launch_bounds__(BS, kForceBlocks) void fused_probe_kernel(DevParams<float> P, int n, FusedProbeArgs A, int run_log,
                                                                     int mode) {
    __shared__ union Lds {
        DensityLds<BS, ROWCAP, Q, true> d;
        ForceLds<ROWCAP> f;
    } lds;
    // on every XCD `stride` density workgroups, then `stride` force workgroups, and so on (mode >> 4; consecutive
    // workgroups of an XCD go to different CUs in turn, so the stride decides which CUs see which kind)
    const int stride = max(mode >> 4, 1);
    mode &= 15;
    const int xcd = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
    const int g = k / (2 * stride), r = k % (2 * stride);
    const int vb = (g * stride + r % stride) * kXcds + xcd;
    const bool first = r < stride;
    const bool dens = mode == 0 ? first : mode == 1;  // mode 1 / 2: only density / only force workgroups do work
    if (mode != 0 && !first) return;
    if (dens)
        density_record_body<BS, ROWCAP, Q, true>(lds.d.sh, lds.d.sg, lds.d.wallSeg, lds.d.wallBits, lds.d.queue, vb, P, n, A.posType, A.velMass, A.restDensity, A.cellStart,
                                                 A.sortedHash, A.hitWords, A.hitCount, A.wallCount, A.wallList, A.wallForce,
                                                 A.density, A.pressure, A.pd, run_log);
    else
        force_rows_body<BS, ROWCAP>(lds.f, vb, P, n, A.posType, A.velMass, A.pd, A.cellStart, A.sortedHash, A.hitWords,
                                    A.hitCount, A.wallCount, A.wallForce, nullptr, A.force, run_log, 0);
}

#endif  // FSA_ABLATE

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
constexpr int kWallTailMax = 4 << 20;  // particles up to which the density pass sums the wall neighbours itself
constexpr int kQueueD = 10;  // density queue: 4-byte slots (with the tile: 19.6 KB of LDS per workgroup)

static inline int tiled_grid(const Solver *s, int run_log) {
    const int unit = kXcds << run_log;  // the grid is rounded up to whole runs
    return ((s->grid_n() + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
}

template <bool EVAL> static int launch_density_record(Solver *s) {
    const int c = s->cur, run_log = s->xcd_run_log | (s->dbg_fast << 8);
    // Where the wall neighbours are summed.  A pass of their own costs about 40 us whatever the size (a chain of
    // dependent loads and a launch); in the tail of this pass they cost nothing fixed, but hold up the wavefronts
    // that have them (16 M particles, disordered: +0.2 ms against +0.05).  Small solvers take the tail.
    const bool in_tail = s->wall_mode == 0 ? s->n <= kWallTailMax : s->wall_mode == 1;  // (n: as last known on the host)
    if (!in_tail) {
        if (!s->wall_list_clean)  // the list of the wall pass starts empty (the force pass leaves it so)
            FSA_HIP(hipMemsetAsync(s->b.wallList, 0, sizeof(int), s->stream));
        s->wall_list_clean = false;
    }
    s->wall_listed = !in_tail;
    hipLaunchKernelGGL((density_record_kernel<kTunedBS, kRowCap, kQueueD, EVAL>), dim3(tiled_grid(s, s->xcd_run_log)),
                       dim3(kTunedBS), 0, s->stream, s->pf, s->grid_n(), (const float4 *)s->b.posType[c],
                       (const float4 *)s->b.velMass[c], (const float *)s->b.restDensity[c], s->b.cellStart,
                       s->b.sortedHash, (uint2 *)s->b.hitWords, s->b.hitCount, s->b.wallCount,
                       in_tail ? nullptr : s->b.wallList, (float4 *)s->b.wallForce, (float *)s->b.density, (float *)s->b.pressure, (float2 *)s->b.pd, run_log);
    FSA_HIP(hipGetLastError());
    s->hits_valid = true;
    return FSA_OK;
}

int launch_density_fast(Solver *s) {
    if (s->grid_n() == 0) return FSA_OK;
    ScopedTimer t(s, "density");
    int rc = launch_density_record<true>(s);
    if (rc) return rc;
    s->pd_valid = true;
    s->pd_fast = true;
    return FSA_OK;
}

int ensure_hit_record(Solver *s) {
    if (s->hits_valid || s->grid_n() == 0) return FSA_OK;
    ScopedTimer t(s, "record_hits");
    return launch_density_record<false>(s);
}

// needs pd[] in the fast layout (the caller sees to that)
int launch_force_fast(Solver *s) {
    if (s->grid_n() == 0) return FSA_OK;
    if (!s->hits_valid) {
        ScopedTimer t(s, "record_hits");
        int rc = launch_density_record<false>(s);
        if (rc) return rc;
    }
    const int c = s->cur, run_log = s->xcd_run_log | (s->dbg_fast << 8), zone = s->force_zone;
    if (zone != 2 && s->wall_listed) {  // (zone 2 follows zone 1, which has run it)
        ScopedTimer t(s, "wall_force");
        hipLaunchKernelGGL((wall_force_kernel<kTunedBS>), dim3(1024), dim3(256), 0, s->stream, s->pf, s->b.wallList,
                           (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                           (const uint2 *)s->b.hitWords, s->b.wallCount, (float4 *)s->b.wallForce);
    }
    ScopedTimer t(s, zone == 1 ? "force_edge" : "force");
    hipLaunchKernelGGL((force_rows_kernel<kTunedBS, kRowCap>), dim3(tiled_grid(s, s->xcd_run_log)), dim3(kTunedBS), 0,
                       s->stream, s->pf, s->grid_n(), (const float4 *)s->b.posType[c], (const float4 *)s->b.velMass[c],
                       (const float2 *)s->b.pd, s->b.cellStart, s->b.sortedHash, (const uint2 *)s->b.hitWords,
                       s->b.hitCount, s->b.wallCount, (const float4 *)s->b.wallForce, s->b.wallList,
                       (Vec3<float> *)s->b.force, run_log, zone);
    FSA_HIP(hipGetLastError());
    s->wall_list_clean = true;
    return FSA_OK;
}

#ifdef FSA_ABLATE
// Timing probe: the density pass and the force replay of the present (sorted, recorded) state in ONE launch, their
// workgroups alternating on every XCD so that each CU holds some of both.  mode 1 / 2: the same grid with only the
// density / only the force workgroups working.  Results are those of the separate passes (both read what the other
// writes, but the values written are the ones already there).
int launch_fused_probe(Solver *s, int mode) {
    if (s->precision != FSA_PRECISION_F32 || s->variant != 3 || !s->hits_valid || !s->pd_valid || s->slab)
        return fail(FSA_ERR_INVALID, "the fused probe needs an fp32 variant-3 solver after a density and a force pass");
    const int c = s->cur;
    FusedProbeArgs A;
    A.posType = (const float4 *)s->b.posType[c];
    A.velMass = (const float4 *)s->b.velMass[c];
    A.restDensity = (const float *)s->b.restDensity[c];
    A.cellStart = s->b.cellStart;
    A.sortedHash = s->b.sortedHash;
    A.hitWords = (uint2 *)s->b.hitWords;
    A.hitCount = s->b.hitCount;
    A.wallCount = s->b.wallCount;
    A.wallList = nullptr;  // wall neighbours in the density workgroups' tail
    A.wallForce = (float4 *)s->b.wallForce;
    A.density = (float *)s->b.density;
    A.pressure = (float *)s->b.pressure;
    A.pd = (float2 *)s->b.pd;
    A.force = (Vec3<float> *)s->b.force;
    hipLaunchKernelGGL((fused_probe_kernel<kTunedBS, kRowCap, kQueueD>), dim3(2 * tiled_grid(s, s->xcd_run_log)), dim3(kTunedBS),
                       0, s->stream, s->pf, s->grid_n(), A, s->xcd_run_log, mode);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

#else
int launch_fused_probe(Solver *, int) {
    return fail(FSA_ERR_INVALID, "the fused probe is part of measurement builds only (-DFSA_ABLATE)");
}
#endif

}  // namespace fsa
