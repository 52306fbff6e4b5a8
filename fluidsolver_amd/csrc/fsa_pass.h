// Runner kernels of the neighbour passes that are written as policies (DFSPH, IISPH and the WCSPH
// density pass of scenes with rigid particles): the policy holds the loop body of one of the
// reference's *Cell functions, the kernels here walk the 27 cells in the reference's order.
#pragma once

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
template <typename T> struct PassArrays {
    const Vec4<T> *posType;
    const Vec4<T> *velMass;
    const T *restDensity;
    const T *density;
    const T *dfFactor;
    const T *pstiffIn;
    const Vec3<T> *vStarIn;
    const Vec4<T> *aux4;  // {v_star, mass} per particle (tuned density-change passes)
    const Vec4<T> *aux4b; // a second packed record per particle (IISPH pressure rounds)
    Vec4<T> *recOut;      // where a pass writes the record the NEXT pass reads of this particle (null: nowhere)
    const T *aux2;        // {pstiff / density, mass} pairs (tuned applyPStiff)
    T *rec2Out;           // like recOut, for a two-word record (null: nowhere)
    T *densityOut;
    T *dfFactorOut;
    T *errorOut;
    T *pstiffOut;
    Vec3<T> *vStarOut;
    // IISPH
    const int *uniqueId;
    const int *groupSorted;   // group of the particle at each sorted position
    const Vec3<T> *normalById;  // indexed by uniqueId
    const T *pressure;
    T *pressureOut;
    Vec3<T> *dii;
    Vec3<T> *dijpjl;
    T *aii;
    T *densityStar;
    Vec3<T> *pressureForce;
};

// A policy may carry the neighbour's position inside its Nb record (`static constexpr bool kPosInNb = true`,
// `Nb::pos()` returning {x, y, z, type bits}): the tiled and the list kernel then skip their own gather of
// posType[j] -- one gather per neighbour instead of two.
template <typename Policy, typename = void> struct pos_in_nb : std::false_type {};
template <typename Policy> struct pos_in_nb<Policy, std::void_t<decltype(Policy::kPosInNb)>>
    : std::bool_constant<Policy::kPosInNb> {};

// `static constexpr bool kDeepReplay = true` makes the list kernel keep two batches of gathers in flight
// instead of one: pays for policies that are light on registers.
template <typename Policy, typename = void> struct deep_replay : std::false_type {};
template <typename Policy> struct deep_replay<Policy, std::void_t<decltype(Policy::kDeepReplay)>>
    : std::bool_constant<Policy::kDeepReplay> {};

// `static constexpr bool kRowReplay = true`: the list pass reads its neighbours from row tiles in LDS (filled with
// coalesced loads of positions and Nb records) instead of gathering every one from global memory
// (list_rows_kernel).  For policies whose `pair` reads nothing of the neighbour beyond Nb and the position.
template <typename Policy, typename = void> struct row_replay : std::false_type {};
template <typename Policy> struct row_replay<Policy, std::void_t<decltype(Policy::kRowReplay)>>
    : std::bool_constant<Policy::kRowReplay> {};

// `static constexpr bool kHitReplay = true`: between two sorts the pass replays the hit record of the WCSPH density
// pass (bit masks per neighbour row, fsa_tile.h) from row tiles in LDS (hit_replay_kernel) instead of a neighbour
// list.  For policies that reach TYPE_FLUID neighbours inside the support only, skip the particle itself, and
// read nothing of a neighbour beyond Nb and the position.
template <typename Policy, typename = void> struct hit_replay : std::false_type {};
template <typename Policy> struct hit_replay<Policy, std::void_t<decltype(Policy::kHitReplay)>>
    : std::bool_constant<Policy::kHitReplay> {};
// ... `static constexpr bool kWallWords = true` beside it: the policy also reaches the TYPE_BOUNDARY particles of the
// 27 cells, wherever they are (reach() == 2); the record lists those for every fluid particle in words of their own.
template <typename Policy, typename = void> struct wall_words : std::false_type {};
template <typename Policy> struct wall_words<Policy, std::void_t<decltype(Policy::kWallWords)>>
    : std::bool_constant<Policy::kWallWords> {};

// One particle against its 27 cells in the reference's order, neighbours from global memory.
template <typename T, typename Policy>
__device__ __forceinline__ void generic_walk(const DevParams<T> &P, const PassArrays<T> &A, int index,
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
                Policy::pair(A, P, S, j == index, tj, j, xx, xy, xz, d2, pj, Policy::load(A, j));
            }
        }
}

template <typename T, typename Policy>
__global__ __launch_bounds__(kBlock) void generic_pass_kernel(DevParams<T> P, int n, PassArrays<T> A,
                                                              const int *__restrict__ cellStart) {
    if (P.halt && *P.halt) return;  // see DevParams::halt
    const int index = blockIdx.x * kBlock + threadIdx.x;
    if (index >= n) return;
    const Vec4<T> pi = A.posType[index];
    if (!Policy::wants(unpack_int(pi.w))) {
        Policy::skip(A, P, index);
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
__global__ __launch_bounds__(BS) void tuned_pass_kernel(DevParams<float> P, int n, PassArrays<float> A,
                                                        const int *__restrict__ cellStart,
                                                        const int *__restrict__ sortedHash, int xcd_log) {
    using Shared = RowShared<BS, ROWCAP, true>;
    __shared__ Shared sh;
    __shared__ unsigned short queue[(Q + 2 * kDrainBatch) * BS];  // the drain reads ahead of the last entry
    if (P.halt && *P.halt) return;  // see DevParams::halt
    const int tid = threadIdx.x;
    const int blk = tile_block<ROWCAP>(P, xcd_log);
    if (blk < 0) return;
    const int b0 = blk * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    unsigned hlo = 0, hhi = 0;
    if (!P.rows) {  // no row table: the rows follow from the cells of the block's first and last particle
        hlo = (unsigned)sortedHash[b0];
        hhi = (unsigned)sortedHash[b0 + nb - 1];
    }
    const int cls = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, hlo, hhi, P.rows);
    if (cls != 0 && cls != row_class<ROWCAP>()) return;  // the other instantiation's block
    const bool staged = cls != 0;

    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    bool active = false;
    if (index < n) {
        pi = A.posType[index];
        active = Policy::wants(unpack_int(pi.w));
        if (!active) Policy::skip(A, P, index);
    }
    if (!__syncthreads_or(active ? 1 : 0)) return;  // e.g. a block of wall particles
    typename Policy::Self S;
    if (active) Policy::init(A, P, index, S);
    // The tiles the cull walks: the block's nine rows, or, for a block whose particles lie in several stretches of
    // cells (class 0 in the grid, fsa_tile.h), segments of its own; a queue entry is then segment << 9 | index.
    __shared__ Segments sg;
    if (!staged) {
        bool sparse = false;
        if constexpr (ROWCAP <= kRowCap) {
            if (P.rows && P.rows[(size_t)blk * kRowTableStride + 19])
                sparse = sparse_segments<ROWCAP>(P, b0, nb, sortedHash, cellStart, reinterpret_cast<int *>(sh.x), sg);
        }
        if (!sparse) {
            if (active) {
                generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
                Policy::finish(A, P, index, S);
            }
            return;
        }
    } else {
        rows_as_segments(sh, sg);
        __syncthreads();
    }
    const int nseg = sg.count;
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
            B.j[u] = s + u < cnt ? sg.start[e >> idx_bits<ROWCAP>()] + (e & ((1 << idx_bits<ROWCAP>()) - 1)) : safe;
        }
#pragma unroll
        for (int u = 0; u < kDrainBatch; u++) {
            B.nb[u] = Policy::load(A, B.j[u]);
            if constexpr (pos_in_nb<Policy>::value) B.pj[u] = B.nb[u].pos();
            else B.pj[u] = A.posType[B.j[u]];
        }
    };
    auto batch_eval = [&](const Batch &B, int s) {
#pragma unroll
        for (int u = 0; u < kDrainBatch; u++)
            if (s + u < cnt) {
                const float xx = pi.x - B.pj[u].x, xy = pi.y - B.pj[u].y, xz = pi.z - B.pj[u].z;
                const float d2 = xx * xx + xy * xy + xz * xz;
                Policy::pair(A, P, S, B.j[u] == index, unpack_int(B.pj[u].w), B.j[u], xx, xy, xz, d2, B.pj[u], B.nb[u]);
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

    float4 e[row_regs<BS, ROWCAP>()];
    int lj, le, nlj = 0, nle = 0;
    tile_fetch<BS, ROWCAP, false>(sg.start[0], sg.len[0], A.posType, e);
    segment_range(P, sg, walk, active, 0, cellStart, lj, le);
    row_store<BS, ROWCAP, true>(sh, 0, e);
    __syncthreads();
    for (int r = 0; r < nseg; r++) {
        const int buf = r & 1;
        if (r + 1 < nseg) {
            tile_fetch<BS, ROWCAP, false>(sg.start[r + 1], sg.len[r + 1], A.posType, e);
            segment_range(P, sg, walk, active, r + 1, cellStart, nlj, nle);
        }
        const int tag = r << idx_bits<ROWCAP>();
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
        if (r + 1 < nseg) {
            row_store<BS, ROWCAP, true>(sh, buf ^ 1, e);
            lj = nlj;
            le = nle;
            __syncthreads();
        }
    }
    drain();
    if (active) Policy::finish(A, P, index, S);
}

// The list flavour (fp32): the hits of every lane were recorded once after the sort (list_build_kernel in
// fsa_lists.hip: the cull of the tiled kernel with the same slack, every particle type, entries in the same
// order) and are replayed here, so a pass costs the gathers and the policy's arithmetic only.  Same `pair`
// calls in the same order as the tiled kernel: same bits.  Blocks the row table marks as not staged, and lanes
// whose list overflowed, walk the cells.
// Workgroups per CU the register budgets of the two list kernels are set for (1: whatever the kernel needs).  Forcing
// more (5 to 8) was measured on the IISPH and DFSPH steps at 16 M particles: the policies with two batches of gathers in
// flight need their 100-odd registers and spill instead (density-change passes 1.8 -> 5.6 ms per launch).
#ifndef FSA_LIST_BLOCKS
#define FSA_LIST_BLOCKS 1
#endif
#ifndef FSA_ROWS_BLOCKS
#define FSA_ROWS_BLOCKS 1
#endif
template <typename Policy, int BS, int K>
__global__ __launch_bounds__(BS, FSA_LIST_BLOCKS) void list_pass_kernel(DevParams<float> P, int n, PassArrays<float> A,
                                                       const int *__restrict__ cellStart,
                                                       const int *__restrict__ sortedHash,
                                                       const unsigned short *__restrict__ list,
                                                       const unsigned short *__restrict__ counts, int xcd_log) {
    __shared__ int rowStartG[32];  // 9 used; up to kMaxSegs by a block with segments of its own
    __shared__ int stagedFlag, sparseFlag;
    __shared__ Segments sg;                          // (blocks with segments of their own only,
    __shared__ int sparseTmp[20 * kSparseRuns + 16];  // see sparse_segments in fsa_tile.h)
    static_assert(kMaxSegs <= 32, "segment table too small");
    if (P.halt && *P.halt) return;  // see DevParams::halt
    const int tid = threadIdx.x;
    const int b0 = xcd_block(xcd_log) * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int index = b0 + tid;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    bool active = false;
    if (index < n) {
        pi = A.posType[index];
        active = Policy::wants(unpack_int(pi.w));
        if (!active) Policy::skip(A, P, index);
    }
    if (!__syncthreads_or(active ? 1 : 0)) return;  // e.g. a block of wall particles
    if (tid < 9) rowStartG[tid] = P.rows[(size_t)(b0 / BS) * kRowTableStride + tid];
    if (tid == 18) stagedFlag = P.rows[(size_t)(b0 / BS) * kRowTableStride + 18];
    if (tid == 19) sparseFlag = P.rows[(size_t)(b0 / BS) * kRowTableStride + 19];
    __syncthreads();
    typename Policy::Self S;
    if (active) Policy::init(A, P, index, S);
    if (!stagedFlag) {
        // a block whose list was built over segments of its own: the same segments, their starts in the table's place
        bool sparse = false;
        if (sparseFlag) sparse = sparse_segments<kRowCap>(P, b0, min(BS, n - b0), sortedHash, cellStart, sparseTmp, sg);
        if (!sparse) {
            if (active) {
                generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
                Policy::finish(A, P, index, S);
            }
            return;
        }
        if (tid < sg.count) rowStartG[tid] = sg.start[tid];
        __syncthreads();
    }
    const int bits = stagedFlag == 2 ? idx_bits<kRowCapBig>() : idx_bits<kRowCap>();  // as recorded for the block's class
    const unsigned mask = (1u << bits) - 1u;
    int cnt = active ? (int)counts[index] : 0;
    const bool overflow = cnt == kListOverflow;
    if (overflow) cnt = 0;
    const int safe = index < n ? index : 0;
    // four consecutive slots of a lane are one 8-byte word; a wavefront reads 512 contiguous bytes per word
    const uint2 *const L = reinterpret_cast<const uint2 *>(list) + (size_t)(b0 / BS) * (K / 4) * BS + tid;
    struct Batch {
        int j[4];
        float4 pj[4];
        typename Policy::Nb nbr[4];
    };
    auto entries = [&](int s) { return s < cnt ? L[(size_t)(s / 4) * BS] : make_uint2(0u, 0u); };
    auto gather = [&](Batch &B, const uint2 w, int s) {
        const unsigned e[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // (the row look-up is unconditional -- slots past the end hold entry 0 -- so that the four LDS reads are
            // issued together and a select, not a branch, picks the lane's own index for an unused slot)
            const int jr = rowStartG[(e[u] >> bits) & 31u] + (int)(e[u] & mask);  // (& 31: stale slots of the last word)
            B.j[u] = s + u < cnt ? jr : safe;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            B.nbr[u] = Policy::load(A, B.j[u]);
            if constexpr (pos_in_nb<Policy>::value) B.pj[u] = B.nbr[u].pos();
            else B.pj[u] = A.posType[B.j[u]];
        }
    };
    auto evaluate = [&](const Batch &B, int s) {
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (s + u < cnt) {
                const int tj = unpack_int(B.pj[u].w);
                if (Policy::reach(tj) != 0) {
                    const float xx = pi.x - B.pj[u].x, xy = pi.y - B.pj[u].y, xz = pi.z - B.pj[u].z;
                    const float d2 = xx * xx + xy * xy + xz * xz;
                    Policy::pair(A, P, S, B.j[u] == index, tj, B.j[u], xx, xy, xz, d2, B.pj[u], B.nbr[u]);
                }
            }
    };
    if constexpr (deep_replay<Policy>::value) {
        // two batches of four in flight: the gathers of one are issued before the other is evaluated, and the
        // entry words run two batches ahead
        Batch B0, B1;
        uint2 w1 = entries(4), w2 = entries(8);
        gather(B0, entries(0), 0);
        for (int s = 0; wave_any(s < cnt); s += 8) {
            gather(B1, w1, s + 4);
            w1 = entries(s + 12);
            evaluate(B0, s);
            if (!wave_any(s + 4 < cnt)) break;
            gather(B0, w2, s + 8);
            w2 = entries(s + 16);
            evaluate(B1, s + 4);
        }
    } else {
        // one batch of four, the next entry word in flight
        Batch B;
        uint2 cur = entries(0);
        for (int s = 0; wave_any(s < cnt); s += 4) {
            const uint2 nxt = entries(s + 4);
            gather(B, cur, s);
            evaluate(B, s);
            cur = nxt;
        }
    }
    if (overflow) generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
    if (active) Policy::finish(A, P, index, S);
}

// The list flavour with the neighbours in LDS.  A gather of 16 bytes per lane costs the vector memory pipeline one
// look-up per distinct cache line, some 25 per instruction here, and a list pass makes one or two per neighbour; a
// tile read costs four LDS cycles.  So the nine neighbour rows of the block stream through LDS once more, this time
// holding everything `pair` reads of a neighbour (position, type, the policy's Nb record; filled with coalesced
// loads, double buffered), and every lane walks its list, which is in row order, taking its neighbours from the
// tiles.  A lane's entries per row vary a lot, so a lane may lag one row behind (both tiles are in LDS): the loop of
// row r runs until nobody has entries of row r - 1 left, then the tile of r + 1 replaces that of r - 1, and it goes
// on until fewer than kRowQuorum lanes have entries of r left.  Same `pair` calls in the same order as
// list_pass_kernel: same bits.  Blocks whose rows need the large tiles, blocks that are not staged and lanes whose
// list overflowed take list_pass_kernel's way.
constexpr int kRowQuorum = 40;
template <typename Policy, int BS, int ROWCAP, int K>
__global__ __launch_bounds__(BS, FSA_ROWS_BLOCKS) void list_rows_kernel(DevParams<float> P, int n, PassArrays<float> A,
                                                       const int *__restrict__ cellStart,
                                                       const int *__restrict__ sortedHash,
                                                       const unsigned short *__restrict__ list,
                                                       const unsigned short *__restrict__ counts, int xcd_log) {
    using Nb = typename Policy::Nb;
    constexpr bool kPos = !pos_in_nb<Policy>::value;  // positions in a tile of their own
    constexpr int kRegs = (ROWCAP + BS - 1) / BS;
    __shared__ int rowStartG[32], rowLen[16];  // 9 used; rowStartG up to kMaxSegs by a block with segments of its own
    __shared__ int stagedFlag, sparseFlag;
    __shared__ float4 tilePos[kPos ? 2 * ROWCAP : 1];
    __shared__ Nb tileNb[2 * ROWCAP];
    static_assert(kMaxSegs <= 32, "segment table too small");
    static_assert(sizeof(Nb) * 2 * ROWCAP >= sizeof(int) * (20 * kSparseRuns + 16) + sizeof(Segments) + 16,
                  "the tile is the scratch space of sparse_segments");
    if (P.halt && *P.halt) return;  // see DevParams::halt
    const int tid = threadIdx.x;
    const int b0 = xcd_block(xcd_log) * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int index = b0 + tid;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    bool active = false;
    if (index < n) {
        pi = A.posType[index];
        active = Policy::wants(unpack_int(pi.w));
        if (!active) Policy::skip(A, P, index);
    }
    if (!__syncthreads_or(active ? 1 : 0)) return;  // e.g. a block of wall particles
    if (tid < 9) rowStartG[tid] = P.rows[(size_t)(b0 / BS) * kRowTableStride + tid];
    else if (tid < 18) rowLen[tid - 9] = P.rows[(size_t)(b0 / BS) * kRowTableStride + tid];
    else if (tid == 18) stagedFlag = P.rows[(size_t)(b0 / BS) * kRowTableStride + 18];
    else if (tid == 19) sparseFlag = P.rows[(size_t)(b0 / BS) * kRowTableStride + 19];
    __syncthreads();
    typename Policy::Self S;
    if (active) Policy::init(A, P, index, S);
    const int cls = stagedFlag;
    if (cls == 0) {
        // a block whose list was built over segments of its own (sparse_segments, fsa_tile.h; the tiles, which such a
        // block does not use, lend the space): the same segments, their starts in the table's place, gathers
        bool sparse = false;
        if (sparseFlag) {
            int *const tmp = reinterpret_cast<int *>(tileNb);
            Segments &sg = *reinterpret_cast<Segments *>(tmp + 20 * kSparseRuns + 16);
            sparse = sparse_segments<kRowCap>(P, b0, min(BS, n - b0), sortedHash, cellStart, tmp, sg);
            if (sparse && tid < sg.count) rowStartG[tid] = sg.start[tid];
            __syncthreads();
        }
        if (!sparse) {
            if (active) {
                generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
                Policy::finish(A, P, index, S);
            }
            return;
        }
    }
    int cnt = active ? (int)counts[index] : 0;
    const bool overflow = cnt == kListOverflow;
    if (overflow) cnt = 0;
    const uint2 *const L = reinterpret_cast<const uint2 *>(list) + (size_t)(b0 / BS) * (K / 4) * BS + tid;
    auto entries = [&](int s) { return s < cnt ? L[(size_t)(s / 4) * BS] : make_uint2(0u, 0u); };
    auto one_pair = [&](const float4 &pj, const Nb &nb, int j) {
        const int tj = unpack_int(pj.w);
        if (Policy::reach(tj) != 0) {
            const float xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
            const float d2 = xx * xx + xy * xy + xz * xz;
            Policy::pair(A, P, S, j == index, tj, j, xx, xy, xz, d2, pj, nb);
        }
    };
    if (cls != 1) {  // large tiles (entries carry 10 index bits) or segments of the block's own (9): gathers
        const int bits = cls == 2 ? idx_bits<kRowCapBig>() : idx_bits<kRowCap>();
        const unsigned mask = (1u << bits) - 1u;
        for (int s = 0; wave_any(s < cnt); s += 4) {
            const uint2 w = entries(s);
            const unsigned e[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (s + u < cnt) {
                    const int j = rowStartG[(e[u] >> bits) & 31u] + (int)(e[u] & mask);
                    const Nb nb = Policy::load(A, j);
                    float4 pj;
                    if constexpr (kPos) pj = A.posType[j]; else pj = nb.pos();
                    one_pair(pj, nb, j);
                }
        }
    } else {
        constexpr int bits = idx_bits<kRowCap>();
        constexpr unsigned mask = (1u << bits) - 1u;
        static_assert(ROWCAP >= kRowCap, "tile too small for the rows of a class-1 block");
        struct Stage {
            float4 p;
            Nb nb;
        };
        Stage e[kRegs];
        auto fetch = [&](int r) {
            const int s0 = rowStartG[r], len = rowLen[r];
            const int last = len > 0 ? len - 1 : 0;
#pragma unroll
            for (int k = 0; k < kRegs; k++) {
                const int j = s0 + min(tid + k * BS, last);
                if constexpr (kPos) e[k].p = A.posType[j];
                e[k].nb = Policy::load(A, j);
            }
        };
        auto store = [&](int r) {
            const int buf = (r & 1) * ROWCAP, len = rowLen[r];
#pragma unroll
            for (int k = 0; k < kRegs; k++) {
                const int t = tid + k * BS;
                if (t < len) {
                    if constexpr (kPos) tilePos[buf + t] = e[k].p;
                    tileNb[buf + t] = e[k].nb;
                }
            }
        };
        // the lane's entry stream: four entries per word, the next word in flight
        int s = 0;               // entries taken so far
        unsigned long long cur;  // the entries of the word at hand that are still to come, lowest first
        uint2 nxt;
        {
            const uint2 w = entries(0);
            cur = ((unsigned long long)w.y << 32) | w.x;
            nxt = entries(4);
        }
        struct Taken {
            bool has;
            int j;
            float4 pj;
            Nb nb;
        };
        fetch(0);
        store(0);
        __syncthreads();
        fetch(1);
        for (int r = 0; r < 9; r++) {
            // Takes the lane's next entry if it belongs to a row up to r and starts the tile reads.  `part` 0: until
            // no lane of the wavefront has an entry of a row before r; 1: until fewer than kRowQuorum lanes have
            // one of row r (the last row: none).
            auto next = [&](Taken &T, int part) -> bool {
                const unsigned ent = (unsigned)cur & 0xffffu;
                const int row = (int)(ent >> bits);
                const bool has = s < cnt && row <= r;
                if (part == 0) {
                    if (!wave_any(s < cnt && row < r)) return false;
                } else {
                    const unsigned long long busy = __ballot(has);
                    if (busy == 0ull || (r + 1 < 9 && __popcll(busy) < kRowQuorum)) return false;
                }
                const int t = has ? (int)(ent & mask) : 0, at = has ? (row & 1) * ROWCAP + t : 0;
                T.has = has;
                T.j = rowStartG[has ? row : 0] + t;
                T.nb = tileNb[at];
                if constexpr (kPos) T.pj = tilePos[at];
                if (has) {
                    s++;
                    cur >>= 16;
                    if ((s & 3) == 0) {
                        cur = ((unsigned long long)nxt.y << 32) | nxt.x;
                        nxt = entries(s + 4);
                    }
                }
                return true;
            };
            auto eval = [&](const Taken &T) {
                if (!T.has) return;
                float4 pj;
                if constexpr (kPos) pj = T.pj; else pj = T.nb.pos();
                one_pair(pj, T.nb, T.j);
            };
            // two stages, roles rotating: the next neighbour's tile reads are in flight while one is evaluated
            auto run = [&](int part) {
                Taken T0, T1;
                bool go = next(T0, part);
                while (go) {
                    const bool go1 = next(T1, part);
                    eval(T0);
                    if (!go1) break;
                    go = next(T0, part);
                    eval(T1);
                }
            };
            run(0);
            if (r + 1 < 9) {
                __syncthreads();  // everyone is done with the tile of row r - 1
                store(r + 1);
                if (r + 2 < 9) fetch(r + 2);
            }
            run(1);
            if (r + 1 < 9) __syncthreads();  // the tile of row r + 1 is visible
        }
    }
    if (overflow) generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
    if (active) Policy::finish(A, P, index, S);
}

// The replay of the recorded hits (see hit_replay above).  The record of fsa_sph_fast.hip marks, for every particle
// and every neighbour row (or segment) of its workgroup's block, which of the row's fluid particles lie inside the
// support on the fused squared distance, the particle itself left out: one 57-candidate word per row in registers,
// nothing searched, no list entries to decode, and no load inside the loops but the tile reads.  The rows stream
// through LDS as in list_rows_kernel (positions and Nb records, coalesced loads, double buffered; a lane may lag one
// row behind the block), the policy's `pair` applies the reference's own tests to every marked neighbour, in
// neighbour order.  Particles without a record (kHitOverflow; every particle of a block that cannot be tiled) walk
// the cells.  Blocks per CU: as the WCSPH force replay, which has this structure and is bound by latency.
#ifndef FSA_REPLAY_BLOCKS
#define FSA_REPLAY_BLOCKS 6
#endif
// `static constexpr int kReplayPer = 2`: a lane takes two neighbours per trip of the loop (the trip's tests and
// ballots are shared; pays for a light `pair` with few registers).
template <typename Policy, typename = void> struct replay_per : std::integral_constant<int, 1> {};
template <typename Policy> struct replay_per<Policy, std::void_t<decltype(Policy::kReplayPer)>>
    : std::integral_constant<int, Policy::kReplayPer> {};
template <typename Nb, int ROWCAP, bool POS> struct ReplayLds {
    RowShared<1, 4, false> sh;  // only its row table is used here
    Segments sg;
    float4 pos[POS ? 2 * ROWCAP : 1];  // two tiles: segment s lives in tile s & 1
    Nb nb[2 * ROWCAP];
};
template <typename Policy, int BS, int ROWCAP>
__global__ __launch_bounds__(BS, FSA_REPLAY_BLOCKS) void hit_replay_kernel(DevParams<float> P, int n, PassArrays<float> A,
                                                        const int *__restrict__ cellStart,
                                                        const int *__restrict__ sortedHash,
                                                        const uint2 *__restrict__ hitWords,
                                                        const unsigned char *__restrict__ hitCount,
                                                        const unsigned char *__restrict__ wallCount, int run_log) {
    using Nb = typename Policy::Nb;
    constexpr bool kPos = !pos_in_nb<Policy>::value;  // positions in a tile of their own
    constexpr int kRegs = row_regs<BS, ROWCAP - kChunk>();
    constexpr int kQuorum = 40;
    constexpr int kPer = replay_per<Policy>::value;  // neighbours a lane takes per trip
    __shared__ ReplayLds<Nb, ROWCAP, kPos> lds;
    static_assert((kPos ? sizeof(float4) : sizeof(Nb)) * 2 * ROWCAP >= sizeof(int) * (20 * kSparseRuns + 16),
                  "a tile is the scratch space of sparse_segments");
    if (P.halt && *P.halt) return;  // see DevParams::halt
    Segments &sg = lds.sg;
    const int tid = threadIdx.x;
    const int blk = xcd_block(run_log);
    const int b0 = blk * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    if (P.nz != P.gres[2]) {
        // a slab solver: the record leaves out the blocks that lie entirely in the outer halo layers, whose results
        // nobody reads (the halo layers get every pass's results from the slabs that own them)
        const unsigned hlo = (unsigned)sortedHash[b0], hhi = (unsigned)sortedHash[b0 + nb - 1];
        if (hhi != (unsigned)FSA_GRID_UNDEF && run_unneeded(P, hlo, hhi, 1)) return;
    }
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    bool active = false;
    if (index < n) {
        pi = A.posType[index];
        active = Policy::wants(unpack_int(pi.w));
        if (!active) Policy::skip(A, P, index);
    }
    if (!__syncthreads_or(active ? 1 : 0)) return;  // e.g. a block of wall particles
    // the same tiles as the pass that recorded: the block's rows, or segments of its own
    const int cls = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, lds.sh, 0u, 0u, P.rows);
    bool staged = cls != 0;
    if (staged) setup_segments<ROWCAP>(lds.sh, cls, sg);
    else if (P.rows[(size_t)blk * kRowTableStride + 19])
        staged = sparse_segments<ROWCAP>(P, b0, nb, sortedHash, cellStart,
                                         reinterpret_cast<int *>(kPos ? (void *)lds.pos : (void *)lds.nb), sg);
    typename Policy::Self S;
    int cnt = 0;       // extra words of the lane's record
    bool rec = false;  // the lane has a record
    if (active) {
        Policy::init(A, P, index, S);
        cnt = hitCount[index];
        rec = cnt != kHitOverflow;
        if (!rec) generic_walk<float, Policy>(P, A, index, pi, cellStart, S);
    }
    if (!rec) cnt = 0;
    if (staged) {
        const uint2 *const wp = hitWords + (size_t)blk * kHitSlots * BS + tid;
        constexpr unsigned kNoSeg = 0xffffu;
        auto extra_at = [&](int k) { return k < cnt ? wp[(size_t)(kMaxSegs + k) * BS] : make_uint2(0u, kNoSeg << kHitSegShift); };
        int xk = 0;                  // extra words taken so far
        uint2 xw = extra_at(0);      // the next one
        unsigned cur = 0, pend = 0;  // bits at hand (32 candidates from `cbase` on), and the 25 after them
        int cseg = -1;               // the segment they belong to
        int cbase = 0;               // tile entry of bit 31 of `cur`, the two tiles counted through
        struct Stage {
            float4 p;
            Nb nb;
        };
        Stage e[kRegs];
        auto fetch = [&](int s, int len) {
            const int last = len > 0 ? len - 1 : 0;
#pragma unroll
            for (int k = 0; k < kRegs; k++) {
                const int j = s + min(tid + k * BS, last);
                if constexpr (kPos) e[k].p = A.posType[j];
                e[k].nb = Policy::load(A, j);
            }
        };
        auto store = [&](int buf, int len) {
#pragma unroll
            for (int k = 0; k < kRegs; k++) {
                const int i = tid + k * BS;
                if (i < len) {
                    if constexpr (kPos) lds.pos[buf * ROWCAP + i] = e[k].p;
                    lds.nb[buf * ROWCAP + i] = e[k].nb;
                }
            }
        };
        __syncthreads();  // the segment table is visible
        fetch(sg.start[0], sg.len[0]);
        uint2 fw = rec ? wp[0] : make_uint2(0u, 0u);  // main word of the segment whose loop runs
        store(0, sg.len[0]);
        __syncthreads();
        const int nseg = sg.count;
        if (nseg > 1) fetch(sg.start[1], sg.len[1]);
        for (int r = 0; r < nseg; r++) {
            uint2 fwn = make_uint2(0u, 0u);
            if (r + 1 < nseg && rec) fwn = wp[(size_t)(r + 1) * BS];  // in flight during this segment's loop
            struct Taken {
                bool has[kPer];
                int j[kPer];
                float4 pj[kPer];
                Nb nb[kPer];
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
            // segment r; 1: until fewer than kQuorum lanes have bits left (all of them in the last segment).
            auto next = [&](Taken &T, int part) -> bool {
                refill();
                const bool has = cur != 0;
                if (part == 0) {
                    if (!wave_any(cseg < r)) return false;  // a lane moves on to r when it is through with r - 1
                } else {
                    const unsigned long long busy = __ballot(has);
                    if (busy == 0ull || (r + 1 < nseg && __popcll(busy) < kQuorum)) return false;
                }
#pragma unroll
                for (int u = 0; u < kPer; u++) {
                    if (u > 0) {  // the bits beyond the first 32 of the main word follow without a trip of their own
                        const bool on = cur == 0 && pend != 0;
                        cur = on ? pend : cur;
                        cbase += on ? 32 : 0;
                        pend = on ? 0u : pend;
                    }
                    const bool hu = cur != 0;
                    const int k = __builtin_clz(cur | 1u);  // cur == 0: nothing to clear
                    cur &= ~(0x80000000u >> k);
                    const int t = hu ? cbase + k : 0;
                    T.has[u] = hu;
                    T.j[u] = hu ? sg.start[cseg] + t - (cseg & 1) * ROWCAP : 0;  // (dropped for policies that ignore it)
                    T.nb[u] = lds.nb[t];
                    if constexpr (kPos) T.pj[u] = lds.pos[t];
                }
                return true;
            };
            auto eval = [&](const Taken &T) {
#pragma unroll
                for (int u = 0; u < kPer; u++) {
                    if (!T.has[u]) continue;
                    float4 pj;
                    if constexpr (kPos) pj = T.pj[u]; else pj = T.nb[u].pos();
                    const float xx = pi.x - pj.x, xy = pi.y - pj.y, xz = pi.z - pj.z;
                    const float d2 = xx * xx + xy * xy + xz * xz;
                    Policy::pair(A, P, S, false, FSA_TYPE_FLUID, T.j[u], xx, xy, xz, d2, pj, T.nb[u]);
                }
            };
            // two stages, roles rotating: the next neighbour's tile reads are in flight while one is evaluated
            auto run = [&](int part) {
                Taken T0, T1;
                bool go = next(T0, part);
                while (go) {
                    const bool go1 = next(T1, part);
                    eval(T0);
                    if (!go1) break;
                    go = next(T0, part);
                    eval(T1);
                }
            };
            run(0);
            if (r + 1 < nseg) {
                __syncthreads();  // everyone is done with the tile of segment r - 1
                store((r + 1) & 1, sg.len[r + 1]);
                if (r + 2 < nseg) fetch(sg.start[r + 2], sg.len[r + 2]);
            }
            run(1);
            if (r + 1 < nseg) {
                fw = fwn;
                __syncthreads();  // the tile of segment r + 1 is visible
            }
        }
    }
    if constexpr (wall_words<Policy>::value) {
        // the wall neighbours, in neighbour order (no barrier follows: a wavefront with particles at a wall ends later)
        if (rec) {
            const int ww = wallCount[index];
            const uint2 *const wwp = hitWords + (size_t)blk * kHitSlots * BS + (size_t)(kMaxSegs + kHitExtra) * BS + tid;
            for (int k = 0; k < ww; k++) {
                const uint2 w = wwp[(size_t)k * BS];
                unsigned m = w.x;
                while (m) {  // four neighbours' loads in flight at a time
                    constexpr int kAhead = 4;
                    int js[kAhead];
                    bool has[kAhead];
#pragma unroll
                    for (int u = 0; u < kAhead; u++) {
                        has[u] = m != 0;
                        const int b = __builtin_clz(m | 1u);
                        if (has[u]) m &= ~(0x80000000u >> b);
                        js[u] = has[u] ? (int)w.y + b : index;
                    }
                    float4 pj[kAhead];
                    Nb nbr[kAhead];
#pragma unroll
                    for (int u = 0; u < kAhead; u++) {
                        nbr[u] = Policy::load(A, js[u]);
                        if constexpr (kPos) pj[u] = A.posType[js[u]]; else pj[u] = nbr[u].pos();
                    }
#pragma unroll
                    for (int u = 0; u < kAhead; u++) {
                        if (!has[u]) continue;
                        const float xx = pi.x - pj[u].x, xy = pi.y - pj[u].y, xz = pi.z - pj[u].z;
                        const float d2 = xx * xx + xy * xy + xz * xz;
                        Policy::pair(A, P, S, false, FSA_TYPE_BOUNDARY, js[u], xx, xy, xz, d2, pj[u], nbr[u]);
                    }
                }
            }
        }
    }
    if (active) Policy::finish(A, P, index, S);
}

// The elements i, i + stride, i + 2 stride, ... below n, folded into `acc` in that order -- eight loads in flight at a
// time (one dependent load per trip reads at a quarter of the memory rate); the order, and so every bit of a
// floating-point sum, is that of the plain loop.
template <typename T, typename Acc, typename Fold>
__device__ __forceinline__ void strided_fold(const T *__restrict__ in, int i, int stride, int n, Acc &acc, Fold fold) {
    constexpr int kAhead = 8;
    for (; (long long)i + (long long)(kAhead - 1) * stride < n; i += kAhead * stride) {
        T v[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < kAhead; u++) fold(acc, v[u]);
    }
    for (; i < n; i += stride) fold(acc, in[i]);
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

template <typename T> static PassArrays<T> arrays_of(Solver *s) {
    const int c = s->cur;
    PassArrays<T> A{};
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
    A.uniqueId = s->b.uniqueId[c];
    A.groupSorted = s->b.groupSorted;
    A.normalById = (const Vec3<T> *)s->b.normal;
    A.pressure = (const T *)s->b.pressure;
    A.pressureOut = (T *)s->b.pressure;
    A.dii = (Vec3<T> *)s->b.dii;
    A.dijpjl = (Vec3<T> *)s->b.dijpjl;
    A.aii = (T *)s->b.aii;
    A.densityStar = (T *)s->b.density_star;
    A.pressureForce = (Vec3<T> *)s->b.pressureForce;
    return A;
}

constexpr int kQueueDf = 48;  // 16-bit slots, as in the WCSPH force pass

static bool use_tiles(const Solver *s) { return s->precision == FSA_PRECISION_F32 && s->variant >= 1; }

// Does Policy replay the hit record in this solver?  (fsa_set_neighbour_lists(s, 0) makes every pass search.)
template <typename Policy> static bool replays_hits(const Solver *s) {
    return hit_replay<Policy>::value && s->use_lists && s->use_replay && s->precision == FSA_PRECISION_F32 &&
           s->variant == 3 && s->sorted_valid && s->pf.rows && s->b.hitWords;
}
// What a pass that repeats between two sorts replays: the hit record or the neighbour lists, made here if stale.
template <typename Policy> static int ensure_neighbours(Solver *s, bool lists = true) {
    if (replays_hits<Policy>(s)) return ensure_hit_record(s);
    return lists ? ensure_lists(s) : FSA_OK;
}

// Policy<float> over arrays A through the tiled kernel, or, with `lists`, through the hit record or the recorded
// neighbour lists (worth it for passes that run many times per sort; the policy must not reach any type beyond the
// support).
template <typename Policy> static int run_tiled(Solver *s, const PassArrays<float> &A, bool lists) {
    const int unit = kXcds << s->xcd_run_log;
    const int nb = ((s->n + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
    if (replays_hits<Policy>(s) && s->hits_valid) {
        if constexpr (hit_replay<Policy>::value)
            hipLaunchKernelGGL((hit_replay_kernel<Policy, kTunedBS, kRowCap>), dim3(nb), dim3(kTunedBS), 0, s->stream,
                               s->pf, s->n, A, s->b.cellStart, s->b.sortedHash, (const uint2 *)s->b.hitWords,
                               s->b.hitCount, s->b.wallCount, s->xcd_run_log);
    } else if (lists && s->use_lists && s->lists_valid) {
        if constexpr (row_replay<Policy>::value)
            hipLaunchKernelGGL((list_rows_kernel<Policy, kTunedBS, kRowCap, kListSlots>), dim3(nb), dim3(kTunedBS), 0,
                               s->stream, s->pf, s->n, A, s->b.cellStart, s->b.sortedHash, s->b.nbList, s->b.nbCount,
                               s->xcd_run_log);
        else
            hipLaunchKernelGGL((list_pass_kernel<Policy, kTunedBS, kListSlots>), dim3(nb), dim3(kTunedBS), 0, s->stream,
                               s->pf, s->n, A, s->b.cellStart, s->b.sortedHash, s->b.nbList, s->b.nbCount, s->xcd_run_log);
    } else
    {
        hipLaunchKernelGGL((tuned_pass_kernel<Policy, kTunedBS, kRowCap, kQueueDf>), dim3(nb), dim3(kTunedBS), 0,
                           s->stream, s->pf, s->n, A, s->b.cellStart, s->b.sortedHash, s->xcd_run_log);
        if (s->pf.big)  // blocks with rows beyond kRowCap: large tiles, launched over the list of such blocks
            hipLaunchKernelGGL((tuned_pass_kernel<Policy, kTunedBS, kRowCapBig, kQueueDf>),
                               dim3((s->n + kTunedBS - 1) / kTunedBS), dim3(kTunedBS), 0, s->stream, s->pf, s->n, A,
                               s->b.cellStart, s->b.sortedHash, s->xcd_run_log);
    }
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

template <typename Policy> static int launch_tuned(Solver *s, const char *tag, bool lists = false) {
    const int rc = ensure_neighbours<Policy>(s, lists);
    if (rc) return rc;
    ScopedTimer t(s, tag);
    return run_tiled<Policy>(s, arrays_of<float>(s), lists);
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

}  // namespace fsa
