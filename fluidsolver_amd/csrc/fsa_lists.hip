// Neighbour lists for the passes that repeat between two sorts (the relaxation rounds of the IISPH and DFSPH
// steps run 20-odd neighbour passes over the same positions): the cull of the tiled pass kernel
// (fsa_pass.h) is done once, its hits are written out, and the passes replay them (list_pass_kernel).
//
// The list of a particle holds, in the order of the reference's 27-cell walk (z, y, x cell loops, ascending
// index inside a cell; solver/sph_kernel_impl.cuh:255-260), every particle of any type whose fused squared
// distance is below the support radius squared times 1.000002, itself included, as 16-bit entries relative to
// the row table of its block (row << 9 | index in the row; row << 10 | index for a block with large tiles;
// segment << 9 | index in the segment for a block with segments of its own, sparse_segments in fsa_tile.h).
// Layout and limits: fsa_tile.h.
#include <type_traits>

#include "fsa_pair.h"
#include "fsa_solver.h"
#include "fsa_tile.h"

namespace fsa {

template <int BS, int ROWCAP, int Q, int K>
__global__ __launch_bounds__(BS) void list_build_kernel(DevParams<float> P, int n,
                                                        const float4 *__restrict__ posType,
                                                        const int *__restrict__ cellStart,
                                                        const int *__restrict__ sortedHash,
                                                        unsigned short *__restrict__ list,
                                                        unsigned short *__restrict__ counts, int xcd_log) {
    using Shared = RowShared<BS, ROWCAP, false>;
    __shared__ Shared sh;
    __shared__ unsigned short queue[(Q + 4) * BS];  // the flush reads up to three slots past the last entry
    const int tid = threadIdx.x;
    const int blk = tile_block<ROWCAP>(P, xcd_log);
    if (blk < 0) return;
    const int b0 = blk * BS;
    if (b0 >= n) return;  // padding blocks of the rounded-up grid
    const int nb = min(BS, n - b0);
    const int index = b0 + tid;
    const bool active = index < n;
    const int cls = setup_rows<ROWCAP>(P, b0, nb, cellStart, sortedHash, sh, 0u, 0u, P.rows);
    if (cls != 0 && cls != row_class<ROWCAP>()) return;  // the other instantiation's block
    // The tiles the cull walks: the block's nine rows, or, for a block whose particles lie in several stretches of
    // cells (class 0 in the grid, fsa_tile.h), segments of its own; an entry is then segment << 9 | index.
    __shared__ Segments sg;
    if (cls == 0) {
        bool sparse = false;
        if constexpr (ROWCAP <= kRowCap) {
            if (P.rows && P.rows[(size_t)blk * kRowTableStride + 19])
                sparse = sparse_segments<ROWCAP>(P, b0, nb, sortedHash, cellStart, reinterpret_cast<int *>(sh.x), sg);
        }
        if (!sparse) {  // the passes walk the cells for this block
            if (active) counts[index] = (unsigned short)kListOverflow;
            return;
        }
    } else {
        rows_as_segments(sh, sg);
        __syncthreads();
    }
    const int nseg = sg.count;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) pi = posType[index];
    const float sr2c = P.sr * P.sr * 1.000002f;
    const RowWalk<float> walk(P, pi.x, pi.y, pi.z);
    unsigned short *const q = queue + tid;
    uint2 *const L = reinterpret_cast<uint2 *>(list) + (size_t)(b0 / BS) * (K / 4) * BS + tid;
    int cnt = 0, total = 0;  // entries in the queue; entries written out (a multiple of four until the last flush)
    // Whole words (four entries, 8 bytes per lane: the 64 lanes of a wavefront store 512 contiguous bytes) go
    // out; up to three left-over entries move to the head of the queue and wait for the next flush.
    auto flush = [&](bool last) {
        const int full = last ? cnt : (cnt & ~3);
        for (int w = 0; wave_any(4 * w < full); w++) {
            unsigned e4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned v = q[(4 * w + u) * BS];
                e4[u] = 4 * w + u < full ? v : 0u;
            }
            const int slot = total + 4 * w;
            if (4 * w < full && slot < K) L[(size_t)(slot >> 2) * BS] = make_uint2(e4[0] | (e4[1] << 16), e4[2] | (e4[3] << 16));
        }
        const int rem = cnt - full;  // 0 .. 3 (0 after the last flush)
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const unsigned short v = q[(full + u) * BS];
            if (u < rem) q[u * BS] = v;
        }
        total += full;
        cnt = rem;
    };

    float4 e[row_regs<BS, ROWCAP>()];
    int lj, le, nlj = 0, nle = 0;
    tile_fetch<BS, ROWCAP, false>(sg.start[0], sg.len[0], posType, e);
    segment_range(P, sg, walk, active, 0, cellStart, lj, le);
    row_store<BS, ROWCAP, false>(sh, 0, e);
    __syncthreads();
    for (int r = 0; r < nseg; r++) {
        const int buf = r & 1;
        if (r + 1 < nseg) {  // in flight during the cull below
            tile_fetch<BS, ROWCAP, false>(sg.start[r + 1], sg.len[r + 1], posType, e);
            segment_range(P, sg, walk, active, r + 1, cellStart, nlj, nle);
        }
        const int tag = r << idx_bits<ROWCAP>();
        // One aligned group of four candidates; those with lo <= u < hi belong to the lane.
        auto cull_group = [&](int g, int lo, int hi, auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            float d2[kChunk];
            group_d2<true>(sh.x[buf][g], sh.y[buf][g], sh.z[buf][g], pi.x, pi.y, pi.z, d2);
            const int etag = tag | (g * kChunk);
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                q[cnt * BS] = (unsigned short)(etag + u);  // a miss is overwritten by the next store
                cnt += (d2[u] < sr2c && (!MASKED || (u >= lo && u < hi))) ? 1 : 0;
            }
        };
        const bool has = lj < le;
        int base = lj & ~(kChunk - 1);
        if (wave_any(has)) {  // first group of the range
            cull_group(has ? base / kChunk : Shared::kPad, lj - base, le - base, std::true_type());
            if (wave_any(cnt > Q - kChunk)) flush(false);
        }
        base = has ? base + kChunk : le;
        while (wave_any(base + kChunk <= le)) {  // whole groups
            const bool go = base + kChunk <= le;
            cull_group(go ? base / kChunk : Shared::kPad, 0, kChunk, std::false_type());
            base += go ? kChunk : 0;
            if (wave_any(cnt > Q - kChunk)) flush(false);
        }
        if (wave_any(base < le)) {  // last group of the range
            const int rem = le - base;
            cull_group(rem > 0 ? base / kChunk : Shared::kPad, 0, rem, std::true_type());
            if (wave_any(cnt > Q - kChunk)) flush(false);
        }
        if (r + 1 < nseg) {
            row_store<BS, ROWCAP, false>(sh, buf ^ 1, e);
            lj = nlj;
            le = nle;
            __syncthreads();
        }
    }
    flush(true);
    if (active) counts[index] = (unsigned short)(total > K ? kListOverflow : total);
}

constexpr int kQueueL = 16;

int ensure_lists(Solver *s) {
    if (!s->use_lists || s->lists_valid || s->n == 0) return FSA_OK;
    if (s->precision != FSA_PRECISION_F32 || !s->sorted_valid || !s->pf.rows) {
        s->lists_valid = false;
        return FSA_OK;  // the passes use the tiles
    }
    if (!s->b.nbList || !s->b.nbCount) {  // sized for the solver's capacity, kept until fsa_destroy
        const size_t blocks = ((size_t)s->capacity + kTunedBS - 1) / kTunedBS + 1;
        void *lists = nullptr, *counts = nullptr;
        hipError_t e = hipMalloc(&lists, blocks * kListSlots * kTunedBS * sizeof(unsigned short));
        if (e == hipSuccess) e = hipMalloc(&counts, blocks * kTunedBS * sizeof(unsigned short));
        if (e != hipSuccess) {
            // no memory for the lists: the searching kernels give the same bits without them, so carry on with
            // those (fsa_last_error keeps the note; fsa_set_neighbour_lists(s, 1) tries again)
            if (lists) (void)hipFree(lists);
            (void)hipGetLastError();
            s->use_lists = false;
            s->lists_valid = false;
            (void)fail(FSA_ERR_NOMEM, "hipMalloc of the neighbour lists (%zu blocks of %d particles) failed: %s; "
                       "the passes search instead", blocks, kTunedBS, hipGetErrorString(e));
            return FSA_OK;
        }
        s->b.nbList = (unsigned short *)lists;
        s->b.nbCount = (unsigned short *)counts;
    }
    ScopedTimer t(s, "list_build");
    const int unit = kXcds << s->xcd_run_log;
    const int nb = ((s->n + kTunedBS - 1) / kTunedBS + unit - 1) / unit * unit;
    hipLaunchKernelGGL((list_build_kernel<kTunedBS, kRowCap, kQueueL, kListSlots>), dim3(nb), dim3(kTunedBS), 0,
                       s->stream, s->pf, s->n, (const float4 *)s->b.posType[s->cur], s->b.cellStart, s->b.sortedHash,
                       s->b.nbList, s->b.nbCount, s->xcd_run_log);
    if (s->pf.big)  // blocks with rows beyond kRowCap: large tiles, entries with 10 index bits
        hipLaunchKernelGGL((list_build_kernel<kTunedBS, kRowCapBig, kQueueL, kListSlots>), dim3((s->n + kTunedBS - 1) / kTunedBS),
                           dim3(kTunedBS), 0, s->stream, s->pf, s->n, (const float4 *)s->b.posType[s->cur],
                           s->b.cellStart, s->b.sortedHash, s->b.nbList, s->b.nbCount, s->xcd_run_log);
    FSA_HIP(hipGetLastError());
    s->lists_valid = true;
    return FSA_OK;
}

}  // namespace fsa
