// Shared machinery of the LDS-tiled neighbour passes (density, force, DFSPH): the row tile in LDS, the
// per-block row table, row fetch / store, the per-lane cell range, and the workgroup -> XCD mapping.
// See fsa_sph_tuned.hip for the description of the scheme.
#pragma once

#include "fsa_pair.h"

namespace fsa {

constexpr int kTunedBS = 256;
#ifndef FSA_ROWCAP
#define FSA_ROWCAP 360
#endif
// typical row: 34 cells x 8 particles = 272.  360 entries is what lets SIX workgroups of the WCSPH force replay share a
// CU's 160 KB of LDS (two tiles of 36-byte entries each; with 384 it was five: 1.43 -> 1.27 ms at 16 M particles)
constexpr int kRowCap = FSA_ROWCAP;
// A lattice that lines up with the cell faces puts three particle layers into some cells and one into their
// neighbours: rows of up to 34 x 2 x 9 = 612 particles.  Blocks with such rows run in a second instantiation
// of the tiled kernels with larger tiles (fewer blocks per CU, but there are few of them).
constexpr int kRowCapBig = 640;
template <int ROWCAP> constexpr int idx_bits() { return ROWCAP <= 512 ? 9 : 10; }  // entry = row << bits | index
template <int ROWCAP> constexpr int row_class() { return ROWCAP <= kRowCap ? 1 : 2; }
constexpr int kChunk = 4;  // candidates fetched per lane and loop trip
constexpr int kDrainBatch = 2;  // force drain: neighbours per batch (two batches in flight)
constexpr float kFar = 3.0e18f;
template <int BS, int ROWCAP> constexpr int row_regs() { return (ROWCAP + kChunk + BS - 1) / BS; }  // tile entries per thread

// Row tile in LDS, coordinates apart so that one 16-byte read yields four candidates.  The group
// after the last full one (index ROWCAP / 4) only ever holds far-away padding: lanes that have nothing
// left to do read it.
template <int BS, int ROWCAP, bool WITH_TYPE> struct RowShared {
    static constexpr int kGroups = (ROWCAP + kChunk) / kChunk;
    static constexpr int kPad = ROWCAP / kChunk;
    float4 x[2][kGroups], y[2][kGroups], z[2][kGroups];
    int4 t[WITH_TYPE ? 2 : 1][WITH_TYPE ? kGroups : 1];
    int rowStartG[9];  // first particle (global index) of each row
    int rowLen[9];
    int staged;
};

// The row table of every block of kTunedBS consecutive sorted particles, computed once per sort and read by
// every tiled pass until the next one: kRowTableStride ints per block = first particle of each of the 9 rows,
// their lengths, and the block's class: 1 = every row in the grid and at most kRowCap long, 2 = at most
// kRowCapBig long, 0 = neither (the block walks the cells particle by particle).  The ids of the class-2
// blocks are also kept as a list (DevParams::big: count, then ids) for the launch that handles them.  The entry's
// last int is 1 for a class-0 block that lies in the grid (its rows are too long as ranges of cells; the WCSPH
// record kernels then try segments of their own, sparse_segments below).
constexpr int kRowTableStride = 20;

// Recorded neighbour lists: at most kListSlots entries per particle, 16 bits each (row << 9 | index in the row,
// as in the hit queues), four consecutive slots of a lane packed into one 8-byte word:
//   word index = (block * kListSlots / 4 + slot / 4) * kTunedBS + lane
// and one 16-bit count per particle, kListOverflow when the list did not fit (or the block is not staged).
constexpr int kListSlots = 64;
constexpr int kListOverflow = 0xffff;

// Recorded hits of the WCSPH passes (fsa_sph_fast.hip), per block of kTunedBS particles as [slot][lane] (a
// wavefront reads and writes 512 contiguous bytes per slot), kHitSlots 8-byte slots per particle:
//   slot s < kMaxSegs       : the main word of segment s, written for every segment the block has: a bit mask of the
//                             57 candidates from group g0 of the segment's tile on (first candidate at bit 31 of .x,
//                             the 33rd at bit 31 of .y) and g0 in the low kHitMainShift bits of .y;
//   the kHitExtra after     : extra words in neighbour order, {32-candidate mask (first candidate highest),
//                             segment << kHitSegShift | first candidate's index in the tile}: what a lane's range
//                             holds beyond 57 candidates;
//   the kHitWall after that : TYPE_BOUNDARY neighbours, which count wherever they are in the 27 cells, in neighbour
//                             order: {32-candidate mask, index of the first candidate in the sorted arrays}.  They are
//                             not replayed from the tiles: particles that have any are listed (wallList) and a
//                             pass of their own, where every lane is such a particle, sums their wall forces.
// hitCount = number of extra words, kHitOverflow for a particle without a record, which walks the cells instead;
// wallCount = number of wall words.
constexpr int kMaxSegs = 24;  // a block's nine rows, the long ones as two tiles each (18), or the segments of a
                              // block whose particles lie in several stretches of cells (see Segments below)
constexpr int kHitExtra = 6;
constexpr int kHitWall = 14;
constexpr int kHitSlots = kMaxSegs + kHitExtra + kHitWall;
constexpr int kHitMainShift = 7;
constexpr unsigned kHitMainMask = (1u << kHitMainShift) - 1u;
constexpr int kHitOverflow = 255;
constexpr int kHitSegShift = 16;

__device__ __forceinline__ void block_row(const DevParams<float> &P, unsigned hlo, unsigned hhi, bool grid_ok, int r,
                                          const int *__restrict__ cellStart, int &s, int &len) {
    s = 0;
    len = -1;
    if (grid_ok) {
        const int C = P.num_cells;
        const int dz = r / 3 - 1, dy = r % 3 - 1;
        const long long off = (long long)dy * P.gres[0] * P.nz + (long long)dz * P.gres[0];
        long long a = (long long)hlo + off - 1, b = (long long)hhi + off + 2;
        a = a < 0 ? 0 : (a > C ? C : a);
        b = b < 0 ? 0 : (b > C ? C : b);
        s = cellStart[a];
        len = cellStart[b] - s;
    }
}

// Block-uniform set-up.  Returns the block's class (see above).  With a row table (`rows` != nullptr) the
// block copies its entry; otherwise it derives the rows itself from the cells of its first and last particle
// (hlo, hhi) and knows the classes 0 and 1 only.
template <int ROWCAP, typename Shared>
__device__ __forceinline__ int setup_rows(const DevParams<float> &P, int b0, int nb,
                                          const int *__restrict__ cellStart,
                                          const int *__restrict__ sortedHash, Shared &sh, unsigned hlo,
                                          unsigned hhi, const int *__restrict__ rows = nullptr) {
    const int tid = threadIdx.x;
    if (rows) {
        if (tid < 19) {
            const int v = rows[(size_t)(b0 / kTunedBS) * kRowTableStride + tid];
            if (tid < 9) sh.rowStartG[tid] = v;
            else if (tid < 18) sh.rowLen[tid - 9] = v;
            else sh.staged = v;
        }
        __syncthreads();
        return sh.staged;
    }
    const int C = P.num_cells;
    const bool grid_ok = hhi != (unsigned)FSA_GRID_UNDEF;
    if (tid < 9) {
        int s = 0, len = -1;
        if (grid_ok) {
            const int dz = tid / 3 - 1, dy = tid % 3 - 1;
            const long long off = (long long)dy * P.gres[0] * P.nz + (long long)dz * P.gres[0];
            long long a = (long long)hlo + off - 1, b = (long long)hhi + off + 2;
            a = a < 0 ? 0 : (a > C ? C : a);
            b = b < 0 ? 0 : (b > C ? C : b);
            s = cellStart[a];
            len = cellStart[b] - s;
        }
        sh.rowStartG[tid] = s;
        sh.rowLen[tid] = len;
    }
    __syncthreads();
    if (tid == 0) {
        bool ok = grid_ok;
        for (int r = 0; r < 9; r++)
            if (sh.rowLen[r] < 0 || sh.rowLen[r] > kRowCap) ok = false;
        sh.staged = ok ? 1 : 0;
    }
    __syncthreads();
    return sh.staged;
}

// Which block of kTunedBS particles this workgroup works on, or -1 for none: the regular instantiation maps the
// grid onto all blocks (XCD-aware, below) and leaves class-2 blocks alone; the large-tile instantiation is
// launched over the list of class-2 blocks.
__device__ __forceinline__ int xcd_block(int run_log);
template <int ROWCAP> __device__ __forceinline__ int tile_block(const DevParams<float> &P, int run_log) {
    if constexpr (ROWCAP <= kRowCap) {
        return xcd_block(run_log);
    } else {
        return (int)blockIdx.x < P.big[0] ? P.big[1 + blockIdx.x] : -1;
    }
}

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

// `len` consecutive particles from `s` on, for one tile: every thread fetches its entries (two, or three for the
// large tiles) into registers ...
// FLUID_ONLY: particles of any other type are moved far away, so that the cull needs no type test.
template <int BS, int ROWCAP, bool FLUID_ONLY>
__device__ __forceinline__ void tile_fetch(int s, int len, const float4 *__restrict__ posType,
                                           float4 (&e)[row_regs<BS, ROWCAP>()]) {
    const int tid = threadIdx.x;
    // entries past the end read some valid address and are then replaced by far-away padding
    const int last = len > 0 ? len - 1 : 0;
    const float4 far = make_float4(kFar, kFar, kFar, 0.0f);
#pragma unroll
    for (int k = 0; k < row_regs<BS, ROWCAP>(); k++) {
        const int i = tid + k * BS;
        e[k] = posType[s + min(i, last)];
        if (i >= len || (FLUID_ONLY && unpack_int(e[k].w) != FSA_TYPE_FLUID)) e[k] = far;
    }
}
// Row r of the block as one tile.
template <int BS, int ROWCAP, bool FLUID_ONLY, typename Shared>
__device__ __forceinline__ void row_fetch(const Shared &sh, int r, const float4 *__restrict__ posType,
                                          float4 (&e)[row_regs<BS, ROWCAP>()]) {
    tile_fetch<BS, ROWCAP, FLUID_ONLY>(sh.rowStartG[r], sh.rowLen[r], posType, e);
}
// ... and stores them later, after the previous row has been examined.
template <int BS, int ROWCAP, bool WITH_TYPE, typename Shared>
__device__ __forceinline__ void row_store(Shared &sh, int buf, const float4 (&e)[row_regs<BS, ROWCAP>()]) {
    const int tid = threadIdx.x;
    float *const x = reinterpret_cast<float *>(sh.x[buf]);
    float *const y = reinterpret_cast<float *>(sh.y[buf]);
    float *const z = reinterpret_cast<float *>(sh.z[buf]);
#pragma unroll
    for (int k = 0; k < row_regs<BS, ROWCAP>(); k++) {
        const int i = tid + k * BS;
        if (i < ROWCAP + kChunk) {
            x[i] = e[k].x;
            y[i] = e[k].y;
            z[i] = e[k].z;
            if (WITH_TYPE) reinterpret_cast<int *>(sh.t[buf])[i] = unpack_int(e[k].w);
        }
    }
}

// Segments (WCSPH density and force kernels): a row longer than a tile is walked as two tiles.  A block of class 1
// has its nine rows as nine segments; a block of class 2 (rows up to kRowCapBig, see above) up to 18.
struct Segments {
    int start[kMaxSegs];  // first particle (global index)
    int len[kMaxSegs];
    int row[kMaxSegs];    // which of the nine rows
    int count;
};
// Fills the segment table (call after setup_rows; visible after the next barrier).
template <int ROWCAP, typename Shared>
__device__ __forceinline__ void setup_segments(const Shared &sh, int cls, Segments &sg) {
    const int tid = threadIdx.x;
    if (cls == 1) {
        if (tid < 9) {
            sg.start[tid] = sh.rowStartG[tid];
            sg.len[tid] = sh.rowLen[tid];
            sg.row[tid] = tid;
        }
        if (tid == 9) sg.count = 9;
    } else if (tid == 0) {
        int k = 0;
        for (int r = 0; r < 9; r++) {
            const int len = sh.rowLen[r], st = sh.rowStartG[r];
            int o = 0;
            do {
                sg.start[k] = st + o;
                sg.len[k] = min(ROWCAP, len - o);
                sg.row[k] = r;
                k++;
                o += ROWCAP;
            } while (o < len && k < kMaxSegs);
        }
        sg.count = k;
    }
}

// Segments of a block whose rows, taken as one range of cells from its first particle's cell to its last one's, are
// too long for the tiles although the block needs little of them: 256 particles that lie in several stretches of
// cells with other particles' cells in between (the end of one wall row, a nearly empty row and the start of a fluid
// row; a splash).  The block's particles are cut into runs of adjacent cells (a new run where the cell index jumps
// by more than 2, so that the neighbour cells of two runs never overlap); in each of the nine neighbour rows every
// run needs the cells from one before its first to one after its last, and consecutive runs share a segment as long
// as the range from the first's start to the last's end fits a tile (the particles in between ride along), else a new
// segment starts; a single range longer than a tile is walked as several.  Segments come out in neighbour order
// (row, then ascending index).  `scratch`: 20 * kSparseRuns + 16 ints of LDS that nobody else uses meanwhile.
// Block-uniform result: false if there are more than kSparseRuns runs or more than kMaxSegs segments (the block then
// walks the cells).  Ends with a barrier: the segment table is visible on return.
constexpr int kSparseRuns = 16;
template <int ROWCAP>
__device__ __forceinline__ bool sparse_segments(const DevParams<float> &P, int b0, int nb,
                                                const int *__restrict__ sortedHash, const int *__restrict__ cellStart,
                                                int *scratch, Segments &sg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *const lo = scratch, *const hi = lo + kSparseRuns;  // cell of the first / last particle of each run
    int *const A = hi + kSparseRuns, *const B = A + 9 * kSparseRuns;  // [run][row]: the range a run needs of a row
    int *const misc = B + 9 * kSparseRuns;  // [0..8] segments per row, then their prefix; [9] number of segments, 0 = give up; [10..13] heads per wavefront
    const bool in = tid < nb;
    const int h = in ? sortedHash[b0 + tid] : 0;
    const bool head = in && (tid == 0 || h - sortedHash[b0 + tid - 1] > 2);
    const bool tail = in && (tid == nb - 1 || sortedHash[b0 + tid + 1] - h > 2);
    const unsigned long long hb = __ballot(head);
    if (lane == 0) misc[10 + wave] = __popcll(hb);
    __syncthreads();
    static_assert(kTunedBS % 64 == 0 && kTunedBS / 64 <= 4 && 9 * kSparseRuns <= kTunedBS,
                  "sparse_segments is written for the workgroups of the tiled kernels (kTunedBS threads)");
    int before = 0, nruns = 0;
    for (int w = 0; w < kTunedBS / 64; w++) {
        before += w < wave ? misc[10 + w] : 0;
        nruns += misc[10 + w];
    }
    if (nruns > kSparseRuns) {
        __syncthreads();  // (everyone has read misc before it may be reused)
        return false;
    }
    const int upto = before + __popcll(hb & ((2ull << lane) - 1ull));  // heads up to and including this thread
    if (head) lo[upto - 1] = h;
    if (tail) hi[upto - 1] = h;
    __syncthreads();
    if (tid < 9 * nruns) {
        const int k = tid / 9, r = tid % 9, C = P.num_cells;
        const int dz = r / 3 - 1, dy = r % 3 - 1;
        const long long off = (long long)dy * P.gres[0] * P.nz + (long long)dz * P.gres[0];
        long long a = (long long)lo[k] + off - 1, b = (long long)hi[k] + off + 2;
        a = a < 0 ? 0 : (a > C ? C : a);
        b = b < 0 ? 0 : (b > C ? C : b);
        A[tid] = cellStart[a];
        B[tid] = cellStart[b];
    }
    __syncthreads();
    // the same walk twice: first the segments of each row are counted, then written behind those of the rows before
    auto walk = [&](int r, auto &&emit) {
        auto tiles = [&](int a, int b) {
            for (int o = a; o < b; o += ROWCAP) emit(o, min(ROWCAP, b - o));
        };
        int ca = A[r], cb = B[r];
        for (int k = 1; k < nruns; k++) {
            const int a = A[9 * k + r], b = B[9 * k + r];
            if (b - ca <= ROWCAP) {
                cb = b;
            } else {
                tiles(ca, cb);
                ca = a;
                cb = b;
            }
        }
        tiles(ca, cb);
    };
    if (tid < 9) {
        int c = 0;
        walk(tid, [&](int, int) { c++; });
        misc[tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int total = 0;
        for (int r = 0; r < 9; r++) {
            const int c = misc[r];
            misc[r] = total;
            total += c;
        }
        misc[9] = (total >= 1 && total <= kMaxSegs) ? total : 0;
    }
    __syncthreads();
    const int total = misc[9];
    if (total && tid < 9) {
        int at = misc[tid];
        walk(tid, [&](int s, int len) {
            sg.start[at] = s;
            sg.len[at] = len;
            sg.row[at] = tid;
            at++;
        });
    }
    if (total && tid == 9) sg.count = total;
    __syncthreads();
    return total != 0;
}

// The nine rows of a staged block as nine segments (visible after the next barrier).
template <typename Shared> __device__ __forceinline__ void rows_as_segments(const Shared &sh, Segments &sg) {
    const int tid = threadIdx.x;
    if (tid < 9) {
        sg.start[tid] = sh.rowStartG[tid];
        sg.len[tid] = sh.rowLen[tid];
        sg.row[tid] = tid;
    }
    if (tid == 9) sg.count = 9;
}

// The lane's index range in segment k, relative to the segment's first particle: the cells of that row next to the
// lane's own, as far as the segment holds them (two loads of the cell table).
__device__ __forceinline__ void segment_range(const DevParams<float> &P, const Segments &sg, const RowWalk<float> &walk,
                                              bool active, int k, const int *__restrict__ cellStart, int &lj, int &le) {
    lj = 0;
    le = 0;
    if (active) {
        const int row = sg.row[k];
        const int c = walk.row_cell(P, row % 3 - 1, row / 3 - 1);
        if (c >= 0) {
            const int s0 = sg.start[k], len = sg.len[k];
            lj = min(max(cellStart[c] - s0, 0), len);
            le = min(max(cellStart[c + (walk.xhi - walk.xlo) + 1] - s0, 0), len);
        }
    }
}

// The lane's index range in row r, relative to the row's first particle (two loads of the cell table).
template <typename Shared>
__device__ __forceinline__ void lane_range(const DevParams<float> &P, const Shared &sh, const RowWalk<float> &walk,
                                           bool active, int r, const int *__restrict__ cellStart, int &lj, int &le) {
    lj = 0;
    le = 0;
    if (active) {
        const int c = walk.row_cell(P, r % 3 - 1, r / 3 - 1);
        if (c >= 0) {
            const int s = sh.rowStartG[r];
            lj = cellStart[c] - s;
            le = cellStart[c + (walk.xhi - walk.xlo) + 1] - s;
        }
    }
}

// Squared distances of the four candidates of group g to the point (px, py, pz).
template <bool FAST>
__device__ __forceinline__ void group_d2(const float4 &X, const float4 &Y, const float4 &Z, float px, float py,
                                         float pz, float (&d2)[kChunk]) {
    const float xs[kChunk] = {X.x, X.y, X.z, X.w}, ys[kChunk] = {Y.x, Y.y, Y.z, Y.w}, zs[kChunk] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
    for (int u = 0; u < kChunk; u++) {
        const float xx = px - xs[u], xy = py - ys[u], xz = pz - zs[u];
        d2[u] = FAST ? __builtin_fmaf(xz, xz, __builtin_fmaf(xy, xy, xx * xx)) : xx * xx + xy * xy + xz * xz;
    }
}

// Workgroups are handed to the 8 XCDs round-robin (workgroup i runs on XCD i % 8) and every XCD has
// its own L2.  Blocks of consecutive sorted particles share most of their neighbour rows, so runs of
// 64 consecutive blocks go to one XCD instead of every eighth block (measured: 5 % on the force pass;
// one contiguous eighth of the range per XCD is slower, the eighths differ too much in cost).
constexpr int kXcds = 8;
__device__ __forceinline__ int xcd_block_of(int b, int run_log) {  // b: the workgroup's index in the grid
    const int x = b % kXcds, k = b / kXcds;
    return ((k >> run_log) * kXcds + x) * (1 << run_log) + (k & ((1 << run_log) - 1));
}
__device__ __forceinline__ int xcd_block(int run_log) { return xcd_block_of((int)blockIdx.x, run_log); }

}  // namespace fsa
