// Uniform-grid neighbour search for gfx950: cell hash, per-cell counts, wavefront-scan
// prefix sum, stable counting sort, SoA reorder.
//
// Replaces calcHashD / thrust::sort_by_key / reorderDataAndFindCellStartD
// (solver/sph_kernel_impl.cuh:83-112, 116-200; solver/sph_solver.cu:30-77 under
// /root/reference/particle_solver).  The reference sorts (hash, index) pairs with a stable radix
// sort, so its order is: ascending cell, ties in previous array order, GRID_UNDEF last.  The key
// range here is the cell count, so one counting sort does it:
//   1. hash_count   : hash[i]; arrival slot inside the cell from an integer atomic (order inside a
//                     cell is arbitrary at this point, the counts are exact); out-of-grid particles
//                     get a stable rank inside their 256-particle block from wave ballots.
//   2. scan         : exclusive prefix sum of the counts (and of the per-block out-of-grid counts).
//   3. scatter      : tmpIndex[cellStart[h] + slot] = i.
//   4. rank         : inside each cell, rank = number of entries with a smaller source index, which
//                     turns the arbitrary arrival order into the stable one, deterministically.
//   5. reorder      : gather the hot arrays through the permutation into the other buffer copy.
#include <mutex>
#include <utility>

#include "fsa_advect.h"
#include "fsa_hash.h"
#include "fsa_solver.h"
#include "fsa_tile.h"

namespace fsa {

// ------------------------------------------------------------------------------------------------
// 1. hash + count
// ------------------------------------------------------------------------------------------------
// PosT: Vec4<T> ({x, y, z, type bits}, the solver's layout) or Vec3<T> (the reference's cfloat3 layout)
template <typename T, typename PosT>
__global__ __launch_bounds__(kBlock) void hash_count_kernel(DevParams<T> P, const PosT *__restrict__ posType,
                                                            int n, int *__restrict__ hash, int *__restrict__ slot,
                                                            int *__restrict__ cellCount,
                                                            int *__restrict__ undefBlockCount) {
    __shared__ int waveCount[kBlock / 64];
    n = live_count(P, n);
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < n;
    T x = 0, y = 0, z = 0;
    if (valid) {
        const PosT p = posType[i];
        x = p.x;
        y = p.y;
        z = p.z;
    }
    hash_and_count(P, valid, i, x, y, z, hash, slot, cellCount, undefBlockCount, waveCount);
}

// AdvectKernel and the calcHash of the next step in one pass: the new position goes from the integration straight
// into the hash (cellCount must be all zero on entry).
template <typename T>
__global__ __launch_bounds__(kBlock) void advect_hash_kernel(DevParams<T> P, int n, Vec4<T> *__restrict__ posType,
                                                             Vec4<T> *__restrict__ velMass,
                                                             const Vec3<T> *__restrict__ force,
                                                             Vec3<T> *__restrict__ v_star, int *__restrict__ hash,
                                                             int *__restrict__ slot, int *__restrict__ cellCount,
                                                             int *__restrict__ undefBlockCount) {
    __shared__ int waveCount[kBlock / 64];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < n;
    Vec4<T> p;
    p.x = p.y = p.z = p.w = 0;
    if (valid) {
        p = posType[i];
        if (advect_moves(P, unpack_int(p.w))) {
            Vec4<T> v = velMass[i];
            Vec3<T> vs;
            advect_particle(P, p, v, force[i], vs);
            v_star[i] = vs;
            posType[i] = p;
            velMass[i] = v;
        }
    }
    hash_and_count(P, valid, i, p.x, p.y, p.z, hash, slot, cellCount, undefBlockCount, waveCount);
}

// ------------------------------------------------------------------------------------------------
// 2. exclusive prefix sum of an int array (three launches: tile sums, scan of the sums, apply)
// ------------------------------------------------------------------------------------------------
constexpr int kScanPerThread = 8;
constexpr int kScanTile = kBlock * kScanPerThread;

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over the block; returns the block total in `total`
__device__ __forceinline__ int block_exclusive_scan(int v, int &total, int *smem /* >= blockDim/64 + 1 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = wave_inclusive_scan(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nw; w++) {
        int c = smem[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kBlock) void scan_tile_sums_kernel(const int *__restrict__ in, int n,
                                                                int *__restrict__ tileSums) {
    __shared__ int smem[kBlock / 64 + 1];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanPerThread;
    int sum = 0;
    if (base + kScanPerThread <= n) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base);
        const int4 b = *reinterpret_cast<const int4 *>(in + base + 4);
        sum = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (int k = 0; k < kScanPerThread; k++)
            if (base + k < n) sum += in[base + k];
    }
    int total;
    block_exclusive_scan(sum, total, smem);
    if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}

// one block: in-place exclusive scan of tileSums[0..m), total to tileSums[m]
__global__ __launch_bounds__(1024) void scan_sums_kernel(int *__restrict__ tileSums, int m) {
    __shared__ int smem[1024 / 64 + 1];
    int carry = 0;
    for (int base = 0; base < m; base += 1024) {
        const int idx = base + threadIdx.x;
        const int v = idx < m ? tileSums[idx] : 0;
        int total;
        const int ex = block_exclusive_scan(v, total, smem);
        if (idx < m) tileSums[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) tileSums[m] = carry;
}

// out[i] = sum of in[0..i); the thread that owns position n also writes out[n] = total
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const int *__restrict__ in, int n,
                                                            const int *__restrict__ tileSums,
                                                            int *__restrict__ out) {
    __shared__ int smem[kBlock / 64 + 1];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    const bool full = base + kScanPerThread <= n;
    if (full) {
        const int4 a = *reinterpret_cast<const int4 *>(in + base);
        const int4 b = *reinterpret_cast<const int4 *>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; k++) v[k] = (base + k < n) ? in[base + k] : 0;
    }
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) sum += v[k];
    int total;
    int run = tileSums[blockIdx.x] + block_exclusive_scan(sum, total, smem);
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) {
        // position n (one past the end) receives the grand total
        if (base + k <= n) out[base + k] = run;
        run += v[k];
    }
}

static int exclusive_scan(Solver *s, const int *in, int n, int *out, int *tmp, const char *tag) {
    const int tiles = (n + 1 + kScanTile - 1) / kScanTile;  // +1: out has n + 1 entries
    ScopedTimer t(s, tag);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tiles), dim3(kBlock), 0, s->stream, in, n, tmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s->stream, tmp, tiles);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(kBlock), 0, s->stream, in, n, tmp, out);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// ------------------------------------------------------------------------------------------------
// 3. scatter source indices into their cells (arbitrary order inside a cell)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void scatter_kernel(int n, int num_cells, const int *__restrict__ hash,
                                                         const int *__restrict__ slot,
                                                         const int *__restrict__ cellStart,
                                                         const int *__restrict__ undefBlockStart,
                                                         int *__restrict__ tmpIndex, int *__restrict__ tmpHash) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned h = (unsigned)hash[i];
    int p;
    if (h != FSA_GRID_UNDEF)
        p = cellStart[h] + slot[i];
    else
        p = cellStart[num_cells] + undefBlockStart[blockIdx.x] + slot[i];
    tmpIndex[p] = i;
    tmpHash[p] = (int)h;
}

// ------------------------------------------------------------------------------------------------
// 4. stable rank inside each cell
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void rank_kernel(int n, const int *__restrict__ tmpIndex,
                                                      const int *__restrict__ tmpHash,
                                                      const int *__restrict__ cellStart,
                                                      int *__restrict__ sortedIndex, int *__restrict__ sortedHash) {
    const int sIdx = blockIdx.x * kBlock + threadIdx.x;
    if (sIdx >= n) return;
    const int i = tmpIndex[sIdx];
    const unsigned h = (unsigned)tmpHash[sIdx];
    int p = sIdx;  // out-of-grid particles are already in stable order
    if (h != FSA_GRID_UNDEF) {
        const int a = cellStart[h], b = cellStart[h + 1];
        int r = 0;
        for (int t = a; t < b; t++) r += (tmpIndex[t] < i) ? 1 : 0;
        p = a + r;
    }
    sortedIndex[p] = i;
    sortedHash[p] = (int)h;
}

// ------------------------------------------------------------------------------------------------
// 5. reorder: gather through the permutation
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void reorder_kernel(int n, const int *__restrict__ sortedIndex,
                                                         const Vec4<T> *__restrict__ posTypeIn,
                                                         const Vec4<T> *__restrict__ velMassIn,
                                                         const T *__restrict__ restDensityIn,
                                                         const int *__restrict__ uniqueIdIn,
                                                         const Vec3<T> *__restrict__ vStarIn,
                                                         Vec4<T> *__restrict__ posTypeOut,
                                                         Vec4<T> *__restrict__ velMassOut,
                                                         T *__restrict__ restDensityOut,
                                                         int *__restrict__ uniqueIdOut,
                                                         Vec3<T> *__restrict__ vStarOut) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int i = sortedIndex[p];
    posTypeOut[p] = posTypeIn[i];
    velMassOut[p] = velMassIn[i];
    restDensityOut[p] = restDensityIn[i];
    uniqueIdOut[p] = uniqueIdIn[i];
    vStarOut[p] = vStarIn[i];  // sortedV_star, solver/sph_kernel_impl.cuh:168
}

// gridCellStart / gridCellEnd in the reference's format (solver/sph_kernel_impl.cuh:151-163 after the
// 0xff memset of solver/sph_solver.cu:65): start = 0xffffffff for an empty cell, end untouched there.
__global__ __launch_bounds__(kBlock) void ref_cells_kernel(int num_cells, const int *__restrict__ cellStart,
                                                           int *__restrict__ gridCellStart,
                                                           int *__restrict__ gridCellEnd) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= num_cells) return;
    const int a = cellStart[c], b = cellStart[c + 1];
    if (b > a) {
        gridCellStart[c] = a;
        gridCellEnd[c] = b;
    } else {
        gridCellStart[c] = (int)0xffffffffu;
    }
}

// indexTable[uniqueId[p]] = p, solver/sph_kernel_impl.cuh:179
__global__ __launch_bounds__(kBlock) void index_table_kernel(int n, const int *__restrict__ uniqueId,
                                                             int *__restrict__ indexTable) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    indexTable[uniqueId[p]] = p;
}

// row table of the tiled neighbour passes (fsa_tile.h), one entry per block of kTunedBS sorted particles
// 16 lanes per block entry: lanes 0..8 look up one row each, the verdict is folded over the 16 lanes
__global__ __launch_bounds__(256) void block_rows_kernel(DevParams<float> P, int n, const int *__restrict__ cellStart,
                                                         const int *__restrict__ sortedHash,
                                                         int *__restrict__ table, int *__restrict__ big) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int blk = gid >> 4, r = gid & 15;
    const int nblk = (n + kTunedBS - 1) / kTunedBS;
    const bool live = blk < nblk;
    int bad = 0, longest = 0, in_grid = 0;
    if (live) {
        const int b0 = blk * kTunedBS, nb = min(kTunedBS, n - b0);
        const unsigned hlo = (unsigned)sortedHash[b0], hhi = (unsigned)sortedHash[b0 + nb - 1];
        const bool grid_ok = hhi != (unsigned)FSA_GRID_UNDEF;
        bad = grid_ok ? 0 : 1;
        in_grid = grid_ok ? 1 : 0;
        if (r < 9) {
            int s, len;
            block_row(P, hlo, hhi, grid_ok, r, cellStart, s, len);
            int *row = table + (size_t)blk * kRowTableStride;
            row[r] = s;
            row[9 + r] = len;
            if (len < 0 || len > kRowCapBig) bad = 1;
            longest = len;
        }
    }
    for (int d = 8; d >= 1; d >>= 1) {
        bad |= __shfl_xor(bad, d, 16);
        longest = max(longest, __shfl_xor(longest, d, 16));
    }
    if (live && r == 0) {
        int *row = table + (size_t)blk * kRowTableStride;
        const int cls = bad ? 0 : (longest <= kRowCap ? 1 : 2);
        row[18] = cls;
        row[19] = (cls == 0 && in_grid) ? 1 : 0;  // rows too long as one range of cells: see sparse_segments (fsa_tile.h)
        if (cls == 2) big[1 + atomicAdd(&big[0], 1)] = blk;  // any order: the blocks are independent
    }
}

// ------------------------------------------------------------------------------------------------
// The sort of a whole step in four launches (single solvers; z-slab solvers rank by order key and keep the
// launches above): tile sums and apply of BOTH prefix sums (cell counts, out-of-grid block counts), scatter,
// rank + reorder + row table.
// ------------------------------------------------------------------------------------------------
struct Scan2 {
    const int *in[2];
    int *out[2];
    int n[2];
    int tiles[2];
    const int *count;  // when set: n[1] = number of kBlock-blocks of *count particles (see DevParams::count)
};
__device__ __forceinline__ int scan2_n(const Scan2 &a, int k) {
    return (k == 1 && a.count) ? (*a.count + kBlock - 1) / kBlock : a.n[k];
}

__global__ __launch_bounds__(kBlock) void scan2_tile_sums_kernel(Scan2 a, int *__restrict__ tileSums,
                                                                 int *__restrict__ big) {
    __shared__ int smem[kBlock / 64 + 1];
    if (big && blockIdx.x == 0 && threadIdx.x == 0) big[0] = 0;  // the list of large-tile blocks starts empty
    const int k = (int)blockIdx.x < a.tiles[0] ? 0 : 1;
    const int tile = (int)blockIdx.x - (k ? a.tiles[0] : 0);
    const int *in = a.in[k];
    const int n = scan2_n(a, k);
    const long long base = (long long)tile * kScanTile + (long long)threadIdx.x * kScanPerThread;
    int sum = 0;
    if (base + kScanPerThread <= n) {
        const int4 u = *reinterpret_cast<const int4 *>(in + base);
        const int4 v = *reinterpret_cast<const int4 *>(in + base + 4);
        sum = u.x + u.y + u.z + u.w + v.x + v.y + v.z + v.w;
    } else {
        for (int j = 0; j < kScanPerThread; j++)
            if (base + j < n) sum += in[base + j];
    }
    int total;
    block_exclusive_scan(sum, total, smem);
    if (threadIdx.x == 0) tileSums[blockIdx.x] = total;
}

// Every block adds up the sums of the tiles before it in its sequence (a few thousand ints out of L2) instead of a
// single-block launch in between.  Sequence 0's input is set back to zero once read: the cell counts are then
// ready for the next hash pass.
__global__ __launch_bounds__(kBlock) void scan2_apply_kernel(Scan2 a, const int *__restrict__ tileSums,
                                                             int *__restrict__ zero_in0) {
    __shared__ int smem[kBlock / 64 + 1];
    const int k = (int)blockIdx.x < a.tiles[0] ? 0 : 1;
    const int first = k ? a.tiles[0] : 0;
    const int tile = (int)blockIdx.x - first;
    const int *in = a.in[k];
    int *out = a.out[k];
    const int n = scan2_n(a, k);
    int before = 0;
    for (int j = threadIdx.x; j < tile; j += kBlock) before += tileSums[first + j];
    int carry;
    block_exclusive_scan(before, carry, smem);
    const long long base = (long long)tile * kScanTile + (long long)threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    const bool full = base + kScanPerThread <= n;
    if (full) {
        const int4 u = *reinterpret_cast<const int4 *>(in + base);
        const int4 w = *reinterpret_cast<const int4 *>(in + base + 4);
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
        v[4] = w.x; v[5] = w.y; v[6] = w.z; v[7] = w.w;
    } else {
#pragma unroll
        for (int j = 0; j < kScanPerThread; j++) v[j] = (base + j < n) ? in[base + j] : 0;
    }
    if (k == 0 && zero_in0) {
        if (full) {
            *reinterpret_cast<int4 *>(zero_in0 + base) = make_int4(0, 0, 0, 0);
            *reinterpret_cast<int4 *>(zero_in0 + base + 4) = make_int4(0, 0, 0, 0);
        } else {
            for (int j = 0; j < kScanPerThread; j++)
                if (base + j < n) zero_in0[base + j] = 0;
        }
    }
    int sum = 0;
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) sum += v[j];
    int total;
    int run = carry + block_exclusive_scan(sum, total, smem);
#pragma unroll
    for (int j = 0; j < kScanPerThread; j++) {
        if (base + j <= n) out[base + j] = run;  // position n (one past the end) receives the grand total
        run += v[j];
    }
}

// scatter_kernel, and the first arrival of every cell also notes which blocks of kTunedBS sorted positions begin
// and end inside its cell: the cell of a block's first and of its last particle (blockLo / blockHi), which the row
// table needs, known here before the particles are in place.
__global__ __launch_bounds__(kBlock) void scatter_bounds_kernel(int n, const int *__restrict__ count, int num_cells,
                                                                const int *__restrict__ hash,
                                                                const int *__restrict__ slot,
                                                                const int *__restrict__ cellStart,
                                                                const int *__restrict__ undefBlockStart,
                                                                int *__restrict__ tmpIndex,
                                                                int *__restrict__ tmpHash, int *__restrict__ blockLo,
                                                                int *__restrict__ blockHi) {
    if (count) n = *count;  // see DevParams::count
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned h = (unsigned)hash[i];
    const int sl = slot[i];      // (both loads in flight before the test)
    if (h == kHashDrop) return;  // the particle has left this solver (fsa_hash.h)
    int p;
    if (h != FSA_GRID_UNDEF) {
        const int a = cellStart[h];
        p = a + sl;
        if (sl == 0) {
            const int b = cellStart[h + 1], last = min(n, cellStart[num_cells]) - 1;
            for (int m = (a + kTunedBS - 1) / kTunedBS * kTunedBS; m < b; m += kTunedBS) blockLo[m / kTunedBS] = (int)h;
            for (int e = a | (kTunedBS - 1); e < b; e += kTunedBS) blockHi[e / kTunedBS] = (int)h;
            if (last >= a && last < b) blockHi[last / kTunedBS] = (int)h;
        }
    } else {
        p = cellStart[num_cells] + undefBlockStart[blockIdx.x] + sl;
    }
    tmpIndex[p] = i;
    tmpHash[p] = (int)h;
}

// rank_kernel + reorder_kernel + block_rows_kernel in one pass: the thread that finds the stable place p of source
// particle i moves the particle there itself (reads as in the gather form, writes permuted inside one cell only),
// and the first 16 threads of each block write the row-table entry of the block of sorted positions they cover.
// VSTAR_ALL = false: v_star only travels for particles that are not TYPE_FLUID (the integration rewrites it for
// every fluid particle before anybody can read it).
// BY_KEY (z-slab solvers, fsa_slab.hip): ties inside a cell are broken by the 64-bit order key each particle
// carries instead of by the source index, and the key of the new order is written on the way.
// slots / undefBlockStart / count_out: see the start of the kernel (null for everything but z-slab groups).
template <typename T, bool VSTAR_ALL, bool BY_KEY>
__global__ __launch_bounds__(kBlock) void rank_reorder_kernel(
    DevParams<float> Pf, int n, const int *__restrict__ tmpIndex, const int *__restrict__ tmpHash,
    const int *__restrict__ cellStart, int *__restrict__ sortedIndex, int *__restrict__ sortedHash,
    const Vec4<T> *__restrict__ posTypeIn, const Vec4<T> *__restrict__ velMassIn, const T *__restrict__ restDensityIn,
    const int *__restrict__ uniqueIdIn, const Vec3<T> *__restrict__ vStarIn, Vec4<T> *__restrict__ posTypeOut,
    Vec4<T> *__restrict__ velMassOut, T *__restrict__ restDensityOut, int *__restrict__ uniqueIdOut,
    Vec3<T> *__restrict__ vStarOut, const int *__restrict__ blockLo, const int *__restrict__ blockHi,
    int *__restrict__ table, int *__restrict__ big, const long long *__restrict__ keyIn,
    long long *__restrict__ keyOut, const int *__restrict__ slots, const int *__restrict__ undefBlockStart,
    int *__restrict__ count_out) {
    static_assert(kTunedBS == kBlock, "one row-table entry per block of this kernel");
    if (slots) {
        // z-slab groups: the source arrays have `*slots` slots, some of them dropped; what the sort keeps is what
        // the two prefix sums counted, and that becomes the particle count of the kernels that follow
        n = cellStart[Pf.num_cells] + undefBlockStart[(*slots + kBlock - 1) / kBlock];
        if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = n;
    } else {
        n = live_count(Pf, n);
    }
    if ((int)blockIdx.x * kBlock >= n) return;
    const int sIdx = blockIdx.x * kBlock + threadIdx.x;
    if (table && threadIdx.x < 64) {  // row table (fp32 solvers), as block_rows_kernel
        const int blk = blockIdx.x, r = threadIdx.x & 15;
        const int b0 = blk * kTunedBS, nb = min(kTunedBS, n - b0);
        const int in_grid = cellStart[Pf.num_cells];
        const unsigned hlo = b0 < in_grid ? (unsigned)blockLo[blk] : (unsigned)FSA_GRID_UNDEF;
        const unsigned hhi = b0 + nb - 1 < in_grid ? (unsigned)blockHi[blk] : (unsigned)FSA_GRID_UNDEF;
        const bool grid_ok = hhi != (unsigned)FSA_GRID_UNDEF;
        int bad = grid_ok ? 0 : 1, longest = 0;
        int *row = table + (size_t)blk * kRowTableStride;
        if (r < 9) {
            int st, len;
            block_row(Pf, hlo, hhi, grid_ok, r, cellStart, st, len);
            if (threadIdx.x < 16) {
                row[r] = st;
                row[9 + r] = len;
            }
            if (len < 0 || len > kRowCapBig) bad = 1;
            longest = len;
        }
        for (int d = 8; d >= 1; d >>= 1) {
            bad |= __shfl_xor(bad, d, 16);
            longest = max(longest, __shfl_xor(longest, d, 16));
        }
        if (threadIdx.x == 0) {
            const int cls = bad ? 0 : (longest <= kRowCap ? 1 : 2);
            row[18] = cls;
            row[19] = (cls == 0 && grid_ok) ? 1 : 0;  // rows too long as one range of cells: see sparse_segments
            if (cls == 2) big[1 + atomicAdd(&big[0], 1)] = blk;  // any order: the blocks are independent
        }
    }
    // BY_KEY: the keys of the block's 256 entries go through LDS -- a cell's entries are consecutive, so nearly all
    // the keys a thread compares its own with are there (a gather per comparison otherwise)
    __shared__ unsigned long long keyS[BY_KEY ? kBlock : 1];
    __shared__ int idxS[BY_KEY ? kBlock : 1];
    const int i = sIdx < n ? tmpIndex[sIdx] : 0;
    unsigned long long mine = 0;
    if (BY_KEY) {
        if (sIdx < n) mine = (unsigned long long)keyIn[i];
        keyS[threadIdx.x] = mine;
        idxS[threadIdx.x] = i;
        __syncthreads();
    }
    if (sIdx >= n) return;
    const unsigned h = (unsigned)tmpHash[sIdx];
    int p = sIdx;  // out-of-grid particles are already in stable order
    int first = cellStart[Pf.num_cells];  // of the particle's cell (out of grid: of the tail)
    if (h != FSA_GRID_UNDEF) {
        const int a = cellStart[h], b = cellStart[h + 1];
        int r = 0;
        if (BY_KEY) {
            const int b0 = blockIdx.x * kBlock;
            for (int t = a; t < b; t++) {
                int j;
                unsigned long long other;
                if (t >= b0 && t < b0 + kBlock) {
                    j = idxS[t - b0];
                    other = keyS[t - b0];
                } else {
                    j = tmpIndex[t];
                    other = (unsigned long long)keyIn[j];
                }
                r += (other < mine || (other == mine && j < i)) ? 1 : 0;
            }
        } else {
            for (int t = a; t < b; t++) r += (tmpIndex[t] < i) ? 1 : 0;
        }
        p = a + r;
        first = a;
    }
    sortedIndex[p] = i;
    sortedHash[p] = (int)h;
    if (BY_KEY) {  // (global cell id << 32) | rank inside the cell; out of grid: (GRID_UNDEF << 32) | rank in the tail
        unsigned long long hi = FSA_GRID_UNDEF;
        if (h != FSA_GRID_UNDEF) {
            const unsigned gx = (unsigned)Pf.gres[0];
            const unsigned x = h % gx, t = h / gx;
            const unsigned lz = t % (unsigned)Pf.nz, y = t / (unsigned)Pf.nz;
            hi = ((unsigned long long)y * (unsigned)Pf.gres[2] + (lz + (unsigned)Pf.z0)) * gx + x;
        }
        keyOut[p] = (long long)((hi << 32) | (unsigned)(p - first));
    }
    const Vec4<T> pt = posTypeIn[i];
    posTypeOut[p] = pt;
    velMassOut[p] = velMassIn[i];
    restDensityOut[p] = restDensityIn[i];
    uniqueIdOut[p] = uniqueIdIn[i];
    if (VSTAR_ALL || unpack_int(pt.w) != FSA_TYPE_FLUID) vStarOut[p] = vStarIn[i];  // sortedV_star, :168
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

int launch_hash_count(Solver *s) {
    const int n = s->grid_n(), nb = blocks_for(n);
    if (!s->counts_clean) {
        ScopedTimer t(s, "clear_counts");
        FSA_HIP(hipMemsetAsync(s->b.cellCount, 0, (size_t)s->num_cells * sizeof(int), s->stream));
    }
    s->counts_clean = n == 0;
    if (n == 0) return FSA_OK;
    ScopedTimer t(s, "hash_count");
    if (s->precision == FSA_PRECISION_F32)
        hipLaunchKernelGGL((hash_count_kernel<float, float4>), dim3(nb), dim3(kBlock), 0, s->stream, s->pf,
                           (const float4 *)s->b.posType[s->cur], n, s->b.hash, s->b.slot, s->b.cellCount,
                           s->b.undefBlockCount);
    else
        hipLaunchKernelGGL((hash_count_kernel<double, double4>), dim3(nb), dim3(kBlock), 0, s->stream, s->pd,
                           (const double4 *)s->b.posType[s->cur], n, s->b.hash, s->b.slot, s->b.cellCount,
                           s->b.undefBlockCount);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_scan_cells(Solver *s) {
    int rc = exclusive_scan(s, s->b.cellCount, s->num_cells, s->b.cellStart, s->b.scanTmp, "scan_cells");
    if (rc) return rc;
    if (s->n == 0) return FSA_OK;
    return exclusive_scan(s, s->b.undefBlockCount, blocks_for(s->n), s->b.undefBlockStart, s->b.scanTmp,
                          "scan_undef");
}

int launch_scatter(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "scatter");
    hipLaunchKernelGGL(scatter_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n, s->num_cells,
                       s->b.hash, s->b.slot, s->b.cellStart, s->b.undefBlockStart, s->b.tmpIndex, s->b.tmpHash);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_rank(Solver *s) {
    if (s->n == 0) return FSA_OK;
    if (s->slab) return launch_rank_key(s);
    ScopedTimer t(s, "rank");
    hipLaunchKernelGGL(rank_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n, s->b.tmpIndex,
                       s->b.tmpHash, s->b.cellStart, s->b.sortedIndex, s->b.sortedHash);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_reorder(Solver *s) {
    if (s->n == 0) return FSA_OK;
    const int in = s->cur, out = 1 - s->cur;
    {
    ScopedTimer t(s, "reorder");
    if (s->precision == FSA_PRECISION_F32)
        hipLaunchKernelGGL(reorder_kernel<float>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                           s->b.sortedIndex, (const float4 *)s->b.posType[in], (const float4 *)s->b.velMass[in],
                           (const float *)s->b.restDensity[in], s->b.uniqueId[in], (const Vec3<float> *)s->b.v_star,
                           (float4 *)s->b.posType[out], (float4 *)s->b.velMass[out], (float *)s->b.restDensity[out],
                           s->b.uniqueId[out], (Vec3<float> *)s->b.v_star_alt);
    else
        hipLaunchKernelGGL(reorder_kernel<double>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                           s->b.sortedIndex, (const double4 *)s->b.posType[in], (const double4 *)s->b.velMass[in],
                           (const double *)s->b.restDensity[in], s->b.uniqueId[in], (const Vec3<double> *)s->b.v_star,
                           (double4 *)s->b.posType[out], (double4 *)s->b.velMass[out], (double *)s->b.restDensity[out],
                           s->b.uniqueId[out], (Vec3<double> *)s->b.v_star_alt);
    FSA_HIP(hipGetLastError());
    }
    std::swap(s->b.v_star, s->b.v_star_alt);
    if (s->precision == FSA_PRECISION_F32) {
        ScopedTimer t(s, "block_rows");
        const int nblk = (s->n + kTunedBS - 1) / kTunedBS;
        FSA_HIP(hipMemsetAsync(s->b.bigBlocks, 0, sizeof(int), s->stream));
        hipLaunchKernelGGL(block_rows_kernel, dim3((nblk * 16 + 255) / 256), dim3(256), 0, s->stream, s->pf, s->n,
                           s->b.cellStart, s->b.sortedHash, s->b.blockRows, s->b.bigBlocks);
        FSA_HIP(hipGetLastError());
    }
    if (s->slab) {
        int rc = launch_slab_keys(s);
        if (rc) return rc;
    }
    if (s->iisph) return launch_iisph_carry(s, out);
    return FSA_OK;
}

// hash + counts of the current positions; `with_advect`: integrate first, in the same pass
int launch_hash_fused(Solver *s, bool with_advect) {
    const int n = s->n, nb = blocks_for(n);
    if (!s->counts_clean) {
        ScopedTimer t(s, "clear_counts");
        FSA_HIP(hipMemsetAsync(s->b.cellCount, 0, (size_t)s->num_cells * sizeof(int), s->stream));
        s->counts_clean = true;
    }
    if (n == 0) return FSA_OK;
    if (!with_advect) return launch_hash_count(s);
    ScopedTimer t(s, "advect_hash");
    s->lists_valid = false;  // positions move
    s->hits_valid = false;
    s->counts_clean = false;
    const int c = s->cur;
    if (s->precision == FSA_PRECISION_F32)
        hipLaunchKernelGGL(advect_hash_kernel<float>, dim3(nb), dim3(kBlock), 0, s->stream, s->pf, n,
                           (float4 *)s->b.posType[c], (float4 *)s->b.velMass[c], (const Vec3<float> *)s->b.force,
                           (Vec3<float> *)s->b.v_star, s->b.hash, s->b.slot, s->b.cellCount, s->b.undefBlockCount);
    else
        hipLaunchKernelGGL(advect_hash_kernel<double>, dim3(nb), dim3(kBlock), 0, s->stream, s->pd, n,
                           (double4 *)s->b.posType[c], (double4 *)s->b.velMass[c], (const Vec3<double> *)s->b.force,
                           (Vec3<double> *)s->b.v_star, s->b.hash, s->b.slot, s->b.cellCount, s->b.undefBlockCount);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// fsa_sort_particle + fsa_reorder_data_and_find_cell_start of a single (not z-slab) solver in four launches;
// hash / slot / cellCount must describe the current arrays.  Leaves cellCount zeroed.
int launch_sort_fused(Solver *s, bool vstar_all) {
    const int n = s->grid_n(), nb = blocks_for(n), C = s->num_cells;
    const int *slots = s->slab_slots;  // z-slab groups: the number of source slots, on the device
    Scan2 a;
    a.count = slots ? slots : s->pf.count;
    a.in[0] = s->b.cellCount;
    a.out[0] = s->b.cellStart;
    a.n[0] = C;
    a.tiles[0] = (C + 1 + kScanTile - 1) / kScanTile;
    a.in[1] = s->b.undefBlockCount;
    a.out[1] = s->b.undefBlockStart;
    a.n[1] = nb;
    a.tiles[1] = n ? (nb + 1 + kScanTile - 1) / kScanTile : 0;  // with a device-side count: sized for the bound
    {
        ScopedTimer t(s, "scan_sums");
        hipLaunchKernelGGL(scan2_tile_sums_kernel, dim3(a.tiles[0] + a.tiles[1]), dim3(kBlock), 0, s->stream, a,
                           s->b.scanTmp, s->b.bigBlocks);
    }
    {
        ScopedTimer t(s, "scan_apply");
        hipLaunchKernelGGL(scan2_apply_kernel, dim3(a.tiles[0] + a.tiles[1]), dim3(kBlock), 0, s->stream, a,
                           (const int *)s->b.scanTmp, s->b.cellCount);
    }
    FSA_HIP(hipGetLastError());
    s->counts_clean = true;
    if (n == 0) return FSA_OK;
    {
        ScopedTimer t(s, "scatter");
        hipLaunchKernelGGL(scatter_bounds_kernel, dim3(nb), dim3(kBlock), 0, s->stream, n, a.count, C, s->b.hash, s->b.slot,
                           s->b.cellStart, s->b.undefBlockStart, s->b.tmpIndex, s->b.tmpHash, s->b.blockLo,
                           s->b.blockHi);
    }
    const int in = s->cur, out = 1 - s->cur;
    {
        ScopedTimer t(s, "rank_reorder");
        const bool f32 = s->precision == FSA_PRECISION_F32;
#define FSA_RR(T, T4, ALL, KEY)                                                                                     \
    hipLaunchKernelGGL((rank_reorder_kernel<T, ALL, KEY>), dim3(nb), dim3(kBlock), 0, s->stream, s->pf, n,          \
                       s->b.tmpIndex, s->b.tmpHash, s->b.cellStart, s->b.sortedIndex, s->b.sortedHash,              \
                       (const T4 *)s->b.posType[in], (const T4 *)s->b.velMass[in], (const T *)s->b.restDensity[in], \
                       s->b.uniqueId[in], (const Vec3<T> *)s->b.v_star, (T4 *)s->b.posType[out],                    \
                       (T4 *)s->b.velMass[out], (T *)s->b.restDensity[out], s->b.uniqueId[out],                     \
                       (Vec3<T> *)s->b.v_star_alt, s->b.blockLo, s->b.blockHi, f32 ? s->b.blockRows : nullptr,      \
                       s->b.bigBlocks, s->b.orderKey[in], s->b.orderKey[out], slots, s->b.undefBlockStart,              \
                       slots ? s->b.slabVars + kVarN : nullptr)
        if (s->slab && vstar_all && s->slab_carry_vstar) {  // (the DFSPH step on slabs goes on from v_star after the sort)
            if (f32) FSA_RR(float, float4, true, true); else FSA_RR(double, double4, true, true);
        } else if (s->slab) {
            if (f32) FSA_RR(float, float4, false, true); else FSA_RR(double, double4, false, true);
        } else if (f32) {
            if (vstar_all) FSA_RR(float, float4, true, false); else FSA_RR(float, float4, false, false);
        } else {
            if (vstar_all) FSA_RR(double, double4, true, false); else FSA_RR(double, double4, false, false);
        }
#undef FSA_RR
        FSA_HIP(hipGetLastError());
    }
    std::swap(s->b.v_star, s->b.v_star_alt);
    if (s->iisph) return launch_iisph_carry(s, out);
    return FSA_OK;
}

int launch_ref_cells(Solver *s) {
    ScopedTimer t(s, "ref_cells");
    hipLaunchKernelGGL(ref_cells_kernel, dim3(blocks_for(s->num_cells)), dim3(kBlock), 0, s->stream, s->num_cells,
                       s->b.cellStart, s->b.gridCellStart, s->b.gridCellEnd);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_index_table(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "index_table");
    hipLaunchKernelGGL(index_table_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                       s->b.uniqueId[s->cur], s->b.indexTable);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}


// ------------------------------------------------------------------------------------------------
// calcHash + sortParticle on caller-owned device arrays, for callers that cannot allocate
// ------------------------------------------------------------------------------------------------
// Everything the counting sort needs beyond its input and its two outputs lives in one file-scope device
// array: every GPU the library runs on holds a copy of it (at its own address) until the process ends.
constexpr int kStaticMaxParticles = FSA_SORT_STATIC_MAX_PARTICLES;
constexpr int kStaticMaxCells = FSA_SORT_STATIC_MAX_CELLS;
constexpr size_t kStaticCells = (size_t)kStaticMaxCells + 8;                       // cellCount, cellStart
constexpr size_t kStaticBlocks = (size_t)kStaticMaxParticles / kBlock + 8;         // undefBlockCount / Start
constexpr size_t kStaticTiles = kStaticCells / kScanTile + 8;                      // scan tile sums
constexpr size_t kStaticInts = 2 * kStaticCells + 2 * (size_t)kStaticMaxParticles + 2 * kStaticBlocks + kStaticTiles;
static __device__ __align__(256) int scratch[kStaticInts];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static int *g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each GPU, looked up once per GPU

// The device address of this GPU's copy of `scratch` (the array's name is only a host-side handle).
static int static_scratch(int device, int **out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        FSA_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(FSA_ERR_HIP, "hipGetSymbolAddress gave a null address on device %d", device);
        g_scratch_ptr[device] = (int *)ptr;
    }
    *out = g_scratch_ptr[device];
    return FSA_OK;
}

}  // namespace fsa

using namespace fsa;

extern "C" int fsa_sort_particle_static(const fsa_params *p, int device, void *hip_stream, int n, const void *d_pos3,
                                        int *d_particleHash, int *d_particleIndex) {
    int rc = check_params(p);
    if (rc) return rc;
    if (n < 0 || n > kStaticMaxParticles)
        return fail(FSA_ERR_CAPACITY, "n = %d: fsa_sort_particle_static takes at most %d particles", n,
                    kStaticMaxParticles);
    DevParams<float> P;
    make_dev_params(*p, P);
    if (P.num_cells > kStaticMaxCells)
        return fail(FSA_ERR_CAPACITY, "%d cells: fsa_sort_particle_static takes at most %d", P.num_cells,
                    kStaticMaxCells);
    if (n > 0 && (!d_pos3 || !d_particleHash || !d_particleIndex)) return fail(FSA_ERR_INVALID, "NULL device array");
    int ndev = 0;
    FSA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= kMaxDevices) return fail(FSA_ERR_INVALID, "device %d of %d", device, ndev);
    if (n == 0) return FSA_OK;
    // the calling thread's current device is put back before returning, whatever happens in between
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    int prev = -1;
    FSA_HIP(hipGetDevice(&prev));
    if (prev != device) {
        FSA_HIP(hipSetDevice(device));
        guard.prev = prev;
    }
    int *base = nullptr;
    rc = static_scratch(device, &base);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)hip_stream;

    // carve the scratch array; the two outputs double as hash[] / slot[] until the scatter has read them
    const int C = P.num_cells, nb = (n + kBlock - 1) / kBlock;
    int *cellCount = base, *cellStart = cellCount + kStaticCells;
    int *tmpIndex = cellStart + kStaticCells, *tmpHash = tmpIndex + kStaticMaxParticles;
    int *undefBlockCount = tmpHash + kStaticMaxParticles, *undefBlockStart = undefBlockCount + kStaticBlocks;
    int *scanTmp = undefBlockStart + kStaticBlocks;
    int *hash = d_particleHash, *slot = d_particleIndex;

    // the array keeps the previous call's contents: the counts must start at zero, the rest is overwritten
    FSA_HIP(hipMemsetAsync(cellCount, 0, (size_t)C * sizeof(int), st));
    hipLaunchKernelGGL((hash_count_kernel<float, Vec3<float>>), dim3(nb), dim3(kBlock), 0, st, P,
                       (const Vec3<float> *)d_pos3, n, hash, slot, cellCount, undefBlockCount);
    const int tilesC = (C + 1 + kScanTile - 1) / kScanTile, tilesU = (nb + 1 + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tilesC), dim3(kBlock), 0, st, cellCount, C, scanTmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, scanTmp, tilesC);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(tilesC), dim3(kBlock), 0, st, cellCount, C, scanTmp, cellStart);
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(tilesU), dim3(kBlock), 0, st, undefBlockCount, nb, scanTmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, scanTmp, tilesU);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(tilesU), dim3(kBlock), 0, st, undefBlockCount, nb, scanTmp,
                       undefBlockStart);
    hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(kBlock), 0, st, n, C, hash, slot, cellStart, undefBlockStart,
                       tmpIndex, tmpHash);
    hipLaunchKernelGGL(rank_kernel, dim3(nb), dim3(kBlock), 0, st, n, tmpIndex, tmpHash, cellStart, d_particleIndex,
                       d_particleHash);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}
