// z-slab decomposition support: one solver per GPU holds the cell layers [z0, z0 + nz) of the global
// grid (its own layers plus halo layers copied from the neighbouring slabs).  The reference runs on
// one device only; what has to be preserved is its sorted order -- ascending cell, ties in the order
// of the previous step's array (solver/sph_solver.cu:46-53 under /root/reference/particle_solver) --
// because the neighbour sums of DensityCell / ForceCell are accumulated in that order.  A slab only
// sees part of the previous array, so every particle carries a 64-bit order key
//     (global cell id at the last sort) << 32 | (rank inside that cell)
// whose ordering equals the previous global array order; ties inside a cell are broken by it.
//
//   slab_pack    after the integration: particles this solver owned at the last sort are classified by
//                the cell layer they are in now; the ones inside the keep range are compacted into the
//                other buffer copy, the ones inside a neighbour's range are written as records into
//                the send buffers (a particle can go to several places: it stays as a halo copy here and
//                becomes an owned particle there).  Halo copies of the last step are dropped.
//   slab_unpack  records received from a neighbour are appended.
//   rank_key     stable rank inside each cell by order key (replaces the by-source-index rank).
//   slab_keys    new keys after the reorder.
//
// Slab groups (fsa_group.hip) run the same exchange without moving the particles that stay: one pass integrates,
// classifies, writes the records and computes the next step's hash (dropped slots are marked, not compacted away),
// a second appends the received records with their hash, and the sort that follows gathers what is left.
#include <algorithm>
#include <vector>

#include "fsa_advect.h"
#include "fsa_hash.h"
#include "fsa_solver.h"

namespace fsa {

template <typename T> struct SlabRecord;
template <> struct SlabRecord<float> {  // 48 bytes
    float4 posType;
    float4 velMass;
    long long orderKey;
    float restDensity;
    int uniqueId;
};
template <> struct SlabRecord<double> {  // 96 bytes
    double4 posType;
    double4 velMass;
    long long orderKey;
    double restDensity;
    int uniqueId;
    int pad0;
    long long pad1;
};
static_assert(sizeof(SlabRecord<float>) == 48, "record layout");
static_assert(sizeof(SlabRecord<double>) == 96, "record layout");

int slab_record_bytes(const Solver *s) {
    return s->precision == FSA_PRECISION_F32 ? (int)sizeof(SlabRecord<float>) : (int)sizeof(SlabRecord<double>);
}

struct SlabRanges {
    int own_lo, own_hi;      // layers owned at the last sort
    int keep_lo, keep_hi;    // layers this solver will hold next (halo included)
    int left_lo, left_hi;    // layers the left neighbour will hold next
    int right_lo, right_hi;  // same for the right neighbour
    int next_lo, next_hi;    // layers this solver will own next (see slab_ranges)
};

// The layers a solver owns next follow from what it and its neighbours hold next: a neighbour's range reaches `halo`
// layers into this solver's own, this solver's range `halo` layers beyond it (slabs are at least halo + 1 thick, so
// nothing of that is clipped by the grid); without a neighbour the own layers end where the held ones do.
static SlabRanges slab_ranges(const int own[2], const int keep[2], const int left[2], const int right[2]) {
    SlabRanges R{own[0], own[1], keep[0], keep[1], left[0], left[1], right[0], right[1], keep[0], keep[1]};
    if (left[1] > left[0]) R.next_lo = (keep[0] + left[1]) / 2;
    if (right[1] > right[0]) R.next_hi = (right[0] + keep[1]) / 2;
    return R;
}

// A particle outside the grid is nobody's neighbour (it is in no cell) and is never sent as a halo copy, but its own
// sums run over the cells around it.  So it is filed under the cell layer of its z coordinate, clamped into the grid,
// lives with the slab that owns that layer, and moves on -- one slab per step -- when it drifts past a cut.
template <typename T> __device__ __forceinline__ int outside_layer(const DevParams<T> &P, T z) {
    const int cz = to_int_sat(r_floor((z - P.gmin[2]) / P.dx));
    return min(max(cz, 0), P.gres[2] - 1);
}
// bit 0 keep, 1 left, 2 right
__device__ __forceinline__ unsigned route_outside(const SlabRanges &R, int layer) {
    return layer < R.next_lo ? 2u : (layer >= R.next_hi ? 4u : 1u);
}

constexpr int kPackItems = 4;  // particles per thread: one atomic per stream for 1024 particles

// One more field per particle that travels with it, `w` reals wide (the pressure of the IISPH step, whose relaxation
// starts from the last step's values; the v_star of the DFSPH step, which the divergence loop after the sort goes on
// from): kept particles take it to the other copy, records to a buffer beside theirs, slot for slot.  All null
// when nothing is carried.
template <typename T> struct SlabExtra {
    const T *in;
    T *out, *left, *right;
    int w;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void slab_pack_kernel(
    DevParams<T> P, int n, int all_owned, SlabRanges R, const int *__restrict__ sortedHash,
    const Vec4<T> *__restrict__ posTypeIn, const Vec4<T> *__restrict__ velMassIn, const T *__restrict__ restDensityIn,
    const int *__restrict__ uniqueIdIn, const long long *__restrict__ keyIn, Vec4<T> *__restrict__ posTypeOut,
    Vec4<T> *__restrict__ velMassOut, T *__restrict__ restDensityOut, int *__restrict__ uniqueIdOut,
    long long *__restrict__ keyOut, int capKeep, SlabRecord<T> *__restrict__ sendLeft, int capLeft,
    SlabRecord<T> *__restrict__ sendRight, int capRight, int *__restrict__ counters, SlabExtra<T> X) {
    constexpr int kWaves = kBlock / 64;
    // per stream (keep, left, right): particles flagged by wave w in item round k, then their prefix
    __shared__ int part[3][kPackItems * kWaves];
    __shared__ int base[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = blockIdx.x * (kBlock * kPackItems);
    n = live_count(P, n);
    if (first >= n) return;
    Vec4<T> p[kPackItems];
    unsigned flags[kPackItems];  // bit 0 keep, 1 left, 2 right
    int before[kPackItems][3];   // flagged lanes below this one in its wave
    int lost = 0;
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        const int i = first + k * kBlock + threadIdx.x;
        flags[k] = 0;
        p[k] = Vec4<T>{};
        if (i < n) {
            bool owned = all_owned != 0;
            if (!owned) {
                const unsigned h = (unsigned)sortedHash[i];
                if (h == FSA_GRID_UNDEF) {
                    owned = true;  // halo copies are always inside the grid when they arrive
                } else {
                    const int gz = (int)((h / (unsigned)P.gres[0]) % (unsigned)P.nz) + P.z0;
                    owned = gz >= R.own_lo && gz < R.own_hi;
                }
            }
            if (owned) {
                p[k] = posTypeIn[i];
                bool inside = in_domain(P, p[k].x, p[k].y, p[k].z);
                int cx = 0, cy = 0, cz = 0;
                if (inside) {
                    calc_grid_pos(P, p[k].x, p[k].y, p[k].z, cx, cy, cz);
                    inside = cx >= 0 && cx < P.gres[0] && cy >= 0 && cy < P.gres[1] && cz >= 0 && cz < P.gres[2];
                }
                if (!inside) {
                    flags[k] = route_outside(R, outside_layer(P, p[k].z));
                } else {
                    if (cz >= R.keep_lo && cz < R.keep_hi) flags[k] |= 1;
                    if (cz >= R.left_lo && cz < R.left_hi) flags[k] |= 2;
                    if (cz >= R.right_lo && cz < R.right_hi) flags[k] |= 4;
                    if (flags[k] == 0) lost++;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned long long m = __ballot((flags[k] >> c) & 1u);
            before[k][c] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) part[c][k * kWaves + wave] = __popcll(m);
        }
    }
    if (lost) atomicAdd(counters + 3, lost);
    __syncthreads();
    if (threadIdx.x < 3) {  // exclusive prefix over (round, wave) and one atomic for the block
        const int c = threadIdx.x;
        int run = 0;
        for (int e = 0; e < kPackItems * kWaves; e++) {
            const int v = part[c][e];
            part[c][e] = run;
            run += v;
        }
        base[c] = run ? atomicAdd(counters + c, run) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        if (flags[k] == 0) continue;
        const int i = first + k * kBlock + threadIdx.x;
        const Vec4<T> v = velMassIn[i];
        const T rd = restDensityIn[i];
        const int uid = uniqueIdIn[i];
        const long long key = keyIn[i];
        T extra[3] = {(T)0, (T)0, (T)0};
        if (X.in)
            for (int c = 0; c < X.w; c++) extra[c] = X.in[(size_t)i * X.w + c];
        const int e = k * kWaves + wave;
        if (flags[k] & 1u) {
            const int slot = base[0] + part[0][e] + before[k][0];
            if (slot < capKeep) {
                posTypeOut[slot] = p[k];
                velMassOut[slot] = v;
                restDensityOut[slot] = rd;
                uniqueIdOut[slot] = uid;
                keyOut[slot] = key;
                if (X.in)
                    for (int c = 0; c < X.w; c++) X.out[(size_t)slot * X.w + c] = extra[c];
            }
        }
        if (flags[k] & 6u) {
            SlabRecord<T> rec{};
            rec.posType = p[k];
            rec.velMass = v;
            rec.orderKey = key;
            rec.restDensity = rd;
            rec.uniqueId = uid;
            if (flags[k] & 2u) {
                const int slot = base[1] + part[1][e] + before[k][1];
                if (slot < capLeft) {
                    sendLeft[slot] = rec;
                    if (X.in)
                        for (int c = 0; c < X.w; c++) X.left[(size_t)slot * X.w + c] = extra[c];
                }
            }
            if (flags[k] & 4u) {
                const int slot = base[2] + part[2][e] + before[k][2];
                if (slot < capRight) {
                    sendRight[slot] = rec;
                    if (X.in)
                        for (int c = 0; c < X.w; c++) X.right[(size_t)slot * X.w + c] = extra[c];
                }
            }
        }
    }
}

// Every particle owned at the last sort (out-of-grid ones included) as a record; the solver's arrays
// stay as they are.  One atomic per wavefront: this runs when the slabs are re-cut, not every step.
template <typename T>
__global__ __launch_bounds__(kBlock) void slab_export_kernel(DevParams<T> P, int n, int all_owned, int own_lo, int own_hi,
                                                             const int *__restrict__ sortedHash,
                                                             const Vec4<T> *__restrict__ posType,
                                                             const Vec4<T> *__restrict__ velMass,
                                                             const T *__restrict__ restDensity,
                                                             const int *__restrict__ uniqueId,
                                                             const long long *__restrict__ key,
                                                             SlabRecord<T> *__restrict__ out, int cap,
                                                             int *__restrict__ counter) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool owned = false;
    if (i < n) {
        owned = all_owned != 0;
        if (!owned) {
            const unsigned h = (unsigned)sortedHash[i];
            if (h == FSA_GRID_UNDEF) {
                owned = true;
            } else {
                const int gz = (int)((h / (unsigned)P.gres[0]) % (unsigned)P.nz) + P.z0;
                owned = gz >= own_lo && gz < own_hi;
            }
        }
    }
    const unsigned long long m = __ballot(owned);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    if (!owned) return;
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (slot >= cap) return;
    SlabRecord<T> rec{};
    rec.posType = posType[i];
    rec.velMass = velMass[i];
    rec.orderKey = key[i];
    rec.restDensity = restDensity[i];
    rec.uniqueId = uniqueId[i];
    out[slot] = rec;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void slab_unpack_kernel(int count, int at, const SlabRecord<T> *__restrict__ recs,
                                                             Vec4<T> *__restrict__ posType,
                                                             Vec4<T> *__restrict__ velMass,
                                                             T *__restrict__ restDensity, int *__restrict__ uniqueId,
                                                             long long *__restrict__ key,
                                                             const T *__restrict__ extras, T *__restrict__ extra,
                                                             int w) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const SlabRecord<T> r = recs[k];
    if (extras)
        for (int c = 0; c < w; c++) extra[(size_t)(at + k) * w + c] = extras[(size_t)k * w + c];
    posType[at + k] = r.posType;
    velMass[at + k] = r.velMass;
    restDensity[at + k] = r.restDensity;
    uniqueId[at + k] = r.uniqueId;
    key[at + k] = r.orderKey;
}

// Stable rank inside each cell by order key (keys are unique).
__global__ __launch_bounds__(kBlock) void rank_key_kernel(int n, const int *__restrict__ tmpIndex,
                                                          const int *__restrict__ tmpHash,
                                                          const int *__restrict__ cellStart,
                                                          const long long *__restrict__ key,
                                                          int *__restrict__ sortedIndex, int *__restrict__ sortedHash) {
    const int sIdx = blockIdx.x * kBlock + threadIdx.x;
    if (sIdx >= n) return;
    const int i = tmpIndex[sIdx];
    const unsigned h = (unsigned)tmpHash[sIdx];
    int p = sIdx;
    if (h != FSA_GRID_UNDEF) {
        const int a = cellStart[h], b = cellStart[h + 1];
        const unsigned long long mine = (unsigned long long)key[i];
        int r = 0;
        for (int t = a; t < b; t++) {
            const int j = tmpIndex[t];
            const unsigned long long other = (unsigned long long)key[j];
            r += (other < mine || (other == mine && j < i)) ? 1 : 0;
        }
        p = a + r;
    }
    sortedIndex[p] = i;
    sortedHash[p] = (int)h;
}

// key[p] = (global cell id << 32) | rank inside the cell; out-of-grid: (GRID_UNDEF << 32) | rank in the tail
template <typename T>
__global__ __launch_bounds__(kBlock) void slab_keys_kernel(DevParams<T> P, int n, const int *__restrict__ sortedHash,
                                                           const int *__restrict__ cellStart,
                                                           long long *__restrict__ key) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const unsigned h = (unsigned)sortedHash[p];
    unsigned long long hi, lo;
    if (h == FSA_GRID_UNDEF) {
        hi = FSA_GRID_UNDEF;
        lo = (unsigned)(p - cellStart[P.num_cells]);
    } else {
        const unsigned gx = (unsigned)P.gres[0];
        const unsigned x = h % gx, t = h / gx;
        const unsigned lz = t % (unsigned)P.nz, y = t / (unsigned)P.nz;
        hi = ((unsigned long long)y * (unsigned)P.gres[2] + (lz + (unsigned)P.z0)) * gx + x;
        lo = (unsigned)(p - cellStart[h]);
    }
    key[p] = (long long)((hi << 32) | lo);
}

__global__ __launch_bounds__(kBlock) void slab_init_keys_kernel(int n, const int *__restrict__ uniqueId,
                                                                long long *__restrict__ key) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) key[i] = (long long)(unsigned)uniqueId[i];
}

// ---- exchange without host round trips (slab groups, fsa_group.hip) --------------------------------------------
// (slabVars[]: fsa_solver.h)

// After slab_edge_send_kernel: the counts leave the atomic counters (which go back to zero for the next pack), each
// send buffer gets its count into its header slot, and a copy goes to `mirror` (pinned host memory) for the host to
// read some steps later.  Counts beyond what the messages take are cut and flagged in slabVars[kVarError]: bits
// 1 / 2 send buffer left / right.  The particle count stays: nothing has been moved.
__global__ void slab_after_pack_kernel(int *__restrict__ counters, int *__restrict__ vars, int sizeLeft,
                                       int sizeRight, long long *__restrict__ hdrLeft,
                                       long long *__restrict__ hdrRight, int *__restrict__ mirror) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int l = counters[1], r = counters[2], err = 0;
    if (l > sizeLeft) { l = sizeLeft; err |= 2; }
    if (r > sizeRight) { r = sizeRight; err |= 4; }
    vars[kVarOutLeft] = l;
    vars[kVarOutRight] = r;
    if (err) atomicOr(vars + kVarError, err);  // (the keep pass may be flagging on another stream)
    if (hdrLeft) hdrLeft[0] = l;
    if (hdrRight) hdrRight[0] = r;
    if (mirror) {
        mirror[0] = l;
        mirror[1] = r;
        mirror[2] = vars[kVarN];
        mirror[3] = err;
    }
    counters[1] = counters[2] = 0;
}

// The counts of the two receive buffers as the unpack uses them (cut to what the message and the capacity take).
__device__ __forceinline__ void incoming_counts(const long long *hdrLeft, const long long *hdrRight, int boundLeft,
                                                int boundRight, int at, int capacity, int &cl, int &cr, int &err) {
    err = 0;
    long long l = hdrLeft ? hdrLeft[0] : 0, r = hdrRight ? hdrRight[0] : 0;
    if (l < 0 || l > boundLeft) { l = l < 0 ? 0 : boundLeft; err |= 8; }
    if (r < 0 || r > boundRight) { r = r < 0 ? 0 : boundRight; err |= 16; }
    const long long room = (long long)capacity - at;
    if (l + r > room) {
        err |= 1;
        if (l > room) l = room;
        r = room - l;
    }
    cl = (int)l;
    cr = (int)r;
}

// Slab groups split the work after the force pass in two, so that the records are on their way while most of the
// step is still being computed:
//   edge send   particles that were in a layer next to a cut at the last sort (DevParams::edge) -- the only ones
//               that can have anything to do with a neighbour -- are integrated in registers (ADVECT; AdvectKernel
//               as in fsa_advect.h) and written as records to the neighbours whose layers they are in now.  The
//               arrays stay as they are: the force pass of the other blocks is still reading them;
//   keep + hash every slot of the current arrays: the particles owned at the last sort are integrated (those of the
//               first pass once more, to the same bits), and every one that is still inside the layers held here
//               gets its hash and arrival slot, its cell counted.  Every other slot -- halo copies of the last
//               step, particles that left the layers held here -- gets hash kHashDrop, which the sort skips.
//               Nothing moves.  cellCount must be all zero on entry.
// Before the first sort of a solver nothing tells owned from halo and edge from interior: everything is owned and
// goes through the edge pass.
__device__ __forceinline__ void owned_and_edge(const DevParams<float> &P, const SlabRanges &R, int all_owned,
                                               const int *__restrict__ sortedHash, int i, bool &owned, bool &edge) {
    owned = edge = all_owned != 0;
    if (!all_owned) {
        const unsigned h = (unsigned)sortedHash[i];
        if (h == FSA_GRID_UNDEF) {
            // (halo copies are always inside the grid when they arrive.)  Where such a particle drifts, no layer of
            // the last sort tells: it goes through the edge pass
            owned = edge = true;
        } else {
            const int gz = (int)((h / (unsigned)P.gres[0]) % (unsigned)P.nz) + P.z0;
            owned = gz >= R.own_lo && gz < R.own_hi;
            edge = owned && layer_in_edge(P, gz);
        }
    }
}

// bit 0 keep, 1 left, 2 right; `lost`: inside the grid but in nobody's layers
__device__ __forceinline__ unsigned classify(const DevParams<float> &P, const SlabRanges &R, const float4 &p, bool &lost) {
    lost = false;
    bool inside = in_domain(P, p.x, p.y, p.z);
    int cx = 0, cy = 0, cz = 0;
    if (inside) {
        calc_grid_pos(P, p.x, p.y, p.z, cx, cy, cz);
        inside = cx >= 0 && cx < P.gres[0] && cy >= 0 && cy < P.gres[1] && cz >= 0 && cz < P.gres[2];
    }
    if (!inside) return route_outside(R, outside_layer(P, p.z));
    unsigned flags = 0;
    if (cz >= R.keep_lo && cz < R.keep_hi) flags |= 1u;
    if (cz >= R.left_lo && cz < R.left_hi) flags |= 2u;
    if (cz >= R.right_lo && cz < R.right_hi) flags |= 4u;
    lost = flags == 0;
    return flags;
}

template <bool ADVECT>
__global__ __launch_bounds__(kBlock) void slab_edge_send_kernel(
    DevParams<float> P, int bound, int all_owned, SlabRanges R, const int *__restrict__ sortedHash,
    const float4 *__restrict__ posType, const float4 *__restrict__ velMass, const Vec3<float> *__restrict__ force,
    const float *__restrict__ restDensity, const int *__restrict__ uniqueId, const long long *__restrict__ key,
    SlabRecord<float> *__restrict__ sendLeft, int capLeft, SlabRecord<float> *__restrict__ sendRight, int capRight,
    int *__restrict__ counters) {
    constexpr int kWaves = kBlock / 64;
    __shared__ int part[2][kWaves];  // left, right: particles flagged by each wave, then their prefix
    __shared__ int base[2];
    const int n = live_count(P, bound);
    if ((int)blockIdx.x * kBlock >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool owned = false, edge = false;
    if (i < n) owned_and_edge(P, R, all_owned, sortedHash, i, owned, edge);
    if (!__syncthreads_or(edge ? 1 : 0)) return;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p;
    unsigned flags = 0;
    if (edge) {
        p = posType[i];
        v = velMass[i];
        if (ADVECT && advect_moves(P, unpack_int(p.w))) {
            Vec3<float> vs;
            advect_particle(P, p, v, force[i], vs);
        }
        bool lost;
        flags = classify(P, R, p, lost);
    }
    int before[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const unsigned long long m = __ballot((flags >> (c + 1)) & 1u);
        before[c] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) part[c][wave] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 2) {  // exclusive prefix over the waves and one atomic for a block that sends anything
        const int c = threadIdx.x;
        int run = 0;
        for (int w = 0; w < kWaves; w++) {
            const int t = part[c][w];
            part[c][w] = run;
            run += t;
        }
        base[c] = run ? atomicAdd(counters + 1 + c, run) : 0;
    }
    __syncthreads();
    if (flags & 6u) {
        SlabRecord<float> rec{};
        rec.posType = p;
        rec.velMass = v;
        rec.orderKey = key[i];
        rec.restDensity = restDensity[i];
        rec.uniqueId = uniqueId[i];
        if (flags & 2u) {
            const int at = base[0] + part[0][wave] + before[0];
            if (at < capLeft) sendLeft[at] = rec;
        }
        if (flags & 4u) {
            const int at = base[1] + part[1][wave] + before[1];
            if (at < capRight) sendRight[at] = rec;
        }
    }
}

// vars[kVarError] bit 5: a particle outside the edge layers ended up in a neighbour's layers (it crossed a whole
// cell layer in one step); its record was not sent and the run is invalid.
template <bool ADVECT>
__global__ __launch_bounds__(kBlock) void slab_keep_hash_kernel(
    DevParams<float> P, int bound, int all_owned, SlabRanges R, const int *__restrict__ sortedHash,
    float4 *__restrict__ posType, float4 *__restrict__ velMass, const Vec3<float> *__restrict__ force,
    Vec3<float> *__restrict__ v_star, int *__restrict__ counters, int *__restrict__ vars, int *__restrict__ hash,
    int *__restrict__ slot, int *__restrict__ cellCount, int *__restrict__ undefBlockCount) {
    __shared__ int waveCount[kBlock / 64];
    const int n = live_count(P, bound);
    if ((int)blockIdx.x * kBlock >= n) {
        // slots the next unpack may append to: their blocks start without out-of-grid particles
        if (threadIdx.x == 0) undefBlockCount[blockIdx.x] = 0;
        return;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned flags = 0;
    if (valid) {
        bool owned, edge;
        owned_and_edge(P, R, all_owned, sortedHash, i, owned, edge);
        if (owned) {
            p = posType[i];
            if (ADVECT && advect_moves(P, unpack_int(p.w))) {
                float4 v = velMass[i];
                Vec3<float> vs;
                advect_particle(P, p, v, force[i], vs);
                v_star[i] = vs;
                posType[i] = p;
                velMass[i] = v;
            }
            bool lost;
            flags = classify(P, R, p, lost);
            if (lost) atomicAdd(vars + kVarLost, 1);
            if (!edge && (flags & 6u)) atomicOr(vars + kVarError, 32);
        }
    }
    const bool keep = (flags & 1u) != 0;
    hash_and_count(P, keep, i, p.x, p.y, p.z, hash, slot, cellCount, undefBlockCount, waveCount);
    if (valid && !keep) hash[i] = (int)kHashDrop;
}

// Appends the records of both receive buffers (left neighbour's first) behind the slots the arrays hold, with
// their hash and arrival slots (they lie in the layers this solver holds: the sender selected them by that).
__global__ __launch_bounds__(kBlock) void slab_unpack_hash_kernel(
    DevParams<float> P, const char *__restrict__ recvLeft, const char *__restrict__ recvRight, int boundLeft,
    int boundRight, int *__restrict__ vars, int capacity, float4 *__restrict__ posType,
    float4 *__restrict__ velMass, float *__restrict__ restDensity, int *__restrict__ uniqueId,
    long long *__restrict__ key, int *__restrict__ hash, int *__restrict__ slot, int *__restrict__ cellCount,
    int *__restrict__ undefBlockCount, int *__restrict__ mirror) {
    const int at = vars[kVarN];  // (not written by this kernel)
    int cl, cr, err;
    incoming_counts((const long long *)recvLeft, (const long long *)recvRight, boundLeft, boundRight, at, capacity, cl,
                    cr, err);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the number of slots after the append (what the sort reads), the counts, and their mirror for the host
        vars[kVarSlots] = at + cl + cr;
        vars[kVarInLeft] = cl;
        vars[kVarInRight] = cr;
        if (err) atomicOr(vars + kVarError, err);
        if (mirror) {
            // the counts as the neighbours sent them (cut to what the messages hold), not as the capacity let them
            // in: the neighbours size their next messages from the same numbers, and both ends must arrive at the
            // same size
            const long long hl = recvLeft ? ((const long long *)recvLeft)[0] : 0,
                            hr = recvRight ? ((const long long *)recvRight)[0] : 0;
            mirror[4] = (int)(hl < 0 ? 0 : (hl > boundLeft ? boundLeft : hl));
            mirror[5] = (int)(hr < 0 ? 0 : (hr > boundRight ? boundRight : hr));
            mirror[6] = at + cl + cr;
            mirror[7] = err;
        }
    }
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if ((int)blockIdx.x * kBlock >= cl + cr) return;
    const bool valid = k < cl + cr;
    const int i = at + k;
    unsigned h = kHashDrop;
    if (valid) {
        const SlabRecord<float> *src = k < cl ? (const SlabRecord<float> *)(recvLeft + sizeof(SlabRecord<float>)) + k
                                              : (const SlabRecord<float> *)(recvRight + sizeof(SlabRecord<float>)) + (k - cl);
        const SlabRecord<float> r = *src;
        posType[i] = r.posType;
        velMass[i] = r.velMass;
        restDensity[i] = r.restDensity;
        uniqueId[i] = r.uniqueId;
        key[i] = r.orderKey;
        bool inside = in_domain(P, r.posType.x, r.posType.y, r.posType.z);
        int cx = 0, cy = 0, cz = 0;
        if (inside) {
            calc_grid_pos(P, r.posType.x, r.posType.y, r.posType.z, cx, cy, cz);
            inside = cx >= 0 && cx < P.gres[0] && cy >= 0 && cy < P.gres[1] && cz >= 0 && cz < P.gres[2];
        }
        if (inside) {
            h = calc_grid_hash(P, cx, cy, cz);
            if (h == FSA_GRID_UNDEF) {  // not in the layers held here: nobody should have sent it
                h = kHashDrop;
                atomicAdd(vars + kVarLost, 1);
            }
        } else {
            h = FSA_GRID_UNDEF;  // an out-of-grid particle that drifted past the cut: it lives here now
        }
    }
    const bool in_cell = valid && h != kHashDrop && h != FSA_GRID_UNDEF;
    int my = count_in_cell(in_cell, h, cellCount);
    // out-of-grid ranks are kept per block of kBlock slots (the keep pass has counted the slots below `at` and
    // zeroed the blocks above): an arrival takes the next rank of its block
    if (valid && h == FSA_GRID_UNDEF) my = atomicAdd(undefBlockCount + i / kBlock, 1);
    if (valid) {
        hash[i] = (int)h;
        slot[i] = my;
    }
}

static inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

int launch_slab_init_keys(Solver *s) {
    if (s->n == 0) return FSA_OK;
    hipLaunchKernelGGL(slab_init_keys_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n,
                       s->b.uniqueId[s->cur], s->b.orderKey[s->cur]);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

int launch_rank_key(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "rank");
    hipLaunchKernelGGL(rank_key_kernel, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->n, s->b.tmpIndex,
                       s->b.tmpHash, s->b.cellStart, s->b.orderKey[s->cur], s->b.sortedIndex, s->b.sortedHash);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// after launch_reorder, before `cur` flips: keys of the new order go to the other copy
int launch_slab_keys(Solver *s) {
    if (s->n == 0) return FSA_OK;
    ScopedTimer t(s, "slab_keys");
    const int out = 1 - s->cur;
    if (s->precision == FSA_PRECISION_F32)
        hipLaunchKernelGGL(slab_keys_kernel<float>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->pf, s->n,
                           s->b.sortedHash, s->b.cellStart, s->b.orderKey[out]);
    else
        hipLaunchKernelGGL(slab_keys_kernel<double>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->pd, s->n,
                           s->b.sortedHash, s->b.cellStart, s->b.orderKey[out]);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

template <typename T>
static int pack_t(Solver *s, const DevParams<T> &P, const SlabRanges &R, void *sendLeft, int capLeft, void *sendRight,
                  int capRight, void *extraLeft, void *extraRight) {
    const int in = s->cur, out = 1 - s->cur;
    SlabExtra<T> X{nullptr, nullptr, nullptr, nullptr, 0};
    if (s->slab_carry) X = SlabExtra<T>{(const T *)s->b.pressure, (T *)s->b.pressure_alt, (T *)extraLeft, (T *)extraRight, 1};
    if (s->slab_carry_vstar) X = SlabExtra<T>{(const T *)s->b.v_star, (T *)s->b.v_star_alt, (T *)extraLeft, (T *)extraRight, 3};
    hipLaunchKernelGGL(slab_pack_kernel<T>, dim3((s->grid_n() + kBlock * kPackItems - 1) / (kBlock * kPackItems)), dim3(kBlock), 0, s->stream, P, s->grid_n(),
                       s->owned_from_sort ? 0 : 1, R, s->b.sortedHash, (const Vec4<T> *)s->b.posType[in],
                       (const Vec4<T> *)s->b.velMass[in], (const T *)s->b.restDensity[in], s->b.uniqueId[in],
                       s->b.orderKey[in], (Vec4<T> *)s->b.posType[out], (Vec4<T> *)s->b.velMass[out],
                       (T *)s->b.restDensity[out], s->b.uniqueId[out], s->b.orderKey[out], s->capacity,
                       (SlabRecord<T> *)sendLeft, capLeft, (SlabRecord<T> *)sendRight, capRight, s->b.slabCounters, X);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// extraLeft / extraRight: with Solver::slab_carry, where the carried scalar of each record goes (capLeft / capRight
// reals), slot for slot beside the records
int launch_slab_pack(Solver *s, const int own[2], const int keep[2], const int left[2], const int right[2],
                     void *sendLeft, int capLeft, void *sendRight, int capRight, int counts[4], void *extraLeft,
                     void *extraRight) {
    FSA_HIP(hipMemsetAsync(s->b.slabCounters, 0, 4 * sizeof(int), s->stream));
    if (s->n > 0) {
        ScopedTimer t(s, "slab_pack");
        const SlabRanges R = slab_ranges(own, keep, left, right);
        int rc = s->precision == FSA_PRECISION_F32
                     ? pack_t<float>(s, s->pf, R, sendLeft, capLeft, sendRight, capRight, extraLeft, extraRight)
                     : pack_t<double>(s, s->pd, R, sendLeft, capLeft, sendRight, capRight, extraLeft, extraRight);
        if (rc) return rc;
        if (s->slab_carry) std::swap(s->b.pressure, s->b.pressure_alt);  // the kept particles' values went there
        if (s->slab_carry_vstar) std::swap(s->b.v_star, s->b.v_star_alt);
    }
    FSA_HIP(hipMemcpyAsync(counts, s->b.slabCounters, 4 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    return FSA_OK;
}

int launch_slab_export(Solver *s, const int own[2], void *recs, int cap, int *count) {
    FSA_HIP(hipMemsetAsync(s->b.slabCounters, 0, 4 * sizeof(int), s->stream));
    if (s->n > 0) {
        ScopedTimer t(s, "slab_export");
        const int c = s->cur, all_owned = s->owned_from_sort ? 0 : 1;
        if (s->precision == FSA_PRECISION_F32)
            hipLaunchKernelGGL(slab_export_kernel<float>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->pf, s->n,
                               all_owned, own[0], own[1], s->b.sortedHash, (const float4 *)s->b.posType[c],
                               (const float4 *)s->b.velMass[c], (const float *)s->b.restDensity[c], s->b.uniqueId[c],
                               s->b.orderKey[c], (SlabRecord<float> *)recs, cap, s->b.slabCounters);
        else
            hipLaunchKernelGGL(slab_export_kernel<double>, dim3(blocks_for(s->n)), dim3(kBlock), 0, s->stream, s->pd,
                               s->n, all_owned, own[0], own[1], s->b.sortedHash, (const double4 *)s->b.posType[c],
                               (const double4 *)s->b.velMass[c], (const double *)s->b.restDensity[c],
                               s->b.uniqueId[c], s->b.orderKey[c], (SlabRecord<double> *)recs, cap,
                               s->b.slabCounters);
        FSA_HIP(hipGetLastError());
    }
    FSA_HIP(hipMemcpyAsync(count, s->b.slabCounters, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    return FSA_OK;
}

int launch_slab_unpack(Solver *s, const void *recs, int count, const void *extras) {
    if (count == 0) return FSA_OK;
    ScopedTimer t(s, "slab_unpack");
    const int c = s->cur;
    if (s->precision == FSA_PRECISION_F32)
        hipLaunchKernelGGL(slab_unpack_kernel<float>, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream, count, s->n,
                           (const SlabRecord<float> *)recs, (float4 *)s->b.posType[c], (float4 *)s->b.velMass[c],
                           (float *)s->b.restDensity[c], s->b.uniqueId[c], s->b.orderKey[c], (const float *)extras,
                           (float *)(s->slab_carry_vstar ? s->b.v_star : s->b.pressure), s->slab_carry_vstar ? 3 : 1);
    else
        hipLaunchKernelGGL(slab_unpack_kernel<double>, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream, count,
                           s->n, (const SlabRecord<double> *)recs, (double4 *)s->b.posType[c],
                           (double4 *)s->b.velMass[c], (double *)s->b.restDensity[c], s->b.uniqueId[c],
                           s->b.orderKey[c], (const double *)extras,
                           (double *)(s->slab_carry_vstar ? s->b.v_star : s->b.pressure), s->slab_carry_vstar ? 3 : 1);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// The exchange of a group step, sending side (no counts come back to the host): the edge pass and the message
// headers.  sendLeft / sendRight are whole send buffers (header slot first); sizeLeft / sizeRight the records this
// step's messages take.  with_advect: the integration of the step before runs in the same passes.
int launch_slab_edge_send(Solver *s, bool with_advect, const int own[2], const int left[2], const int right[2],
                          void *sendLeft, int sizeLeft, void *sendRight, int sizeRight) {
    const int rb = slab_record_bytes(s), c = s->cur, n = s->grid_n();
    ScopedTimer t(s, "edge_send");
    const int held[2] = {s->pf.z0, s->pf.z0 + s->pf.nz};
    const SlabRanges R = slab_ranges(own, held, left, right);
    char *l = (char *)sendLeft, *r = (char *)sendRight;
#define FSA_ES(ADV)                                                                                                  \
    hipLaunchKernelGGL(slab_edge_send_kernel<ADV>, dim3(blocks_for(n)), dim3(kBlock), 0, s->stream, s->pf, n,        \
                       s->owned_from_sort ? 0 : 1, R, s->b.sortedHash, (const float4 *)s->b.posType[c],              \
                       (const float4 *)s->b.velMass[c], (const Vec3<float> *)s->b.force,                             \
                       (const float *)s->b.restDensity[c], s->b.uniqueId[c], s->b.orderKey[c],                       \
                       (SlabRecord<float> *)(l ? l + rb : nullptr), l ? sizeLeft : 0,                                \
                       (SlabRecord<float> *)(r ? r + rb : nullptr), r ? sizeRight : 0, s->b.slabCounters)
    if (n > 0 && (l || r)) {  // (no neighbour, no records)
        if (with_advect) FSA_ES(true); else FSA_ES(false);
    }
#undef FSA_ES
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// The message headers of the edge pass, on `stream` (where it must follow the edge pass).
int launch_slab_pack_tail(Solver *s, void *stream, void *sendLeft, int sizeLeft, void *sendRight, int sizeRight,
                          int *mirror) {
    char *l = (char *)sendLeft, *r = (char *)sendRight;
    hipLaunchKernelGGL(slab_after_pack_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s->b.slabCounters,
                       s->b.slabVars, l ? sizeLeft : 0, r ? sizeRight : 0, (long long *)l, (long long *)r, mirror);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// ... and the pass over everything: who stays, and the hash of the next sort.
int launch_slab_keep_hash(Solver *s, bool with_advect, const int own[2], const int keep[2], const int left[2],
                          const int right[2]) {
    const int c = s->cur, n = s->grid_n();
    if (!s->counts_clean) {
        ScopedTimer t(s, "clear_counts");
        FSA_HIP(hipMemsetAsync(s->b.cellCount, 0, (size_t)s->num_cells * sizeof(int), s->stream));
    }
    s->counts_clean = false;
    ScopedTimer t(s, with_advect ? "advect_keep_hash" : "keep_hash");
    const SlabRanges R = slab_ranges(own, keep, left, right);
#define FSA_KH(ADV)                                                                                                  \
    hipLaunchKernelGGL(slab_keep_hash_kernel<ADV>, dim3(blocks_for(n)), dim3(kBlock), 0, s->stream, s->pf, n,        \
                       s->owned_from_sort ? 0 : 1, R, s->b.sortedHash, (float4 *)s->b.posType[c],                    \
                       (float4 *)s->b.velMass[c], (const Vec3<float> *)s->b.force, (Vec3<float> *)s->b.v_star,       \
                       s->b.slabCounters, s->b.slabVars, s->b.hash, s->b.slot, s->b.cellCount, s->b.undefBlockCount)
    if (n > 0) {
        if (with_advect) FSA_KH(true); else FSA_KH(false);
    }
#undef FSA_KH
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

// Receiving side: append, hash, and the slot count for the sort.
int launch_slab_unpack_hash(Solver *s, const void *recvLeft, int boundLeft, const void *recvRight, int boundRight,
                            int *mirror) {
    ScopedTimer t(s, "unpack_hash");
    const int c = s->cur, bound = (recvLeft ? boundLeft : 0) + (recvRight ? boundRight : 0);
    // (at least one block: its first thread also publishes the counts)
    hipLaunchKernelGGL(slab_unpack_hash_kernel, dim3(std::max(blocks_for(bound), 1)), dim3(kBlock), 0, s->stream, s->pf,
                       (const char *)recvLeft, (const char *)recvRight, boundLeft, boundRight, s->b.slabVars,
                       s->capacity, (float4 *)s->b.posType[c], (float4 *)s->b.velMass[c],
                       (float *)s->b.restDensity[c], s->b.uniqueId[c], s->b.orderKey[c], s->b.hash, s->b.slot,
                       s->b.cellCount, s->b.undefBlockCount, mirror);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}


// ---- field refresh between the neighbour passes of a step that runs many of them (IISPH on slabs) ------------
// The sorted order of a slab solver is (y, layer, x, order key).  The in-grid particles of the global layers
// [lo, hi) are therefore, for every y, one contiguous run of the sorted arrays, and two solvers that both hold
// those layers hold the same particles in the same order.  A solver that owns the layers gathers a field of
// theirs into a buffer; the neighbour that holds them as halo copies scatters the buffer onto its own: both walk
// the runs in ascending sorted position.

// offsets[y] = particles of the layers in rows before y; offsets[gy] = their number (one thread: gy is a few hundred)
__global__ void layer_runs_kernel(int gx, int gy, int nz, int zl0, int zl1, const int *__restrict__ cellStart,
                                  int *__restrict__ offsets) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int run = 0;
    for (int y = 0; y < gy; y++) {
        offsets[y] = run;
        run += cellStart[((size_t)y * nz + zl1) * gx] - cellStart[((size_t)y * nz + zl0) * gx];
    }
    offsets[gy] = run;
}

__global__ __launch_bounds__(kBlock) void layer_indices_kernel(int gx, int gy, int nz, int zl0, int count,
                                                               const int *__restrict__ cellStart,
                                                               const int *__restrict__ offsets,
                                                               int *__restrict__ idx) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= count) return;
    int lo = 0, hi = gy - 1;  // the last y whose offset is <= e
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= e) lo = mid; else hi = mid - 1;
    }
    idx[e] = cellStart[((size_t)lo * nz + zl0) * gx] + (e - offsets[lo]);
}

template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void field_move_kernel(int count, int words, const int *__restrict__ idx,
                                                            unsigned *__restrict__ field, unsigned *__restrict__ buf) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)count * words) return;
    const int e = (int)(t / words), w = (int)(t % words);
    if (SCATTER) field[(size_t)idx[e] * words + w] = buf[t];
    else buf[t] = field[(size_t)idx[e] * words + w];
}

// Sum of a[] over the particles this solver owns (cell layer in [lo, hi), or outside the grid), in double, block
// partial sums folded in a fixed order.
// (MAX: the maximum instead, from `start` on: the stop test of the DFSPH loops, solver/sph_solver.cu:176-180)
template <typename T, bool MAX>
__global__ __launch_bounds__(kBlock) void owned_sum_kernel(DevParams<T> P, int n, int lo, int hi,
                                                           const int *__restrict__ sortedHash, const T *__restrict__ a,
                                                           double *__restrict__ part_out, double start) {
    __shared__ double part[kBlock / 64];
    double acc = MAX ? start : 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned h = (unsigned)sortedHash[i];
        bool owned = h == (unsigned)FSA_GRID_UNDEF;
        if (!owned) {
            const int gz = (int)((h / (unsigned)P.gres[0]) % (unsigned)P.nz) + P.z0;
            owned = gz >= lo && gz < hi;
        }
        if (owned) acc = MAX ? fmax(acc, (double)a[i]) : acc + (double)a[i];
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(acc, d, 64);
        acc = MAX ? fmax(acc, o) : acc + o;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) acc = MAX ? fmax(acc, part[w]) : acc + part[w];
        part_out[blockIdx.x] = acc;
    }
}

int launch_slab_layer_indices(Solver *s, int lo, int hi, int *d_idx, int cap, int *count) {
    const int gx = s->pf.gres[0], gy = s->pf.gres[1], nz = s->pf.nz, z0 = s->pf.z0;
    *count = 0;
    if (lo >= hi) return FSA_OK;
    // the prefix table lives in the scratch of the block maxima (gy + 1 ints of its 2048 doubles)
    if (gy + 1 > 4096) return fail(FSA_ERR_INVALID, "gridres[1] = %d: more rows than the run table takes", gy);
    int *offsets = (int *)s->b.reduce_tmp;
    hipLaunchKernelGGL(layer_runs_kernel, dim3(1), dim3(64), 0, s->stream, gx, gy, nz, lo - z0, hi - z0, s->b.cellStart, offsets);
    FSA_HIP(hipGetLastError());
    int total = 0;
    FSA_HIP(hipMemcpyAsync(&total, offsets + gy, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    *count = total;
    if (total > cap) return fail(FSA_ERR_CAPACITY, "%d particles in layers [%d, %d) for %d index slots", total, lo, hi, cap);
    if (total == 0) return FSA_OK;
    hipLaunchKernelGGL(layer_indices_kernel, dim3(blocks_for(total)), dim3(kBlock), 0, s->stream, gx, gy, nz, lo - z0, total,
                       s->b.cellStart, offsets, d_idx);
    FSA_HIP(hipGetLastError());
    // (the table's scratch is free again once the kernel has run; the passes that use it come later on this stream)
    return FSA_OK;
}

int launch_field_move(Solver *s, bool scatter, void *field, int words, const int *d_idx, int count, void *d_buf) {
    if (count == 0) return FSA_OK;
    const long long total = (long long)count * words;
    if (scatter)
        hipLaunchKernelGGL(field_move_kernel<true>, dim3(blocks_for(total)), dim3(kBlock), 0, s->stream, count, words, d_idx,
                           (unsigned *)field, (unsigned *)d_buf);
    else
        hipLaunchKernelGGL(field_move_kernel<false>, dim3(blocks_for(total)), dim3(kBlock), 0, s->stream, count, words, d_idx,
                           (unsigned *)field, (unsigned *)d_buf);
    FSA_HIP(hipGetLastError());
    return FSA_OK;
}

static int owned_fold(Solver *s, int lo, int hi, const void *a, bool is_max, double start, double *out) {
    *out = is_max ? start : 0.0;
    if (s->n == 0) return FSA_OK;
    const int nb = std::min(1024, blocks_for(s->n));
    double *tmp = (double *)s->b.reduce_tmp;
#define FSA_FOLD(T, P, MAX)                                                                                       \
    hipLaunchKernelGGL((owned_sum_kernel<T, MAX>), dim3(nb), dim3(kBlock), 0, s->stream, P, s->n, lo, hi, s->b.sortedHash, \
                       (const T *)a, tmp, start)
    if (s->precision == FSA_PRECISION_F32) {
        if (is_max) FSA_FOLD(float, s->pf, true); else FSA_FOLD(float, s->pf, false);
    } else {
        if (is_max) FSA_FOLD(double, s->pd, true); else FSA_FOLD(double, s->pd, false);
    }
#undef FSA_FOLD
    FSA_HIP(hipGetLastError());
    std::vector<double> h(nb);
    FSA_HIP(hipMemcpyAsync(h.data(), tmp, nb * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    double acc = is_max ? start : 0.0;
    for (int b = 0; b < nb; b++) acc = is_max ? std::max(acc, h[b]) : acc + h[b];
    *out = acc;
    return FSA_OK;
}

int launch_owned_sum(Solver *s, int lo, int hi, const void *a, double *out) {
    return owned_fold(s, lo, hi, a, false, 0.0, out);
}

int launch_owned_max(Solver *s, int lo, int hi, const void *a, double start, double *out) {
    return owned_fold(s, lo, hi, a, true, start, out);
}

}  // namespace fsa
