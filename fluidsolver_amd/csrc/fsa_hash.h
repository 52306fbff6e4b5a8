// Cell hash and per-cell arrival slots of the counting sort (calcHashD, solver/sph_kernel_impl.cuh:83-112 under
// /root/reference/particle_solver), shared by the kernels that compute them: the plain hash pass, the integration
// that goes straight on to the next step's hash, and the z-slab exchange kernels.
#pragma once

#include "fsa_device.h"

namespace fsa {

// hash[] of a slot whose particle has left this solver (z-slab groups): it is not counted and the sort skips it.
constexpr unsigned kHashDrop = 0xfffffffeu;

// Arrival slot of a particle inside cell h (`counts` false: the lane has nothing to count; every lane of the
// wavefront calls this).  One atomic per run of equal cells inside the wavefront instead of one per particle: after
// the first step the input is almost sorted, so a run is usually a whole cell.  The run leader adds the run length
// and hands the base to the others; the order inside a cell is settled later (rank).
__device__ __forceinline__ int count_in_cell(bool counts, unsigned h, int *__restrict__ cellCount) {
    const int lane = threadIdx.x & 63;
    const unsigned hprev = (unsigned)__shfl_up((int)h, 1, 64);
    const bool start = lane == 0 || !counts || h != hprev;
    const unsigned long long starts = __ballot(start);
    const unsigned long long upto = starts & ((2ull << lane) - 1ull);     // run starts at or below this lane
    const unsigned long long above = starts & ~((2ull << lane) - 1ull);  // run starts above it
    const int leader = 63 - __clzll((long long)upto);
    const int next = above ? __ffsll((long long)above) - 1 : 64;
    int base = 0;
    if (counts && lane == leader) base = atomicAdd(&cellCount[h], next - leader);
    base = __shfl(base, leader, 64);
    return counts ? base + (lane - leader) : 0;
}

// The hash and count of particle i at (x, y, z); every thread of the block calls it (`valid` = there is a particle).
template <typename T>
__device__ __forceinline__ void hash_and_count(const DevParams<T> &P, bool valid, int i, T x, T y, T z,
                                               int *__restrict__ hash, int *__restrict__ slot,
                                               int *__restrict__ cellCount, int *__restrict__ undefBlockCount,
                                               int *waveCount /* shared, kBlock / 64 */) {
    unsigned h = 0;
    if (valid) {
        if (!in_domain(P, x, y, z)) {
            h = FSA_GRID_UNDEF;
        } else {
            int cx, cy, cz;
            calc_grid_pos(P, x, y, z, cx, cy, cz);
            h = calc_grid_hash(P, cx, cy, cz);
        }
        hash[i] = (int)h;
    }
    const bool undef = valid && h == FSA_GRID_UNDEF;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int my = count_in_cell(valid && !undef, h, cellCount);
    const unsigned long long m = __ballot(undef);
    if (lane == 0) waveCount[wave] = __popcll(m);
    __syncthreads();
    if (undef) {
        int base = 0;
        for (int w = 0; w < wave; w++) base += waveCount[w];
        my = base + __popcll(m & ((1ull << lane) - 1ull));
    }
    if (valid) slot[i] = my;
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < kBlock / 64; w++) tot += waveCount[w];
        undefBlockCount[blockIdx.x] = tot;
    }
}

}  // namespace fsa
