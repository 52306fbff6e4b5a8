// Host-side solver object behind the C ABI (include/fluidsolver_amd.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <map>
#include <string>
#include <vector>

#include "../../include/fluidsolver_amd.h"
#include "fsa_device.h"

namespace fsa {

// Device buffers of one solver.  Per-particle hot data is kept twice (index `cur` and `1 - cur`):
// the counting-sort reorder gathers from one copy into the other and flips `cur`, which replaces the
// reference's sorted->primary device copies (solver/sph_solver.cpp:58-84).
struct Buffers {
    // hot set, sorted order after a sort: {x, y, z, type bits}, {vx, vy, vz, mass}
    void *posType[2] = {nullptr, nullptr};
    void *velMass[2] = {nullptr, nullptr};
    void *restDensity[2] = {nullptr, nullptr};
    int *uniqueId[2] = {nullptr, nullptr};
    // per-step results (index = current device order)
    void *density = nullptr;
    void *pressure = nullptr;
    void *force = nullptr;   // tight 3-vectors
    void *v_star = nullptr;  // tight 3-vectors; reordered with the particles like the hot set ...
    void *v_star_alt = nullptr;  // ... into this copy, then the two pointers swap
    // DFSPH step
    void *DF_factor = nullptr;
    void *pstiff = nullptr;
    void *error = nullptr;
    void *reduce_tmp = nullptr;  // block maxima of error[]
    void *aux4 = nullptr;        // {v_star, mass} per particle for the tiled DFSPH passes
    // IISPH step
    void *pressure_alt = nullptr;  // the sort and the relaxed-Jacobi pressure update write here, then swap
    void *dii = nullptr;           // tight 3-vectors
    void *dijpjl = nullptr;        // tight 3-vectors
    void *pressureForce = nullptr; // tight 3-vectors
    void *aii = nullptr;
    void *density_star = nullptr;
    int *groupSorted = nullptr;    // group[] gathered into sorted order after each sort
    void *vFrac = nullptr;         // typenum volume fractions per particle, uniqueId order (set on demand)
    void *densArr = nullptr;       // typenum phase densities
    void *pd = nullptr;      // {pressure / density^2, density} pairs for the tuned force pass
    // neighbour search
    int *hash = nullptr;         // cell of particle i (device order before the sort)
    int *slot = nullptr;         // arrival slot inside its cell / rank among out-of-grid particles of its block
    int *tmpIndex = nullptr;     // cell-grouped, unordered inside a cell
    int *tmpHash = nullptr;
    int *sortedIndex = nullptr;  // stable permutation ("particleIndex" after sortParticle)
    int *sortedHash = nullptr;   // "particleHash" after sortParticle
    int *cellCount = nullptr;    // num_cells
    int *cellStart = nullptr;    // num_cells + 1, exclusive prefix of cellCount; [num_cells] = particles in grid
    int *gridCellStart = nullptr;  // reference format (0xffffffff for an empty cell)
    int *gridCellEnd = nullptr;
    int *blockRows = nullptr;    // row table of the tiled passes, rebuilt by every reorder (fp32)
    int *bigBlocks = nullptr;    // count, then ids of the blocks that need the large tiles
    int *blockLo = nullptr;      // cell of the first / last particle of every block of kTunedBS sorted positions
    int *blockHi = nullptr;
    unsigned short *nbList = nullptr;   // recorded neighbour lists (allocated on first use), see fsa_tile.h
    unsigned short *nbCount = nullptr;  // entries per particle
    void *hitWords = nullptr;           // recorded hits of the WCSPH passes (fp32), see fsa_tile.h
    unsigned char *hitCount = nullptr;  // words per particle
    unsigned char *wallCount = nullptr; // wall words per particle
    int *wallList = nullptr;            // [0] = count, then the particles that have wall words
    void *wallForce = nullptr;          // float4 per particle: the wall pass's sum (read where wallCount > 0)
    void *applyRec = nullptr;           // float4 per particle: what applyPStiff reads of a neighbour while the hit record is in use
    int *scanTmp = nullptr;
    int *undefBlockCount = nullptr;
    int *undefBlockStart = nullptr;
    // attributes that never change on this path, stored once in uniqueId order
    void *color = nullptr;   // 4 reals
    void *normal = nullptr;  // 3 reals
    int *group = nullptr;
    int *indexTable = nullptr;
    // z-slab runs only: order key of every particle (two copies like the hot set), pack counters
    long long *orderKey[2] = {nullptr, nullptr};
    int *slabCounters = nullptr;
    int *slabVars = nullptr;     // particle count and exchange counts of a slab group step (fsa_slab.hip)
};

struct TimedLaunch {
    int kernel_id;
    hipEvent_t start, stop;
};

struct Solver {
    int precision = FSA_PRECISION_F32;
    int device = 0;
    int capacity = 0;
    int n = 0;
    int num_cells = 0;
    int cur = 0;
    fsa_params hp{};  // as given by the caller
    DevParams<float> pf{};
    DevParams<double> pd{};
    Buffers b;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // call-order state
    bool hash_valid = false;    // hash/slot/cellCount describe the current arrays
    bool hash_ahead = false;    // the same, computed by the last step's integration pass (fsa_step); not what
                                // the reference's particleHash holds after Advect, so read-back ignores it
    bool count_stale = false;   // group steps have run since `n` was last read from the device (pf.count set)
    bool counts_clean = true;   // cellCount[] is all zero (as allocated; the fused scan leaves it so)
    bool perm_valid = false;    // sortedIndex/sortedHash/cellStart computed from hash
    bool sorted_valid = false;  // current arrays are in sorted order and cellStart matches them
    bool index_table_valid = false;
    bool count_valid = false;   // cellStart[num_cells] holds the in-grid count of the last sort
    bool ref_cells_valid = false;
    bool pd_valid = false;  // pd[] matches density / pressure / mass of the current arrays ...
    bool pd_fast = false;   // ... in the fast flavour's layout
    bool use_lists = true;      // passes that repeat between two sorts replay recorded neighbour lists
    bool use_replay = true;     // ... or, where the pass allows it, the hit record of the variant-3 density pass
                                // (hit_replay_kernel in fsa_pass.h); FSA_HIT_REPLAY=0, read once by fsa_create, turns it off
    bool lists_valid = false;   // nbList / nbCount describe the current arrays and positions
    bool hits_valid = false;    // hitWords / hitCount describe the current arrays and positions
    int wall_mode = 0;             // where the variant-3 passes sum the wall neighbours: 0 by size (fsa_sph_fast.hip), 1 in
                                   // the density pass's tail, 2 in a pass of their own; FSA_WALL_PASS=0 / 1, read once by
                                   // fsa_create, forces the tail / the pass
    bool wall_listed = false;      // the last record listed its wall particles: the wall pass has to run
    bool wall_list_clean = false;  // wallList[0] is zero (the variant-3 force pass resets it once the wall pass has run)
    bool has_rigid = false;     // some particle is TYPE_RIGID: the density pass needs the rigid term
    int typenum = 0;            // columns of vFrac (0: no volume fractions given)
    bool iisph = false;         // the sort also carries pressure[] and refreshes groupSorted[]
    bool slab_carry = false;    // slab solver set up for the IISPH step: pack / unpack carry pressure[] (fsa_slab_enable_iisph)
    bool slab_carry_vstar = false;  // z-slab solver prepared for the DFSPH step: v_star travels with the particles
    int table_n = 0;            // ... and its group / normal tables cover this many (global) unique ids
    int iisph_chain = 0;        // inside fsa_iisph_solve_pressure: 1 first round, 2 later rounds (fsa_iisph.hip)
    bool fluid_mass_uniform = false;  // fp32: every TYPE_FLUID particle has the same mass (checked where the host hands the masses over)
    int dfsph_chain = 0;        // inside the two DFSPH correction loops: 1 before the first pass, 2 after (fsa_dfsph.hip)
    bool slab = false;             // holds a z-slab of the grid: order keys are maintained
    bool owned_from_sort = false;  // sortedHash describes the current arrays (tells owned from halo copies)
    int force_zone = 0;  // variant-3 force pass: 0 all blocks, 1 only those that touch an edge (DevParams::edge), 2 the rest
    const int *slab_slots = nullptr;  // member of a slab group: number of slots the arrays hold before the sort (the
                                      // sort drops the particles that left; device memory, slabVars[kVarSlots])
    int variant = 3;  // 0 generic, 1 tuned structure + exact arithmetic, 2 tuned structure + fast fp32,
                      // 3 as 2 with one neighbour search per step (fsa_sph_fast.hip; same bits as 2)
    // number of particles the launch grids must cover: n, or the capacity while n lives on the device
    int grid_n() const { return pf.count ? capacity : n; }
    int xcd_run_log = 6;  // log2 of the run of consecutive blocks handed to one XCD (fsa_tile.h); FSA_XCD_RUN_LOG,
                          // read once by fsa_create, overrides it for experiments
    // timing experiments (debug_time_pass): parts of the tuned passes switched off, results to scratch arrays
    int dbg_ablate = 0;
    void *dbg_out0 = nullptr, *dbg_out1 = nullptr;
    int dbg_fast = 0;     // FSA_FAST_DBG, likewise read once: experiment flags of the variant-3 kernels
    // timing
    bool timing = false;
    std::vector<std::string> kernel_names;
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> event_pool;

    size_t real_size() const { return precision == FSA_PRECISION_F64 ? sizeof(double) : sizeof(float); }
};

// slabVars[] of a z-slab solver in a group (device memory): the particle count and the counts of the last exchange
enum { kVarN = 0, kVarKept, kVarOutLeft, kVarOutRight, kVarLost, kVarError, kVarInLeft, kVarInRight, kVarSlots, kVarCount };

// Every place where the library makes the host wait for the device goes through these, which count the waits
// (fsa_debug_host_waits: tests assert that a slab group step makes none).
hipError_t counted_stream_sync(hipStream_t st);
hipError_t counted_device_sync();
hipError_t counted_event_sync(hipEvent_t ev);  // counted apart, and only if the event had not completed yet

// Sets the calling thread's error message and returns `code`.
int fail(int code, const char *fmt, ...);
// Argument check of a caller's parameter block, and its fp32 device form (fsa_capi.hip).
int check_params(const fsa_params *p);
void make_dev_params(const fsa_params &h, DevParams<float> &d);

#define FSA_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::fsa::fail(FSA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                                 \
    } while (0)

// Brackets one kernel launch with an event pair when timing is on.
struct ScopedTimer {
    Solver *s;
    hipEvent_t stop = nullptr;
    ScopedTimer(Solver *s, const char *name);
    ~ScopedTimer();
};

// ---- launchers (fsa_grid.hip, fsa_sph.hip) ---------------------------------
int launch_hash_count(Solver *s);
int launch_hash_fused(Solver *s, bool with_advect);
int launch_sort_fused(Solver *s, bool vstar_all);
int launch_scan_cells(Solver *s);
int launch_scatter(Solver *s);
int launch_rank(Solver *s);
int launch_reorder(Solver *s);
int launch_ref_cells(Solver *s);
int launch_index_table(Solver *s);
int launch_density(Solver *s);
int launch_force(Solver *s);
int launch_advect(Solver *s);
int launch_rigid_volume(Solver *s);
int launch_density_alpha(Solver *s);
int launch_non_pressure_force(Solver *s);
int launch_solve_density_stiff(Solver *s);
int launch_solve_divergence_stiff(Solver *s);
int launch_apply_pstiff(Solver *s);
int launch_update_position(Solver *s);
int launch_update_velocities(Solver *s);
int launch_max_error(Solver *s, double start, double *out);  // synchronises
int launch_iisph_carry(Solver *s, int uid_buffer);  // part of the reorder when s->iisph
int launch_effective_mass(Solver *s);
int launch_npf_multiphase(Solver *s);
int launch_iisph_factor(Solver *s);
int launch_iisph_predict_density(Solver *s);
int launch_iisph_dijpjl(Solver *s);
int launch_iisph_new_pressure(Solver *s);
int launch_iisph_pressure_force(Solver *s);
int launch_iisph_integrate(Solver *s);
int launch_mean_error(Solver *s, double *out);  // synchronises
int launch_dfsph_loop_begin(Solver *s);  // the DFSPH correction loops with their stop test on the device ...
int launch_dfsph_round_end(Solver *s, double start, double ethres);  // ... maximum of error[] and the test
void dfsph_loop_end(Solver *s);
int launch_dfsph_loop_result(Solver *s, double start, int *iterations, double *last_error);  // synchronises once
int launch_iisph_rounds_begin(Solver *s);  // the pressure rounds with their stop test on the device ...
int launch_iisph_round_end(Solver *s);     // ... mean of error[] and the test, after each round
int launch_iisph_rounds_finish(Solver *s, int launched, int *done, double *mean);  // synchronises once
int ensure_lists(Solver *s);  // builds the neighbour lists if they are enabled and stale
int launch_density_tuned(Solver *s);
int launch_density_fast(Solver *s);
int launch_force_fast(Solver *s);
int ensure_hit_record(Solver *s);  // records the hits of the sorted state if the record is stale (no density written)
int launch_fused_probe(Solver *s, int mode);  // measurement only (fsa_sph_fast.hip)
int launch_density_rigid_tuned(Solver *s);
int launch_force_tuned(Solver *s);
int debug_time_pass(Solver *s, int which, int ablate, int reps);
int slab_record_bytes(const Solver *s);
int launch_slab_init_keys(Solver *s);
int launch_rank_key(Solver *s);
int launch_slab_keys(Solver *s);
int launch_slab_pack(Solver *s, const int own[2], const int keep[2], const int left[2], const int right[2],
                     void *sendLeft, int capLeft, void *sendRight, int capRight, int counts[4],
                     void *extraLeft = nullptr, void *extraRight = nullptr);
int launch_slab_unpack(Solver *s, const void *recs, int count, const void *extras = nullptr);
int launch_slab_layer_indices(Solver *s, int lo, int hi, int *d_idx, int cap, int *count);  // synchronises
int launch_field_move(Solver *s, bool scatter, void *field, int words, const int *d_idx, int count, void *d_buf);
int launch_owned_max(Solver *s, int lo, int hi, const void *a, double start, double *out);
int launch_owned_sum(Solver *s, int lo, int hi, const void *a, double *out);  // synchronises
int launch_slab_edge_send(Solver *s, bool with_advect, const int own[2], const int left[2], const int right[2],
                          void *sendLeft, int sizeLeft, void *sendRight, int sizeRight);
int launch_slab_pack_tail(Solver *s, void *stream, void *sendLeft, int sizeLeft, void *sendRight, int sizeRight,
                          int *mirror);
int launch_slab_keep_hash(Solver *s, bool with_advect, const int own[2], const int keep[2], const int left[2],
                          const int right[2]);
int launch_slab_unpack_hash(Solver *s, const void *recvLeft, int boundLeft, const void *recvRight, int boundRight,
                            int *mirror);
int launch_slab_export(Solver *s, const int own[2], void *recs, int cap, int *count);
int launch_pack(Solver *s, int n, const void *d_pos3, const void *d_vel3, const int *d_type, const void *d_mass);
int launch_unpack_vec3(Solver *s, const void *src4, void *dst3, int n);
int launch_unpack_w(Solver *s, const void *src4, void *dst, int n, bool as_int);
int launch_gather_by_uid(Solver *s, const void *table, void *dst, int n, int words);
int launch_set_xyz(Solver *s, void *dst4, const void *src3, int n);
int launch_set_w(Solver *s, void *dst4, const void *src, int n, bool as_int);

}  // namespace fsa
