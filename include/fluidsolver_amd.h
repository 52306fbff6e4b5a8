/*
 * fluidsolver_amd.h -- C ABI of the MI355X (gfx950) particle-solver step.
 *
 * This is the drop-in boundary: plain pointers and sizes, int status codes, an
 * opaque solver handle.  It replaces the operator seam of the reference,
 * particle_solver/solver/sph_solver.cuh:197-220 (CopyParam2Device, calcHash,
 * sortParticle, reorderDataAndFindCellStart, ComputePressure, computeForce,
 * Advect -- free functions taking the SimData_SPH struct of device pointers by
 * value), plus the cudaMalloc / cudaMemcpy calls the reference's solver
 * classes make inline (MultiphaseSPHSolver.cpp:10-27, 871-977;
 * sph_solver.cpp:35-38).  All citations are relative to
 * /root/reference/particle_solver.
 *
 * Host array layouts are the reference's: cfloat3 = 3 tightly packed reals,
 * cfloat4 = 4 (catpaw/geometry.h:58-60, 519-520).  With precision
 * FSA_PRECISION_F32 a "real" is a float; with FSA_PRECISION_F64 it is a double
 * (the reference has no fp64 path; that mode exists for validation).
 *
 * Every function returns FSA_OK (0) or a negative FSA_ERR_* code;
 * fsa_last_error() gives the message of the calling thread's last failure.
 * Nothing here synchronises the device unless its comment says so.  A solver
 * must be used from one host thread at a time.
 */
#ifndef FLUIDSOLVER_AMD_H
#define FLUIDSOLVER_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSA_OK 0
#define FSA_ERR_INVALID (-1)      /* bad argument / call order */
#define FSA_ERR_HIP (-2)          /* a HIP runtime call failed */
#define FSA_ERR_NOMEM (-3)
#define FSA_ERR_UNKNOWN_NAME (-4) /* buffer name not known */
#define FSA_ERR_CAPACITY (-5)

/* particle types, solver/particle_common.h:6-16 */
#define FSA_TYPE_FLUID 0
#define FSA_TYPE_CLOTH 1
#define FSA_TYPE_BOUNDARY 2
#define FSA_TYPE_EMITTER 3
#define FSA_TYPE_RIGID 4
#define FSA_TYPE_DEFORMABLE 5
#define FSA_TYPE_GRANULAR 6

/* solver/cuda_common.cuh:10-11 */
#define FSA_GRID_UNDEF 99999999

#define FSA_PRECISION_F32 0
#define FSA_PRECISION_F64 1

/* integration variants */
#define FSA_BOUNDARY_VERBATIM 0     /* AdvectKernel, solver/sph_kernel_impl.cuh:377-391 */
#define FSA_BOUNDARY_LEGACY_CLAMP 1 /* + skip TYPE_BOUNDARY and clamp to the grid box:
                                       advectAndCollision / clampBoundary of the older snapshot,
                                       solver/新建文件夹/sph_kernel_impl.cuh:365-403 */

/*
 * Hot-path fields of SimParam_SPH (solver/sph_solver.cuh:10-77), always in
 * double here; the solver rounds them to its precision.  Replaces hParam /
 * dParam (solver/sph_kernel_impl.cuh:13-14).
 */
typedef struct fsa_params {
    double dx;
    double gridxmin[3];
    double gridxmax[3];
    int gridres[3];
    double gravity[3];
    double dt;
    double viscosity;
    double spacing;
    double smoothradius;
    double kspikydiff;
    double kernel_cubic;
    double kernel_cubic_gradient;
    double pressureK;
    double boundary_visc;
    double acceleration_limit;
    int boundary_mode; /* FSA_BOUNDARY_* */
    /* z-slab owned by this solver, in global cell layers [slab_z0, slab_z0 + slab_nz),
     * ghost layers included; slab_nz = 0 means the whole grid. */
    int slab_z0;
    int slab_nz;
    double kpoly6; /* DFSPH density kernel constant (solver/MultiphaseSPHSolver.cpp:464) */
    /* IISPH non-pressure forces: <SurfaceTension> and <Friction> (solver/MultiphaseSPHSolver.cpp:393, 429) */
    double surface_tension;
    double boundary_friction;
} fsa_params;

typedef struct fsa_solver fsa_solver;

const char *fsa_last_error(void);
const char *fsa_version(void);

/*
 * Derived constants of MultiphaseSPHSolver::LoadParam
 * (solver/MultiphaseSPHSolver.cpp:381, 452-469): smoothradius, dx, gridres and
 * the kernel normalisations, evaluated as a C++ compiler evaluates the
 * reference's expressions for `float` (precision F32) or all in double (F64).
 * Other fields of *p are left untouched.
 */
int fsa_derive_params(fsa_params *p, int precision, double spacing, double smoothratio,
                      const double vmin[3], const double vmax[3]);

/*
 * Replaces SetupDeviceBuffer (MultiphaseSPHSolver.cpp:871-977) +
 * CopyParam2Device (solver/sph_solver.cu:18-20).  `capacity` = maximum number
 * of particles; `device` = HIP device ordinal.  Allocates the device buffers of
 * the WCSPH step for that capacity (about 480 bytes per particle in fp32, the
 * recorded hits of kernel variant 3 among them) and 20 bytes per grid cell; a
 * step allocates nothing.  Two things are allocated later, once, when first
 * used: the neighbour lists of the IISPH / DFSPH steps (fsa_set_neighbour_lists;
 * if that allocation fails the passes search instead) and the table of
 * fsa_set_volume_fractions.  Read-back and upload calls use temporary staging
 * buffers.
 */
int fsa_create(const fsa_params *p, int precision, int capacity, int device, fsa_solver **out);
int fsa_destroy(fsa_solver *s);

/* CopyParam2Device, solver/sph_solver.cu:18-20.  The grid resolution and slab may not change. */
int fsa_copy_param_to_device(fsa_solver *s, const fsa_params *p);

/* Use the caller's hipStream_t for all work of this solver (NULL = the solver's own stream). */
int fsa_set_stream(fsa_solver *s, void *hip_stream);
int fsa_synchronize(fsa_solver *s);

/*
 * Replaces CopyParticleDataToDevice (solver/sph_solver.cpp:10-19,
 * MultiphaseSPHSolver.cpp:10-27): host arrays in the reference layouts.
 * color, normal, group may be NULL (zeros).  uniqueId may be NULL (0..n-1);
 * otherwise it must be a permutation of 0..n-1 (the reference indexes
 * indexTable with it).  Synchronises.
 */
int fsa_upload(fsa_solver *s, int n, const void *pos3, const void *vel3, const void *color4,
               const void *normal3, const int *type, const int *group, const int *uniqueId,
               const void *mass, const void *restDensity);

int fsa_num_particles(fsa_solver *s, int *n);
/* number of particles whose hash was GRID_UNDEF at the last sort (synchronises) */
int fsa_num_undef(fsa_solver *s, int *n);

/* ---- the operator seam, one entry per reference launcher ---------------- */
/* calcHash, solver/sph_solver.cu:30-42 */
int fsa_calc_hash(fsa_solver *s);
/* sortParticle, solver/sph_solver.cu:46-53: afterwards "particleHash" is sorted ascending and
 * "particleIndex" is the stable permutation. */
int fsa_sort_particle(fsa_solver *s);
/* reorderDataAndFindCellStart, solver/sph_solver.cu:57-77, together with the sorted->primary
 * device copies of SPHSolver::SortParticles (solver/sph_solver.cpp:58-84), which become a
 * buffer swap: afterwards "pos", "vel", ... are in sorted order. */
int fsa_reorder_data_and_find_cell_start(fsa_solver *s);
/* SPHSolver::SortParticles (solver/sph_solver.cpp:50-85) in one call */
int fsa_sort_particles(fsa_solver *s);
/* ComputePressure / computeForce / Advect, solver/sph_solver.cu:80-108 */
int fsa_compute_pressure(fsa_solver *s);
int fsa_compute_force(fsa_solver *s);
int fsa_advect(fsa_solver *s);
/* RigidParticleVolume (solver/sph_kernel_impl.cuh:2423-2473): restDensity of TYPE_RIGID particles
 * = 1 / sum of W over rigid neighbours.  Needs a sorted state. */
int fsa_rigid_particle_volume(fsa_solver *s);

/* nsteps x SPHSolver::SolveSPH (solver/sph_solver.cpp:87-98) without the host read-back;
 * fully asynchronous on the solver's stream. */
int fsa_step(fsa_solver *s, int nsteps);

/*
 * CopyPosColorFromDevice (solver/sph_solver.cpp:35-38): positions and colours
 * in the current device order (the order of the last sort).  Either pointer
 * may be NULL.  Synchronises.
 */
int fsa_readback_pos_color(fsa_solver *s, void *pos3, void *color4, int *n);

/*
 * Debug / parity access to one named array in the reference's layout, current
 * device order.  Names: pos vel force v_star normal (n x 3 reals); color
 * (n x 4 reals); mass density pressure restDensity (n reals); type group
 * uniqueId indexTable particleHash particleIndex (n ints); gridCellStart
 * gridCellEnd (num_cells ints; gridCellEnd is only meaningful where
 * gridCellStart != 0xffffffff, as in the reference); cellStartDense
 * (num_cells + 1 ints, exclusive prefix of the per-cell counts); orderKey (n 64-bit ints, slab
 * solvers only); DF_factor pstiff error (n reals, DFSPH); the IISPH buffers listed further down;
 * neighbourCount (n ints, get only: entries of each particle's recorded neighbour list, -1 for a particle
 * that walks the cells instead; records the lists if they are stale).
 * `bytes` must be the exact size.  Both synchronise.
 */
int fsa_get_buffer(fsa_solver *s, const char *name, void *dst, size_t bytes);
/* settable: pos vel force v_star mass density pressure restDensity type DF_factor pstiff error */
int fsa_set_buffer(fsa_solver *s, const char *name, const void *src, size_t bytes);

/* ---- single-phase DFSPH step ------------------------------------------
 *
 * The other step of the reference's SPHSolver (solver/sph_solver.cpp:100-120), one entry per
 * launcher of solver/sph_solver.cu:119-256.  Extra named buffers for fsa_get_buffer /
 * fsa_set_buffer: DF_factor, pstiff, error (n reals each).  All need a sorted state except the two
 * integration kernels.  In slab solvers only after fsa_slab_enable_dfsph (further down).
 */
/* computeDensityAlpha: poly6 density and the DFSPH factor */
int fsa_compute_density_alpha(fsa_solver *s);
/* computeNonPForce: viscosity + wall penalty + gravity -> v_star */
int fsa_compute_non_pressure_force(fsa_solver *s);
/* the kernels the two correction loops launch: solveDensityStiff, solveDivergenceStiff, applyPStiff,
 * updatePosition, UpdateVelocities (solver/sph_kernel_impl.cuh:818-1018) */
int fsa_solve_density_stiff(fsa_solver *s);
int fsa_solve_divergence_stiff(fsa_solver *s);
int fsa_apply_pstiff(fsa_solver *s);
int fsa_update_position(fsa_solver *s);
int fsa_update_velocities(fsa_solver *s);
/* correctDensityError / correctDivergenceError (solver/sph_solver.cu:144-256): Jacobi rounds until the
 * maximum of error[] drops below ethres or maxiter rounds are done, then updatePosition /
 * UpdateVelocities.  All rounds are queued at once; the maximum and the test against the threshold run on the
 * device (the reference copies error[] to the host every round), and the passes of the rounds after the stop
 * return at once.  *iterations = rounds of applyPStiff done, *last_error = last maximum; either pointer may be
 * NULL, and only a call that asks for one of them waits for the device (one read-back). */
int fsa_correct_density_error(fsa_solver *s, int maxiter, double ethres, int *iterations, double *last_error);
int fsa_correct_divergence_error(fsa_solver *s, int maxiter, double ethres, int *iterations, double *last_error);
/* SetupDFSPH after the scene is uploaded (solver/sph_solver.cpp:102-106): sort + computeDensityAlpha */
int fsa_setup_dfsph(fsa_solver *s);
/* nsteps x SPHSolver::SolveDFSPH (solver/sph_solver.cpp:108-120, 5 rounds, threshold 1) without the
 * host read-back of positions */
int fsa_step_dfsph(fsa_solver *s, int nsteps);

/* ---- IISPH step (fluid and rigid particles) ----------------------------
 *
 * The step MultiphaseSPHSolver::Step runs by default (SolveIISPH, solver/MultiphaseSPHSolver.cpp:173-188),
 * one entry per launcher of solver/sph_solver.cu:300-310 and :1388-1457.  Covered particle types:
 * TYPE_FLUID and TYPE_RIGID (deformable / cloth bodies and their kernels are not).  Extra named buffers
 * for fsa_get_buffer / fsa_set_buffer: dii dijpjl pressureForce (n x 3 reals), aii density_star (n reals).
 * From the first of these calls after an upload on, every sort also carries pressure[] (the
 * reference's sorted_pressure copy).  All need a sorted state.  Not available in slab solvers.
 *
 * density[] of TYPE_RIGID particles is never written by this step but read by the surface-tension term
 * for a wall in the fluid's group (solver/sph_kernel_impl.cuh:1173, 1214-1232); the reference leaves it
 * uninitialised, here it is whatever fsa_upload / fsa_set_buffer left (zero after an upload).
 */
/* Volume fractions for EffectiveMass: typenum (1..10) fractions per particle, rows in the current device
 * order (n x typenum reals), and <DensityArray>.  Kept per particle identity until the next fsa_upload; this
 * call allocates the table (see fsa_create for what else is allocated after it).  Synchronises. */
int fsa_set_volume_fractions(fsa_solver *s, int typenum, const double *dens_arr, const void *vfrac);
/* EffectiveMass (solver/sph_solver.cu:288-298, kernel solver/sph_kernel_impl.cuh:1722-1755): mass =
 * rest density x spacing^3, rest density = sum of fraction x phase density, colour = the first three
 * fractions, for every particle that is not TYPE_RIGID.  With fixed fractions this only re-rounds what the
 * scene setup computed; SetupIISPH runs it once. */
int fsa_effective_mass(fsa_solver *s);
/* NonPressureForce_Multiphase: XSPH viscosity, surface tension, wall viscosity + friction, gravity -> v_star */
int fsa_non_pressure_force_multiphase(fsa_solver *s);
/* IISPHFactor: cubic-spline density and d_ii */
int fsa_iisph_factor(fsa_solver *s);
/* IISPHPredictDensity: pressure /= 2 (warm start), density_star, a_ii */
int fsa_iisph_predict_density(fsa_solver *s);
/* the two kernels of one relaxation round (CalcDIJPJLKernel, CalcNewPressureKernel).  The reference's
 * CalcNewPressureKernel updates pressure[] in place while neighbours read it (a race); here every
 * particle reads the previous round's values. */
int fsa_iisph_calc_dijpjl(fsa_solver *s);
int fsa_iisph_calc_new_pressure(fsa_solver *s);
/* IISPHSolvePressure (solver/sph_solver.cu:1409-1443, maxiter = 10 there): rounds until the mean of error[]
 * is below 1e-6 in magnitude or changes by less than 1e-6.  The mean is reduced on the device in double (the
 * reference adds the array up on the host in one float, in index order; since the second test fires when the
 * mean stalls within float rounding, the two can differ in the number of rounds).  All rounds are queued at
 * once and the stop test runs on the device; one read-back at the end of the solve, so it synchronises.  *rounds = rounds run, *last_mean = last mean; either pointer may be NULL. */
int fsa_iisph_solve_pressure(fsa_solver *s, int maxiter, int *rounds, double *last_mean);
/* CalcPressureForceKernel alone */
int fsa_iisph_calc_pressure_force(fsa_solver *s);
/* IISPHUpdate: CalcPressureForceKernel + IISPHUpdateKernel (vel = v_star + F dt / m, pos += vel dt) */
int fsa_iisph_update(fsa_solver *s);
/* SetupIISPH after the scene is uploaded (solver/MultiphaseSPHSolver.cpp:850-860): sort, IISPHFactor,
 * RigidParticleVolume (EffectiveMass, which sits between the sort and IISPHFactor in the reference, is the
 * caller's to run first when it has volume fractions: fsa_sort_particles, fsa_effective_mass, then this;
 * InitializeDeformable acts on deformable solids only) */
int fsa_setup_iisph(fsa_solver *s);
/* nsteps x SolveIISPH without the host read-back of positions */
int fsa_step_iisph(fsa_solver *s, int nsteps);

/* ---- counting sort without any allocation --------------------------------
 *
 * No counterpart in the reference.  For callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc -- this is fsa_calc_hash + fsa_sort_particle
 * (calcHash + sortParticle, solver/sph_solver.cu:30-53) on the caller's own device arrays, fp32: d_pos3 holds
 * n positions in the cfloat3 layout; afterwards d_particleHash is sorted ascending (GRID_UNDEF last) and
 * d_particleIndex is the stable permutation, bit for bit what the two solver calls give.  The grid comes from
 * *p (dx, gridxmin/max, gridres, slab_z0/slab_nz).  Work is queued on hip_stream (NULL = the default stream)
 * of `device`; nothing is synchronised.
 *
 * The call allocates nothing: the cell counts, their prefix sums and the cell-grouped intermediate order
 * live in a fixed device array inside the library (about 17 MB, held on every GPU the library runs on until
 * the process ends).  Hence the limits: n <= FSA_SORT_STATIC_MAX_PARTICLES and gridres[0] * gridres[1] *
 * layers <= FSA_SORT_STATIC_MAX_CELLS, beyond which the call fails with FSA_ERR_CAPACITY.  All calls on one
 * GPU share that array: calls that overlap on one GPU (two streams, two threads) race.
 *
 * The first call on a GPU looks up the array's device address (hipGetSymbolAddress, which may load the
 * library's code on that GPU): make that first call outside a stream capture; every later call on that GPU is
 * kernel launches and one hipMemsetAsync only and can be captured.  The calling thread's current device is
 * left as it was.  The rank inside a cell compares every pair of its particles (the solver's own rank pass
 * does the same): a cell crowded with k particles costs k^2 comparisons, so positions that pile thousands of
 * particles into one cell make the call slow (the solvers' scenes hold 8 to 30 per cell).
 */
#define FSA_SORT_STATIC_MAX_PARTICLES 1048576
#define FSA_SORT_STATIC_MAX_CELLS 1048576
int fsa_sort_particle_static(const fsa_params *p, int device, void *hip_stream, int n, const void *d_pos3,
                             int *d_particleHash, int *d_particleIndex);

/* ---- measurement ------------------------------------------------------ */
#define FSA_KERNEL_NAME_MAX 32
typedef struct fsa_kernel_time {
    char name[FSA_KERNEL_NAME_MAX];
    int launches;
    double total_ms;
} fsa_kernel_time;
/* Record a HIP event pair around every kernel launched on the solver's stream. */
int fsa_timing_enable(fsa_solver *s, int on);
int fsa_timing_reset(fsa_solver *s);
/* Synchronises; fills up to `cap` entries, *count = number of distinct kernels seen. */
int fsa_timing_read(fsa_solver *s, fsa_kernel_time *out, int cap, int *count);

/* Kernel variants of the density and force passes (fp32; fp64 always runs the generic kernels):
 *   0  generic: one thread per particle reading neighbours from global memory, arithmetic
 *      operation for operation as the reference's expressions;
 *   1  tuned structure (LDS row tiles + per-lane hit queues), same exact arithmetic;
 *   2  tuned structure, fast fp32 arithmetic (hardware sqrt/rsq/rcp, fused multiply-adds): about
 *      1e-7 relative away from variant 1 per neighbour term; TYPE_BOUNDARY neighbours of the force pass in
 *      fp32 with reciprocals as well.  The IISPH passes (all but the non-pressure force) and the three kernels
 *      of the DFSPH correction loops have such a flavour too; the remaining DFSPH and IISPH passes run the
 *      exact arithmetic in every variant;
 *   3  as 2, but the WCSPH step searches its neighbours once: the density pass also records one bit per
 *      candidate and the force pass replays the set bits from row tiles that hold everything a pair needs.
 *      Same arithmetic in the same order: bit for bit the results of variant 2.  Default.  Uses 256 bytes of
 *      device memory per particle of the solver's capacity (allocated by fsa_create). */
int fsa_set_variant(fsa_solver *s, int variant);

/* Recorded neighbour lists (fp32, variants 1 to 3; no counterpart in the reference, which walks the 27
 * cells in every kernel).  The IISPH and DFSPH steps run 20-odd neighbour passes between two sorts over the
 * same positions: with lists enabled (the default) the first such pass after a sort records, per particle
 * and in the reference's neighbour order, every particle within the support radius, and the passes replay
 * that list instead of searching again.  Same pair arithmetic in the same order: results are bit for bit
 * those of the searching kernels.  Costs 128 bytes of device memory per particle of the solver's capacity,
 * allocated when first used.  The WCSPH step (two passes per sort) does not use them.
 *
 * On kernel variant 3 the passes of the DFSPH step replay the hit record
 * of the variant-3 density pass instead (one bit per candidate of each neighbour row, made once per sort
 * without computing a density; the neighbours come from row tiles in LDS as in the WCSPH force pass) and no
 * list is built: same pair arithmetic in the same order.  The record marks a neighbour on the fused squared
 * distance and the list on the same with a slack of 2e-6, so a pair whose two roundings of the distance
 * straddle the support radius (its term is about 1e-14 of the others) is the only thing the two can differ in.
 *
 * enabled: 0 = every pass searches, 1 = lists, and the hit record where it applies (default), 2 = lists only. */
int fsa_set_neighbour_lists(fsa_solver *s, int enabled);

/* ---- z-slab decomposition: one solver per GPU ---------------------------
 *
 * The reference runs on one device; these entry points have no counterpart in it.  They split the
 * step of SPHSolver::SolveSPH (solver/sph_solver.cpp:87-98) over several solvers, each created with
 * fsa_params.slab_z0 / slab_nz naming the global cell layers it holds (its own layers plus halo
 * layers that mirror the neighbouring slabs), so that every solver reproduces, for the particles it
 * owns, exactly the neighbour sums the single-device run computes: ties inside a cell are broken by a
 * 64-bit order key (global cell at the last sort, rank inside that cell) that travels with each
 * particle and orders particles as the previous step's global array did.  Before the first sort the
 * key is the particle's uniqueId, so a slab run matches a single-device run whose uniqueId is
 * 0..n-1 in upload order.
 *
 * A slab solver does not store color / normal / group / indexTable (uniqueId is a global id there):
 * fsa_upload wants NULL for them and skips the permutation check.  A particle outside the grid is nobody's
 * neighbour and is never a halo copy, but its own sums run over the cells around it: it is filed under the cell
 * layer of its z coordinate, clamped into the grid, lives with the slab that owns that layer and moves on, one slab
 * per step, when it drifts past a cut (a driver hands such particles out by the same rule at the start).  The order
 * of out-of-grid particles at the tail of the arrays is unspecified in a slab run.
 *
 * Per step, between fsa_advect and the next fsa_calc_hash:
 *   fsa_slab_pack    -> send the two record buffers to the neighbours (RCCL send/recv, the caller's job)
 *   fsa_slab_unpack  for each buffer received.
 */
/* Size of one exchange record: 48 bytes (fp32) / 96 bytes (fp64). */
int fsa_slab_record_bytes(fsa_solver *s, int *bytes);
/*
 * Classifies the particles this solver OWNED at the last sort (cell layer in [own[0], own[1]) then, or
 * outside the grid; all particles if no sort has run since the upload) by the cell layer they are in
 * now.  Those in [keep[0], keep[1]) remain; those in
 * [left[0], left[1]) / [right[0], right[1]) are written as records to the device buffers
 * d_send_left / d_send_right (cap_* records each); a particle may do several of these.  A particle outside
 * the grid does exactly one of them, by the layer it is filed under (above) and the layers this solver owns
 * next, which follow from the three ranges.  Everything
 * else the solver held (last step's halo copies) is dropped.  counts = {remaining, records left,
 * records right, lost}: `lost` counts owned particles that fit none of the ranges.  Synchronises.
 */
int fsa_slab_pack(fsa_solver *s, const int own[2], const int keep[2], const int left[2], const int right[2],
                  void *d_send_left, int cap_left, void *d_send_right, int cap_right, int counts[4]);
/*
 * Tells the solver which of the layers it holds it owns, [own[0], own[1]).  From then on the density
 * pass only covers particles within one layer of that range and the force pass the range itself (the
 * halo beyond feeds nobody); density / pressure / force of the other particles held are unspecified.
 */
int fsa_slab_set_owned(fsa_solver *s, const int own[2]);
/*
 * Writes every particle this solver owned at the last sort (out-of-grid ones included; all particles
 * if no sort has run) as a record to d_records without changing the solver: what a driver needs to
 * re-cut the slabs (it then builds new solvers and hands each its particles with fsa_slab_unpack).
 * Synchronises.
 */
int fsa_slab_export_owned(fsa_solver *s, const int own[2], void *d_records, int cap, int *count);
/* Appends `count` records from device memory (written by a neighbour's fsa_slab_pack). */
int fsa_slab_unpack(fsa_solver *s, const void *d_records, int count);

/* ---- the IISPH step on z-slabs ------------------------------------------------------------------------------
 *
 * MultiphaseSPHSolver::SolveIISPH (solver/MultiphaseSPHSolver.cpp:173-188; the pressure rounds:
 * solver/sph_solver.cu:1409-1443) runs some 25 neighbour passes between two sorts, each reading what the pass
 * before wrote at the neighbours: a halo of copies goes stale after every pass.  On slabs the driver therefore
 * refreshes, after each pass, the fields that pass wrote for the particles in the halo layers from the slab that
 * owns them, with the calls below (no counterpart in the reference, which runs on one device).  The sorted order
 * of a slab solver is (y, layer, x, order key), so two solvers that both hold a range of layers hold its in-grid
 * particles in the same order: fsa_slab_layer_indices lists their sorted positions, the owner gathers a field
 * through its list, the neighbour scatters the buffer through its own.  The stop test of the pressure rounds needs
 * the mean of error[] over the whole scene: fsa_slab_owned_sum gives a slab's part of the sum.
 */
/* Prepares a slab solver for the fsa_iisph_* calls: allocates the arrays of that step, makes fsa_slab_pack_carry /
 * fsa_slab_unpack_carry available, and hands over group[] and normal[] (cfloat3 layout, in the solver's precision) of
 * the WHOLE scene, indexed by the global unique ids the solver's particles carry (n_global of them). */
int fsa_slab_enable_iisph(fsa_solver *s, int n_global, const int *group, const void *normal3);
/* fsa_slab_pack / fsa_slab_unpack that also carry pressure[] (the relaxation of a step starts from the pressures
 * of the step before): the pressure of each record goes to d_extra_left / d_extra_right (cap_left / cap_right
 * reals), slot for slot beside the records, and comes in through d_extras likewise. */
int fsa_slab_pack_carry(fsa_solver *s, const int own[2], const int keep[2], const int left[2], const int right[2],
                        void *d_send_left, void *d_extra_left, int cap_left, void *d_send_right, void *d_extra_right,
                        int cap_right, int counts[4]);
int fsa_slab_unpack_carry(fsa_solver *s, const void *d_records, const void *d_extras, int count);
/* Sorted positions (ascending) of the in-grid particles in the global cell layers [layer_lo, layer_hi), which
 * must be layers this solver holds, into the device array d_idx (cap ints); needs a sorted state.  Synchronises. */
int fsa_slab_layer_indices(fsa_solver *s, int layer_lo, int layer_hi, int *d_idx, int cap, int *count);
/* d_out[e] = field[d_idx[e]] / field[d_idx[e]] = d_in[e], device buffers; fields: density, pressure, restDensity,
 * error, pstiff, DF_factor (one real per particle), v_star, dii, dijpjl (three).  Queued on the solver's stream. */
int fsa_slab_gather_field(fsa_solver *s, const char *name, const int *d_idx, int count, void *d_out);
int fsa_slab_scatter_field(fsa_solver *s, const char *name, const int *d_idx, int count, const void *d_in);
/* Sum (in double) of a scalar field over the particles this solver owns: cell layer in [own[0], own[1]) at the
 * last sort, or outside the grid.  Synchronises. */
int fsa_slab_owned_sum(fsa_solver *s, const char *name, const int own[2], double *sum);
/* ... and the maximum of `start` and the field's values over the same particles.  Synchronises. */
int fsa_slab_owned_max(fsa_solver *s, const char *name, const int own[2], double start, double *max);

/* The DFSPH step on z-slabs (SPHSolver::SolveDFSPH, solver/sph_solver.cpp:108-120, for a scene cut along z).  As for
 * the IISPH step, the fields a neighbour pass wrote are refreshed in the halo layers after the pass with
 * fsa_slab_layer_indices / fsa_slab_gather_field / fsa_slab_scatter_field: v_star after computeNonPForce and after
 * every applyPStiff, pstiff after every solveDensityStiff / solveDivergenceStiff, density and DF_factor after
 * computeDensityAlpha.  fsa_slab_enable_dfsph opens the DFSPH stage calls to a slab solver and makes v_star travel
 * with the particles: the sort reorders it for every particle, and fsa_slab_pack_carry / fsa_slab_unpack_carry carry
 * it beside the records (three reals per record; the divergence loop after the sort goes on from it).  The stop
 * test of a correction loop is the maximum of error[] over the particles every slab owns (fsa_slab_owned_max, then
 * the maximum over the slabs).  fluidsolver_amd/slabs.py (DfsphSlabs) drives it; a solver is set up for one of the
 * two steps, not both. */
int fsa_slab_enable_dfsph(fsa_solver *s);
/* ... and for the slabs of one process the library drives it, as fsa_slab_iisph_* does the IISPH step (further
 * down): solvers prepared by fsa_slab_enable_dfsph, `maxiter` rounds per correction loop at most, threshold `ethres`
 * (the reference: 5 and 1).  setup = SetupDFSPH (hands out the halo copies, sort, computeDensityAlpha).  stats waits
 * for the members' streams; density_rounds / divergence_rounds (either may be NULL): applyPStiff passes of the two
 * loops in the last rounds_cap steps, oldest first. */
typedef struct fsa_slab_iisph fsa_slab_dfsph; /* the same kind of team, running the other step */
int fsa_slab_dfsph_create_local(int world, fsa_solver **solvers, const int *cuts, int halo, int record_cap, int maxiter,
                                double ethres, fsa_slab_dfsph **out);
int fsa_slab_dfsph_setup(fsa_slab_dfsph *t);
int fsa_slab_dfsph_step(fsa_slab_dfsph *t, int nsteps);
int fsa_slab_dfsph_stats(fsa_slab_dfsph *t, long long *steps, long long *lost, long long *records_sent, int *density_rounds,
                         int *divergence_rounds, int rounds_cap);
int fsa_slab_dfsph_destroy(fsa_slab_dfsph *t);

/* The whole of that for slabs that all belong to solvers of THIS process (one host program driving a node, as the
 * reference's main.cpp drives one GPU): replaces SetupIISPH / SolveIISPH (solver/MultiphaseSPHSolver.cpp:850-860,
 * 173-188) for a scene cut into z-slabs.  The solvers: fp32 or fp64 slab solvers prepared by fsa_slab_enable_iisph,
 * each holding the layers [cuts[r] - halo, cuts[r+1] + halo) (clipped to the grid) and, at this point, the particles
 * of the layers it owns, as uploaded (unique ids of the whole scene, n_global of them).  fsa_slab_iisph_setup hands
 * out the halo copies, sorts, and computes the factors and rigid particle volumes; fsa_slab_iisph_step runs whole
 * steps (each ends, like SolveIISPH, with the sort and the factors of the next).  Particle counts are kept on the
 * host: a step waits for the device a few dozen times (it is a dozen times longer than the WCSPH step).  Results for
 * every particle equal the single solver's bit for bit, with the same number of pressure rounds. */
typedef struct fsa_slab_iisph fsa_slab_iisph;
int fsa_slab_iisph_create_local(int world, fsa_solver **solvers, const int *cuts /* world + 1 */, int halo,
                                int record_cap, int n_global, fsa_slab_iisph **out);
int fsa_slab_iisph_setup(fsa_slab_iisph *t);
int fsa_slab_iisph_step(fsa_slab_iisph *t, int nsteps);
/* Waits for the queued work; any pointer may be NULL.  rounds[0 .. rounds_cap): pressure rounds of the last
 * rounds_cap steps, oldest first (0 where there was no such step). */
int fsa_slab_iisph_stats(fsa_slab_iisph *t, long long *steps, long long *lost, long long *records_sent, int *rounds,
                         int rounds_cap);
/* The solvers live on. */
int fsa_slab_iisph_destroy(fsa_slab_iisph *t);

/* ---- z-slab groups: the whole slab step, exchange included, without a host round trip ------------------
 *
 * Replaces the step loop of main.cpp:11-19 -> MultiphaseSPHSolver::Step (solver/MultiphaseSPHSolver.cpp:199-217)
 * for a scene cut into z-slabs.  A group drives, per step and per slab: append what the neighbours sent -> sort
 * (which drops the halo copies of the step before and the particles that left) -> density -> force ->
 * integration, fsa_slab_pack's classification and the records for the neighbours -> transport, all queued on the
 * solvers' streams; the particles that stay are never copied for the exchange.  The force pass handles the blocks
 * next to the cuts first, so that the records of the next step travel (on a stream of the group's own) while the
 * rest of the step is computed; FSA_SLAB_OVERLAP=0 in the environment when the group is made keeps everything on
 * the solver's stream.  The particle counts stay in device memory (they change with every exchange):
 * fsa_slab_group_step returns without waiting, and any other entry point called on a member solver first waits for
 * the queued steps.  Between two fsa_slab_group_step calls a member's particles may be read, not changed (upload,
 * parameter changes, the solver's own step): the next call then fails with FSA_ERR_INVALID.
 * Members must be fp32 slab solvers on kernel variant 3 whose scene has no TYPE_RIGID particles; each must hold
 * exactly the layers [cuts[r] - halo, cuts[r+1] + halo) (clipped to the grid) of its rank r.
 *
 *   local : all `world` slabs belong to solvers of THIS process, on one GPU or on several (one host program
 *           driving a node, as the reference's main.cpp drives one GPU); records move by device-to-device copies.
 *   rccl  : this process holds the slab of `rank`; its z-neighbours are other processes of the same
 *           communicator.  Records move by ncclSend / ncclRecv (RCCL; xGMI inside a node), both neighbours in
 *           one group call per step.  `id` comes from fsa_slab_group_unique_id on one rank and reaches the
 *           others by whatever the host program uses (torch.distributed, MPI, a file).
 *
 * Messages are sized from the counts of three steps earlier plus head room (record_cap = the most records one
 * message may ever carry).  If a count outgrows its message or a solver's capacity, the excess is dropped,
 * flagged on the device, and fsa_slab_group_sync (or the next entry point used on the solver) returns
 * FSA_ERR_CAPACITY: the state is then invalid.  The same happens when a particle crosses a whole cell layer
 * in one step next to a cut (only the particles of the halo + 1 layers next to a cut are looked at for sending).
 */
typedef struct fsa_slab_group fsa_slab_group;
#define FSA_SLAB_GROUP_ID_BYTES 128
typedef struct fsa_slab_group_stats {
    long long steps;         /* steps queued since the group was made */
    long long particles;     /* held by this process's members now (halo copies included) */
    long long max_particles; /* the largest member */
    long long lost;          /* particles that left every range they could go to (no neighbour holds their layer) */
    long long records_sent;  /* by this process's members, up to three steps ago */
    int error_bits;          /* 0, or what overflowed: 1 capacity, 2/4 message out left/right, 8/16 in left/right,
                                32 a particle crossed a whole cell layer in one step */
} fsa_slab_group_stats;
int fsa_slab_group_create_local(int world, fsa_solver **solvers, const int *cuts /* world + 1 */, int halo,
                                int record_cap, fsa_slab_group **out);
/* A caller-supplied way between the processes of a slab group or a grid team, for hosts without RCCL between them
 * (MPI, sockets ...) and
 * for several processes on one GPU, which RCCL refuses.  Both functions work on HOST memory and return 0 when the
 * data has arrived.  sendrecv: send `send_bytes` to rank `peer` and receive `recv_bytes` from it (every rank
 * calls it for the slab below first, then for the slab above, with the matching sizes on the other side).
 * allreduce: in place over all ranks; what = 0: sum of `count` doubles, 1: sum of floats, 2: maximum of ints. */
typedef struct fsa_host_transport {
    void *ctx;
    int (*sendrecv)(void *ctx, int peer, const void *send, size_t send_bytes, void *recv, size_t recv_bytes);
    int (*allreduce)(void *ctx, void *buf, size_t count, int what);
} fsa_host_transport;
int fsa_slab_group_unique_id(char id[FSA_SLAB_GROUP_ID_BYTES]);
int fsa_slab_group_create_rccl(fsa_solver *solver, int rank, int world, const char id[FSA_SLAB_GROUP_ID_BYTES],
                               const int *cuts /* world + 1 */, int halo, int record_cap, fsa_slab_group **out);
/* as fsa_slab_group_create_rccl, the records travelling through the caller's transport: every step waits for its
 * messages on the host (the one-slab-per-process step for hosts without RCCL between their processes, and for
 * several processes on one GPU) */
int fsa_slab_group_create_hosted(fsa_solver *solver, int rank, int world, const fsa_host_transport *via, const int *cuts,
                                 int halo, int record_cap, fsa_slab_group **out);
/* Queues nsteps steps; returns without waiting for them. */
int fsa_slab_group_step(fsa_slab_group *g, int nsteps);
/* Waits for everything queued, brings the member solvers' host-side counts up to date, fills *st (may be NULL). */
int fsa_slab_group_sync(fsa_slab_group *g, fsa_slab_group_stats *st);
/* Self test of the RCCL transport: sends `bytes` from d_src to this very rank's d_dst through the group's
 * communicator on the member's stream, and waits. */
int fsa_slab_group_rccl_loopback(fsa_slab_group *g, const void *d_src, void *d_dst, size_t bytes);
/* Waits, hands the counts back to the solvers (which live on), frees the group's buffers. */
int fsa_slab_group_destroy(fsa_slab_group *g);
/* Overrides the scan of `type` that fsa_upload does: whether a TYPE_RIGID particle can ever be held by
 * this solver (in a slab run one may arrive from a neighbour later). */
int fsa_set_scene_has_rigid(fsa_solver *s, int has_rigid);

/* ---- Eulerian grid solver ---------------------------------------------------
 *
 * The reference's other fluid solver: GridSolver (solver/grid_solver.h:52-108, solver/grid_solver.cpp), a
 * staggered-grid solver with semi-Lagrangian advection, gravity, pressure projection by conjugate
 * gradients and passive tracer particles, which the reference runs in host loops.  It is independent of the
 * particle solvers above and has its own handle.  One entry per method of the class; `precision` as for
 * fsa_create (the reference's `scalar` is float).
 *
 * Field names for fsa_grid_get_field / fsa_grid_set_field: u v w uadv vadv wadv (face-centred, padded, laid
 * out as Grid::uId / vId / wId index them, solver/grid_solver.h:39-49) and p divU b Aq r q (cell-centred,
 * dim.x * dim.y * dim.z, x fastest, Grid::cellId :36-38).
 */
typedef struct fsa_grid fsa_grid;
/* loadConfig + allocate (solver/grid_solver.cpp:14-66) with the values of config/Eulerian.xml passed in:
 * the low corner is (-dim.x/2*h, 0, -dim.z/2*h) as loadConfig sets it.  padding must be >= 1. */
int fsa_grid_create(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                    fsa_grid **out);
int fsa_grid_destroy(fsa_grid *g);
/* sizes = {uSize, vSize, wSize, dimSize} (Grid::setSize, solver/grid_solver.h:23-35); either may be NULL */
int fsa_grid_sizes(fsa_grid *g, int sizes[4], double xmin[3]);
int fsa_grid_synchronize(fsa_grid *g);
int fsa_grid_get_field(fsa_grid *g, const char *name, void *dst, size_t bytes); /* synchronises */
int fsa_grid_set_field(fsa_grid *g, const char *name, const void *src, size_t bytes);
/* tracer particles (addParticles :625-637 builds the reference's block; here the caller passes positions,
 * n x 3 reals; colours start as 1,1,1,1) */
int fsa_grid_set_particles(fsa_grid *g, int n, const void *pos3);
int fsa_grid_get_particles(fsa_grid *g, void *pos3, void *color4, int *n); /* synchronises */
/* advect :456-565 (u, v, w <- their advected copies), advectParticles :639-651, bodyForce :579-588,
 * testcase :590-606 */
int fsa_grid_advect(fsa_grid *g);
int fsa_grid_advect_particles(fsa_grid *g);
int fsa_grid_body_force(fsa_grid *g);
int fsa_grid_testcase(fsa_grid *g);
/* divVelocity :124-141: fills divU; *div_u_sum (may be NULL; asking for it synchronises) = sum of div^2 */
int fsa_grid_div_velocity(fsa_grid *g, double *div_u_sum);
/* makeRHS :143-160 (runs divVelocity first, as the reference does) */
int fsa_grid_make_rhs(fsa_grid *g);
/* solve :162-214: conjugate gradients on p from its current value until the mean |r| drops below 1e-6.
 * The reference's loop has no bound; maxiter >= 0 adds one (counted like its `iter`), maxiter < 0 keeps it
 * unbounded.  *iterations = final `iter`, *residual = last mean |r|; either may be NULL.  The dot products
 * are reduced in double on the device; the loop's scalars stay there, so the host only looks at the state
 * every 16 rounds.  A residual that is not a number ends the loop with FSA_ERR_INVALID (the reference would
 * spin forever).  Synchronises. */
int fsa_grid_solve(fsa_grid *g, int maxiter, int *iterations, double *residual);
/* updateU :216-266 */
int fsa_grid_update_u(fsa_grid *g);
/* mvproduct :72-115 and dotproduct :117-122 on named cell-centred fields */
int fsa_grid_mvproduct(fsa_grid *g, const char *src, const char *dst);
int fsa_grid_dotproduct(fsa_grid *g, const char *a, const char *b, double *out); /* synchronises */
/* nsteps x GridSolver::Step :608-619 */
int fsa_grid_step(fsa_grid *g, int nsteps, int maxiter);

/* ------------------------------------------------------------------------------------------------
 * The grid solver on z-slabs (no counterpart in the reference, whose GridSolver is one grid on one host:
 * solver/grid_solver.cpp).  The grid is cut into slabs of whole z planes, cuts[0] = 0 < ... < cuts[world] =
 * dim[2], one solver per slab.  A slab keeps copies of its neighbours' planes: `reach` + 1 planes of the face
 * arrays (refreshed after the advection and after the velocity update; `reach` is how many planes the
 * back-trace of the advection may cross: a sample that needs more raises an error bit that fsa_grid_team_sync
 * reports, and the state is then invalid), one plane of p and of the search direction q (refreshed before
 * every stencil of the conjugate-gradient loop).  The two dot products and the residual of a round and the
 * divergence sum are totals over all slabs, so every slab runs the same rounds.  Every slab needs at least
 * max(padding, reach + 1) + 1 planes; at most 16 slabs in one process.
 *
 * fsa_grid_team_create_local: all slabs in this process, on one GPU and one stream (refreshes are device
 * copies); fsa_grid_team_create_rccl: this process holds slab `rank` of `world` (one process per GPU), refreshes
 * are ncclSend / ncclRecv with the two z-neighbours and the totals an ncclAllReduce of two doubles, all queued
 * on the team's stream; `id` comes from fsa_slab_group_unique_id on rank 0.  Tracer arrays are whole on every
 * rank (each moves the tracers inside its planes; the rest arrives by an all-reduce with zeros).
 *
 * Fields are exchanged with the caller in the layout of the single solver (fsa_grid_get_field, the whole
 * grid): get fills the planes this process's slabs own and leaves the rest of dst alone (a local team fills
 * all of it), set takes what its slabs hold, copies of the neighbours' planes included.  Cell-centred results
 * and velocities equal the single solver's bit for bit as long as the totals round to the same number in the
 * solver's precision (they are added up in double in another order: always, in practice, for fp32; to 1e-10
 * for fp64).  The other calls are the team's forms of the fsa_grid_* calls of the same name.
 */
typedef struct fsa_grid_team fsa_grid_team;
int fsa_grid_team_create_local(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                               int world, const int *cuts, int reach, fsa_grid_team **out);
int fsa_grid_team_create_rccl(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                              int rank, int world, const char id[FSA_SLAB_GROUP_ID_BYTES], const int *cuts, int reach,
                              fsa_grid_team **out);
/* as fsa_grid_team_create_rccl, the bytes travelling through `via` (copied): every exchange is staged through pinned
 * host memory and waits for the stream */
int fsa_grid_team_create_hosted(const int dim[3], int padding, double h, double dt, double rho, int precision, int device,
                                int rank, int world, const fsa_host_transport *via, const int *cuts, int reach,
                                fsa_grid_team **out);
int fsa_grid_team_destroy(fsa_grid_team *t);
int fsa_grid_team_get_field(fsa_grid_team *t, const char *name, void *dst, size_t bytes); /* synchronises */
int fsa_grid_team_set_field(fsa_grid_team *t, const char *name, const void *src, size_t bytes);
int fsa_grid_team_set_particles(fsa_grid_team *t, int n, const void *pos3);
int fsa_grid_team_get_particles(fsa_grid_team *t, void *pos3, void *color4, int *n); /* synchronises */
int fsa_grid_team_advect(fsa_grid_team *t);
int fsa_grid_team_advect_particles(fsa_grid_team *t);
int fsa_grid_team_body_force(fsa_grid_team *t);
int fsa_grid_team_make_rhs(fsa_grid_team *t);
int fsa_grid_team_div_velocity(fsa_grid_team *t, double *div_u_sum);
int fsa_grid_team_solve(fsa_grid_team *t, int maxiter, int *iterations, double *residual);
int fsa_grid_team_update_u(fsa_grid_team *t);
int fsa_grid_team_step(fsa_grid_team *t, int nsteps, int maxiter);
/* waits for the team's stream; *error_bits (may be NULL): 1 = a sample needed a plane its slab does not hold (between
 * processes: on any rank; collective there, every rank calls it) */
int fsa_grid_team_sync(fsa_grid_team *t, int *error_bits);

#ifdef __cplusplus
}
#endif
#endif
