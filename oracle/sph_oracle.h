/*
 * sph_oracle.h -- CPU restatement of the reference's WCSPH particle-solver step.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared with
 * (and the timed "port" CPU baseline).  Nothing in the product links or calls
 * it; the product fails loudly when its HIP library is missing.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path, and its sources need the CUDA toolkit headers, Thrust and
 * MSVC extensions that this image does not have, so it cannot be compiled here
 * without writing stand-ins for them.  The restatement below follows the
 * reference text expression by expression (each function cites file:line under
 * /root/reference/particle_solver) and is anchored only by the few known
 * answers listed in tests/test_oracle_known_answers.py.
 *
 * Arithmetic conventions (what a C++ compiler does with the reference text):
 *   - `float` variables, but unsuffixed literals (0.25, 0.666667, 1.5, 0.01,
 *     0.02, 88.5, 0.66666, 9.8) promote the enclosing sub-expression to double;
 *     every such promotion is written out with explicit casts here.
 *   - sqrt/floor of a float resolve to the float overloads (sqrtf/floorf).
 *   - no FMA contraction (built with -ffp-contract=off).
 *
 * Two builds of the same source: real = float (symbols orc_*), and
 * -DORC_REAL_IS_DOUBLE, real = double (symbols orc64_*), in which every
 * float of the reference is widened to double (the reference has no fp64
 * path; that build only serves the fp64 GPU-vs-CPU check).
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORC_REAL_IS_DOUBLE
typedef double orc_real;
#define ORC_NAME(x) orc64_##x
#else
typedef float orc_real;
#define ORC_NAME(x) orc_##x
#endif

/* solver/particle_common.h:6-16 */
#define ORC_TYPE_FLUID 0
#define ORC_TYPE_CLOTH 1
#define ORC_TYPE_BOUNDARY 2
#define ORC_TYPE_EMITTER 3
#define ORC_TYPE_RIGID 4
#define ORC_TYPE_DEFORMABLE 5
#define ORC_TYPE_GRANULAR 6

/* solver/cuda_common.cuh:10-11 */
#define ORC_EPSILON 1e-10
#define ORC_GRID_UNDEF 99999999

/* advect variants */
#define ORC_BOUNDARY_VERBATIM 0      /* AdvectKernel, sph_kernel_impl.cuh:377-391 */
#define ORC_BOUNDARY_LEGACY_CLAMP 1  /* + skip TYPE_BOUNDARY and clampBoundary,
                                        solver/新建文件夹/sph_kernel_impl.cuh:365-403 */

/* Hot-path fields of SimParam_SPH (solver/sph_solver.cuh:10-77). */
typedef struct ORC_NAME(params) {
    orc_real dx;
    orc_real gridxmin[3];
    orc_real gridxmax[3];
    int gridres[3];
    orc_real gravity[3];
    orc_real dt;
    orc_real viscosity;
    orc_real spacing;
    orc_real smoothradius;
    orc_real kspikydiff;
    orc_real kernel_cubic;
    orc_real kernel_cubic_gradient;
    orc_real pressureK;
    orc_real boundary_visc;
    orc_real acceleration_limit;
    int boundary_mode;
    orc_real kpoly6; /* DFSPH density kernel constant, MultiphaseSPHSolver.cpp:464 */
    orc_real surface_tension;   /* <SurfaceTension>, MultiphaseSPHSolver.cpp:393 */
    orc_real boundary_friction; /* <Friction>, MultiphaseSPHSolver.cpp:429 */
} ORC_NAME(params);

/* Hot-path subset of SimData_SPH (solver/sph_solver.cuh:79-190): tight AoS
 * arrays, 3 reals per cfloat3, 4 per cfloat4, exactly as the reference. */
typedef struct ORC_NAME(state) {
    int n;
    int num_cells;
    orc_real *pos, *vel, *normal, *force, *v_star; /* n x 3 */
    orc_real *color;                                /* n x 4 */
    orc_real *mass, *density, *pressure, *restDensity;
    int *type, *group, *uniqueId, *indexTable;
    orc_real *sortedPos, *sortedVel, *sortedNormal, *sortedV_star;
    orc_real *sortedColor;
    orc_real *sortedMass, *sortedRestDensity, *sorted_pressure;
    int *sortedType, *sortedGroup, *sortedUniqueId;
    int *particleHash, *particleIndex, *gridCellStart, *gridCellEnd;
    orc_real *DF_factor, *pstiff, *error; /* DFSPH, n each (solver/sph_solver.cuh:79-190) */
    orc_real *dii, *dijpjl, *pressureForce; /* IISPH, n x 3 (solver/sph_solver.cuh:181-188) */
    orc_real *aii, *density_star;           /* IISPH, n each */
} ORC_NAME(state);

#ifndef ORC_REAL_IS_DOUBLE
/* tests only: the seventh power of the pressure equation as the oracle takes it (product in double, rounded once)
 * beside this host's powf(x, 7) */
void orc_pow7_probe(const float *x, int n, float *product, float *libm);
#endif

ORC_NAME(state) *ORC_NAME(create)(int n, int num_cells);
void ORC_NAME(destroy)(ORC_NAME(state) *s);

/* MultiphaseSPHSolver::LoadParam derived constants, MultiphaseSPHSolver.cpp:452-469 */
void ORC_NAME(derive_params)(ORC_NAME(params) *p, orc_real spacing, orc_real smoothratio,
                             const orc_real vmin[3], const orc_real vmax[3]);

void ORC_NAME(set_threads)(int nthreads);
int ORC_NAME(get_threads)(void);

/* the operator seam, solver/sph_solver.cuh:197-220 */
void ORC_NAME(calc_hash)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(sort_particle)(ORC_NAME(state) *s);
void ORC_NAME(reorder_data_and_find_cell_start)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(copy_sorted_to_primary)(ORC_NAME(state) *s); /* sph_solver.cpp:58-84 */
void ORC_NAME(sort_particles)(const ORC_NAME(params) *p, ORC_NAME(state) *s); /* SPHSolver::SortParticles */
void ORC_NAME(compute_pressure)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(compute_force)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(advect)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(rigid_particle_volume)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
/* SPHSolver::SolveSPH without the D2H copy, sph_solver.cpp:87-98 */
void ORC_NAME(solve_sph)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int nsteps);

/* ---- single-phase DFSPH step (solver/sph_kernel_impl.cuh:582-1018, sph_solver.cu:119-256) ---- */
void ORC_NAME(compute_density_alpha)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(compute_non_pressure_force)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(solve_density_stiff)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(solve_divergence_stiff)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(apply_pstiff)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(update_position)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(update_velocities)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
/* both return the number of applyPStiff rounds done; *last_error = the last maximum of error[] */
int ORC_NAME(correct_density_error)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int maxiter, orc_real ethres,
                                    orc_real *last_error);
int ORC_NAME(correct_divergence_error)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int maxiter,
                                       orc_real ethres, orc_real *last_error);
/* SPHSolver::SolveDFSPH without the D2H copy, sph_solver.cpp:108-120 */
void ORC_NAME(solve_dfsph)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int nsteps);

/* ---- IISPH step, fluid and rigid particles (solver/sph_kernel_impl.cuh:1126-1306, 3663-4076;
 * launchers solver/sph_solver.cu:300-310, 1388-1457; step MultiphaseSPHSolver.cpp:173-188).
 * Deformable and cloth particles are outside this restatement: their kernels
 * (ComputeTensionWithP_Kernel, UpdateSolidStateF_Kernel) do nothing for the types covered. ---- */
/* EffectiveMassKernel, solver/sph_kernel_impl.cuh:1722-1755: mass, rest density and colour of every
 * non-rigid particle from its volume fractions.  `vfrac` holds typenum fractions per particle in uniqueId
 * order (the reference carries the table through the sort; reading it by uniqueId gives the same values),
 * `dens_arr` is <DensityArray>. */
void ORC_NAME(effective_mass)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int typenum, const orc_real *dens_arr,
                              const orc_real *vfrac);
void ORC_NAME(non_pressure_force_multiphase)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(iisph_factor)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(iisph_predict_density)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(iisph_calc_dijpjl)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(iisph_calc_new_pressure)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
/* IISPHSolvePressure: returns the number of rounds run (1..maxiter); *last_mean = last mean of error[].
 * mean_in_double = 0: the reference's sequential float sum; 1: a double sum rounded once. */
int ORC_NAME(iisph_solve_pressure)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int maxiter, int mean_in_double,
                                   orc_real *last_mean);
void ORC_NAME(iisph_calc_pressure_force)(const ORC_NAME(params) *p, ORC_NAME(state) *s);
void ORC_NAME(iisph_update)(const ORC_NAME(params) *p, ORC_NAME(state) *s); /* both kernels of IISPHUpdate */
/* SolveIISPH without the D2H copy; the state must have been prepared as SetupIISPH does
 * (MultiphaseSPHSolver.cpp:850-860): sorted, iisph_factor and rigid_particle_volume run once. */
void ORC_NAME(solve_iisph)(const ORC_NAME(params) *p, ORC_NAME(state) *s, int nsteps, int mean_in_double);

#ifdef __cplusplus
}
#endif
#endif
