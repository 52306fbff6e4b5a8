/*
 * sph_oracle.c -- CPU restatement of the reference WCSPH step (see sph_oracle.h).
 * TEST INFRASTRUCTURE ONLY; parity unpinned (no reference fixtures exist).
 *
 * All file:line citations are relative to /root/reference/particle_solver.
 * Every function keeps the reference's evaluation order and its float/double
 * promotions; `R` marks a value the reference holds in a `float`.
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef orc_real R;
typedef ORC_NAME(params) params_t;
typedef ORC_NAME(state) state_t;

#ifdef ORC_REAL_IS_DOUBLE
#define R_SQRT(x) sqrt(x)
#define R_FLOOR(x) floor(x)
#define R_POW(x, y) pow((x), (y))
/* (the fp64 build, which has no counterpart in the reference: the product the fp64 HIP kernels take, see orc_pow7f
 * below; within four roundings of 2^-53 of pow(x, 7.0)) */
static inline double orc_pow7d(double x) {
    const double d2 = x * x, d4 = d2 * d2;
    return d4 * d2 * x;
}
#define R_POW7(x) orc_pow7d(x)
#define R_ROUND(x) round(x)
#else
#define R_SQRT(x) sqrtf(x)
#define R_FLOOR(x) floorf(x)
#define R_POW(x, y) powf((x), (y))
/* powf(x, 7) of the pressure equation (solver/sph_kernel_impl.cuh:263).  CUDA's powf is specified to a few units in
 * the last place, glibc's misses the correctly rounded value for about one argument in 100 000, ROCm's for one in a
 * thousand: no two libraries agree on every argument.  The fp32 oracle takes the product in double (four roundings of
 * 2^-53) and rounds once -- the correctly rounded power but for ties closer than that -- which every faithful powf
 * is within one unit of, and which the HIP kernels compute the same way (r_pow7), so that a run on the
 * exact-arithmetic kernels can be compared with the oracle's bit for bit over many steps. */
static inline float orc_pow7f(float x) {
    const double d = (double)x, d2 = d * d, d4 = d2 * d2;
    return (float)(d4 * d2 * d);
}
#define R_POW7(x) orc_pow7f(x)
#define R_ROUND(x) roundf(x)
/* for the tests: the seventh power as the pressure equation takes it, beside this host's powf(x, 7) */
void orc_pow7_probe(const float *x, int n, float *product, float *libm) {
    for (int i = 0; i < n; i++) {
        product[i] = orc_pow7f(x[i]);
        libm[i] = powf(x[i], 7);
    }
}
#endif

static int g_threads = 1;

void ORC_NAME(set_threads)(int nthreads) {
    if (nthreads < 1) nthreads = 1;
    g_threads = nthreads;
}
int ORC_NAME(get_threads)(void) { return g_threads; }

/* ------------------------------------------------------------------ */
/* allocation                                                          */
/* ------------------------------------------------------------------ */

static void *zalloc(size_t count, size_t size) {
    void *p = calloc(count ? count : 1, size);
    return p;
}

state_t *ORC_NAME(create)(int n, int num_cells) {
    state_t *s = (state_t *)zalloc(1, sizeof(state_t));
    if (!s) return NULL;
    s->n = n;
    s->num_cells = num_cells;
    size_t N = (size_t)n;
#define A3(f) s->f = (R *)zalloc(N * 3, sizeof(R))
#define A4(f) s->f = (R *)zalloc(N * 4, sizeof(R))
#define A1(f) s->f = (R *)zalloc(N, sizeof(R))
#define AI(f) s->f = (int *)zalloc(N, sizeof(int))
    A3(pos); A3(vel); A3(normal); A3(force); A3(v_star); A4(color);
    A1(mass); A1(density); A1(pressure); A1(restDensity);
    AI(type); AI(group); AI(uniqueId); AI(indexTable);
    A3(sortedPos); A3(sortedVel); A3(sortedNormal); A3(sortedV_star); A4(sortedColor);
    A1(sortedMass); A1(sortedRestDensity); A1(sorted_pressure);
    AI(sortedType); AI(sortedGroup); AI(sortedUniqueId);
    AI(particleHash); AI(particleIndex);
    A1(DF_factor); A1(pstiff); A1(error);
    A3(dii); A3(dijpjl); A3(pressureForce); A1(aii); A1(density_star);
#undef A3
#undef A4
#undef A1
#undef AI
    s->gridCellStart = (int *)zalloc((size_t)num_cells, sizeof(int));
    s->gridCellEnd = (int *)zalloc((size_t)num_cells, sizeof(int));
    return s;
}

void ORC_NAME(destroy)(state_t *s) {
    if (!s) return;
    free(s->pos); free(s->vel); free(s->normal); free(s->force); free(s->v_star); free(s->color);
    free(s->mass); free(s->density); free(s->pressure); free(s->restDensity);
    free(s->type); free(s->group); free(s->uniqueId); free(s->indexTable);
    free(s->sortedPos); free(s->sortedVel); free(s->sortedNormal); free(s->sortedV_star);
    free(s->sortedColor); free(s->sortedMass); free(s->sortedRestDensity); free(s->sorted_pressure);
    free(s->sortedType); free(s->sortedGroup); free(s->sortedUniqueId);
    free(s->particleHash); free(s->particleIndex); free(s->gridCellStart); free(s->gridCellEnd);
    free(s->DF_factor); free(s->pstiff); free(s->error);
    free(s->dii); free(s->dijpjl); free(s->pressureForce); free(s->aii); free(s->density_star);
    free(s);
}

/* ------------------------------------------------------------------ */
/* derived parameters: MultiphaseSPHSolver::LoadParam,                 */
/* solver/MultiphaseSPHSolver.cpp:452-469 (and ParseParam :381)        */
/* ------------------------------------------------------------------ */

void ORC_NAME(derive_params)(params_t *p, R spacing, R smoothratio, const R vmin[3], const R vmax[3]) {
    p->spacing = spacing;
    p->smoothradius = spacing * smoothratio; /* :381 */
    p->dx = p->smoothradius;                 /* :452 */
    for (int a = 0; a < 3; a++) {
        p->gridxmin[a] = vmin[a];
        p->gridxmax[a] = vmax[a];
        p->gridres[a] = (int)R_ROUND((vmax[a] - vmin[a]) / p->dx); /* :453-455 */
    }
    R sr = p->smoothradius;
    /* pow(float, float) is the float overload in C++; pow(float, int) promotes to double. */
    p->kspikydiff = (R)(-45.0f / (3.141592 * (double)R_POW(sr, (R)6.0))); /* :466 */
    p->kernel_cubic = (R)(1 / 3.141593 / pow((double)(sr / 2), 3.0));     /* :468 */
    p->kernel_cubic_gradient = (R)(1.5 / (double)3.141593f / pow((double)(sr / 2), 4.0)); /* :469 */
    p->kpoly6 = (R)(315.0f / (64.0f * 3.141592 * (double)R_POW(sr, (R)9.0)));             /* :464 */
}

/* ------------------------------------------------------------------ */
/* cell coordinates and hash                                           */
/* ------------------------------------------------------------------ */

/* float -> int as cvt.rzi.s32.f32 does it on the reference's hardware:
 * saturating, NaN -> 0 (plain C leaves out-of-range conversions undefined). */
static inline int to_int_sat(R v) {
    if (v != v) return 0;
    if (v >= (R)2147483647.0) return 2147483647;
    if (v <= (R)-2147483648.0) return (-2147483647 - 1);
    return (int)v;
}

/* calcGridPos, solver/sph_kernel_impl.cuh:74-80 */
static inline void calc_grid_pos(const params_t *p, const R *x, int g[3]) {
    g[0] = to_int_sat(R_FLOOR((x[0] - p->gridxmin[0]) / p->dx));
    g[1] = to_int_sat(R_FLOOR((x[1] - p->gridxmin[1]) / p->dx));
    g[2] = to_int_sat(R_FLOOR((x[2] - p->gridxmin[2]) / p->dx));
}

/* calcGridHash, solver/sph_kernel_impl.cuh:58-72 (the :69 guard is unreachable
 * after the range test but is kept). */
static inline unsigned calc_grid_hash(const params_t *p, int cx, int cy, int cz) {
    if (cx < 0 || cx >= p->gridres[0] || cy < 0 || cy >= p->gridres[1] || cz < 0 || cz >= p->gridres[2])
        return ORC_GRID_UNDEF;
    unsigned ret = (unsigned)(cy * p->gridres[0] * p->gridres[2] + cz * p->gridres[0] + cx);
    if (ret > (unsigned)(p->gridres[0] * p->gridres[1] * p->gridres[2])) return ORC_GRID_UNDEF;
    return ret;
}

/* calcHashD, solver/sph_kernel_impl.cuh:83-112; launcher sph_solver.cu:30-42.
 * EPSILON is a double literal, so the domain test compares in double. */
void ORC_NAME(calc_hash)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int i = 0; i < n; i++) {
        const R *x = s->pos + 3 * (size_t)i;
        unsigned hash;
        if (!((double)x[0] < (double)p->gridxmax[0] - ORC_EPSILON &&
              (double)x[0] > (double)p->gridxmin[0] + ORC_EPSILON &&
              (double)x[1] < (double)p->gridxmax[1] - ORC_EPSILON &&
              (double)x[1] > (double)p->gridxmin[1] + ORC_EPSILON &&
              (double)x[2] < (double)p->gridxmax[2] - ORC_EPSILON &&
              (double)x[2] > (double)p->gridxmin[2] + ORC_EPSILON)) {
            hash = ORC_GRID_UNDEF;
        } else {
            int g[3];
            calc_grid_pos(p, x, g);
            hash = calc_grid_hash(p, g[0], g[1], g[2]);
        }
        s->particleHash[i] = (int)hash;
        s->particleIndex[i] = i;
    }
}

/* ------------------------------------------------------------------ */
/* sortParticle, solver/sph_solver.cu:46-53.                           */
/* thrust::sort_by_key on int keys (Thrust of CUDA 9.2, not vendored in */
/* the reference) is a stable least-significant-digit radix sort: the   */
/* result is the ascending order of the keys with ties kept in their    */
/* previous array order.  Restated as an 8-bit LSD radix sort whose     */
/* per-thread chunks are taken in thread order, which keeps it stable.  */
/* Keys are non-negative (cell ids and GRID_UNDEF).                     */
/* ------------------------------------------------------------------ */

void ORC_NAME(sort_particle)(state_t *s) {
    const int n = s->n;
    if (n <= 1) return;
    int *k0 = s->particleHash, *v0 = s->particleIndex;
    int *k1 = (int *)malloc((size_t)n * sizeof(int));
    int *v1 = (int *)malloc((size_t)n * sizeof(int));
    int T = g_threads;
    if (T > n) T = 1;
    size_t *hist = (size_t *)malloc((size_t)T * 256 * sizeof(size_t));
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        memset(hist, 0, (size_t)T * 256 * sizeof(size_t));
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            size_t lo = (size_t)n * t / T, hi = (size_t)n * (t + 1) / T;
            size_t *h = hist + (size_t)t * 256;
            for (size_t i = lo; i < hi; i++) h[((unsigned)k0[i] >> shift) & 255u]++;
#pragma omp barrier
#pragma omp single
            {
                size_t sum = 0;
                for (int d = 0; d < 256; d++)
                    for (int tt = 0; tt < T; tt++) {
                        size_t c = hist[(size_t)tt * 256 + d];
                        hist[(size_t)tt * 256 + d] = sum;
                        sum += c;
                    }
            }
            for (size_t i = lo; i < hi; i++) {
                size_t dst = h[((unsigned)k0[i] >> shift) & 255u]++;
                k1[dst] = k0[i];
                v1[dst] = v0[i];
            }
        }
        int *tk = k0; k0 = k1; k1 = tk;
        int *tv = v0; v0 = v1; v1 = tv;
    }
    /* four passes: the data is back in the caller's arrays */
    free(k1);
    free(v1);
    free(hist);
}

/* ------------------------------------------------------------------ */
/* reorderDataAndFindCellStartD, solver/sph_kernel_impl.cuh:116-200;   */
/* launcher solver/sph_solver.cu:57-77 (memset of gridCellStart only:  */
/* gridCellEnd keeps stale entries for empty cells).                   */
/* ------------------------------------------------------------------ */

void ORC_NAME(reorder_data_and_find_cell_start)(const params_t *p, state_t *s) {
    (void)p;
    const int n = s->n;
    memset(s->gridCellStart, 0xff, (size_t)s->num_cells * sizeof(int));
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        unsigned hash = (unsigned)s->particleHash[index];
        unsigned prev = index > 0 ? (unsigned)s->particleHash[index - 1] : 0u;
        if (index == 0 || hash != prev) {
            if (hash != ORC_GRID_UNDEF) s->gridCellStart[hash] = index;
            if (index > 0) s->gridCellEnd[prev] = index;
        }
        if (index == n - 1) {
            if (hash != ORC_GRID_UNDEF) s->gridCellEnd[hash] = index + 1;
        }
        size_t src = (size_t)s->particleIndex[index], dst = (size_t)index;
        for (int a = 0; a < 3; a++) {
            s->sortedPos[3 * dst + a] = s->pos[3 * src + a];
            s->sortedVel[3 * dst + a] = s->vel[3 * src + a];
            s->sortedNormal[3 * dst + a] = s->normal[3 * src + a];
            s->sortedV_star[3 * dst + a] = s->v_star[3 * src + a];
        }
        for (int a = 0; a < 4; a++) s->sortedColor[4 * dst + a] = s->color[4 * src + a];
        s->sortedMass[dst] = s->mass[src];
        s->sortedType[dst] = s->type[src];
        s->sortedGroup[dst] = s->group[src];
        s->sortedUniqueId[dst] = s->uniqueId[src];
        s->indexTable[s->sortedUniqueId[dst]] = index;
        s->sortedRestDensity[dst] = s->restDensity[src];
        s->sorted_pressure[dst] = s->pressure[src];
    }
}

/* Note: `gridCellEnd[prev] = index` (:156-157) carries no GRID_UNDEF guard and
 * needs none: GRID_UNDEF is the largest key, so a run boundary never has it on
 * the `prev` side (this requires num_cells < GRID_UNDEF). */

/* the 11 device-to-device copies that concern this path, solver/sph_solver.cpp:58-84 */
void ORC_NAME(copy_sorted_to_primary)(state_t *s) {
    size_t N = (size_t)s->n;
    memcpy(s->pos, s->sortedPos, N * 3 * sizeof(R));
    memcpy(s->vel, s->sortedVel, N * 3 * sizeof(R));
    memcpy(s->normal, s->sortedNormal, N * 3 * sizeof(R));
    memcpy(s->color, s->sortedColor, N * 4 * sizeof(R));
    memcpy(s->type, s->sortedType, N * sizeof(int));
    memcpy(s->group, s->sortedGroup, N * sizeof(int));
    memcpy(s->mass, s->sortedMass, N * sizeof(R));
    memcpy(s->uniqueId, s->sortedUniqueId, N * sizeof(int));
    memcpy(s->v_star, s->sortedV_star, N * 3 * sizeof(R));
    memcpy(s->restDensity, s->sortedRestDensity, N * sizeof(R));
    memcpy(s->pressure, s->sorted_pressure, N * sizeof(R));
}

/* SPHSolver::SortParticles, solver/sph_solver.cpp:50-85 */
void ORC_NAME(sort_particles)(const params_t *p, state_t *s) {
    ORC_NAME(calc_hash)(p, s);
    ORC_NAME(sort_particle)(s);
    ORC_NAME(reorder_data_and_find_cell_start)(p, s);
    ORC_NAME(copy_sorted_to_primary)(s);
}

/* ------------------------------------------------------------------ */
/* smoothing kernel                                                    */
/* ------------------------------------------------------------------ */

/* Kernel_Cubic, solver/sph_kernel_impl.cuh:32-40.  q is a float; the two
 * polynomial branches are evaluated in double because of their literals. */
static inline R kernel_cubic(const params_t *p, R h, R x, R y, R z) {
    R q = R_SQRT(x * x + y * y + z * z) / h;
    R fq;
    if (q >= 2) return 0;
    if (q >= 1)
        fq = (R)((double)((2 - q) * (2 - q) * (2 - q)) * 0.25);
    else
        fq = (R)((0.666667 - (double)(q * q) + 0.5 * (double)q * (double)q * (double)q) * 1.5);
    return fq * p->kernel_cubic;
}

/* ------------------------------------------------------------------ */
/* DensityCell + ComputePressureKernel,                                */
/* solver/sph_kernel_impl.cuh:203-244, 246-266; launcher sph_solver.cu:80-88 */
/* ------------------------------------------------------------------ */

static inline void density_cell(const params_t *p, const state_t *s, int cx, int cy, int cz, int index,
                                const R *pos, R *density) {
    unsigned gridHash = calc_grid_hash(p, cx, cy, cz);
    if (gridHash == ORC_GRID_UNDEF) return;
    unsigned startIndex = (unsigned)s->gridCellStart[gridHash];
    if (startIndex == 0xffffffffu) return;
    unsigned endIndex = (unsigned)s->gridCellEnd[gridHash];
    R sr = p->smoothradius;
    R sr2 = sr * sr;
    for (unsigned j = startIndex; j < endIndex; j++) {
        const R *pos2 = s->pos + 3 * (size_t)j;
        R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
        R d2 = xx * xx + xy * xy + xz * xz;
        if (d2 >= sr2) continue;
        R wij = kernel_cubic(p, (R)((double)sr * 0.5), xx, xy, xz);
        if (s->type[j] == ORC_TYPE_FLUID) *density += s->mass[index] * wij;
        if (s->type[j] == ORC_TYPE_RIGID) *density += s->restDensity[index] * s->restDensity[j] * wij;
    }
}

void ORC_NAME(compute_pressure)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R density = 0;
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++)
                    density_cell(p, s, g[0] + x, g[1] + y, g[2] + z, index, pos, &density);
        s->density[index] = density;
        R pr = p->pressureK * (R_POW7(density / s->restDensity[index]) - 1);
        if (pr < 0) pr = 0;
        s->pressure[index] = pr;
    }
}

/* ------------------------------------------------------------------ */
/* ForceCell + computeF,                                               */
/* solver/sph_kernel_impl.cuh:269-355, 357-373; launcher sph_solver.cu:90-98 */
/* ------------------------------------------------------------------ */

static inline void force_cell(const params_t *p, const state_t *s, int cx, int cy, int cz, int index,
                              const R *pos, R *force) {
    unsigned gridHash = calc_grid_hash(p, cx, cy, cz);
    if (gridHash == ORC_GRID_UNDEF) return;
    unsigned startIndex = (unsigned)s->gridCellStart[gridHash];
    R sr = p->smoothradius;
    R sr2 = sr * sr;
    if (startIndex == 0xffffffffu) return;
    unsigned endIndex = (unsigned)s->gridCellEnd[gridHash];
    const R *veli = s->vel + 3 * (size_t)index;

    for (unsigned j = startIndex; j < endIndex; j++) {
        const R *pos2 = s->pos + 3 * (size_t)j;
        const R *velj = s->vel + 3 * (size_t)j;
        R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
        R d2 = xx * xx + xy * xy + xz * xz;
        R d = R_SQRT(d2);

        if (j != (unsigned)index && s->type[j] == ORC_TYPE_FLUID) {
            if (d2 >= sr2) continue;
            R c = sr - d;
            /* pressure, :291-294 */
            R nablaw = p->kspikydiff * c * c / d;
            R pc = nablaw * s->mass[j] *
                   (s->pressure[j] / s->density[j] / s->density[j] +
                    s->pressure[index] / s->density[index] / s->density[index]);
            force[0] += xx * pc * (R)(-1);
            force[1] += xy * pc * (R)(-1);
            force[2] += xz * pc * (R)(-1);
            /* viscosity, :297-299: `0.01 * sr2` makes the quotient and all that follows double */
            R vc = (R)((double)(nablaw * d2) / ((double)d2 + 0.01 * (double)sr2) * (double)s->mass[j] /
                       (double)s->density[j] * 2 * (double)p->viscosity);
            force[0] += (veli[0] - velj[0]) * vc;
            force[1] += (veli[1] - velj[1]) * vc;
            force[2] += (veli[2] - velj[2]) * vc;
        }

        if (s->type[j] == ORC_TYPE_BOUNDARY) {
            /* :316-339 */
            R B = 1;
            R q = d / sr;
            if ((double)q < 0.66666)
                B = (R)((double)B * 0.66666);
            else if (q < 1)
                B = (R)((double)B * ((double)(2 * q) - 1.5 * (double)q * (double)q));
            else if (q < 2)
                B = (R)((double)B * (0.5 * (double)(2 - q) * (double)(2 - q)));
            else
                B = 0;
            B = (R)((double)B * (0.02 * 88.5 * 88.5 / (double)d));
            R magnitude = s->mass[j] / (s->mass[index] + s->mass[j]) * B;
            force[0] += xx * magnitude;
            force[1] += xy * magnitude;
            force[2] += xz * magnitude;
            /* artificial viscosity, :342-350 */
            R vx = veli[0] - velj[0], vy = veli[1] - velj[1], vz = veli[2] - velj[2];
            R xv = vx * xx + vy * xy + vz * xz;
            if (xv < 0) {
                R c = sr - d;
                R nablaw = p->kspikydiff * c * c / d;
                R visc = (R)((double)(2 * p->boundary_visc * p->smoothradius) * 88.5 / 2000);
                R pi = (R)((double)(-visc * xv) / ((double)d2 + 0.01 * (double)sr2));
                force[0] += xx * pi * nablaw * s->mass[j] * (R)(-1);
                force[1] += xy * pi * nablaw * s->mass[j] * (R)(-1);
                force[2] += xz * pi * nablaw * s->mass[j] * (R)(-1);
            }
        }
    }
}

void ORC_NAME(compute_force)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] != ORC_TYPE_FLUID) continue; /* force[index] keeps its old value */
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R force[3] = {0, 0, 0};
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++)
                    force_cell(p, s, g[0] + x, g[1] + y, g[2] + z, index, pos, force);
        /* :372 -- the literal (0,-9.8,0), not dParam.gravity */
        s->force[3 * (size_t)index + 0] = force[0] + (R)0;
        s->force[3 * (size_t)index + 1] = force[1] + (R)-9.8;
        s->force[3 * (size_t)index + 2] = force[2] + (R)0;
    }
}

/* ------------------------------------------------------------------ */
/* AdvectKernel, solver/sph_kernel_impl.cuh:377-391; launcher sph_solver.cu:100-108. */
/* boundary_mode 1 adds the older snapshot's collision handling:       */
/* advectAndCollision + clampBoundary,                                  */
/* solver/新建文件夹/sph_kernel_impl.cuh:365-403 (skip TYPE_BOUNDARY,   */
/* clamp to the grid box in the order y, x, z and zero that velocity    */
/* component; `gridxmin + EPSILON` is a double rounded back to float).  */
/* ------------------------------------------------------------------ */

void ORC_NAME(advect)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        if (p->boundary_mode == ORC_BOUNDARY_LEGACY_CLAMP && s->type[index] == ORC_TYPE_BOUNDARY) continue;
        R *vel = s->vel + 3 * (size_t)index;
        R *pos = s->pos + 3 * (size_t)index;
        const R *force = s->force + 3 * (size_t)index;
        for (int a = 0; a < 3; a++) vel[a] += force[a] * p->dt;
        for (int a = 0; a < 3; a++) pos[a] += vel[a] * p->dt;

        R dr[3];
        for (int a = 0; a < 3; a++) dr[a] = p->gravity[a] - force[a];
        R magnitude = R_SQRT(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
        if (magnitude > p->acceleration_limit)
            for (int a = 0; a < 3; a++) dr[a] = dr[a] / magnitude * p->acceleration_limit;
        for (int a = 0; a < 3; a++) s->v_star[3 * (size_t)index + a] = p->gravity[a] - dr[a];

        if (p->boundary_mode == ORC_BOUNDARY_LEGACY_CLAMP) {
            static const int order[3] = {1, 0, 2};
            for (int k = 0; k < 3; k++) {
                int a = order[k];
                if ((double)pos[a] < (double)p->gridxmin[a] + ORC_EPSILON) {
                    pos[a] = (R)((double)p->gridxmin[a] + ORC_EPSILON);
                    vel[a] = 0;
                }
                if ((double)pos[a] > (double)p->gridxmax[a] - ORC_EPSILON) {
                    pos[a] = (R)((double)p->gridxmax[a] - ORC_EPSILON);
                    vel[a] = 0;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* RigidParticleVolumeCell + RigidParticleVolumeKernel,                */
/* solver/sph_kernel_impl.cuh:2423-2473: restDensity of a rigid        */
/* particle = 1 / sum of W over rigid neighbours (itself included).    */
/* ------------------------------------------------------------------ */

void ORC_NAME(rigid_particle_volume)(const params_t *p, state_t *s) {
    const int n = s->n;
    R sr = p->smoothradius;
    R sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] != ORC_TYPE_RIGID) continue;
        const R *xi = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, xi, g);
        R sum_wij = 0;
        for (int z = -1; z <= 1; z++)
            for (int y = -1; y <= 1; y++)
                for (int x = -1; x <= 1; x++) {
                    unsigned h = calc_grid_hash(p, g[0] + x, g[1] + y, g[2] + z);
                    if (h == ORC_GRID_UNDEF) continue;
                    unsigned start = (unsigned)s->gridCellStart[h];
                    if (start == 0xffffffffu) continue;
                    unsigned end = (unsigned)s->gridCellEnd[h];
                    for (unsigned j = start; j < end; j++) {
                        if (s->type[j] != ORC_TYPE_RIGID) continue;
                        const R *xj = s->pos + 3 * (size_t)j;
                        R xx = xi[0] - xj[0], xy = xi[1] - xj[1], xz = xi[2] - xj[2];
                        R d2 = xx * xx + xy * xy + xz * xz;
                        if (d2 >= sr2) continue;
                        sum_wij += kernel_cubic(p, sr / 2, xx, xy, xz);
                    }
                }
        s->restDensity[index] = 1 / sum_wij;
    }
}

/* SPHSolver::SolveSPH, solver/sph_solver.cpp:87-98 (host copy-back left out) */
void ORC_NAME(solve_sph)(const params_t *p, state_t *s, int nsteps) {
    for (int k = 0; k < nsteps; k++) {
        ORC_NAME(sort_particles)(p, s);
        ORC_NAME(compute_pressure)(p, s);
        ORC_NAME(compute_force)(p, s);
        ORC_NAME(advect)(p, s);
    }
}

/* ================================================================== */
/* Single-phase DFSPH step.                                            */
/* Kernels solver/sph_kernel_impl.cuh:582-1018, launchers               */
/* solver/sph_solver.cu:119-256, step SPHSolver::SolveDFSPH             */
/* solver/sph_solver.cpp:108-120.  All of it is float arithmetic except */
/* where a double literal promotes (noted).                             */
/* ================================================================== */

/* Cell lookup shared by the DFSPH cell functions: false for a cell outside the grid or empty. */
static inline int cell_range(const params_t *p, const state_t *s, int cx, int cy, int cz, unsigned *start,
                             unsigned *end) {
    unsigned gridHash = calc_grid_hash(p, cx, cy, cz);
    if (gridHash == ORC_GRID_UNDEF) return 0;
    *start = (unsigned)s->gridCellStart[gridHash];
    if (*start == 0xffffffffu) return 0;
    *end = (unsigned)s->gridCellEnd[gridHash];
    return 1;
}

#define FOR_27_CELLS(g, ...)                                    \
    for (int z = -1; z <= 1; z++)                               \
        for (int y = -1; y <= 1; y++)                           \
            for (int x = -1; x <= 1; x++) {                     \
                unsigned startIndex, endIndex;                  \
                if (!cell_range(p, s, (g)[0] + x, (g)[1] + y, (g)[2] + z, &startIndex, &endIndex)) continue; \
                for (unsigned j = startIndex; j < endIndex; j++) __VA_ARGS__ \
            }

/* DensityAlphaCell :582-624 + computeDensityAlpha_kernel :626-660: poly6 density and the DFSPH factor */
void ORC_NAME(compute_density_alpha)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R density = 0, mwij = 0, m3[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            if (j == (unsigned)index) continue;
            const R *pos2 = s->pos + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            R d = R_SQRT(d2);
            R c2 = sr2 - d2;
            R c = sr - d;
            R nablaw = p->kspikydiff * c * c / d;
            if (s->type[j] == ORC_TYPE_FLUID) {
                density += c2 * c2 * c2 * s->mass[j];
                R ax = xx * nablaw * s->mass[j], ay = xy * nablaw * s->mass[j], az = xz * nablaw * s->mass[j];
                mwij += ax * ax + ay * ay + az * az;
                m3[0] += ax;
                m3[1] += ay;
                m3[2] += az;
            }
        })
        density += sr2 * sr2 * sr2 * s->mass[index];
        density *= p->kpoly6;
        s->density[index] = density;
        R denom = m3[0] * m3[0] + m3[1] * m3[1] + m3[2] * m3[2] + mwij;
        if ((double)denom < 0.000001) denom = (R)0.000001; /* clamp for stability, :656-657 */
        s->DF_factor[index] = s->density[index] / denom;
    }
}

/* computeNPFCell :663-742 + computeNPF_kernel :744-773: viscosity, wall penalty, gravity -> v_star */
void ORC_NAME(compute_non_pressure_force)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        R *vs = s->v_star + 3 * (size_t)index;
        if (s->type[index] != ORC_TYPE_FLUID) {
            vs[0] = vs[1] = vs[2] = 0;
            continue;
        }
        const R *pos = s->pos + 3 * (size_t)index;
        const R *veli = s->vel + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R force[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            const R *pos2 = s->pos + 3 * (size_t)j;
            const R *velj = s->vel + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            R d = R_SQRT(d2);
            if (j != (unsigned)index && s->type[j] == ORC_TYPE_FLUID) {
                if (d2 >= sr2) continue;
                R c = sr - d;
                R nablaw = p->kspikydiff * c * c / d;
                /* :703-704, `0.01 * sr2` is a double */
                R vc = (R)((double)(nablaw * d2) / ((double)d2 + 0.01 * (double)sr2) * (double)s->mass[j] /
                           (double)s->density[j] * 2 * (double)p->viscosity);
                force[0] += (veli[0] - velj[0]) * vc;
                force[1] += (veli[1] - velj[1]) * vc;
                force[2] += (veli[2] - velj[2]) * vc;
            }
            if (s->type[j] == ORC_TYPE_BOUNDARY) { /* :708-739, the text of ForceCell's branch */
                R B = 1;
                R q = d / sr;
                if ((double)q < 0.66666)
                    B = (R)((double)B * 0.66666);
                else if (q < 1)
                    B = (R)((double)B * ((double)(2 * q) - 1.5 * (double)q * (double)q));
                else if (q < 2)
                    B = (R)((double)B * (0.5 * (double)(2 - q) * (double)(2 - q)));
                else
                    B = 0;
                B = (R)((double)B * (0.02 * 88.5 * 88.5 / (double)d));
                R magnitude = s->mass[j] / (s->mass[index] + s->mass[j]) * B;
                force[0] += xx * magnitude;
                force[1] += xy * magnitude;
                force[2] += xz * magnitude;
                R vx = veli[0] - velj[0], vy = veli[1] - velj[1], vz = veli[2] - velj[2];
                R xv = vx * xx + vy * xy + vz * xz;
                if (xv < 0) {
                    R c = sr - d;
                    R nablaw = p->kspikydiff * c * c / d;
                    R visc = (R)((double)(2 * p->boundary_visc * p->smoothradius) * 88.5 / 2000);
                    R pi = (R)((double)(-visc * xv) / ((double)d2 + 0.01 * (double)sr2));
                    force[0] += xx * pi * nablaw * s->mass[j] * (R)(-1);
                    force[1] += xy * pi * nablaw * s->mass[j] * (R)(-1);
                    force[2] += xz * pi * nablaw * s->mass[j] * (R)(-1);
                }
            }
        })
        force[0] += (R)0;
        force[1] += (R)-9.8; /* :770 */
        force[2] += (R)0;
        vs[0] = veli[0] + force[0] * p->dt;
        vs[1] = veli[1] + force[1] * p->dt;
        vs[2] = veli[2] + force[2] * p->dt;
    }
}

/* sum over fluid neighbours of dot(v*_i - v*_j, grad W_ij m_j) (DensityChangeCell :775-813,
 * PredictDensityCell :846-887); `scale_dt` multiplies every term by dt as the latter does */
static inline R density_change(const params_t *p, const state_t *s, int index, R start, int scale_dt) {
    const R sr = p->smoothradius, sr2 = sr * sr;
    const R *pos = s->pos + 3 * (size_t)index;
    const R *vsi = s->v_star + 3 * (size_t)index;
    int g[3];
    calc_grid_pos(p, pos, g);
    R acc = start;
    FOR_27_CELLS(g, {
        if (j == (unsigned)index) continue;
        const R *pos2 = s->pos + 3 * (size_t)j;
        const R *vsj = s->v_star + 3 * (size_t)j;
        R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
        R vx = vsi[0] - vsj[0], vy = vsi[1] - vsj[1], vz = vsi[2] - vsj[2];
        R d2 = xx * xx + xy * xy + xz * xz;
        if (d2 >= sr2) continue;
        R d = R_SQRT(d2);
        R c = sr - d;
        R nablaw = p->kspikydiff * c * c / d;
        if (s->type[j] == ORC_TYPE_FLUID) {
            R nx = xx * nablaw * s->mass[j], ny = xy * nablaw * s->mass[j], nz = xz * nablaw * s->mass[j];
            R dot = vx * nx + vy * ny + vz * nz;
            if (scale_dt)
                acc += dot * p->dt;
            else
                acc += dot;
        }
    })
    return acc;
}

/* solveDivergenceStiff :818-842 */
void ORC_NAME(solve_divergence_stiff)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] != ORC_TYPE_FLUID) {
            s->error[index] = 0;
            s->pstiff[index] = 0;
            continue;
        }
        R densChange = density_change(p, s, index, 0, 0);
        if (densChange < 0) densChange = 0;
        s->error[index] = densChange / s->restDensity[index];
        s->pstiff[index] = densChange / p->dt * s->DF_factor[index];
    }
}

/* solveDensityStiff :890-919 */
void ORC_NAME(solve_density_stiff)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] != ORC_TYPE_FLUID) {
            s->error[index] = 0;
            s->pstiff[index] = 0;
            continue;
        }
        R density = density_change(p, s, index, s->density[index], 1);
        s->pstiff[index] = (density - s->restDensity[index]) * s->DF_factor[index] / p->dt / p->dt;
        s->error[index] = density - s->restDensity[index];
        if (s->pstiff[index] < 0) {
            s->pstiff[index] = 0;
            s->error[index] = 0;
        }
    }
}

/* applyPressureCell :921-961 + applyPStiff :963-984.  v_star is updated in place while neighbours
 * read pos / pstiff / density only, so the order of the particles does not matter. */
void ORC_NAME(apply_pstiff)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] != ORC_TYPE_FLUID) continue;
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R force[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            if (j == (unsigned)index || s->type[j] != ORC_TYPE_FLUID) continue;
            const R *pos2 = s->pos + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            /* KernelGradient_Spiky :24-30 */
            R d = R_SQRT(xx * xx + xy * xy + xz * xz);
            R c = sr - d;
            R factor = p->kspikydiff * c * c / d;
            R nx = xx * factor * s->mass[j], ny = xy * factor * s->mass[j], nz = xz * factor * s->mass[j];
            R w = s->pstiff[index] / s->density[index] + s->pstiff[j] / s->density[j];
            force[0] += nx * w;
            force[1] += ny * w;
            force[2] += nz * w;
        })
        R *vs = s->v_star + 3 * (size_t)index;
        vs[0] += force[0] * p->dt * (R)(-1);
        vs[1] += force[1] * p->dt * (R)(-1);
        vs[2] += force[2] * p->dt * (R)(-1);
    }
}

/* updatePosition :986-992 */
void ORC_NAME(update_position)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        for (int a = 0; a < 3; a++) s->pos[3 * (size_t)index + a] += s->v_star[3 * (size_t)index + a] * p->dt;
    }
}

/* UpdateVelocities :995-1018 */
void ORC_NAME(update_velocities)(const params_t *p, state_t *s) {
    const int n = s->n;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        R *v = s->vel + 3 * (size_t)index, *vs = s->v_star + 3 * (size_t)index;
        R a[3] = {vs[0] - v[0], vs[1] - v[1], vs[2] - v[2]};
        for (int k = 0; k < 3; k++) a[k] /= p->dt;
        R magnitude = R_SQRT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (magnitude > p->acceleration_limit)
            for (int k = 0; k < 3; k++) a[k] = a[k] / magnitude * p->acceleration_limit;
        for (int k = 0; k < 3; k++) {
            a[k] = p->gravity[k] - a[k];
            v[k] = vs[k];
            vs[k] = a[k];
        }
    }
}

static R max_error(const state_t *s, R start) {
    R e = start;
    for (int i = 0; i < s->n; i++) e = s->error[i] > e ? s->error[i] : e;
    return e;
}

/* correctDensityError, solver/sph_solver.cu:144-210 (the maximum starts from -9999) */
int ORC_NAME(correct_density_error)(const params_t *p, state_t *s, int maxiter, R ethres, R *last_error) {
    int iter = 0;
    R error = 0;
    while (iter < maxiter) {
        ORC_NAME(solve_density_stiff)(p, s);
        error = max_error(s, (R)-9999);
        if (error < ethres) break;
        ORC_NAME(apply_pstiff)(p, s);
        iter++;
    }
    ORC_NAME(update_position)(p, s);
    if (last_error) *last_error = error;
    return iter;
}

/* correctDivergenceError, solver/sph_solver.cu:212-256 (one extra solve first; the maximum starts from 0) */
int ORC_NAME(correct_divergence_error)(const params_t *p, state_t *s, int maxiter, R ethres, R *last_error) {
    int iter = 0;
    R error = 0;
    ORC_NAME(solve_divergence_stiff)(p, s);
    while (iter < maxiter) {
        ORC_NAME(solve_divergence_stiff)(p, s);
        error = max_error(s, 0);
        if (error < ethres) break;
        ORC_NAME(apply_pstiff)(p, s);
        iter++;
    }
    ORC_NAME(update_velocities)(p, s);
    if (last_error) *last_error = error;
    return iter;
}

/* SPHSolver::SolveDFSPH, solver/sph_solver.cpp:108-120; the state must have been prepared as
 * SetupDFSPH does (:102-106): sorted, with compute_density_alpha run once. */
void ORC_NAME(solve_dfsph)(const params_t *p, state_t *s, int nsteps) {
    for (int k = 0; k < nsteps; k++) {
        ORC_NAME(compute_non_pressure_force)(p, s);
        ORC_NAME(correct_density_error)(p, s, 5, 1, NULL);
        ORC_NAME(sort_particles)(p, s);
        ORC_NAME(compute_density_alpha)(p, s);
        ORC_NAME(correct_divergence_error)(p, s, 5, 1, NULL);
    }
}

/* ================================================================== */
/* IISPH step for fluid and rigid particles.                           */
/* Kernels solver/sph_kernel_impl.cuh:1126-1306 (non-pressure forces)  */
/* and :3663-4076, launchers solver/sph_solver.cu:300-310, 1388-1457,  */
/* step MultiphaseSPHSolver::SolveIISPH MultiphaseSPHSolver.cpp:173-188. */
/* ================================================================== */

/* KernelGradient_Cubic, solver/sph_kernel_impl.cuh:42-54: `q > EPSILON` compares in double, both
 * polynomial branches are evaluated in double because of their literals. */
static inline void kernel_gradient_cubic(const params_t *p, R h, R x, R y, R z, R out[3]) {
    R r = R_SQRT(x * x + y * y + z * z);
    R q = r / h;
    if (!(q < 2 && (double)q > ORC_EPSILON)) {
        out[0] = out[1] = out[2] = 0;
        return;
    }
    R df;
    if (q >= 1)
        df = (R)(0.5 * (double)(2 - q) * (double)(2 - q) * (-1));
    else
        df = (R)((double)(-2 * q) + 1.5 * (double)q * (double)q);
    out[0] = x * df * p->kernel_cubic_gradient / r;
    out[1] = y * df * p->kernel_cubic_gradient / r;
    out[2] = z * df * p->kernel_cubic_gradient / r;
}

static inline R dot3(const R *a, const R *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static inline int is_fluid_like(int t) { /* the three `case` labels that share one body */
    return t == ORC_TYPE_FLUID || t == ORC_TYPE_DEFORMABLE || t == ORC_TYPE_CLOTH;
}

/* EffectiveMassKernel :1722-1755 (the printf on a vanishing beta is left out; effective_mass[] = mass[]) */
void ORC_NAME(effective_mass)(const params_t *p, state_t *s, int typenum, const R *dens_arr, const R *vfrac) {
    const int n = s->n;
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        const R *vol_frac = vfrac + (size_t)s->uniqueId[index] * (size_t)typenum;
        R beta = 0, rest_density = 0;
        for (int k = 0; k < typenum; k++) {
            R mass_frac = vol_frac[k] * dens_arr[k] / s->restDensity[index];
            if ((double)mass_frac < ORC_EPSILON) continue;
            beta += vol_frac[k] * vol_frac[k] / mass_frac;
            rest_density += vol_frac[k] * dens_arr[k];
        }
        (void)beta;
        s->mass[index] = rest_density * p->spacing * p->spacing * p->spacing;
        s->restDensity[index] = rest_density;
        /* color.Set(vol_frac[0], vol_frac[1], vol_frac[2], 1): the table has at least three columns there */
        for (int k = 0; k < 3; k++) s->color[4 * (size_t)index + k] = k < typenum ? vol_frac[k] : 0;
        s->color[4 * (size_t)index + 3] = 1;
    }
}

/* NonPressureForceCell_Multiphase :1126-1262 + NonPressureForceKernel_Multiphase :1264-1306.
 * pow(float, int) in `fac` and `sr6` is taken as the double overload (C++11 promotion). */
void ORC_NAME(non_pressure_force_multiphase)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R h = (R)((double)p->smoothradius * 0.5);
    const R h2 = h * h;
    const R support_radius = h * 2;
    const R fac = (R)(32.0f / 3.141593 / pow((double)support_radius, 9.0));
    const R sr6 = (R)(pow((double)p->smoothradius, 6.0) / 64.0);
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        R *vs = s->v_star + 3 * (size_t)index;
        const R *veli = s->vel + 3 * (size_t)index;
        if (s->type[index] == ORC_TYPE_RIGID) {
            vs[0] = vs[1] = vs[2] = 0;
            continue;
        }
        if (s->type[index] == ORC_TYPE_CLOTH) {
            for (int a = 0; a < 3; a++) vs[a] = veli[a] + p->gravity[a] * p->dt;
            continue;
        }
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R force[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            const R *pos2 = s->pos + 3 * (size_t)j;
            const R *velj = s->vel + 3 * (size_t)j;
            R xij[3] = {pos[0] - pos2[0], pos[1] - pos2[1], pos[2] - pos2[2]};
            R d2 = xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2];
            if (d2 >= h2 * 4 || (double)d2 < ORC_EPSILON) continue;
            R d = R_SQRT(d2);
            R nablaw[3];
            kernel_gradient_cubic(p, h, xij[0], xij[1], xij[2], nablaw);
            R vij[3] = {veli[0] - velj[0], veli[1] - velj[1], veli[2] - velj[2]};
            R volj = s->mass[j] / s->density[j];

            if (s->type[index] == ORC_TYPE_FLUID && s->type[j] == ORC_TYPE_FLUID) {
                /* xsph artificial viscosity, :1205-1208 */
                R w = kernel_cubic(p, h, xij[0], xij[1], xij[2]);
                R c;
                if (s->group[index] == 0 || s->group[j] == 0)
                    c = p->viscosity * volj * (R)(-1) * w / p->dt;
                else
                    c = (R)(0.5 * (double)volj * (-1) * (double)w / (double)p->dt);
                for (int a = 0; a < 3; a++) force[a] += vij[a] * c;
            }
            if (s->group[index] == s->group[j]) { /* surface tension, :1214-1232 */
                R c = support_radius - d;
                R sf_kernel;
                if (d < h)
                    sf_kernel = 2 * c * c * c * d * d2 - sr6;
                else if (d < support_radius)
                    sf_kernel = c * c * c * d * d2;
                else
                    sf_kernel = 0;
                sf_kernel *= fac;
                R tmp = p->surface_tension;
                if (s->group[index] == 4) tmp *= 10;
                R k = tmp * volj * sf_kernel / d * (R)(-1);
                for (int a = 0; a < 3; a++) force[a] += xij[a] * k;
            }
            /* :1234-1237 couples two TYPE_DEFORMABLE particles: not covered */
            if (s->type[j] == ORC_TYPE_RIGID && s->group[j] == 0) { /* wall viscosity and friction, :1239-1259 */
                R xv = dot3(vij, xij);
                const R *nj = s->normal + 3 * (size_t)j;
                R ndv = dot3(nj, vij);
                R vt[3] = {vij[0] - nj[0] * ndv, vij[1] - nj[1] * ndv, vij[2] - nj[2] * ndv};
                if (xv < 0) {
                    R visc = (R)((double)(p->boundary_visc * p->smoothradius) * 88.5 / (double)s->density[index] * 0.25);
                    R pi = (R)((double)(visc * xv) / ((double)d2 + 0.01 * (double)h2));
                    R f[3];
                    for (int a = 0; a < 3; a++) f[a] = nablaw[a] * pi * s->restDensity[index] * s->restDensity[j];
                    R fdn = dot3(f, nj);
                    R fn[3] = {nj[0] * fdn, nj[1] * fdn, nj[2] * fdn};
                    R k = R_SQRT(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]) * (R)(-1) * p->boundary_friction;
                    for (int a = 0; a < 3; a++) force[a] += fn[a];
                    for (int a = 0; a < 3; a++) force[a] += vt[a] * k;
                }
            }
        })
        for (int a = 0; a < 3; a++) force[a] += p->gravity[a];
        for (int a = 0; a < 3; a++) vs[a] = veli[a] + force[a] * p->dt;
    }
}

/* mass_j of the IISPH cell functions: mass[i] for a fluid-like neighbour, restDensity[j] *
 * restDensity[i] for a rigid one; 0 = the neighbour's type takes no part (no `case` for it) */
static inline int iisph_mass_j(const state_t *s, int i, unsigned j, R *mass_j) {
    if (is_fluid_like(s->type[j])) {
        *mass_j = s->mass[i];
        return 1;
    }
    if (s->type[j] == ORC_TYPE_RIGID) {
        *mass_j = s->restDensity[j] * s->restDensity[i];
        return 2;
    }
    return 0;
}

/* IISPHFactorCell :3665-3715 + IISPHFactorKernel :3717-3749 */
void ORC_NAME(iisph_factor)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R density = 0, sum[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            const R *pos2 = s->pos + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            R nablaw[3], mass_j;
            kernel_gradient_cubic(p, sr / 2, xx, xy, xz, nablaw);
            if (!iisph_mass_j(s, index, j, &mass_j)) continue;
            for (int a = 0; a < 3; a++) sum[a] += nablaw[a] * mass_j;
            density += kernel_cubic(p, sr / 2, xx, xy, xz) * mass_j;
        })
        s->density[index] = density;
        R k = p->dt * p->dt / density / density * (R)(-1);
        for (int a = 0; a < 3; a++) s->dii[3 * (size_t)index + a] = sum[a] * k;
    }
}

/* d_ji of :3784 / :3925: gradW * -1 * mass_j / density_i / density_i * dt * dt * -1, per component */
static inline void iisph_dji(const params_t *p, const R nablaw[3], R mass_j, R dens_i, R out[3]) {
    for (int a = 0; a < 3; a++)
        out[a] = nablaw[a] * (R)(-1) * mass_j / dens_i / dens_i * p->dt * p->dt * (R)(-1);
}

/* IISPHPredictDensityCell :3751-3805 + IISPHPredictDensityKernel :3807-3833 */
void ORC_NAME(iisph_predict_density)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        s->pressure[index] /= 2;
        R density = s->density[index];
        const R *pos = s->pos + 3 * (size_t)index;
        const R *vsi = s->v_star + 3 * (size_t)index;
        const R *dii = s->dii + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R aii = 0;
        FOR_27_CELLS(g, {
            const R *pos2 = s->pos + 3 * (size_t)j;
            const R *vsj = s->v_star + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R vij[3] = {vsi[0] - vsj[0], vsi[1] - vsj[1], vsi[2] - vsj[2]};
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            R nablaw[3], mass_j, dji[3], diff[3];
            kernel_gradient_cubic(p, sr / 2, xx, xy, xz, nablaw);
            int kind = iisph_mass_j(s, index, j, &mass_j);
            if (!kind) continue;
            if (kind == 1)
                density += p->dt * mass_j * dot3(vij, nablaw);
            else
                density += p->dt * mass_j * dot3(vsi, nablaw);
            iisph_dji(p, nablaw, mass_j, s->density[index], dji);
            for (int a = 0; a < 3; a++) diff[a] = dii[a] - dji[a];
            aii += mass_j * dot3(diff, nablaw);
        })
        s->density_star[index] = density;
        s->aii[index] = aii;
    }
}

/* CalcDIJPJLCell :3835-3880 + CalcDIJPJLKernel :3882-3902 */
void ORC_NAME(iisph_calc_dijpjl)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R acc[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            const R *pos2 = s->pos + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            R nablaw[3];
            kernel_gradient_cubic(p, sr / 2, xx, xy, xz, nablaw);
            if (!is_fluid_like(s->type[j])) continue; /* the TYPE_RIGID case is empty */
            R mass_j = s->mass[index];
            for (int a = 0; a < 3; a++)
                acc[a] += nablaw[a] * mass_j / s->density[j] / s->density[j] * s->pressure[j] * (R)(-1);
        })
        R k = p->dt * p->dt;
        for (int a = 0; a < 3; a++) s->dijpjl[3 * (size_t)index + a] = acc[a] * k;
    }
}

/* CalcNewPressureCell :3904-3959 + CalcNewPressureKernel :3961-3996.
 * The reference kernel overwrites pressure[index] while other threads still read pressure[j] of the
 * same array: what a neighbour sees depends on GPU scheduling.  This restatement takes the
 * deterministic reading -- every particle sees the pressures of the previous round (Jacobi) -- by
 * writing the new values to a copy that replaces pressure[] afterwards. */
void ORC_NAME(iisph_calc_new_pressure)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
    R *newp = (R *)malloc((size_t)(n ? n : 1) * sizeof(R));
    memcpy(newp, s->pressure, (size_t)n * sizeof(R));
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) {
            s->error[index] = 0;
            continue;
        }
        const R *pos = s->pos + 3 * (size_t)index;
        const R *dijpjl_i = s->dijpjl + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R lll = 0;
        FOR_27_CELLS(g, {
            if (j >= (unsigned)n) break;
            const R *pos2 = s->pos + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            R nablaw[3], mass_j;
            kernel_gradient_cubic(p, sr / 2, xx, xy, xz, nablaw);
            int kind = iisph_mass_j(s, index, j, &mass_j);
            if (kind == 1) {
                R dji[3], lt[3];
                iisph_dji(p, nablaw, mass_j, s->density[index], dji);
                const R *dii_j = s->dii + 3 * (size_t)j;
                const R *dijpjl_j = s->dijpjl + 3 * (size_t)j;
                for (int a = 0; a < 3; a++)
                    lt[a] = dijpjl_i[a] - dii_j[a] * s->pressure[j] - dijpjl_j[a] + dji[a] * s->pressure[index];
                lll += mass_j * dot3(lt, nablaw);
            } else if (kind == 2) {
                lll += mass_j * dot3(dijpjl_i, nablaw);
            }
        })
        R prev = s->pressure[index];
        R density_corr = s->density_star[index] + lll;
        R aii = s->aii[index];
        R np;
        if ((double)(aii < 0 ? -aii : aii) > 1e-8) {
            np = (R)(0.5 * (double)prev + 0.5 * (double)(s->restDensity[index] - density_corr) / (double)aii);
            if (np < 0) np = 0;
        } else {
            np = 0;
        }
        newp[index] = np;
        density_corr += prev * aii;
        s->error[index] = density_corr;
    }
    memcpy(s->pressure, newp, (size_t)n * sizeof(R));
    free(newp);
}

/* IISPHSolvePressure, solver/sph_solver.cu:1409-1443: up to `maxiter` (10 there) rounds, stopped when the
 * mean of error[] is below 1e-6 in magnitude or moved by less than 1e-6 since the last round. */
int ORC_NAME(iisph_solve_pressure)(const params_t *p, state_t *s, int maxiter, int mean_in_double, R *last_mean) {
    R err_sum = 0;
    int rounds = 0;
    for (int iter = 0; iter < maxiter; iter++) {
        ORC_NAME(iisph_calc_dijpjl)(p, s);
        ORC_NAME(iisph_calc_new_pressure)(p, s);
        rounds++;
        R last_err_sum = err_sum;
        if (mean_in_double) {
            double acc = 0;
            for (int i = 0; i < s->n; i++) acc += (double)s->error[i];
            err_sum = (R)(acc / (double)s->n);
        } else {
            err_sum = 0;
            for (int i = 0; i < s->n; i++) err_sum += s->error[i];
            err_sum /= s->n;
        }
        R a = err_sum < 0 ? -err_sum : err_sum;
        R b = err_sum - last_err_sum;
        if (b < 0) b = -b;
        if ((double)a < 1e-6 || (double)b < 1e-6) break;
    }
    if (last_mean) *last_mean = err_sum;
    return rounds;
}

/* CalcPressureForceCell :3998-4046 + CalcPressureForceKernel :4048-4065 */
void ORC_NAME(iisph_calc_pressure_force)(const params_t *p, state_t *s) {
    const int n = s->n;
    const R sr = p->smoothradius, sr2 = sr * sr;
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        const R *pos = s->pos + 3 * (size_t)index;
        int g[3];
        calc_grid_pos(p, pos, g);
        R pf[3] = {0, 0, 0};
        FOR_27_CELLS(g, {
            const R *pos2 = s->pos + 3 * (size_t)j;
            R xx = pos[0] - pos2[0], xy = pos[1] - pos2[1], xz = pos[2] - pos2[2];
            R d2 = xx * xx + xy * xy + xz * xz;
            if (d2 >= sr2) continue;
            R nablaw[3], mass_j;
            kernel_gradient_cubic(p, sr / 2, xx, xy, xz, nablaw);
            int kind = iisph_mass_j(s, index, j, &mass_j);
            if (kind == 1) {
                R w = s->pressure[index] / s->density[index] / s->density[index] +
                      s->pressure[j] / s->density[j] / s->density[j];
                for (int a = 0; a < 3; a++) pf[a] += nablaw[a] * (R)(-1) * mass_j * mass_j * w;
            } else if (kind == 2) {
                R w = s->pressure[index] / s->density[index] / s->density[index];
                for (int a = 0; a < 3; a++) pf[a] += nablaw[a] * (R)(-1) * s->mass[index] * mass_j * w;
            }
        })
        for (int a = 0; a < 3; a++) s->pressureForce[3 * (size_t)index + a] = pf[a];
    }
}

/* IISPHUpdate, solver/sph_solver.cu:1445-1457: CalcPressureForceKernel, then IISPHUpdateKernel :4067-4076 */
void ORC_NAME(iisph_update)(const params_t *p, state_t *s) {
    ORC_NAME(iisph_calc_pressure_force)(p, s);
    const int n = s->n;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int index = 0; index < n; index++) {
        if (s->type[index] == ORC_TYPE_RIGID) continue;
        for (int a = 0; a < 3; a++) {
            size_t k = 3 * (size_t)index + a;
            s->vel[k] = s->v_star[k] + s->pressureForce[k] * p->dt / s->mass[index];
            s->pos[k] += s->vel[k] * p->dt;
        }
    }
}

/* MultiphaseSPHSolver::SolveIISPH, MultiphaseSPHSolver.cpp:173-188.  ComputeTension and UpdateSolidState
 * act on TYPE_DEFORMABLE particles only (solver/sph_kernel_impl.cuh:2913, 3209) and are left out;
 * Sort (:48-100) is SortParticles for the arrays kept here. */
void ORC_NAME(solve_iisph)(const params_t *p, state_t *s, int nsteps, int mean_in_double) {
    for (int k = 0; k < nsteps; k++) {
        ORC_NAME(non_pressure_force_multiphase)(p, s);
        ORC_NAME(iisph_predict_density)(p, s);
        ORC_NAME(iisph_solve_pressure)(p, s, 10, mean_in_double, NULL);
        ORC_NAME(iisph_update)(p, s);
        ORC_NAME(sort_particles)(p, s);
        ORC_NAME(iisph_factor)(p, s);
    }
}
