/*
 * grid_oracle.c -- CPU restatement of the reference's Eulerian grid solver (see grid_oracle.h).
 * TEST INFRASTRUCTURE ONLY; parity unpinned (the reference solver cannot be built here and ships no
 * fixtures).  All file:line citations are solver/grid_solver.cpp under /root/reference/particle_solver
 * unless another file is named.  `R` marks a value the reference holds in a `float`.
 */
#include "grid_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef gorc_real R;
typedef GORC_NAME(grid) grid_t;

#ifdef ORC_REAL_IS_DOUBLE
#define R_SQRT(x) sqrt(x)
#define R_FLOOR(x) floor(x)
#define R_ABS(x) fabs(x)
#else
#define R_SQRT(x) sqrtf(x)
#define R_FLOOR(x) floorf(x)
#define R_ABS(x) fabsf(x)
#endif

/* the EPSILON in force from :299 on (sampleU/V/W); locateCell's 1e-6 (grid_solver.h:7) has no caller */
#define SAMPLE_EPSILON 0.000000001

/* Grid::cellId / uId / vId / wId, solver/grid_solver.h:36-49 */
static inline int cell_id(const grid_t *g, int x, int y, int z) { return (z * g->dim[1] + y) * g->dim[0] + x; }
static inline int u_id(const grid_t *g, int x, int y, int z) {
    return ((z + g->padding) * g->ylen + y + g->padding) * g->ulen + x + g->padding;
}
static inline int v_id(const grid_t *g, int x, int y, int z) {
    return ((z + g->padding) * g->vlen + y + g->padding) * g->xlen + x + g->padding;
}
static inline int w_id(const grid_t *g, int x, int y, int z) {
    return ((z + g->padding) * g->ylen + y + g->padding) * g->xlen + x + g->padding;
}

static R *zalloc(size_t n) { return (R *)calloc(n ? n : 1, sizeof(R)); }

grid_t *GORC_NAME(create)(const int dim[3], int padding, R h, R dt, R rho) {
    grid_t *g = (grid_t *)calloc(1, sizeof(grid_t));
    if (!g) return NULL;
    for (int a = 0; a < 3; a++) g->dim[a] = dim[a];
    g->padding = padding;
    g->h = h;
    /* :26-28: `-grid.dim.x/2*grid.h` negates, divides as integers, then multiplies */
    g->xmin[0] = (R)(-dim[0] / 2) * h;
    g->xmin[1] = 0;
    g->xmin[2] = (R)(-dim[2] / 2) * h;
    /* Grid::setSize, solver/grid_solver.h:23-35 */
    g->xlen = dim[0] + padding * 2;
    g->ylen = dim[1] + padding * 2;
    g->zlen = dim[2] + padding * 2;
    g->ulen = g->xlen + 1;
    g->vlen = g->ylen + 1;
    g->wlen = g->zlen + 1;
    g->Size = g->xlen * g->ylen * g->zlen;
    g->uSize = g->ulen * g->ylen * g->zlen;
    g->vSize = g->xlen * g->vlen * g->zlen;
    g->wSize = g->xlen * g->ylen * g->wlen;
    g->dimSize = dim[0] * dim[1] * dim[2];
    g->dt = dt;
    g->rho = rho;
    g->frame = 0;
    /* allocate :42-66 (only the face arrays and p are cleared there; calloc clears the rest too) */
    g->u = zalloc((size_t)g->uSize); g->v = zalloc((size_t)g->vSize); g->w = zalloc((size_t)g->wSize);
    g->uadv = zalloc((size_t)g->uSize); g->vadv = zalloc((size_t)g->vSize); g->wadv = zalloc((size_t)g->wSize);
    g->p = zalloc((size_t)g->dimSize); g->divU = zalloc((size_t)g->dimSize); g->b = zalloc((size_t)g->dimSize);
    g->Aq = zalloc((size_t)g->dimSize); g->r = zalloc((size_t)g->dimSize); g->q = zalloc((size_t)g->dimSize);
    return g;
}

void GORC_NAME(destroy)(grid_t *g) {
    if (!g) return;
    free(g->u); free(g->v); free(g->w); free(g->uadv); free(g->vadv); free(g->wadv);
    free(g->p); free(g->divU); free(g->b); free(g->Aq); free(g->r); free(g->q);
    free(g->pos); free(g->color);
    free(g);
}

void GORC_NAME(set_particles)(grid_t *g, int n, const R *pos3) {
    free(g->pos);
    free(g->color);
    g->n_particles = n;
    g->pos = zalloc((size_t)n * 3);
    g->color = zalloc((size_t)n * 4);
    memcpy(g->pos, pos3, (size_t)n * 3 * sizeof(R));
    for (int i = 0; i < n * 4; i++) g->color[i] = 1;
}

/* mvproduct :72-115: the 7-point Laplacian with p = 0 outside every side except the floor (j = 0),
 * which is a wall; the six terms are added to `res` in the order x-, x+, y-, y+, z-, z+ */
void GORC_NAME(mvproduct)(const grid_t *g, const R *v, R *dst) {
    const int nx = g->dim[0], ny = g->dim[1], nz = g->dim[2];
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                R res = 0;
                const int c = cell_id(g, i, j, k);
                if (i == 0) res += v[c]; else res += v[c] - v[cell_id(g, i - 1, j, k)];
                if (i == nx - 1) res += v[c]; else res += v[c] - v[cell_id(g, i + 1, j, k)];
                if (j == 0) res += 0; else res += v[c] - v[cell_id(g, i, j - 1, k)];
                if (j == ny - 1) res += v[c]; else res += v[c] - v[cell_id(g, i, j + 1, k)];
                if (k == 0) res += v[c]; else res += v[c] - v[cell_id(g, i, j, k - 1)];
                if (k == nz - 1) res += v[c]; else res += v[c] - v[cell_id(g, i, j, k + 1)];
                dst[c] = res;
            }
}

/* dotproduct :117-122: one float accumulator, index order */
R GORC_NAME(dotproduct)(const grid_t *g, const R *a, const R *b) {
    R res = 0;
    for (int i = 0; i < g->dimSize; i++) res += a[i] * b[i];
    return res;
}

/* divVelocity :124-141 */
void GORC_NAME(div_velocity)(grid_t *g) {
    g->divUsum = 0;
    for (int k = 0; k < g->dim[2]; k++)
        for (int j = 0; j < g->dim[1]; j++)
            for (int i = 0; i < g->dim[0]; i++) {
                R div = 0;
                div += g->u[u_id(g, i + 1, j, k)] - g->u[u_id(g, i, j, k)];
                div += g->v[v_id(g, i, j + 1, k)] - g->v[v_id(g, i, j, k)];
                div += g->w[w_id(g, i, j, k + 1)] - g->w[w_id(g, i, j, k)];
                div /= g->h;
                g->divU[cell_id(g, i, j, k)] = div;
                g->divUsum += div * div;
            }
}

/* makeRHS :143-160 */
void GORC_NAME(make_rhs)(grid_t *g) {
    GORC_NAME(div_velocity)(g);
    for (int k = 0; k < g->dim[2]; k++)
        for (int j = 0; j < g->dim[1]; j++)
            for (int i = 0; i < g->dim[0]; i++) {
                const int c = cell_id(g, i, j, k);
                R rhs = -g->divU[c];
                if (j == 0) rhs += (0 - g->v[v_id(g, i, j, k)]) / g->h;
                g->b[c] = rhs * g->rho * g->h * g->h / g->dt;
            }
}

/* solve :162-214 (conjugate gradients from the p of the previous step) */
int GORC_NAME(solve)(grid_t *g, int maxiter, R *residual) {
    const int N = g->dimSize;
    int iter = 0;
    R alpha, beta, rr, qAq, rabs = 0;
    GORC_NAME(mvproduct)(g, g->p, g->Aq);
    for (int i = 0; i < N; i++) {
        g->r[i] = g->b[i] - g->Aq[i];
        g->q[i] = g->r[i];
    }
    while (1) {
        GORC_NAME(mvproduct)(g, g->q, g->Aq);
        rr = GORC_NAME(dotproduct)(g, g->r, g->r);
        qAq = GORC_NAME(dotproduct)(g, g->q, g->Aq);
        if (rr == 0 || qAq == 0) break; /* "trivial all-zero condition" */
        alpha = rr / qAq;
        for (int i = 0; i < N; i++) {
            g->p[i] += alpha * g->q[i];
            g->r[i] -= alpha * g->Aq[i];
        }
        beta = GORC_NAME(dotproduct)(g, g->r, g->r) / rr;
        rabs = 0;
        for (int i = 0; i < N; i++) {
            g->q[i] = g->r[i] + beta * g->q[i];
            rabs += R_ABS(g->r[i]);
        }
        rabs /= N;
        if ((double)rabs < 0.000001) break;
        iter++;
        if (maxiter >= 0 && iter >= maxiter) break; /* the reference has no cap */
    }
    GORC_NAME(mvproduct)(g, g->p, g->Aq);
    if (residual) *residual = rabs;
    return iter;
}

/* updateU :216-266.  Every face is touched by exactly one cell of the reference's loop (face i by cell i,
 * the far face by the last cell), so the update is written per face.  The maximum-velocity estimate of
 * :251-256 feeds only a commented-out time-step update and is left out. */
void GORC_NAME(update_u)(grid_t *g) {
    const int nx = g->dim[0], ny = g->dim[1], nz = g->dim[2];
    const R c = g->dt / g->rho / g->h;
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i <= nx; i++) {
                R *f = &g->u[u_id(g, i, j, k)];
                if (i == 0) *f -= (g->p[cell_id(g, 0, j, k)] - 0) * c;
                else if (i < nx) *f -= (g->p[cell_id(g, i, j, k)] - g->p[cell_id(g, i - 1, j, k)]) * c;
                else *f -= (0 - g->p[cell_id(g, nx - 1, j, k)]) * c;
            }
    for (int k = 0; k < nz; k++)
        for (int j = 0; j <= ny; j++)
            for (int i = 0; i < nx; i++) {
                R *f = &g->v[v_id(g, i, j, k)];
                if (j == 0) *f = 0; /* near solid */
                else if (j < ny) *f -= (g->p[cell_id(g, i, j, k)] - g->p[cell_id(g, i, j - 1, k)]) * c;
                else *f -= (0 - g->p[cell_id(g, i, ny - 1, k)]) * c;
            }
    for (int k = 0; k <= nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                R *f = &g->w[w_id(g, i, j, k)];
                if (k == 0) *f -= (g->p[cell_id(g, i, j, 0)] - 0) * c;
                else if (k < nz) *f -= (g->p[cell_id(g, i, j, k)] - g->p[cell_id(g, i, j, k - 1)]) * c;
                else *f -= (0 - g->p[cell_id(g, i, j, nz - 1)]) * c;
            }
}

/* The common head of sampleU / sampleV / sampleW (:304-330 and its two copies): cell of the point,
 * the clamps at the far faces, 0 outside. */
static inline int sample_cell(const grid_t *g, const R p[3], R x[3], int c[3]) {
    for (int a = 0; a < 3; a++) {
        x[a] = (p[a] - g->xmin[a]) / g->h;
        c[a] = (int)R_FLOOR(x[a]);
    }
    for (int a = 0; a < 3; a++) {
        if ((double)(x[a] - c[a]) < SAMPLE_EPSILON && x[a] == g->dim[a]) c[a] -= 1;
        if ((double)x[a] > -SAMPLE_EPSILON && x[a] == -1) c[a] += 1;
    }
    if (c[0] < 0 || c[0] > g->dim[0] - 1 || c[1] < 0 || c[1] > g->dim[1] - 1 || c[2] < 0 || c[2] > g->dim[2] - 1)
        return 0;
    return 1;
}

/* weight along an axis on which the sampled component lives on faces (`c * h`: float) or on cell
 * centres (`(c + 0.5) * h`: double, rounded to float when stored) */
static inline R weight_face(const grid_t *g, const R p[3], int a, int c) {
    return (p[a] - (g->xmin[a] + c * g->h)) / g->h;
}
static inline R weight_centre(const grid_t *g, const R p[3], int a, int c) {
    return (R)(((double)p[a] - ((double)g->xmin[a] + ((double)c + 0.5) * (double)g->h)) / (double)g->h);
}

#define TRILERP(F, ID)                                                                                    \
    ((1 - w1) * ((1 - w2) * ((1 - w3) * F[ID(g, c[0], c[1], c[2])] + w3 * F[ID(g, c[0], c[1], c[2] + 1)]) +   \
                 w2 * ((1 - w3) * F[ID(g, c[0], c[1] + 1, c[2])] + w3 * F[ID(g, c[0], c[1] + 1, c[2] + 1)])) + \
     w1 * ((1 - w2) * ((1 - w3) * F[ID(g, c[0] + 1, c[1], c[2])] + w3 * F[ID(g, c[0] + 1, c[1], c[2] + 1)]) + \
           w2 * ((1 - w3) * F[ID(g, c[0] + 1, c[1] + 1, c[2])] + w3 * F[ID(g, c[0] + 1, c[1] + 1, c[2] + 1)])))

/* sampleU :301-350 */
R GORC_NAME(sample_u)(const grid_t *g, const R p[3]) {
    R x[3];
    int c[3];
    if (!sample_cell(g, p, x, c)) return 0;
    if ((double)(x[1] - c[1]) < 0.5) c[1]--;
    if ((double)(x[2] - c[2]) < 0.5) c[2]--;
    const R w1 = weight_face(g, p, 0, c[0]), w2 = weight_centre(g, p, 1, c[1]), w3 = weight_centre(g, p, 2, c[2]);
    return TRILERP(g->u, u_id);
}

/* sampleV :352-401 */
R GORC_NAME(sample_v)(const grid_t *g, const R p[3]) {
    R x[3];
    int c[3];
    if (!sample_cell(g, p, x, c)) return 0;
    if ((double)(x[0] - c[0]) < 0.5) c[0]--;
    if ((double)(x[2] - c[2]) < 0.5) c[2]--;
    const R w1 = weight_centre(g, p, 0, c[0]), w2 = weight_face(g, p, 1, c[1]), w3 = weight_centre(g, p, 2, c[2]);
    return TRILERP(g->v, v_id);
}

/* sampleW :403-452 */
R GORC_NAME(sample_w)(const grid_t *g, const R p[3]) {
    R x[3];
    int c[3];
    if (!sample_cell(g, p, x, c)) return 0;
    if ((double)(x[0] - c[0]) < 0.5) c[0]--;
    if ((double)(x[1] - c[1]) < 0.5) c[1]--;
    const R w1 = weight_centre(g, p, 0, c[0]), w2 = weight_centre(g, p, 1, c[1]), w3 = weight_face(g, p, 2, c[2]);
    return TRILERP(g->w, w_id);
}

/* position of a face / cell centre along one axis: `xmin + h * i` in float, `xmin + h * (i + 0.5)` in double */
static inline R at_face(const grid_t *g, int a, int i) { return g->xmin[a] + g->h * i; }
static inline R at_centre(const grid_t *g, int a, int i) {
    return (R)((double)g->xmin[a] + (double)g->h * ((double)i + 0.5));
}
static inline R quarter(R a, R b, R c, R d) { return (R)((double)(a + b + c + d) * 0.25); }

/* xtmp + vel * dt * s, the cfloat3 expression of :477, :483 (s = -0.5 or -1 rounded to float) */
static inline void back_trace(const grid_t *g, const R xtmp[3], const R vel[3], R s, R out[3]) {
    for (int a = 0; a < 3; a++) out[a] = xtmp[a] + vel[a] * g->dt * s;
}

/* advect :456-565: semi-Lagrangian with a midpoint trace for v and w; the u component is traced one
 * full step along x only (`xmid2`, :485-489 -- the midpoint result is computed and not used) */
void GORC_NAME(advect)(grid_t *g) {
    const int nx = g->dim[0], ny = g->dim[1], nz = g->dim[2];
    const R *u = g->u, *v = g->v, *w = g->w;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx + 1; i++) {
                R utmp[3], xtmp[3], x2[3];
                utmp[0] = u[u_id(g, i, j, k)];
                utmp[1] = 0; /* :485-486 */
                utmp[2] = 0;
                xtmp[0] = at_face(g, 0, i);
                xtmp[1] = at_centre(g, 1, j);
                xtmp[2] = at_centre(g, 2, k);
                back_trace(g, xtmp, utmp, (R)-1, x2);
                g->uadv[u_id(g, i, j, k)] = GORC_NAME(sample_u)(g, x2);
            }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz; k++)
        for (int j = 0; j < ny + 1; j++)
            for (int i = 0; i < nx; i++) {
                R utmp[3], utmp1[3], xtmp[3], xmid[3];
                utmp[0] = quarter(u[u_id(g, i, j, k)], u[u_id(g, i + 1, j, k)], u[u_id(g, i, j - 1, k)],
                                  u[u_id(g, i + 1, j - 1, k)]);
                utmp[1] = v[v_id(g, i, j, k)];
                utmp[2] = quarter(w[w_id(g, i, j, k)], w[w_id(g, i, j, k + 1)], w[w_id(g, i, j - 1, k)],
                                  w[w_id(g, i, j - 1, k + 1)]);
                xtmp[0] = at_centre(g, 0, i);
                xtmp[1] = at_face(g, 1, j);
                xtmp[2] = at_centre(g, 2, k);
                back_trace(g, xtmp, utmp, (R)-0.5, xmid);
                utmp1[0] = GORC_NAME(sample_u)(g, xmid);
                utmp1[1] = GORC_NAME(sample_v)(g, xmid);
                utmp1[2] = GORC_NAME(sample_w)(g, xmid);
                back_trace(g, xtmp, utmp1, (R)-1, xmid);
                g->vadv[v_id(g, i, j, k)] = GORC_NAME(sample_v)(g, xmid);
            }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nz + 1; k++)
        for (int j = 0; j < ny; j++)
            for (int i = 0; i < nx; i++) {
                R utmp[3], utmp1[3], xtmp[3], xmid[3];
                utmp[0] = quarter(u[u_id(g, i, j, k)], u[u_id(g, i, j, k - 1)], u[u_id(g, i + 1, j, k)],
                                  u[u_id(g, i + 1, j, k - 1)]);
                utmp[1] = quarter(v[v_id(g, i, j, k)], v[v_id(g, i, j, k - 1)], v[v_id(g, i, j + 1, k)],
                                  v[v_id(g, i, j + 1, k - 1)]);
                utmp[2] = w[w_id(g, i, j, k)];
                xtmp[0] = at_centre(g, 0, i);
                xtmp[1] = at_centre(g, 1, j);
                xtmp[2] = at_face(g, 2, k);
                back_trace(g, xtmp, utmp, (R)-0.5, xmid);
                utmp1[0] = GORC_NAME(sample_u)(g, xmid);
                utmp1[1] = GORC_NAME(sample_v)(g, xmid);
                utmp1[2] = GORC_NAME(sample_w)(g, xmid);
                back_trace(g, xtmp, utmp1, (R)-1, xmid);
                g->wadv[w_id(g, i, j, k)] = GORC_NAME(sample_w)(g, xmid);
            }
    /* swapBuffer :573-577 */
    R *t;
    t = g->u; g->u = g->uadv; g->uadv = t;
    t = g->v; g->v = g->vadv; g->vadv = t;
    t = g->w; g->w = g->wadv; g->wadv = t;
}

/* bodyForce :579-588: `10 * dt` on every v face (gravity points along +y in this solver's sign convention) */
void GORC_NAME(body_force)(grid_t *g) {
    for (int i = 0; i < g->dim[0]; i++)
        for (int j = 0; j < g->dim[1] + 1; j++)
            for (int k = 0; k < g->dim[2]; k++) g->v[v_id(g, i, j, k)] += 10 * g->dt;
}

/* testcase :590-606: u = 2 i, v = 3 j, w = 4 k on the faces */
void GORC_NAME(testcase)(grid_t *g) {
    const int nx = g->dim[0], ny = g->dim[1], nz = g->dim[2];
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++)
            for (int k = 0; k < nz; k++) {
                g->u[u_id(g, i, j, k)] = (R)(i * 2);
                g->v[v_id(g, i, j, k)] = (R)(j * 3);
                g->w[w_id(g, i, j, k)] = (R)(k * 4);
                if (i == nx - 1) g->u[u_id(g, i + 1, j, k)] = (R)((i + 1) * 2);
                if (j == ny - 1) g->v[v_id(g, i, j + 1, k)] = (R)((j + 1) * 3);
                if (k == nz - 1) g->w[w_id(g, i, j, k + 1)] = (R)((k + 1) * 4);
            }
}

/* advectParticles :639-651 */
void GORC_NAME(advect_particles)(grid_t *g) {
    for (int i = 0; i < g->n_particles; i++) {
        R *x = g->pos + 3 * (size_t)i;
        R vel[3];
        vel[0] = GORC_NAME(sample_u)(g, x);
        vel[1] = GORC_NAME(sample_v)(g, x);
        vel[2] = GORC_NAME(sample_w)(g, x);
        for (int a = 0; a < 3; a++) x[a] += vel[a] * g->dt;
        R l = 1;
        g->color[4 * (size_t)i + 0] = (R)((double)R_ABS(vel[0]) / 0.01);
        g->color[4 * (size_t)i + 1] = 0;
        g->color[4 * (size_t)i + 2] = R_ABS(vel[2]) / l;
    }
}

/* Step :608-619 */
void GORC_NAME(step)(grid_t *g, int maxiter) {
    GORC_NAME(advect)(g);
    GORC_NAME(advect_particles)(g);
    GORC_NAME(body_force)(g);
    GORC_NAME(make_rhs)(g);
    GORC_NAME(solve)(g, maxiter, NULL);
    GORC_NAME(update_u)(g);
    GORC_NAME(div_velocity)(g);
    g->frame++;
}
