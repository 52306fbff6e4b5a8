/*
 * grid_oracle.h -- CPU restatement of the reference's Eulerian grid solver
 * (solver/grid_solver.{h,cpp} under /root/reference/particle_solver): semi-Lagrangian
 * advection on a staggered grid, gravity, pressure projection by conjugate
 * gradients, tracer particles.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP grid solver is compared
 * with.  Nothing in the product links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests or fixtures for this solver, and
 * its source does not build in this image (its include chain ends in GL/glu.h,
 * which the image lacks), so nothing here has been run against the reference.
 * The restatement follows the reference text expression by expression (each
 * function cites file:line) and is anchored by the known answers in
 * tests/test_oracle_grid.py.
 *
 * Arithmetic conventions are those of sph_oracle.h: `scalar` variables are
 * float, unsuffixed literals promote the enclosing sub-expression to double,
 * sqrt/floor/abs of a float are the float overloads, no FMA contraction.
 * Two builds: real = float (symbols gorc_*) and -DORC_REAL_IS_DOUBLE
 * (symbols gorc64_*), in which every float is widened to double.
 */
#ifndef GRID_ORACLE_H
#define GRID_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORC_REAL_IS_DOUBLE
typedef double gorc_real;
#define GORC_NAME(x) gorc64_##x
#else
typedef float gorc_real;
#define GORC_NAME(x) gorc_##x
#endif

/* struct Grid + the solver's fields, solver/grid_solver.h:8-82 */
typedef struct GORC_NAME(grid) {
    int dim[3];
    int padding;
    int xlen, ylen, zlen, ulen, vlen, wlen;
    int Size, dimSize, uSize, vSize, wSize;
    gorc_real xmin[3];
    gorc_real h, dt, rho;
    gorc_real *u, *v, *w, *uadv, *vadv, *wadv; /* face-centred, padded */
    gorc_real *p, *divU, *b, *Aq, *r, *q;      /* cell-centred, dimSize */
    gorc_real divUsum;
    int frame;
    int n_particles;
    gorc_real *pos;   /* n x 3 */
    gorc_real *color; /* n x 4 */
} GORC_NAME(grid);

/* loadConfig :14-40 (values instead of the XML file) + allocate :42-66 */
GORC_NAME(grid) *GORC_NAME(create)(const int dim[3], int padding, gorc_real h, gorc_real dt, gorc_real rho);
void GORC_NAME(destroy)(GORC_NAME(grid) *g);
/* tracer particles: copies n positions in, colours start as (1,1,1,1) (addParticles :625-637) */
void GORC_NAME(set_particles)(GORC_NAME(grid) *g, int n, const gorc_real *pos3);

void GORC_NAME(mvproduct)(const GORC_NAME(grid) *g, const gorc_real *v, gorc_real *dst); /* :72-115 */
gorc_real GORC_NAME(dotproduct)(const GORC_NAME(grid) *g, const gorc_real *a, const gorc_real *b); /* :117-122 */
void GORC_NAME(div_velocity)(GORC_NAME(grid) *g);   /* :124-141 */
void GORC_NAME(make_rhs)(GORC_NAME(grid) *g);       /* :143-160 */
/* solve :162-214; returns the value of `iter` when the loop ends (completed rounds - 1 on convergence),
 * *residual = last mean |r|; maxiter < 0 = no cap, as in the reference */
int GORC_NAME(solve)(GORC_NAME(grid) *g, int maxiter, gorc_real *residual);
void GORC_NAME(update_u)(GORC_NAME(grid) *g);       /* :216-266 */
gorc_real GORC_NAME(sample_u)(const GORC_NAME(grid) *g, const gorc_real p[3]); /* :301-350 */
gorc_real GORC_NAME(sample_v)(const GORC_NAME(grid) *g, const gorc_real p[3]); /* :352-401 */
gorc_real GORC_NAME(sample_w)(const GORC_NAME(grid) *g, const gorc_real p[3]); /* :403-452 */
void GORC_NAME(advect)(GORC_NAME(grid) *g);           /* :456-565 */
void GORC_NAME(body_force)(GORC_NAME(grid) *g);       /* :579-588 */
void GORC_NAME(testcase)(GORC_NAME(grid) *g);         /* :590-606 */
void GORC_NAME(advect_particles)(GORC_NAME(grid) *g); /* :639-651 */
void GORC_NAME(step)(GORC_NAME(grid) *g, int maxiter); /* :608-619 */

#ifdef __cplusplus
}
#endif
#endif
