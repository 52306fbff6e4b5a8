// The Eulerian grid solver of the plugin surface (solver/grid_solver.h:8-108 in the reference, under
// /root/reference/particle_solver): same Grid description, same public methods, one Step() per frame; the
// loops run on the GPU behind the fsa_grid_* entries of the C ABI.  The field pointers of the reference
// class (u, v, w, p, ...) are device arrays here; `field()` copies one to the host for inspection.
#pragma once

#include <string>
#include <vector>

#include "../../../include/fluidsolver_amd.h"
#include "ParticleSolver.h"

typedef float scalar;

// struct Grid, solver/grid_solver.h:8-50
struct Grid {
    cint3 dim;    // effective resolution
    int padding = 0;
    int xlen = 0, ylen = 0, zlen = 0;  // cell centre dimension
    int ulen = 0, vlen = 0, wlen = 0;  // face dimension
    int Size = 0;                      // full cell number
    int dimSize = 0;                   // effective cell number
    int uSize = 0, vSize = 0, wSize = 0;
    cfloat3 xmin;  // coordinates of the low corner
    scalar h = 0;  // cell size

    void setSize() {
        xlen = dim.x + padding * 2;
        ylen = dim.y + padding * 2;
        zlen = dim.z + padding * 2;
        ulen = xlen + 1;
        vlen = ylen + 1;
        wlen = zlen + 1;
        Size = xlen * ylen * zlen;
        uSize = ulen * ylen * zlen;
        vSize = xlen * vlen * zlen;
        wSize = xlen * ylen * wlen;
        dimSize = dim.x * dim.y * dim.z;
    }
    int cellId(int x, int y, int z) { return (z * dim.y + y) * dim.x + x; }
    int uId(int x, int y, int z) { return ((z + padding) * ylen + y + padding) * ulen + x + padding; }
    int vId(int x, int y, int z) { return ((z + padding) * vlen + y + padding) * xlen + x + padding; }
    int wId(int x, int y, int z) { return ((z + padding) * ylen + y + padding) * xlen + x + padding; }
};

class GridSolver : public ParticleSolver {
public:
    Grid grid;
    scalar rho = 0;
    scalar divUsum = 0;
    int frame = 0;
    int pad = 0;
    int max_cg_iterations = -1;  // < 0: unbounded, as the reference's loop; FSA_GRID_MAXITER overrides
    int readback_every = 1;      // tracer positions are copied back every k-th Step (FSA_READBACK_EVERY)

    ~GridSolver() override;

    void Step() override;
    void HandleKeyEvent(char key) override;

    void setup();
    void loadConfig();
    void allocate();

    void testcase();
    void divVelocity();
    void makeRHS();
    void solve();
    void updateU();
    void bodyForce();
    void advect();

    void addParticles();
    void advectParticles();

    // host copy of one named field ("u", "v", "w", "p", "divU", "b", ...)
    std::vector<scalar> field(const char *name);
    int last_cg_iterations = 0;
    double last_cg_residual = 0.0;

private:
    fsa_grid *device = nullptr;
    fsa_grid_team *team = nullptr;  // FSA_GRID_SLABS=N: the grid as N z-slabs instead
    void readBackParticles();
};
