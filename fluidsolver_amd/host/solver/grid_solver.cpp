#include "grid_solver.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "xml_lite.h"

namespace {
void check(int rc, const char *what) {
    if (rc != FSA_OK) {
        std::fprintf(stderr, "%s: %s\n", what, fsa_last_error());
        std::exit(EXIT_FAILURE);
    }
}
}  // namespace

GridSolver::~GridSolver() {
    if (device) fsa_grid_destroy(device);
    if (team) fsa_grid_team_destroy(team);
}

// setup, solver/grid_solver.cpp:5-12
void GridSolver::setup() {
    loadConfig();
    allocate();
    pos.clear();
    color.clear();
    addParticles();
}

// loadConfig :14-40: <Param> of config/Eulerian.xml
void GridSolver::loadConfig() {
    const char *dir = std::getenv("FSA_CONFIG_DIR");
    const std::string path = std::string(dir ? dir : "config") + "/Eulerian.xml";
    xml_lite::Document doc;
    if (!doc.load_file(path.c_str())) {
        std::fprintf(stderr, "%s: %s\n", path.c_str(), doc.error().c_str());
        std::exit(EXIT_FAILURE);
    }
    xml_lite::TagReader reader;
    reader.Use(doc.first_child("Param"));
    grid.dim = reader.GetInt3("dim");
    grid.padding = reader.GetInt("padding");
    grid.h = reader.GetFloat("h");
    grid.xmin.x = -grid.dim.x / 2 * grid.h;
    grid.xmin.y = 0;
    grid.xmin.z = -grid.dim.z / 2 * grid.h;
    domainMin = grid.xmin;
    domainMax = cfloat3(grid.xmin.x + grid.dim.x * grid.h, grid.xmin.y + grid.dim.y * grid.h,
                        grid.xmin.z + grid.dim.z * grid.h);
    grid.setSize();
    std::printf("grid dimension: %d %d %d %f \n", grid.dim.x, grid.dim.y, grid.dim.z, grid.h);
    dt = reader.GetFloat("dt");
    rho = reader.GetFloat("rho");
    frame = 0;
    pad = grid.padding;
    if (const char *e = std::getenv("FSA_GRID_MAXITER")) max_cg_iterations = std::atoi(e);
    if (const char *e = std::getenv("FSA_READBACK_EVERY")) readback_every = std::atoi(e) > 0 ? std::atoi(e) : 1;
}

// allocate :42-66: every field on the device, cleared
// FSA_GRID_SLABS=N (this build): the grid as a team of N z-slabs of equal thickness on the one GPU, a rehearsal
// of the one-slab-per-GPU form with the same results; FSA_GRID_REACH = planes a back-trace may cross (default 1).
void GridSolver::allocate() {
    if (device) fsa_grid_destroy(device);
    if (team) fsa_grid_team_destroy(team);
    device = nullptr;
    team = nullptr;
    const int dim[3] = {grid.dim.x, grid.dim.y, grid.dim.z};
    const char *dev = std::getenv("FSA_DEVICE");
    const char *slabs = std::getenv("FSA_GRID_SLABS");
    const int n = slabs ? std::atoi(slabs) : 0;
    if (n > 1) {
        const char *r = std::getenv("FSA_GRID_REACH");
        std::vector<int> cuts((size_t)n + 1);
        for (int k = 0; k <= n; k++) cuts[(size_t)k] = (int)((long long)k * dim[2] / n);
        check(fsa_grid_team_create_local(dim, grid.padding, grid.h, dt, rho, FSA_PRECISION_F32, dev ? std::atoi(dev) : 0, n,
                                         cuts.data(), r ? std::atoi(r) : 1, &team),
              "fsa_grid_team_create_local");
        std::printf("grid on %d z-slabs\n", n);
        return;
    }
    check(fsa_grid_create(dim, grid.padding, grid.h, dt, rho, FSA_PRECISION_F32, dev ? std::atoi(dev) : 0, &device),
          "fsa_grid_create");
}

// addParticles :625-637: a 0.2 x 0.1 x 0.2 block of tracers, float loop counters
void GridSolver::addParticles() {
    const cfloat3 xmin(-0.1f, 0.01f, -0.1f);
    for (float i = 0; i < 0.2; i += 0.01)
        for (float j = 0; j < 0.1; j += 0.01)
            for (float k = 0; k < 0.2; k += 0.01) {
                pos.push_back(cfloat3(xmin.x + i, xmin.y + j, xmin.z + k));
                color.push_back(cfloat4(1, 1, 1, 1));
            }
    numParticles = (int)pos.size();
    if (team) check(fsa_grid_team_set_particles(team, numParticles, pos.data()), "fsa_grid_team_set_particles");
    else check(fsa_grid_set_particles(device, numParticles, pos.data()), "fsa_grid_set_particles");
}

void GridSolver::readBackParticles() {
    int n = 0;
    if (team) {
        check(fsa_grid_team_sync(team, nullptr), "grid team");  // (a back-trace beyond the reach ends the program)
        check(fsa_grid_team_get_particles(team, pos.data(), color.data(), &n), "fsa_grid_team_get_particles");
        return;
    }
    check(fsa_grid_get_particles(device, pos.data(), color.data(), &n), "fsa_grid_get_particles");
}

void GridSolver::testcase() {
    if (team) {
        std::fprintf(stderr, "testcase: not available on z-slabs (FSA_GRID_SLABS)\n");
        std::exit(EXIT_FAILURE);
    }
    check(fsa_grid_testcase(device), "testcase");
}

void GridSolver::divVelocity() {
    double s = 0.0;
    check(team ? fsa_grid_team_div_velocity(team, &s) : fsa_grid_div_velocity(device, &s), "divVelocity");
    divUsum = (scalar)s;
    std::printf("div sum %f\n", divUsum);
}

void GridSolver::makeRHS() { check(team ? fsa_grid_team_make_rhs(team) : fsa_grid_make_rhs(device), "makeRHS"); }

void GridSolver::solve() {
    check(team ? fsa_grid_team_solve(team, max_cg_iterations, &last_cg_iterations, &last_cg_residual)
               : fsa_grid_solve(device, max_cg_iterations, &last_cg_iterations, &last_cg_residual),
          "solve");
    std::printf("turn %d residual: %f\n", last_cg_iterations, last_cg_residual);
}

void GridSolver::updateU() {
    check(team ? fsa_grid_team_update_u(team) : fsa_grid_update_u(device), "updateU");
    std::printf("time step size updated as %f.\n", dt);
}

void GridSolver::bodyForce() { check(team ? fsa_grid_team_body_force(team) : fsa_grid_body_force(device), "bodyForce"); }
void GridSolver::advect() { check(team ? fsa_grid_team_advect(team) : fsa_grid_advect(device), "advect"); }
void GridSolver::advectParticles() {
    check(team ? fsa_grid_team_advect_particles(team) : fsa_grid_advect_particles(device), "advectParticles");
}

// Step :608-619
void GridSolver::Step() {
    advect();
    advectParticles();
    bodyForce();
    makeRHS();
    solve();
    updateU();
    divVelocity();
    frame++;
    if (frame % readback_every == 0) readBackParticles();
}

void GridSolver::HandleKeyEvent(char) {}

std::vector<scalar> GridSolver::field(const char *name) {
    // Grid::setSize, solver/grid_solver.h:23-35
    const int sizes[4] = {grid.uSize, grid.vSize, grid.wSize, grid.dimSize};
    const std::string k(name);
    const int n = k[0] == 'u' ? sizes[0] : k[0] == 'v' ? sizes[1] : k[0] == 'w' ? sizes[2] : sizes[3];
    std::vector<scalar> out((size_t)n, (scalar)0);
    if (team) check(fsa_grid_team_get_field(team, name, out.data(), out.size() * sizeof(scalar)), "fsa_grid_team_get_field");
    else check(fsa_grid_get_field(device, name, out.data(), out.size() * sizeof(scalar)), "fsa_grid_get_field");
    return out;
}
