#include "sph_solver.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

void SPHSolver::check(int rc, const char *what) {
    if (rc == FSA_OK) return;
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, fsa_last_error());
    std::exit(EXIT_FAILURE);
}

SPHSolver::~SPHSolver() {
    LeaveSlabMode();
    if (device) fsa_destroy(device);
}

// AddDefaultParticle, solver/MultiphaseSPHSolver.cpp:483-514: one more particle with default values,
// its unique id is its index at creation.
int SPHSolver::AddDefaultParticle() {
    pos.push_back(cfloat3(0, 0, 0));
    color.push_back(cfloat4(1, 1, 1, 1));
    normal.push_back(cfloat3(0, 0, 0));
    unique_id.push_back((int)pos.size() - 1);
    vel.push_back(cfloat3(0, 0, 0));
    type.push_back(TYPE_FLUID);
    adjacentIndex.push_back(adjacent());
    mass.push_back(0.f);
    group.push_back(0);
    rest_density.push_back(0.f);
    v_star.push_back(cfloat3(0, 0, 0));
    return (int)pos.size() - 1;
}

// SetupDeviceBuffer (solver/MultiphaseSPHSolver.cpp:871-977): every device array is sized by the number
// of particles present at this point.
void SPHSolver::SetupDeviceBuffer() {
    const int maxp = (int)pos.size();
    id_table.resize(maxp);
    if (device) {
        fsa_destroy(device);
        device = nullptr;
    }
    int dev = 0;
    if (const char *e = std::getenv("FSA_DEVICE")) dev = std::atoi(e);
    check(fsa_create(&hParam, FSA_PRECISION_F32, maxp, dev, &device), "fsa_create");
    if (const char *e = std::getenv("FSA_VARIANT")) check(fsa_set_variant(device, std::atoi(e)), "fsa_set_variant");
    if (const char *e = std::getenv("FSA_READBACK_EVERY")) readback_every = std::atoi(e) > 0 ? std::atoi(e) : 1;
}

// CopyParticleDataToDevice, solver/MultiphaseSPHSolver.cpp:10-27
void SPHSolver::CopyParticleDataToDevice() {
    numParticles = (int)pos.size();
    check(fsa_upload(device, numParticles, pos.data(), vel.data(), color.data(), normal.data(), type.data(),
                     group.data(), unique_id.data(), mass.data(), rest_density.data()),
          "fsa_upload");
}

void SPHSolver::CopyParam2Device() { check(fsa_copy_param_to_device(device, &hParam), "fsa_copy_param_to_device"); }

// CopyPosColorFromDevice, solver/sph_solver.cpp:35-38: current device order.
void SPHSolver::CopyPosColorFromDevice() {
    if (slab_group || slab_team) {
        CopyPosColorFromSlabs();
        return;
    }
    int n = 0;
    check(fsa_readback_pos_color(device, pos.data(), color.data(), &n), "fsa_readback_pos_color");
}

// CopySimulationDataFromDevice (solver/MultiphaseSPHSolver.cpp:29-46): everything the text dump prints.
void SPHSolver::CopySimulationDataFromDevice() {
    if (slab_group || slab_team) {
        std::fprintf(stderr, "dumps need velocities, groups and v_star in one place: not available with FSA_SLABS\n");
        std::exit(EXIT_FAILURE);
    }
    const size_t n = (size_t)numParticles;
    check(fsa_get_buffer(device, "pos", pos.data(), n * sizeof(cfloat3)), "get pos");
    check(fsa_get_buffer(device, "vel", vel.data(), n * sizeof(cfloat3)), "get vel");
    check(fsa_get_buffer(device, "color", color.data(), n * sizeof(cfloat4)), "get color");
    check(fsa_get_buffer(device, "type", type.data(), n * sizeof(int)), "get type");
    check(fsa_get_buffer(device, "group", group.data(), n * sizeof(int)), "get group");
    check(fsa_get_buffer(device, "mass", mass.data(), n * sizeof(float)), "get mass");
    check(fsa_get_buffer(device, "uniqueId", unique_id.data(), n * sizeof(int)), "get uniqueId");
    check(fsa_get_buffer(device, "v_star", v_star.data(), n * sizeof(cfloat3)), "get v_star");
}

// SortParticles, solver/sph_solver.cpp:50-85
void SPHSolver::SortParticles() {
    check(fsa_calc_hash(device), "calcHash");
    check(fsa_sort_particle(device), "sortParticle");
    check(fsa_reorder_data_and_find_cell_start(device), "reorderDataAndFindCellStart");
}

// SolveSPH, solver/sph_solver.cpp:87-98 (the host copy is taken in Step)
void SPHSolver::SolveSPH() {
    SortParticles();
    check(fsa_compute_pressure(device), "ComputePressure");
    check(fsa_compute_force(device), "computeForce");
    check(fsa_advect(device), "Advect");
}

// SolveDFSPH, solver/sph_solver.cpp:108-120 (5 Jacobi rounds at most, threshold 1, no debug print)
void SPHSolver::SolveDFSPH() {
    check(fsa_compute_non_pressure_force(device), "computeNonPForce");
    check(fsa_correct_density_error(device, 5, 1, nullptr, nullptr), "correctDensityError");
    SortParticles();
    check(fsa_compute_density_alpha(device), "computeDensityAlpha");
    check(fsa_correct_divergence_error(device, 5, 1, nullptr, nullptr), "correctDivergenceError");
}

// SolveIISPH, solver/MultiphaseSPHSolver.cpp:173-188; ComputeTension and UpdateSolidState act on
// TYPE_DEFORMABLE particles only and have no counterpart here
void SPHSolver::SolveIISPH() {
    check(fsa_non_pressure_force_multiphase(device), "NonPressureForce_Multiphase");
    check(fsa_iisph_predict_density(device), "IISPHPredictDensity");
    check(fsa_iisph_solve_pressure(device, 10, nullptr, nullptr), "IISPHSolvePressure");
    check(fsa_iisph_update(device), "IISPHUpdate");
    SortParticles();
    check(fsa_iisph_factor(device), "IISPHFactor");
}

// Step, solver/sph_solver.cpp:129-142 and solver/MultiphaseSPHSolver.cpp:199-217
void SPHSolver::Step() {
    if (numParticles > 0 && (slab_group || slab_team)) {
        StepSlabs();
    } else if (numParticles > 0) {
        switch (run_mode) {
        case SPH:
            SolveSPH();
            break;
        case DFSPH:
            SolveDFSPH();
            break;
        case IISPH:
            SolveIISPH();
            break;
        default:
            std::fprintf(stderr, "run mode %d is not available in this build (SPH, DFSPH and IISPH are)\n", run_mode);
            std::exit(EXIT_FAILURE);
        }
    }
    frame_count++;
    time += (float)hParam.dt;
    if (numParticles > 0 && frame_count % readback_every == 0) CopyPosColorFromDevice();
}

void SPHSolver::Eval(const char *expression) {
    if (std::strcmp(expression, "DumpRenderData") == 0) DumpRenderData();
}

// key bindings of solver/MultiphaseSPHSolver.cpp:309-331
void SPHSolver::HandleKeyEvent(char key) {
    switch (key) {
    case 'b':
        DumpSimulationDataText();
        break;
    case 'r':
        DumpRenderData();
        break;
    case 'e':
        emit_particle_on = !emit_particle_on;
        std::printf(emit_particle_on ? "Start to emit particles.\n" : "Stop emitting particles.\n");
        break;
    case 'm':
        // the reference also moves its constraint box away here, which belongs to the deformable
        // solver; the rigid particle volumes are what the fluid step sees
        if (device && numParticles > 0) {
            SortParticles();
            check(fsa_rigid_particle_volume(device), "RigidParticleVolume");
        }
        break;
    }
}

static FILE *open_numbered(const char *env, const char *fallback_dir, int number) {
    const char *dir = std::getenv(env);
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%03d.txt", dir ? dir : fallback_dir, number);
    FILE *fp = std::fopen(path, "w+");
    if (!fp) std::printf("error opening file\n");
    return fp;
}

// Text checkpoint, one particle per line (solver/sph_solver.cpp:144-176):
//   n, then per particle "x y z vx vy vz type group mass uniqueId vsx vsy vsz"
void SPHSolver::DumpSimulationDataText() {
    std::printf("Dumping simulation data at frame %d\n", frame_count);
    FILE *fp = open_numbered("FSA_DUMP_DIR", "dump", frame_count);
    if (!fp) return;
    CopySimulationDataFromDevice();
    std::fprintf(fp, "%d\n", numParticles);
    for (int i = 0; i < numParticles; i++) {
        std::fprintf(fp, "%f %f %f ", pos[i].x, pos[i].y, pos[i].z);
        std::fprintf(fp, "%f %f %f ", vel[i].x, vel[i].y, vel[i].z);
        std::fprintf(fp, "%d %d %f %d ", type[i], group[i], mass[i], unique_id[i]);
        std::fprintf(fp, "%f %f %f \n", v_star[i].x, v_star[i].y, v_star[i].z);
    }
    std::fclose(fp);
}

// Renderer hand-off: "frame k", then per particle "index x y z group"
void SPHSolver::DumpRenderData() {
    FILE *fp = open_numbered("FSA_RENDER_DIR", "../particle_data", dump_count++);
    if (!fp) return;
    CopySimulationDataFromDevice();
    std::fprintf(fp, "frame %d\n", frame_count);
    for (int i = 0; i < numParticles; i++)
        std::fprintf(fp, "%d %f %f %f %d\n", i, pos[i].x, pos[i].y, pos[i].z, group[i]);
    std::fclose(fp);
}
