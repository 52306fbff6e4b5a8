// Host side of the particle step on top of the C ABI (include/fluidsolver_amd.h).
//
// Mirrors the reference's SPHSolver (solver/sph_solver.h:18-92, sph_solver.cpp under
// /root/reference/particle_solver): host vectors in the layouts of the plugin surface, one device
// solver, and the call sequence of SortParticles / SolveSPH / Step.  Where the reference keeps a struct
// of raw device pointers (SimData_SPH) and calls CUDA launchers, this class keeps an fsa_solver handle
// and calls the extern "C" entry points; no device code is compiled into the host layer.
#pragma once

#include <vector>

#include "../../../include/fluidsolver_amd.h"
#include "ParticleSolver.h"
#include "scene_setup.h"

// run modes, solver/sph_solver.h:11-15
#define SPH 0
#define DFSPH 1
#define MSPH 2
#define MSPH_REN 3
#define IISPH 4

class SPHSolver : public ParticleSolver {
protected:
    veci unique_id;
    veci id_table;
    vecf3 normal;
    veci type;
    // per-particle material data every particle carries on the device
    veci group;
    vecf mass;
    vecf rest_density;
    vecf3 v_star;

    fsa_solver *device = nullptr;  // replaces SimData_SPH device_data
    // FSA_SLABS=N (slab_mode.cpp): the scene on N z-slab solvers driven as one group, `device` released
    std::vector<fsa_solver *> slab_devices;
    fsa_slab_group *slab_group = nullptr;
    fsa_slab_iisph *slab_team = nullptr;  // ... or, for the IISPH and DFSPH steps, as a team (fsa_slab_iisph_* / _dfsph_*)
    bool slab_team_dfsph = false;
    std::vector<int> slab_cuts;
    vecf4 slab_color;  // colours by slab id
    fsa_params hParam{};           // replaces the global hParam (solver/sph_kernel_impl.cuh:13)
    // parameters of the scene file the device step does not use
    int maxpnum = 0;
    int maxtypenum = 0;
    float restdensity = 0.f;
    float densArr[10] = {0.f};
    float viscArr[10] = {0.f};

    std::vector<fluidvol> fluid_volumes;

    int num_fluid_particles = 0;
    int num_grid_cells = 0;
    int frame_count = 0;
    int dump_count = 0;
    int case_id = 0;
    int run_mode = SPH;
    int readback_every = 1;  // host copy of pos/color every k-th Step (FSA_READBACK_EVERY)
    float time = 0.f;
    bool emit_particle_on = false;

    // prints the library's message and ends the process, as the reference does on a CUDA error
    void check(int rc, const char *what);

public:
    ~SPHSolver() override;

    void Step() override;
    void HandleKeyEvent(char key) override;
    void Eval(const char *expression) override;

    void SetupDeviceBuffer();
    void CopyParticleDataToDevice();
    void CopyPosColorFromDevice();
    void CopySimulationDataFromDevice();
    void CopyParam2Device();

    void SortParticles();
    void SolveSPH();
    void SolveDFSPH();
    void SolveIISPH();

    int AddDefaultParticle();

    void EnterSlabMode(int nslabs);
    void EnterSlabModeIISPH(int nslabs, bool dfsph = false);
    void LeaveSlabMode();
    void StepSlabs();
    void CopyPosColorFromSlabs();

    virtual void DumpSimulationDataText();
    virtual void DumpRenderData();

    int GetFrameCount() const { return frame_count; }
    fsa_solver *GetDeviceSolver() { return device; }
};
