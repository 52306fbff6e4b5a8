// The solver class main.cpp instantiates (solver/MultiphaseSPHSolver.h:5-61 in the reference, under
// /root/reference/particle_solver): scene setup from config/sph_scene.xml and a boundary file, then one
// Step() per frame.
//
// The reference's Setup() always selects its IISPH step; which step runs here is chosen at run time:
// environment variable FSA_RUN_MODE ("SPH" = WCSPH, the default, "DFSPH" or "IISPH"), or a
// <RunMode> tag in the scene with one of those words.  FSA_RESUME=<file> continues from a text checkpoint (key 'b').  Two more optional scene tags exist for the WCSPH step, which has no wall
// response for TYPE_RIGID particles: <WallType> (particle type given to the boundary file's particles,
// default 4 = TYPE_RIGID as LoadPO assigns), <WallRestDensity>, and <BoundaryMode> (0 verbatim
// integration, 1 legacy clamp to the grid box).  <GridMin> / <GridMax> give a scene its own grid box
// instead of the <Fluid> block's, <BoundaryFile> its own boundary file instead of "big box.xml".
#pragma once

#include <string>

#include "sph_solver.h"

class MultiphaseSPHSolver : public SPHSolver {
protected:
    vecf vol_frac;
    veci localid;
    int localid_counter[100] = {0};
    int wall_type = TYPE_RIGID;
    float wall_rest_density = 0.f;
    std::string boundary_file = "big box.xml";

    const char *config_path(const char *file);  // inside FSA_CONFIG_DIR (default "config")

public:
    void Setup();

    void SetupHostBuffer();
    int AddDefaultParticle();
    void AddMultiphaseFluidVolumes();

    /* Rigid bodies */
    void LoadPO(ParticleObject *po);

    // Setup Scenes
    void ParseParam(const char *xmlpath);
    void LoadParam(const char *xmlpath);
    void BuildHostScene();
    void SetupFluidScene();
    void ClearParticles();
    void SetupSPH();
    void SetupDFSPH();
    void SetupIISPH();

    void DumpSimulationDataText() override;
    void LoadSimulationDataText(const char *filepath);
    void DumpRenderData() override;
};
