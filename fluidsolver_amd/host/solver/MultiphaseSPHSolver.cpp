#include "MultiphaseSPHSolver.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using xml_lite::Element;
using xml_lite::TagReader;

const char *MultiphaseSPHSolver::config_path(const char *file) {
    static std::string path;
    const char *dir = std::getenv("FSA_CONFIG_DIR");
    path = std::string(dir ? dir : "config") + "/" + file;
    return path.c_str();
}

// ParseParam, solver/MultiphaseSPHSolver.cpp:336-431: <Fluid> carries gravity, the grid box and the scene
// id; the <MultiScene id=...> of that id carries the rest.  Tags that are absent read as 0.
void MultiphaseSPHSolver::ParseParam(const char *xmlpath) {
    std::printf("Parsing XML.\n");
    xml_lite::Document doc;
    if (!doc.load_file(xmlpath)) {
        std::printf("%s: %s\n", xmlpath, doc.error().c_str());
        return;
    }
    const Element *fluid = doc.first_child("Fluid");
    const Element *scene = doc.first_child("MultiScene");
    const Element *bound = doc.first_child("BoundInfo");
    if (!fluid || !scene) {
        std::printf("missing fluid/scene xml node");
        return;
    }
    TagReader reader;
    reader.Use(fluid);
    const cfloat3 g = reader.GetFloat3("Gravity");
    cfloat3 lo = reader.GetFloat3("VolMin"), hi = reader.GetFloat3("VolMax");
    hParam.gravity[0] = g.x; hParam.gravity[1] = g.y; hParam.gravity[2] = g.z;
    case_id = reader.GetInt("SceneId");
    if (const char *e = std::getenv("FSA_SCENE_ID")) case_id = std::atoi(e);

    for (;;) {
        int id = 0;
        scene->int_attribute("id", &id);
        if (id == case_id) break;
        scene = scene->next_sibling();
        if (!scene) {
            std::printf("scene not found.\n");
            return;
        }
    }
    reader.Use(scene);
    maxpnum = reader.GetInt("MaxPNum");
    maxtypenum = reader.GetInt("TypeNum");
    if (maxtypenum < 0 || maxtypenum > 10) maxtypenum = 10;
    hParam.dt = reader.GetFloat("DT");
    hParam.spacing = reader.GetFloat("PSpacing");
    const float smoothratio = reader.GetFloat("SmoothRatio");
    hParam.viscosity = reader.GetFloat("Viscosity");
    restdensity = reader.GetFloat("RestDensity");
    hParam.pressureK = reader.GetFloat("PressureK");
    reader.GetFloatN(densArr, maxtypenum, "DensityArray");
    reader.GetFloatN(viscArr, maxtypenum, "ViscosityArray");
    hParam.surface_tension = reader.GetFloat("SurfaceTension");
    hParam.acceleration_limit = reader.GetFloat("AccelerationLimit");
    // this build's optional tags
    if (reader.GetText("GridMin")) lo = reader.GetFloat3("GridMin");
    if (reader.GetText("GridMax")) hi = reader.GetFloat3("GridMax");
    domainMin = lo;
    domainMax = hi;
    if (const char *b = reader.GetText("BoundaryFile")) boundary_file = b;
    if (reader.GetText("WallType")) wall_type = reader.GetInt("WallType");
    wall_rest_density = reader.GetFloat("WallRestDensity");
    hParam.boundary_mode = reader.GetInt("BoundaryMode");
    if (const char *m = reader.GetText("RunMode")) {
        const std::string mode(m);
        // surrounding blanks of the tag text do not count
        const size_t a = mode.find_first_not_of(" \t\r\n"), b = mode.find_last_not_of(" \t\r\n");
        const std::string word = a == std::string::npos ? "" : mode.substr(a, b - a + 1);
        run_mode = word == "SPH" ? SPH : word == "DFSPH" ? DFSPH : word == "IISPH" ? IISPH : -1;
    }

    // smoothradius, dx, gridres and the kernel constants exactly as LoadParam evaluates them in float
    const double vmin[3] = {lo.x, lo.y, lo.z}, vmax[3] = {hi.x, hi.y, hi.z};
    check(fsa_derive_params(&hParam, FSA_PRECISION_F32, hParam.spacing, smoothratio, vmin, vmax), "fsa_derive_params");

    loadFluidVolume(scene, maxtypenum, fluid_volumes);

    reader.Use(bound);
    hParam.boundary_visc = reader.GetFloat("Viscosity");
    hParam.boundary_friction = reader.GetFloat("Friction");
}

// LoadParam, solver/MultiphaseSPHSolver.cpp:435-473
void MultiphaseSPHSolver::LoadParam(const char *xmlpath) {
    frame_count = 0;
    dump_count = 0;
    numParticles = 0;
    num_fluid_particles = 0;
    for (int &c : localid_counter) c = 0;
    emit_particle_on = false;
    hParam = fsa_params{};
    fluid_volumes.clear();
    boundary_file = "big box.xml";
    wall_type = TYPE_RIGID;
    wall_rest_density = 0.f;

    ParseParam(xmlpath);

    num_grid_cells = hParam.gridres[0] * hParam.gridres[1] * hParam.gridres[2];
    dt = (float)hParam.dt;
}

void MultiphaseSPHSolver::SetupHostBuffer() { ClearParticles(); }

void MultiphaseSPHSolver::ClearParticles() {
    pos.clear(); color.clear(); normal.clear(); unique_id.clear(); vel.clear(); type.clear();
    adjacentIndex.clear(); mass.clear(); group.clear(); rest_density.clear(); v_star.clear();
    localid.clear(); vol_frac.clear();
}

int MultiphaseSPHSolver::AddDefaultParticle() {
    const int pid = SPHSolver::AddDefaultParticle();
    localid.push_back(0);
    for (int t = 0; t < maxtypenum; t++) vol_frac.push_back(0.f);
    return pid;
}

// AddMultiphaseFluidVolumes, solver/MultiphaseSPHSolver.cpp:517-745: rest density of a block is the
// volume-fraction mix of DensityArray, particle mass = spacing^3 * density, colour = the fractions.
void MultiphaseSPHSolver::AddMultiphaseFluidVolumes() {
    for (size_t i = 0; i < fluid_volumes.size(); i++) {
        const fluidvol &fv = fluid_volumes[i];
        if (!(fv.type == TYPE_FLUID || fv.type == TYPE_DEFORMABLE || fv.type == TYPE_GRANULAR)) {
            std::printf("Error: wrong volume type.\n");
            continue;
        }
        const float spacing = (float)hParam.spacing;
        float density = 0.f;
        for (int t = 0; t < maxtypenum; t++) density += densArr[t] * fv.volfrac[t];
        const float mp = spacing * spacing * spacing * density;

        std::vector<cfloat3> points;
        fill_volume_points(fv, spacing, points);
        for (const cfloat3 &p : points) {
            const int pid = AddDefaultParticle();
            pos[pid] = p;
            color[pid] = cfloat4(fv.volfrac[0], fv.volfrac[1], fv.volfrac[2], 1);
            type[pid] = fv.type;
            group[pid] = fv.group;
            mass[pid] = mp;
            rest_density[pid] = density;
            localid[pid] = localid_counter[fv.type]++;
            for (int t = 0; t < maxtypenum; t++) vol_frac[(size_t)pid * maxtypenum + t] = fv.volfrac[t];
        }
        const int added = (int)points.size();
        if (fv.type == TYPE_FLUID) {
            std::printf("Block No. %d, type: fluid, particle num: %d\n", (int)i, added);
            num_fluid_particles += added;
        } else if (fv.type == TYPE_DEFORMABLE) {
            std::printf("Block No. %d, type: deformable, particle num: %d\n", (int)i, added);
        } else {
            std::printf("Block No. %d, type: granular, particle num: %d\n", (int)i, added);
        }
    }
}

// LoadPO, solver/MultiphaseSPHSolver.cpp:747-766: boundary particles weigh twice a fluid particle of
// RestDensity; their rest density stays 0 until RigidParticleVolume fills it.
void MultiphaseSPHSolver::LoadPO(ParticleObject *po) {
    const float spacing = (float)hParam.spacing;
    const float mp = spacing * spacing * spacing * restdensity * 2;
    for (size_t i = 0; i < po->pos.size(); i++) {
        const int pid = AddDefaultParticle();
        pos[pid] = po->pos[i];
        color[pid] = cfloat4(1, 1, 1, 0.5);
        type[pid] = wall_type;
        normal[pid] = po->normal[i];
        mass[pid] = mp;
        group[pid] = po->id[i];
        rest_density[pid] = wall_rest_density;
    }
}

// The host half of SetupFluidScene (solver/MultiphaseSPHSolver.cpp:815-827): parameters and particles.
void MultiphaseSPHSolver::BuildHostScene() {
    LoadParam(config_path("sph_scene.xml"));
    if (hParam.gridres[0] <= 0 || !(hParam.spacing > 0)) {
        std::fprintf(stderr, "no usable scene in %s\n", config_path("sph_scene.xml"));
        std::exit(EXIT_FAILURE);
    }
    SetupHostBuffer();

    AddMultiphaseFluidVolumes();

    BoundaryGenerator bg;
    if (const char *bfile = std::getenv("FSA_BOUNDARY_FILE")) boundary_file = bfile;
    ParticleObject *boundary = bg.loadxml(config_path(boundary_file.c_str()));
    if (boundary) {
        LoadPO(boundary);
        delete boundary;
    }
    numParticles = (int)pos.size();
}

// SetupFluidScene, solver/MultiphaseSPHSolver.cpp:815-835
void MultiphaseSPHSolver::SetupFluidScene() {
    BuildHostScene();
    SetupDeviceBuffer();
    CopyParticleDataToDevice();
    CopyParam2Device();
}

// The WCSPH counterpart of SetupIISPH (solver/MultiphaseSPHSolver.cpp:850-860): scene, first sort, and the
// rigid particle volumes the density pass multiplies in.
void MultiphaseSPHSolver::SetupSPH() {
    SetupFluidScene();
    run_mode = SPH;
    if (numParticles > 0) {
        SortParticles();
        check(fsa_rigid_particle_volume(device), "RigidParticleVolume");
        CopyPosColorFromDevice();
    }
}

// SetupDFSPH, solver/sph_solver.cpp:102-106
void MultiphaseSPHSolver::SetupDFSPH() {
    SetupFluidScene();
    run_mode = DFSPH;
    if (numParticles > 0) {
        // several z-slabs under this one host program (slab_mode.cpp): the set-up runs on them
        if (const char *slabs = std::getenv("FSA_SLABS"))
            if (std::atoi(slabs) > 0) {
                EnterSlabModeIISPH(std::atoi(slabs), true);
                CopyPosColorFromDevice();
                return;
            }
        check(fsa_setup_dfsph(device), "SetupDFSPH");
        CopyPosColorFromDevice();
    }
}

// SetupIISPH, solver/MultiphaseSPHSolver.cpp:850-860: Sort, EffectiveMass, IISPHFactor, RigidParticleVolume.
// InitializeDeformable acts on TYPE_DEFORMABLE particles only and has no counterpart.
void MultiphaseSPHSolver::SetupIISPH() {
    SetupFluidScene();
    run_mode = IISPH;
    if (numParticles > 0) {
        SortParticles();
        // vol_frac is in creation order = uniqueId order; the library wants rows in the device's order
        std::vector<int> uid((size_t)numParticles);
        check(fsa_get_buffer(device, "uniqueId", uid.data(), uid.size() * sizeof(int)), "get uniqueId");
        vecf rows((size_t)numParticles * maxtypenum);
        for (int i = 0; i < numParticles; i++)
            for (int t = 0; t < maxtypenum; t++)
                rows[(size_t)i * maxtypenum + t] = vol_frac[(size_t)uid[i] * maxtypenum + t];
        double dens[10];
        for (int t = 0; t < maxtypenum; t++) dens[t] = densArr[t];
        check(fsa_set_volume_fractions(device, maxtypenum, dens, rows.data()), "fsa_set_volume_fractions");
        check(fsa_effective_mass(device), "EffectiveMass");
        // several z-slabs under this one host program (slab_mode.cpp): the rest of the set-up runs on them
        if (const char *slabs = std::getenv("FSA_SLABS"))
            if (std::atoi(slabs) > 0) {
                EnterSlabModeIISPH(std::atoi(slabs));
                CopyPosColorFromDevice();
                return;
            }
        check(fsa_iisph_factor(device), "IISPHFactor");
        check(fsa_rigid_particle_volume(device), "RigidParticleVolume");
        CopyPosColorFromDevice();
    }
}

// The reference's Setup() is `SetupIISPH();` (solver/MultiphaseSPHSolver.cpp:862-866).  Here the step is
// picked by FSA_RUN_MODE, else by the scene's <RunMode> tag, else it is the WCSPH step.
void MultiphaseSPHSolver::Setup() {
    int want = SPH;
    if (const char *mode = std::getenv("FSA_RUN_MODE")) {
        if (std::strcmp(mode, "SPH") == 0) want = SPH;
        else if (std::strcmp(mode, "DFSPH") == 0) want = DFSPH;
        else if (std::strcmp(mode, "IISPH") == 0) want = IISPH;
        else {
            std::fprintf(stderr, "FSA_RUN_MODE=%s: this build has the SPH (WCSPH), DFSPH and IISPH steps\n", mode);
            std::exit(EXIT_FAILURE);
        }
    } else {
        // the tag is only known once the scene file is read: peek at it
        run_mode = SPH;
        LoadParam(config_path("sph_scene.xml"));
        if (run_mode < 0) {
            std::fprintf(stderr, "<RunMode> of the scene asks for a step this build does not have\n");
            std::exit(EXIT_FAILURE);
        }
        want = run_mode;
    }
    if ((want == DFSPH || want == IISPH) && std::getenv("FSA_RESUME")) {
        // the text checkpoint carries neither the pressures nor the DFSPH factors these steps start from
        std::fprintf(stderr, "FSA_RESUME: a text checkpoint can only be resumed by the SPH (WCSPH) step\n");
        std::exit(EXIT_FAILURE);
    }
    if (want == DFSPH) {
        SetupDFSPH();
        return;
    }
    if (want == IISPH) {
        SetupIISPH();
        return;
    }
    SetupSPH();
    // resume from a text checkpoint written by key 'b' (DumpSimulationDataText)
    if (const char *resume = std::getenv("FSA_RESUME")) LoadSimulationDataText(resume);
    // several z-slabs under this one host program (slab_mode.cpp)
    if (const char *slabs = std::getenv("FSA_SLABS"))
        if (std::atoi(slabs) > 0 && numParticles > 0) EnterSlabMode(std::atoi(slabs));
}

// Text checkpoint of solver/MultiphaseSPHSolver.cpp:219-247: the single-phase line plus the volume fractions.
void MultiphaseSPHSolver::DumpSimulationDataText() {
    std::printf("Dumping simulation data at frame %d\n", frame_count);
    const char *dir = std::getenv("FSA_DUMP_DIR");
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%03d.txt", dir ? dir : "dump", frame_count);
    FILE *fp = std::fopen(path, "w+");
    if (!fp) {
        std::printf("error opening file\n");
        return;
    }
    CopySimulationDataFromDevice();
    std::fprintf(fp, "%d\n", numParticles);
    for (int i = 0; i < numParticles; i++) {
        std::fprintf(fp, "%f %f %f ", pos[i].x, pos[i].y, pos[i].z);
        std::fprintf(fp, "%f %f %f ", vel[i].x, vel[i].y, vel[i].z);
        std::fprintf(fp, "%d ", type[i]);
        std::fprintf(fp, "%d ", group[i]);
        std::fprintf(fp, "%f ", mass[i]);
        std::fprintf(fp, "%d ", unique_id[i]);
        std::fprintf(fp, "%f %f %f ", v_star[i].x, v_star[i].y, v_star[i].z);
        // volume fractions stay with the particle's identity, not with its slot in the sorted arrays
        const size_t u = (size_t)unique_id[i];
        for (int k = 0; k < maxtypenum; k++) std::fprintf(fp, "%f ", vol_frac[u * maxtypenum + k]);
        std::fprintf(fp, "\n");
    }
    std::fclose(fp);
}

// Renderer hand-off of solver/MultiphaseSPHSolver.cpp:249-279: "frame k", then per particle
// "index x y z f0 f1 f2 group".
void MultiphaseSPHSolver::DumpRenderData() {
    const char *dir = std::getenv("FSA_RENDER_DIR");
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%03d.txt", dir ? dir : "../particle_data", dump_count++);
    FILE *fp = std::fopen(path, "w+");
    if (!fp) {
        std::printf("error opening file\n");
        return;
    }
    CopySimulationDataFromDevice();
    std::fprintf(fp, "frame %d\n", frame_count);
    for (int i = 0; i < numParticles; i++) {
        const size_t u = (size_t)unique_id[i];
        float f[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3 && k < maxtypenum; k++) f[k] = vol_frac[u * maxtypenum + k];
        std::fprintf(fp, "%d %f %f %f %f %f %f %d\n", i, pos[i].x, pos[i].y, pos[i].z, f[0], f[1], f[2], group[i]);
    }
    std::fclose(fp);
}

// Reads a text checkpoint back (solver/MultiphaseSPHSolver.cpp:281-305) and makes it the device state.
// A solver must have been set up (parameters and device buffers) with at least as many particles.
void MultiphaseSPHSolver::LoadSimulationDataText(const char *filepath) {
    std::printf("Loading simulation data text from %s\n", filepath);
    FILE *fp = std::fopen(filepath, "r");
    if (!fp) {
        std::printf("error opening file\n");
        return;
    }
    // The file is read into temporaries and checked as a whole before any solver state is touched: a truncated
    // or hand-edited dump must end in "malformed checkpoint", not in a damaged heap.
    std::fseek(fp, 0, SEEK_END);
    const long file_bytes = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    int n = 0;
    // a particle line has at least 14 numbers of at least two characters each
    if (std::fscanf(fp, "%d\n", &n) != 1 || n < 0 || (long)n > file_bytes / 28) {
        std::printf("malformed checkpoint\n");
        std::fclose(fp);
        return;
    }
    struct Line {
        cfloat3 pos, vel, v_star;
        int type, group, id;
        float mass;
    };
    std::vector<Line> lines((size_t)n);
    vecf frac((size_t)n * maxtypenum);
    std::vector<char> seen((size_t)n, 0);
    for (int k = 0; k < n; k++) {
        Line &L = lines[k];
        bool ok = std::fscanf(fp, "%f %f %f ", &L.pos.x, &L.pos.y, &L.pos.z) == 3;
        ok = ok && std::fscanf(fp, "%f %f %f ", &L.vel.x, &L.vel.y, &L.vel.z) == 3;
        ok = ok && std::fscanf(fp, "%d %d %f %d ", &L.type, &L.group, &L.mass, &L.id) == 4;
        ok = ok && std::fscanf(fp, "%f %f %f ", &L.v_star.x, &L.v_star.y, &L.v_star.z) == 3;
        for (int t = 0; ok && t < maxtypenum; t++) ok = std::fscanf(fp, "%f ", &frac[(size_t)k * maxtypenum + t]) == 1;
        // ids index the per-identity tables: each of 0 .. n-1 exactly once
        ok = ok && L.id >= 0 && L.id < n && !seen[L.id];
        if (!ok) {
            std::printf("malformed checkpoint at particle %d\n", k);
            std::fclose(fp);
            return;
        }
        seen[L.id] = 1;
    }
    ClearParticles();
    for (int k = 0; k < n; k++) {
        const int i = AddDefaultParticle();
        const Line &L = lines[k];
        pos[i] = L.pos;
        vel[i] = L.vel;
        v_star[i] = L.v_star;
        type[i] = L.type;
        group[i] = L.group;
        mass[i] = L.mass;
        unique_id[i] = L.id;
        for (int t = 0; t < maxtypenum; t++) vol_frac[(size_t)i * maxtypenum + t] = frac[(size_t)k * maxtypenum + t];
        // the dump does not carry the rest density: rebuild it as the setup does
        float density = 0.f;
        for (int t = 0; t < maxtypenum; t++) density += densArr[t] * vol_frac[(size_t)i * maxtypenum + t];
        const bool volume_particle = type[i] == TYPE_FLUID || type[i] == TYPE_DEFORMABLE || type[i] == TYPE_GRANULAR;
        rest_density[i] = volume_particle ? density : wall_rest_density;
        color[i] = type[i] == TYPE_FLUID
                       ? cfloat4(vol_frac[(size_t)i * maxtypenum], maxtypenum > 1 ? vol_frac[(size_t)i * maxtypenum + 1] : 0.f,
                                 maxtypenum > 2 ? vol_frac[(size_t)i * maxtypenum + 2] : 0.f, 1)
                       : cfloat4(1, 1, 1, 0.5);
    }
    std::fclose(fp);
    // the file is in the device order of the dump; ids must again be a permutation for indexTable
    // and the volume fractions are stored per id
    {
        vecf by_id(vol_frac.size());
        for (int i = 0; i < n; i++)
            for (int t = 0; t < maxtypenum; t++)
                by_id[(size_t)unique_id[i] * maxtypenum + t] = vol_frac[(size_t)i * maxtypenum + t];
        vol_frac.swap(by_id);
    }
    SetupDeviceBuffer();
    CopyParticleDataToDevice();
    CopyParam2Device();
    if (numParticles > 0) {
        check(fsa_set_buffer(device, "v_star", v_star.data(), (size_t)numParticles * sizeof(cfloat3)), "set v_star");
        SortParticles();
        check(fsa_rigid_particle_volume(device), "RigidParticleVolume");
        CopyPosColorFromDevice();
    }
}
