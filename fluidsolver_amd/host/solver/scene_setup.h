// Scene description read from the config files: fluid volumes of a <MultiScene> and the boundary
// shapes of a <Boundary> file.  Restates, for the particle types the WCSPH step handles, what the
// reference does in loadFluidVolume (solver/particle_common.cpp:7-70), AddMultiphaseFluidVolumes
// (solver/MultiphaseSPHSolver.cpp:517-745) and BoundaryGenerator (solver/ParticleGenerator.h:20-155)
// under /root/reference/particle_solver.  All lattice loops accumulate in fp32 exactly as written there
// (`for (float x = lo; x < hi; x += spacing)`), which fixes particle counts and coordinates.
#pragma once

#include <vector>

#include "../catpaw/geometry.h"
#include "xml_lite.h"

// particle types, solver/particle_common.h:6-16
#define TYPE_FLUID 0
#define TYPE_CLOTH 1
#define TYPE_BOUNDARY 2
#define TYPE_EMITTER 3
#define TYPE_RIGID 4
#define TYPE_DEFORMABLE 5
#define TYPE_GRANULAR 6

struct fluidvol {
    cfloat3 xmin;
    cfloat3 xmax;
    float volfrac[10] = {0.f};
    int group = 0;
    int type = TYPE_FLUID;
    bool empty = false;  // only the six faces of the block
};

// Every <Vol> child of a scene element, in file order.
void loadFluidVolume(const xml_lite::Element *scene, int typenum, std::vector<fluidvol> &out);

// Lattice points of one volume: x outermost, z innermost for a solid block; face pairs (z, y, x
// normals in that order) for a hollow one.
void fill_volume_points(const fluidvol &fv, float spacing, std::vector<cfloat3> &points);

struct ParticleObject {
    std::vector<cfloat3> pos;
    std::vector<cfloat3> normal;
    std::vector<int> type;
    std::vector<int> id;
};

// <Cube>, <Box>, <OpenBox> children of the <Boundary> element of a file.
class BoundaryGenerator {
public:
    // nullptr when the file cannot be read or has no <Boundary> element; the caller owns the result
    ParticleObject *loadxml(const char *filepath);
    void parseNode(const xml_lite::Element *e, ParticleObject *out);
};
