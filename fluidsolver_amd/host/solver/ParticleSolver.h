// Particle flavour of the plugin surface (solver/ParticleSolver.h:5-37 in the reference).  The GUI keeps
// the addresses of the position and colour vectors (not of their storage), so they may be resized;
// their contents are valid whenever Step() has returned, in the device's current (sorted) order.
#pragma once

#include "../Solver.h"

class ParticleSolver : public Solver {
protected:
    int numParticles = 0;
    vecf3 pos;
    vecf4 color;
    vecf3 vel;
    vecadj adjacentIndex;
    cfloat3 domainMin;
    cfloat3 domainMax;
    float dt = 0.f;

public:
    vecf3 *GetPos() { return &pos; }
    vecf4 *GetColor() { return &color; }
    int GetNumParticles() { return numParticles; }
    float GetTimeStep() override { return dt; }
    cfloat3 GetDomainMin() override { return domainMin; }
    cfloat3 GetDomainMax() override { return domainMax; }
};
