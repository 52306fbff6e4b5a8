// Plugin base class the GUI drives (Solver.h:5-17 in the reference): one Step() per frame, key and
// expression hooks, time step and domain box queries.
#pragma once

#include "catpaw/vec_define.h"

class Solver {
public:
    virtual ~Solver() = default;

    virtual void Step() {}
    virtual void HandleKeyEvent(char /*key*/) {}
    virtual void Eval(const char * /*expression*/) {}

    virtual float GetTimeStep() = 0;
    virtual cfloat3 GetDomainMin() = 0;
    virtual cfloat3 GetDomainMax() = 0;
};
