// Headless stand-in for the reference's GLUT window (Interface/SolverGUI.h:83-110, SolverGUI.cpp:719-752
// under /root/reference/particle_solver): the same four entry points main.cpp uses -- Initialize,
// bindSolver, setParticlesz, Run -- and the same per-frame conversation with the solver (Step, then
// GetTimeStep, optionally Eval("DumpRenderData"), then the position / colour vectors), without OpenGL.
//
// Run() advances a fixed number of frames and prints the rate.  Command line (or environment):
//   --frames N        (FSA_FRAMES, default 100)
//   --warmup N        untimed frames first (FSA_WARMUP, default 0)
//   --dump-render     Eval("DumpRenderData") after every frame, as the GUI's dump toggle does
//   --keys STRING     key events delivered after the last frame (e.g. "b" writes a text checkpoint)
//   --save-final FILE binary snapshot after the last frame: int32 n, n x 3 float32 positions,
//                     n x 4 float32 colours, in the solver's current order
#pragma once

#include <string>

#include "../solver/ParticleSolver.h"

class SolverGUI {
private:
    ParticleSolver *solver = nullptr;
    vecf3 *hPos = nullptr;
    vecf4 *hColor = nullptr;
    cfloat3 boxMin, boxMax;
    float particleSize = 0.f;
    int rendermode = 0;
    int frameNo = 0;
    int frames = 100;
    int warmup = 0;
    bool dump_renderdata = false;
    std::string keys;
    std::string save_final;

public:
    void Initialize(int argc, char **argv);
    void GetBoundingBox();
    void Run();
    void Exit();

    void setRenderMode(int mode) { rendermode = mode; }
    void setParticlesz(float sz) { particleSize = sz; }
    void keyDown(unsigned char key);

    void bindSolver(ParticleSolver *solver_) {
        solver = solver_;
        hPos = solver->GetPos();
        hColor = solver->GetColor();
        GetBoundingBox();
    }

    int GetFrameNo() const { return frameNo; }
};
