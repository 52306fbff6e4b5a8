// Driver of the headless build: the same sequence as the reference's main.cpp:11-33 (construct the
// solver, Setup, bind it to the GUI object, Run).  The reference's own main.cpp compiles against these
// headers unchanged (`make refmain REF=/path/to/particle_solver`); this file exists so that the
// program can be built where the reference tree is absent.
#include "Interface/SolverGUI.h"
#include "solver/MultiphaseSPHSolver.h"

static SolverGUI gui;

int main(int argc, char **argv) {
    gui.Initialize(argc, argv);
    MultiphaseSPHSolver solver;
    solver.Setup();
    gui.bindSolver(&solver);
    gui.setParticlesz(0.005f);
    gui.Run();
    return 0;
}
