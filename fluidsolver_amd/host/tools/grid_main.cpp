// Headless run of the Eulerian grid solver through the plugin class:
//   grid_solver [--frames N] [--save-final FILE]
// FILE receives n (int32), the tracer positions (n x 3 float) and the fields u, v, w, p (float each, sizes
// uSize, vSize, wSize, dimSize as int32 in front).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "solver/grid_solver.h"

int main(int argc, char **argv) {
    int frames = 10;
    const char *save = nullptr;
    for (int a = 1; a < argc; a++) {
        if (!std::strcmp(argv[a], "--frames") && a + 1 < argc) frames = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "--save-final") && a + 1 < argc) save = argv[++a];
        else {
            std::fprintf(stderr, "usage: %s [--frames N] [--save-final FILE]\n", argv[0]);
            return 2;
        }
    }
    GridSolver solver;
    solver.setup();
    for (int f = 0; f < frames; f++) solver.Step();
    if (save) {
        FILE *fp = std::fopen(save, "wb");
        if (!fp) {
            std::perror(save);
            return 1;
        }
        const int n = solver.GetNumParticles();
        std::fwrite(&n, sizeof(int), 1, fp);
        std::fwrite(solver.GetPos()->data(), sizeof(cfloat3), (size_t)n, fp);
        for (const char *name : {"u", "v", "w", "p"}) {
            const std::vector<scalar> f = solver.field(name);
            const int m = (int)f.size();
            std::fwrite(&m, sizeof(int), 1, fp);
            std::fwrite(f.data(), sizeof(scalar), f.size(), fp);
        }
        std::fclose(fp);
    }
    std::printf("frames %d, last solve: %d iterations, residual %g, div sum %g\n", frames, solver.last_cg_iterations,
                solver.last_cg_residual, (double)solver.divUsum);
    return 0;
}
