#include "SolverGUI.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static int env_int(const char *name, int fallback) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) : fallback;
}

void SolverGUI::Initialize(int argc, char **argv) {
    frames = env_int("FSA_FRAMES", 100);
    warmup = env_int("FSA_WARMUP", 0);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        auto value = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (std::strcmp(a, "--frames") == 0) frames = std::atoi(value());
        else if (std::strcmp(a, "--warmup") == 0) warmup = std::atoi(value());
        else if (std::strcmp(a, "--dump-render") == 0) dump_renderdata = true;
        else if (std::strcmp(a, "--keys") == 0) keys = value();
        else if (std::strcmp(a, "--save-final") == 0) save_final = value();
        else std::fprintf(stderr, "unknown option %s ignored\n", a);
    }
    if (frames < 0) frames = 0;
    if (warmup < 0) warmup = 0;
}

void SolverGUI::GetBoundingBox() {
    boxMin = solver->GetDomainMin();
    boxMax = solver->GetDomainMax();
}

void SolverGUI::keyDown(unsigned char key) {
    if (solver) solver->HandleKeyEvent((char)key);
}

void SolverGUI::Exit() { std::exit(0); }

void SolverGUI::Run() {
    if (!solver) {
        std::fprintf(stderr, "SolverGUI::Run without a bound solver\n");
        return;
    }
    for (int k = 0; k < warmup; k++) solver->Step();
    const auto t0 = std::chrono::steady_clock::now();
    double sim_time = 0.0;
    for (int k = 0; k < frames; k++) {
        solver->Step();
        sim_time += solver->GetTimeStep();
        if (dump_renderdata) solver->Eval("DumpRenderData");
        frameNo++;
    }
    // the vectors the renderer would upload now
    const int n = solver->GetNumParticles();
    const cfloat3 *p = hPos->data();
    const cfloat4 *c = hColor->data();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("frames %d  particles %d  simulated %.4f s  wall %.3f s  %.3f ms/frame  %.3e particle-steps/s\n", frames,
                n, sim_time, secs, frames ? secs / frames * 1e3 : 0.0, secs > 0 ? (double)n * frames / secs : 0.0);
    if (n > 0)
        std::printf("first particle (%f, %f, %f) colour (%g, %g, %g, %g); box (%g, %g, %g)-(%g, %g, %g)\n", p[0].x, p[0].y,
                    p[0].z, c[0].x, c[0].y, c[0].z, c[0].w, boxMin.x, boxMin.y, boxMin.z, boxMax.x, boxMax.y, boxMax.z);
    for (char k : keys) keyDown((unsigned char)k);
    if (!save_final.empty()) {
        FILE *fp = std::fopen(save_final.c_str(), "wb");
        if (!fp) {
            std::fprintf(stderr, "cannot write %s\n", save_final.c_str());
            return;
        }
        const int32_t count = n;
        std::fwrite(&count, sizeof(count), 1, fp);
        std::fwrite(p, sizeof(cfloat3), (size_t)n, fp);
        std::fwrite(c, sizeof(cfloat4), (size_t)n, fp);
        std::fclose(fp);
    }
}
