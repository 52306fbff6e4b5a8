// Builds a scene from the config files on the host only (no device) and writes it as one binary
// file, so that the C++ scene setup can be compared with other generators of the same scene:
//   int32 n, int32 gridres[3], float64 dx, float64 kernel_cubic, float64 kspikydiff,
//   then n x 3 float32 pos, n x 4 float32 color, n x 3 float32 normal, n int32 type, n int32 group,
//   n float32 mass, n float32 restDensity
#include <cstdint>
#include <cstdio>

#include "solver/MultiphaseSPHSolver.h"

class SceneDump : public MultiphaseSPHSolver {
public:
    int write(const char *path) {
        BuildHostScene();
        FILE *fp = std::fopen(path, "wb");
        if (!fp) return 1;
        const int32_t n = (int32_t)pos.size();
        const int32_t res[3] = {hParam.gridres[0], hParam.gridres[1], hParam.gridres[2]};
        const double consts[3] = {hParam.dx, hParam.kernel_cubic, hParam.kspikydiff};
        std::fwrite(&n, 4, 1, fp);
        std::fwrite(res, 4, 3, fp);
        std::fwrite(consts, 8, 3, fp);
        std::fwrite(pos.data(), sizeof(cfloat3), n, fp);
        std::fwrite(color.data(), sizeof(cfloat4), n, fp);
        std::fwrite(normal.data(), sizeof(cfloat3), n, fp);
        std::fwrite(type.data(), 4, n, fp);
        std::fwrite(group.data(), 4, n, fp);
        std::fwrite(mass.data(), 4, n, fp);
        std::fwrite(rest_density.data(), 4, n, fp);
        std::fclose(fp);
        std::printf("%d particles written to %s\n", n, path);
        return 0;
    }
};

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: scene_dump OUTPUT   (scene chosen by FSA_CONFIG_DIR / FSA_SCENE_ID)\n");
        return 2;
    }
    SceneDump d;
    return d.write(argv[1]);
}
