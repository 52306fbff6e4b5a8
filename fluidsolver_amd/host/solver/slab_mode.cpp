// Several z-slabs driven by ONE host program (FSA_SLABS=N): the plugin surface of main.cpp stays as it is, the
// SPH (WCSPH) step underneath runs as a slab group of the library (fsa_slab_group_create_local): N device solvers,
// on one GPU or spread over FSA_SLAB_DEVICES GPUs, exchanging migrants and halo copies by device-to-device copies
// with no host round trip per step.  Replaces the single `device` of SPHSolver::Step
// (solver/sph_solver.cpp:129-142 under /root/reference/particle_solver) for scenes too large or too slow for one
// GPU.  Results are those of the single-solver run, bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "sph_solver.h"

namespace {

constexpr int kHalo = 2;
constexpr unsigned kUndef = 99999999u;
double g_headroom = 1.5;  // FSA_SLAB_HEADROOM as the slabs were made with (for the message when one is outgrown)

// cell layer (z index) of a position with the arithmetic of calcGridPos / calcHashD
// (solver/sph_kernel_impl.cuh:74-100): fp32 subtract, divide, floor; -1 outside the domain
int cell_layer(const cfloat3 &p, const fsa_params &P) {
    const float x[3] = {p.x, p.y, p.z};
    int cell[3];
    for (int a = 0; a < 3; a++) {
        const float lo = (float)P.gridxmin[a], hi = (float)P.gridxmax[a];
        if (!((double)x[a] < (double)hi - 1e-10 && (double)x[a] > (double)lo + 1e-10)) return -1;
        cell[a] = (int)std::floor((x[a] - lo) / (float)P.dx);
        if (cell[a] < 0 || cell[a] >= P.gridres[a]) return -1;
    }
    return cell[2];
}

// the layer a particle outside the grid is filed under: that of its z coordinate, clamped into the grid (it lives
// with the slab that owns this layer, as in the library's exchange kernels)
int outside_layer(const cfloat3 &p, const fsa_params &P) {
    const float c = std::floor((p.z - (float)P.gridxmin[2]) / (float)P.dx);
    if (!(c == c) || c < 0.f) return 0;
    return c >= (float)P.gridres[2] ? P.gridres[2] - 1 : (int)c;
}

// cut indices that balance the particle counts, every slab at least min_layers thick
std::vector<int> balanced_cuts(const std::vector<long long> &hist, int nranks, int min_layers) {
    const int gz = (int)hist.size();
    std::vector<double> pre(gz + 1, 0.0);
    for (int z = 0; z < gz; z++) pre[z + 1] = pre[z] + (double)hist[z];
    std::vector<int> cuts{0};
    for (int r = 1; r < nranks; r++) {
        const double target = pre[gz] * r / nranks;
        int c = (int)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin());
        if (c > 0 && std::fabs(pre[c - 1] - target) <= std::fabs(pre[std::min(c, gz)] - target)) c--;
        c = std::max(c, cuts.back() + min_layers);
        c = std::min(c, gz - (nranks - r) * min_layers);
        cuts.push_back(c);
    }
    cuts.push_back(gz);
    return cuts;
}

}  // namespace

void SPHSolver::LeaveSlabMode() {
    if (slab_group) fsa_slab_group_destroy(slab_group);
    slab_group = nullptr;
    if (slab_team) fsa_slab_iisph_destroy(slab_team);
    slab_team = nullptr;
    for (fsa_solver *s : slab_devices) fsa_destroy(s);
    slab_devices.clear();
}

// Called at the end of the set-up, with the scene on `device` in its sorted order: that state is cut into
// nslabs slabs and `device` is released.
void SPHSolver::EnterSlabMode(int nslabs) {
    if (run_mode != SPH) {
        std::fprintf(stderr, "FSA_SLABS: this entry is the SPH (WCSPH) step's (the IISPH and DFSPH steps enter slab mode from their own set-up)\n");
        std::exit(EXIT_FAILURE);
    }
    const int n = numParticles;
    const int gz = hParam.gridres[2];
    if (nslabs < 1 || gz < nslabs * (kHalo + 1)) {
        std::fprintf(stderr, "FSA_SLABS=%d: %d cell layers cannot be cut into that many slabs of at least %d\n", nslabs, gz,
                     kHalo + 1);
        std::exit(EXIT_FAILURE);
    }
    // the state in device order; a particle's slab id is its place in that order, which makes the order keys of
    // the slabs' first sort break ties exactly as the single solver's next sort would
    vecf3 p((size_t)n), v((size_t)n);
    vecf m((size_t)n), rd((size_t)n);
    veci t((size_t)n);
    slab_color.resize((size_t)n);
    int got = 0;
    check(fsa_readback_pos_color(device, p.data(), slab_color.data(), &got), "fsa_readback_pos_color");
    check(fsa_get_buffer(device, "vel", v.data(), (size_t)n * sizeof(cfloat3)), "get vel");
    check(fsa_get_buffer(device, "type", t.data(), (size_t)n * sizeof(int)), "get type");
    check(fsa_get_buffer(device, "mass", m.data(), (size_t)n * sizeof(float)), "get mass");
    check(fsa_get_buffer(device, "restDensity", rd.data(), (size_t)n * sizeof(float)), "get restDensity");
    for (int i = 0; i < n; i++)
        if (t[i] == TYPE_RIGID) {
            std::fprintf(stderr, "FSA_SLABS: scenes with TYPE_RIGID particles run on one solver only\n");
            std::exit(EXIT_FAILURE);
        }
    fsa_destroy(device);
    device = nullptr;

    veci layer((size_t)n);  // the layer each particle is filed under
    std::vector<long long> hist((size_t)gz, 0);
    long long outside = 0;
    for (int i = 0; i < n; i++) {
        layer[i] = cell_layer(p[i], hParam);
        if (layer[i] >= 0) {
            hist[layer[i]]++;
        } else {
            outside++;
            layer[i] = outside_layer(p[i], hParam);
        }
    }
    slab_cuts = nslabs > 1 ? balanced_cuts(hist, nslabs, kHalo + 1) : std::vector<int>{0, gz};
    std::vector<long long> pre((size_t)gz + 1, 0);
    for (int z = 0; z < gz; z++) pre[z + 1] = pre[z] + hist[z];
    auto held = [&](int r, int out[2]) {
        out[0] = std::max(slab_cuts[r] - kHalo, 0);
        out[1] = std::min(slab_cuts[r + 1] + kHalo, gz);
    };
    long long most = 0, widest = 0;
    for (int r = 0; r < nslabs; r++) {
        int h[2];
        held(r, h);
        most = std::max(most, pre[h[1]] - pre[h[0]]);
    }
    for (int r = 1; r < nslabs; r++) {
        const int c = slab_cuts[r];
        widest = std::max(widest, pre[std::min(c + 2 * kHalo, gz)] - pre[std::max(c - 2 * kHalo, 0)]);
    }
    // head room over the fullest slab's present count: the slabs are not re-cut during a run, so a scene whose mass
    // moves along z needs more (FSA_SLAB_HEADROOM, default 1.5; an outgrown slab ends the run with FSA_ERR_CAPACITY)
    double headroom = 1.5;
    long long spare = 4096;  // particles on top of the factor (FSA_SLAB_SPARE; small scenes jiggle by more than a factor covers)
    if (const char *e = std::getenv("FSA_SLAB_HEADROOM")) headroom = std::max(1.0, std::atof(e));
    if (const char *e = std::getenv("FSA_SLAB_SPARE")) spare = std::max(0LL, std::atoll(e));
    g_headroom = headroom;
    const long long capacity64 = (long long)((double)most * headroom) + outside + spare;
    const long long record_cap64 = (long long)((double)widest * headroom) + spare;
    if (capacity64 > 2000000000LL || record_cap64 > 2000000000LL) {
        std::fprintf(stderr, "FSA_SLABS=%d: a slab would hold %lld particles, more than one solver takes; use more slabs\n", nslabs,
                     capacity64);
        std::exit(EXIT_FAILURE);
    }
    const int capacity = (int)capacity64, record_cap = (int)record_cap64;
    int ndev = 1, dev0 = 0;
    if (const char *e = std::getenv("FSA_SLAB_DEVICES")) ndev = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("FSA_DEVICE")) dev0 = std::atoi(e);
    for (int r = 0; r < nslabs; r++) {
        int h[2];
        held(r, h);
        fsa_params P = hParam;
        P.slab_z0 = h[0];
        P.slab_nz = h[1] - h[0];
        fsa_solver *s = nullptr;
        check(fsa_create(&P, FSA_PRECISION_F32, capacity, dev0 + r % ndev, &s), "fsa_create (slab)");
        slab_devices.push_back(s);
        vecf3 sp, sv;
        vecf sm, srd;
        veci st, sid;
        for (int i = 0; i < n; i++) {
            const bool mine = layer[i] >= slab_cuts[r] && layer[i] < slab_cuts[r + 1];
            if (!mine) continue;
            sp.push_back(p[i]);
            sv.push_back(v[i]);
            sm.push_back(m[i]);
            srd.push_back(rd[i]);
            st.push_back(t[i]);
            sid.push_back(i);
        }
        check(fsa_upload(s, (int)sp.size(), sp.data(), sv.data(), nullptr, nullptr, st.data(), nullptr, sid.data(),
                         sm.data(), srd.data()),
              "fsa_upload (slab)");
        check(fsa_set_scene_has_rigid(s, 0), "fsa_set_scene_has_rigid");
    }
    check(fsa_slab_group_create_local(nslabs, slab_devices.data(), slab_cuts.data(), kHalo, record_cap, &slab_group),
          "fsa_slab_group_create_local");
    std::printf("%d particles on %d z-slabs (cuts", n, nslabs);
    for (int c : slab_cuts) std::printf(" %d", c);
    std::printf("), capacity %d each (head room %.2f: FSA_SLAB_HEADROOM)\n", capacity, headroom);
}

// The IISPH step (what the reference's main.cpp runs) on nslabs z-slabs.  Called in the middle of SetupIISPH, with the
// scene on `device` sorted once and the effective masses in place (solver/MultiphaseSPHSolver.cpp:855-856): that state
// is cut into slabs, `device` is released, and the rest of the set-up (factors, rigid particle volumes: :857-858)
// runs on the slabs.  A particle keeps its unique id; group and normal, which the step reads of a neighbour and which
// never change, go to every slab as tables over all ids.
// dfsph: the DFSPH step instead (called at the start of SetupDFSPH with the scene as uploaded; v_star rides with the
// particles, no tables); the set-up that follows (sort, computeDensityAlpha) runs on the slabs.
void SPHSolver::EnterSlabModeIISPH(int nslabs, bool dfsph) {
    const int n = numParticles;
    const int gz = hParam.gridres[2];
    if (nslabs < 1 || gz < nslabs * (kHalo + 1)) {
        std::fprintf(stderr, "FSA_SLABS=%d: %d cell layers cannot be cut into that many slabs of at least %d\n", nslabs, gz,
                     kHalo + 1);
        std::exit(EXIT_FAILURE);
    }
    vecf3 p((size_t)n), v((size_t)n);
    vecf m((size_t)n), rd((size_t)n);
    veci t((size_t)n), uid((size_t)n);
    vecf4 col((size_t)n);
    int got = 0;
    check(fsa_readback_pos_color(device, p.data(), col.data(), &got), "fsa_readback_pos_color");
    check(fsa_get_buffer(device, "vel", v.data(), (size_t)n * sizeof(cfloat3)), "get vel");
    check(fsa_get_buffer(device, "type", t.data(), (size_t)n * sizeof(int)), "get type");
    check(fsa_get_buffer(device, "mass", m.data(), (size_t)n * sizeof(float)), "get mass");
    check(fsa_get_buffer(device, "restDensity", rd.data(), (size_t)n * sizeof(float)), "get restDensity");
    check(fsa_get_buffer(device, "uniqueId", uid.data(), (size_t)n * sizeof(int)), "get uniqueId");
    fsa_destroy(device);
    device = nullptr;
    slab_color.resize((size_t)n);
    bool rigid = false;
    for (int i = 0; i < n; i++) {
        slab_color[(size_t)uid[i]] = col[i];
        rigid |= t[i] == TYPE_RIGID;
    }

    veci layer((size_t)n);
    std::vector<long long> hist((size_t)gz, 0);
    long long outside = 0;
    for (int i = 0; i < n; i++) {
        layer[i] = cell_layer(p[i], hParam);
        if (layer[i] >= 0) {
            hist[layer[i]]++;
        } else {
            outside++;
            layer[i] = outside_layer(p[i], hParam);
        }
    }
    slab_cuts = nslabs > 1 ? balanced_cuts(hist, nslabs, kHalo + 1) : std::vector<int>{0, gz};
    std::vector<long long> pre((size_t)gz + 1, 0);
    for (int z = 0; z < gz; z++) pre[z + 1] = pre[z] + hist[z];
    auto held = [&](int r, int out[2]) {
        out[0] = std::max(slab_cuts[r] - kHalo, 0);
        out[1] = std::min(slab_cuts[r + 1] + kHalo, gz);
    };
    long long most = 0, widest = 0;
    for (int r = 0; r < nslabs; r++) {
        int h[2];
        held(r, h);
        most = std::max(most, pre[h[1]] - pre[h[0]]);
    }
    for (int r = 1; r < nslabs; r++) {
        const int c = slab_cuts[r];
        widest = std::max(widest, pre[std::min(c + 2 * kHalo, gz)] - pre[std::max(c - 2 * kHalo, 0)]);
    }
    const long long capacity = most * 3 / 2 + outside + 4096, record_cap = widest * 3 / 2 + 4096;
    if (capacity > 2000000000LL || record_cap > 2000000000LL) {
        std::fprintf(stderr, "FSA_SLABS=%d: a slab would hold %lld particles, more than one solver takes; use more slabs\n", nslabs, capacity);
        std::exit(EXIT_FAILURE);
    }
    int ndev = 1, dev0 = 0;
    if (const char *e = std::getenv("FSA_SLAB_DEVICES")) ndev = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("FSA_DEVICE")) dev0 = std::atoi(e);
    for (int r = 0; r < nslabs; r++) {
        int h[2];
        held(r, h);
        fsa_params P = hParam;
        P.slab_z0 = h[0];
        P.slab_nz = h[1] - h[0];
        fsa_solver *s = nullptr;
        check(fsa_create(&P, FSA_PRECISION_F32, (int)capacity, dev0 + r % ndev, &s), "fsa_create (slab)");
        slab_devices.push_back(s);
        if (const char *e = std::getenv("FSA_VARIANT")) check(fsa_set_variant(s, std::atoi(e)), "fsa_set_variant");
        vecf3 sp, sv;
        vecf sm, srd;
        veci st, sid;
        for (int i = 0; i < n; i++) {
            if (layer[i] < slab_cuts[r] || layer[i] >= slab_cuts[r + 1]) continue;
            sp.push_back(p[i]);
            sv.push_back(v[i]);
            sm.push_back(m[i]);
            srd.push_back(rd[i]);
            st.push_back(t[i]);
            sid.push_back(uid[i]);
        }
        check(fsa_upload(s, (int)sp.size(), sp.data(), sv.data(), nullptr, nullptr, st.data(), nullptr, sid.data(),
                         sm.data(), srd.data()),
              "fsa_upload (slab)");
        check(fsa_set_scene_has_rigid(s, rigid ? 1 : 0), "fsa_set_scene_has_rigid");
        if (dfsph) check(fsa_slab_enable_dfsph(s), "fsa_slab_enable_dfsph");
        else check(fsa_slab_enable_iisph(s, n, group.data(), normal.data()), "fsa_slab_enable_iisph");
    }
    slab_team_dfsph = dfsph;
    if (dfsph) {  // 5 rounds at most, threshold 1: SolveDFSPH, solver/sph_solver.cpp:108-120
        check(fsa_slab_dfsph_create_local(nslabs, slab_devices.data(), slab_cuts.data(), kHalo, (int)record_cap, 5, 1.0,
                                          &slab_team),
              "fsa_slab_dfsph_create_local");
        check(fsa_slab_dfsph_setup(slab_team), "fsa_slab_dfsph_setup");
    } else {
        check(fsa_slab_iisph_create_local(nslabs, slab_devices.data(), slab_cuts.data(), kHalo, (int)record_cap, n, &slab_team),
              "fsa_slab_iisph_create_local");
        check(fsa_slab_iisph_setup(slab_team), "fsa_slab_iisph_setup");
    }
    std::printf("%d particles on %d z-slabs (cuts", n, nslabs);
    for (int c : slab_cuts) std::printf(" %d", c);
    std::printf("), capacity %lld each, %s step\n", capacity, dfsph ? "DFSPH" : "IISPH");
}

void SPHSolver::StepSlabs() {
    if (slab_team && slab_team_dfsph) check(fsa_slab_dfsph_step(slab_team, 1), "fsa_slab_dfsph_step");
    else if (slab_team) check(fsa_slab_iisph_step(slab_team, 1), "fsa_slab_iisph_step");
    else check(fsa_slab_group_step(slab_group, 1), "fsa_slab_group_step");
}

// Positions and colours of all slabs in the order a single solver would hold them: ascending global cell, inside a
// cell the slab's own order, out-of-grid particles last.
void SPHSolver::CopyPosColorFromSlabs() {
    fsa_slab_group_stats st{};
    if (slab_team) check(fsa_slab_iisph_stats(slab_team, nullptr, &st.lost, nullptr, nullptr, 0), "fsa_slab_iisph_stats");
    else {
        const int rc = fsa_slab_group_sync(slab_group, &st);
        if (rc == FSA_ERR_CAPACITY)
            std::fprintf(stderr, "FSA_SLABS: the slabs are sized when the run starts and not re-cut; raise FSA_SLAB_HEADROOM "
                                 "(now %.2f) for a scene whose mass moves along z\n", g_headroom);
        check(rc, "fsa_slab_group_sync");
    }
    struct Item {
        unsigned long long cell;
        cfloat3 pos;
        int id;
    };
    std::vector<Item> items;
    items.reserve((size_t)numParticles);
    const unsigned gx = (unsigned)hParam.gridres[0], gzz = (unsigned)hParam.gridres[2];
    for (size_t r = 0; r < slab_devices.size(); r++) {
        fsa_solver *s = slab_devices[r];
        int n = 0;
        check(fsa_num_particles(s, &n), "fsa_num_particles");
        vecf3 p((size_t)n);
        veci id((size_t)n), hash((size_t)n);
        check(fsa_get_buffer(s, "pos", p.data(), (size_t)n * sizeof(cfloat3)), "get pos");
        check(fsa_get_buffer(s, "uniqueId", id.data(), (size_t)n * sizeof(int)), "get uniqueId");
        check(fsa_get_buffer(s, "particleHash", hash.data(), (size_t)n * sizeof(int)), "get particleHash");
        const int z0 = std::max(slab_cuts[r] - kHalo, 0);
        const unsigned nz = (unsigned)(std::min(slab_cuts[r + 1] + kHalo, (int)gzz) - z0);
        for (int i = 0; i < n; i++) {
            const unsigned h = (unsigned)hash[i];
            if (h == kUndef) {
                items.push_back({~0ull, p[i], id[i]});
                continue;
            }
            const unsigned x = h % gx, t = h / gx, lz = t % nz, y = t / nz;
            const int z = (int)lz + z0;
            if (z < slab_cuts[r] || z >= slab_cuts[r + 1]) continue;  // a halo copy
            items.push_back({((unsigned long long)y * gzz + (unsigned)z) * gx + x, p[i], id[i]});
        }
    }
    if ((int)items.size() != numParticles) {
        std::fprintf(stderr, "slab read-back: %zu particles owned, %d expected (%lld lost)\n", items.size(), numParticles,
                     st.lost);
        std::exit(EXIT_FAILURE);
    }
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.cell < b.cell; });
    for (int i = 0; i < numParticles; i++) {
        pos[i] = items[i].pos;
        color[i] = slab_color[(size_t)items[i].id];
    }
}
