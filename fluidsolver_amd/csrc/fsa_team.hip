// The IISPH step on z-slabs, all slabs in this process: MultiphaseSPHSolver::SolveIISPH
// (solver/MultiphaseSPHSolver.cpp:173-188 under /root/reference/particle_solver; the pressure rounds:
// solver/sph_solver.cu:1409-1443) for a scene cut along z, one slab solver per slab, on one GPU or several.
//
// The step runs some 25 neighbour passes between two sorts, each reading what the pass before wrote at the
// neighbours.  A slab's halo copies go stale with every pass, so after each pass the fields it wrote are refreshed in
// the halo layers from the slab that owns the particles: the sorted order of a slab solver is (y, layer, x, order
// key), both slabs therefore hold the particles of the layers next to a cut in the same order, and a refresh is a
// gather through the owner's list of sorted positions and a scatter through the neighbour's (fsa_slab.hip); the two
// are ordered by events, the host does not wait.  The particle exchange after the integration is the one of the
// WCSPH slab step with the pressure riding along (the relaxation of a step starts from the pressures of the step
// before).  The mean of error[] that ends the pressure rounds is summed over the particles each slab owns, then over
// the slabs.  Particle counts are known on the host here: one wait per exchange, per sort and per round -- the step
// is a dozen times longer than the WCSPH step, whose slab groups (fsa_group.hip) avoid them.
//
// Results equal the single solver's for every particle, bit for bit (ties inside a cell by order key, as in every
// slab run), and the rounds are the same in number.
//
// The DFSPH step (SPHSolver::SolveDFSPH, solver/sph_solver.cpp:108-120; loops: solver/sph_solver.cu:144-256) runs
// on the same machinery with other fields: v_star rides with the records (three reals) and is refreshed after
// computeNonPForce and every applyPStiff, pstiff after every solve*Stiff, density and DF_factor after
// computeDensityAlpha; a correction loop ends on the maximum of error[] over all slabs' owned particles.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "fsa_solver.h"

namespace fsa {

struct TeamMember {
    Solver *s = nullptr;
    int rank = 0;
    int own[2] = {0, 0}, held[2] = {0, 0}, left[2] = {0, 0}, right[2] = {0, 0};
    bool has[2] = {false, false};        // a neighbour on the left / right
    char *send[2] = {nullptr, nullptr};  // records for that neighbour ...
    char *sendx[2] = {nullptr, nullptr}; // ... and their pressures, slot for slot
    char *recv[2] = {nullptr, nullptr};  // what that neighbour sent (only used across devices)
    char *recvx[2] = {nullptr, nullptr};
    int count_out[2] = {0, 0};
    int *edge[2] = {nullptr, nullptr};   // sorted positions of the owned particles in the layers next to that cut
    int *halo[2] = {nullptr, nullptr};   // ... and of this slab's copies of what the neighbour owns there
    int nedge[2] = {0, 0}, nhalo[2] = {0, 0};
    char *fout[2] = {nullptr, nullptr};  // gathered fields for that neighbour
    char *fin[2] = {nullptr, nullptr};   // (across devices) the neighbour's gathered fields
    hipEvent_t gathered[2] = {nullptr, nullptr};   // fout[d] is written
    hipEvent_t scattered[2] = {nullptr, nullptr};  // this slab has read the neighbour's buffer of side d
    bool scattered_valid[2] = {false, false};
};

struct Team {
    int mode = 0;     // 0 the IISPH step, 1 the DFSPH step
    int carry_w = 1;  // reals beside a record: the pressure / v_star
    int maxiter = 5;  // DFSPH: rounds of a correction loop at most, and its threshold
    double ethres = 1.0;
    std::vector<int> rounds2;  // DFSPH: rounds of the divergence loop (`rounds`: of the density loop)
    int world = 0, halo = 0, record_cap = 0, rb = 0, n_global = 0;
    size_t real = 4;
    std::vector<int> cuts;
    std::vector<TeamMember> m;
    std::vector<int> rounds;  // pressure rounds of every step so far
    long long lost = 0, sent = 0, steps = 0;
    bool set_up = false;
};

static void team_held(const Team &t, int r, int out[2]) {
    if (r < 0 || r >= t.world) {
        out[0] = out[1] = 0;
        return;
    }
    out[0] = std::max(t.cuts[r] - t.halo, 0);
    out[1] = std::min(t.cuts[r + 1] + t.halo, t.cuts.back());
}

static void team_free(Team &t) {
    for (TeamMember &mb : t.m) {
        if (!mb.s) continue;
        (void)hipSetDevice(mb.s->device);
        (void)counted_stream_sync(mb.s->stream);
        for (int d = 0; d < 2; d++) {
            for (void *p : {(void *)mb.send[d], (void *)mb.sendx[d], (void *)mb.recv[d], (void *)mb.recvx[d],
                            (void *)mb.edge[d], (void *)mb.halo[d], (void *)mb.fout[d], (void *)mb.fin[d]})
                if (p) (void)hipFree(p);
            if (mb.gathered[d]) (void)hipEventDestroy(mb.gathered[d]);
            if (mb.scattered[d]) (void)hipEventDestroy(mb.scattered[d]);
        }
    }
}

constexpr int kFieldRealsMax = 7;  // the widest refresh: density + dii + restDensity = 5 reals; head room

static int team_member_setup(Team &t, TeamMember &mb, Solver *s, int rank) {
    if (!s) return fail(FSA_ERR_INVALID, "solver is NULL");
    if (t.mode == 0 && (!s->slab || !s->slab_carry))
        return fail(FSA_ERR_INVALID, "slab %d: the IISPH step on slabs needs slab solvers prepared by fsa_slab_enable_iisph", rank);
    if (t.mode == 1 && (!s->slab || !s->slab_carry_vstar))
        return fail(FSA_ERR_INVALID, "slab %d: the DFSPH step on slabs needs slab solvers prepared by fsa_slab_enable_dfsph", rank);
    if (s->pf.count) return fail(FSA_ERR_INVALID, "slab %d: the solver is a member of a slab group", rank);
    mb.s = s;
    mb.rank = rank;
    mb.own[0] = t.cuts[rank];
    mb.own[1] = t.cuts[rank + 1];
    team_held(t, rank, mb.held);
    team_held(t, rank - 1, mb.left);
    team_held(t, rank + 1, mb.right);
    mb.has[0] = rank > 0;
    mb.has[1] = rank + 1 < t.world;
    if (mb.held[0] != s->pf.z0 || mb.held[1] != s->pf.z0 + s->pf.nz)
        return fail(FSA_ERR_INVALID, "slab %d: the solver holds layers [%d, %d), the cuts ask for [%d, %d)", rank, s->pf.z0,
                    s->pf.z0 + s->pf.nz, mb.held[0], mb.held[1]);
    FSA_HIP(hipSetDevice(s->device));
    s->pf.own_lo = s->pd.own_lo = mb.own[0];
    s->pf.own_hi = s->pd.own_hi = mb.own[1];
    const size_t cap = (size_t)std::max(t.record_cap, 1);
    for (int d = 0; d < 2; d++) {
        if (!mb.has[d]) continue;
        FSA_HIP(hipMalloc((void **)&mb.send[d], cap * t.rb));
        FSA_HIP(hipMalloc((void **)&mb.sendx[d], cap * t.real * t.carry_w));
        FSA_HIP(hipMalloc((void **)&mb.recv[d], cap * t.rb));
        FSA_HIP(hipMalloc((void **)&mb.recvx[d], cap * t.real * t.carry_w));
        // the layers next to a cut hold fewer particles than a message may carry (record_cap covers 2 halo widths)
        FSA_HIP(hipMalloc((void **)&mb.edge[d], cap * sizeof(int)));
        FSA_HIP(hipMalloc((void **)&mb.halo[d], cap * sizeof(int)));
        FSA_HIP(hipMalloc((void **)&mb.fout[d], cap * kFieldRealsMax * t.real));
        FSA_HIP(hipMalloc((void **)&mb.fin[d], cap * kFieldRealsMax * t.real));
        FSA_HIP(hipEventCreateWithFlags(&mb.gathered[d], hipEventDisableTiming));
        FSA_HIP(hipEventCreateWithFlags(&mb.scattered[d], hipEventDisableTiming));
    }
    return FSA_OK;
}

// migration and fresh halo copies, the pressures with them
static int team_exchange(Team &t) {
    for (TeamMember &mb : t.m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        int counts[4] = {0, 0, 0, 0};
        int rc = launch_slab_pack(s, mb.own, mb.held, mb.left, mb.right, mb.send[0], mb.has[0] ? t.record_cap : 0,
                                  mb.send[1], mb.has[1] ? t.record_cap : 0, counts, mb.sendx[0], mb.sendx[1]);  // waits
        if (rc) return rc;
        if (counts[0] > s->capacity) return fail(FSA_ERR_CAPACITY, "slab %d: %d particles stay, capacity %d", mb.rank, counts[0], s->capacity);
        if (counts[1] > t.record_cap || counts[2] > t.record_cap)
            return fail(FSA_ERR_CAPACITY, "slab %d: %d / %d records for messages of %d", mb.rank, counts[1], counts[2], t.record_cap);
        s->n = counts[0];
        s->cur = 1 - s->cur;
        s->sorted_valid = s->hash_valid = s->hash_ahead = s->perm_valid = false;
        s->pd_valid = s->lists_valid = s->hits_valid = s->index_table_valid = s->ref_cells_valid = false;
        s->count_valid = false;
        s->owned_from_sort = false;
        mb.count_out[0] = counts[1];
        mb.count_out[1] = counts[2];
        t.lost += counts[3];
        t.sent += (long long)counts[1] + counts[2];
    }
    for (TeamMember &mb : t.m) {  // left neighbour's records first
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        for (int d = 0; d < 2; d++) {
            if (!mb.has[d]) continue;
            TeamMember &peer = t.m[mb.rank + (d == 0 ? -1 : 1)];
            const int cnt = peer.count_out[1 - d];
            if (cnt == 0) continue;
            if ((long long)s->n + cnt > s->capacity)
                return fail(FSA_ERR_CAPACITY, "slab %d: %d + %d particles exceed capacity %d", mb.rank, s->n, cnt, s->capacity);
            const char *recs = peer.send[1 - d], *extras = peer.sendx[1 - d];
            if (peer.s->device != s->device) {  // (the sender's pack has finished: it waited for its counts)
                FSA_HIP(hipMemcpyPeerAsync(mb.recv[d], s->device, recs, peer.s->device, (size_t)cnt * t.rb, s->stream));
                FSA_HIP(hipMemcpyPeerAsync(mb.recvx[d], s->device, extras, peer.s->device, (size_t)cnt * t.real * t.carry_w,
                                           s->stream));
                recs = mb.recv[d];
                extras = mb.recvx[d];
            }
            int rc = launch_slab_unpack(s, recs, cnt, extras);
            if (rc) return rc;
            s->n += cnt;
        }
    }
    // a sender packs again only after the next sort, which follows these unpacks on the receivers' streams only if
    // the host has waited for them: it does, for the lists of the layers (team_after_sort)
    return FSA_OK;
}

static int team_after_sort(Team &t) {
    for (TeamMember &mb : t.m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        for (int d = 0; d < 2; d++) {
            mb.nedge[d] = mb.nhalo[d] = 0;
            if (!mb.has[d]) continue;
            int e[2], h[2];
            if (d == 0) {
                e[0] = mb.own[0]; e[1] = std::min(mb.own[0] + t.halo, mb.own[1]);
                h[0] = mb.own[0] - t.halo; h[1] = mb.own[0];
            } else {
                e[0] = std::max(mb.own[1] - t.halo, mb.own[0]); e[1] = mb.own[1];
                h[0] = mb.own[1]; h[1] = mb.own[1] + t.halo;
            }
            int rc = launch_slab_layer_indices(s, e[0], e[1], mb.edge[d], t.record_cap, &mb.nedge[d]);  // waits
            if (!rc) rc = launch_slab_layer_indices(s, h[0], h[1], mb.halo[d], t.record_cap, &mb.nhalo[d]);
            if (rc) return rc;
        }
    }
    for (TeamMember &mb : t.m)
        for (int d = 0; d < 2; d++) {
            if (!mb.has[d]) continue;
            const TeamMember &peer = t.m[mb.rank + (d == 0 ? -1 : 1)];
            if (peer.nedge[1 - d] != mb.nhalo[d])
                return fail(FSA_ERR_INVALID, "slab %d holds %d copies of the layers slab %d owns next to the cut, that slab "
                                             "%d particles", mb.rank, mb.nhalo[d], peer.rank, peer.nedge[1 - d]);
        }
    return FSA_OK;
}

struct FieldRef {
    void *ptr;
    int words;
};

static int team_field(Solver *s, const char *name, FieldRef &f) {
    const int w = (int)(s->real_size() / 4);
    const std::string k(name);
    f.ptr = nullptr;
    f.words = w;
    if (k == "density") f.ptr = s->b.density;
    else if (k == "pressure") f.ptr = s->b.pressure;
    else if (k == "restDensity") f.ptr = s->b.restDensity[s->cur];
    else if (k == "v_star") { f.ptr = s->b.v_star; f.words = 3 * w; }
    else if (k == "dii") { f.ptr = s->b.dii; f.words = 3 * w; }
    else if (k == "dijpjl") { f.ptr = s->b.dijpjl; f.words = 3 * w; }
    else if (k == "pstiff") f.ptr = s->b.pstiff;
    else if (k == "DF_factor") f.ptr = s->b.DF_factor;
    if (!f.ptr) return fail(FSA_ERR_UNKNOWN_NAME, "field '%s'", name);
    return FSA_OK;
}

// The named fields of the halo copies become what their owners hold.
static int team_refresh(Team &t, std::initializer_list<const char *> fields) {
    for (TeamMember &mb : t.m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        for (int d = 0; d < 2; d++) {
            if (!mb.has[d]) continue;
            TeamMember &peer = t.m[mb.rank + (d == 0 ? -1 : 1)];
            // the neighbour must have read what the last refresh put into this buffer
            if (peer.scattered_valid[1 - d]) FSA_HIP(hipStreamWaitEvent(s->stream, peer.scattered[1 - d], 0));
            size_t at = 0;
            for (const char *name : fields) {
                FieldRef f;
                int rc = team_field(s, name, f);
                if (!rc) rc = launch_field_move(s, false, f.ptr, f.words, mb.edge[d], mb.nedge[d], mb.fout[d] + at);
                if (rc) return rc;
                at += (size_t)mb.nedge[d] * f.words * 4;
            }
            FSA_HIP(hipEventRecord(mb.gathered[d], s->stream));
        }
    }
    for (TeamMember &mb : t.m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        for (int d = 0; d < 2; d++) {
            if (!mb.has[d]) continue;
            TeamMember &peer = t.m[mb.rank + (d == 0 ? -1 : 1)];
            FSA_HIP(hipStreamWaitEvent(s->stream, peer.gathered[1 - d], 0));
            const char *src = peer.fout[1 - d];
            size_t total = 0;
            for (const char *name : fields) {
                FieldRef f;
                int rc = team_field(s, name, f);
                if (rc) return rc;
                total += (size_t)mb.nhalo[d] * f.words * 4;
            }
            if (peer.s->device != s->device) {
                if (total) FSA_HIP(hipMemcpyPeerAsync(mb.fin[d], s->device, src, peer.s->device, total, s->stream));
                src = mb.fin[d];
            }
            size_t at = 0;
            for (const char *name : fields) {
                FieldRef f;
                int rc = team_field(s, name, f);
                if (!rc) rc = launch_field_move(s, true, f.ptr, f.words, mb.halo[d], mb.nhalo[d], const_cast<char *>(src) + at);
                if (rc) return rc;
                at += (size_t)mb.nhalo[d] * f.words * 4;
            }
            s->pd_valid = false;
            FSA_HIP(hipEventRecord(mb.scattered[d], s->stream));
            mb.scattered_valid[d] = true;
        }
    }
    return FSA_OK;
}

template <typename F> static int team_each(Team &t, F f) {
    for (TeamMember &mb : t.m) {
        FSA_HIP(hipSetDevice(mb.s->device));
        int rc = f(reinterpret_cast<fsa_solver *>(mb.s));
        if (rc) return rc;
    }
    return FSA_OK;
}

static int team_step(Team &t) {
    int rc = team_each(t, fsa_non_pressure_force_multiphase);
    if (!rc) rc = team_refresh(t, {"v_star"});
    if (!rc) rc = team_each(t, fsa_iisph_predict_density);
    if (rc) return rc;
    double last = 0.0;
    int rounds = 0;
    for (int iter = 0; iter < 10; iter++) {  // "temp 10", solver/sph_solver.cu:1416
        rc = team_each(t, fsa_iisph_calc_dijpjl);
        if (!rc) rc = team_refresh(t, {"dijpjl"});
        if (!rc) rc = team_each(t, fsa_iisph_calc_new_pressure);
        if (!rc) rc = team_refresh(t, {"pressure"});
        if (rc) return rc;
        double total = 0.0;
        for (TeamMember &mb : t.m) {
            FSA_HIP(hipSetDevice(mb.s->device));
            double part = 0.0;
            rc = launch_owned_sum(mb.s, mb.own[0], mb.own[1], mb.s->b.error, &part);  // waits
            if (rc) return rc;
            total += part;
        }
        // the mean and the two tests of solver/sph_solver.cu:1440, on values of the solver's precision (as the
        // single solver's round-end kernel takes them)
        const double m = total / (double)t.n_global;
        const double mean = t.real == 4 ? (double)(float)m : m;
        const double a = std::fabs(mean);
        const double b = t.real == 4 ? (double)std::fabs((float)mean - (float)last) : std::fabs(mean - last);
        rounds++;
        if (a < 1e-6 || b < 1e-6) break;
        last = mean;
    }
    if (t.rounds.size() >= 8192) t.rounds.erase(t.rounds.begin(), t.rounds.begin() + 4096);  // (only the latest are asked for)
    t.rounds.push_back(rounds);
    rc = team_each(t, fsa_iisph_update);
    if (!rc) rc = team_exchange(t);
    if (!rc) rc = team_each(t, fsa_sort_particles);
    if (!rc) rc = team_each(t, fsa_iisph_factor);
    if (!rc) rc = team_after_sort(t);
    if (!rc) rc = team_refresh(t, {"density", "dii"});
    if (rc) return rc;
    t.steps++;
    return FSA_OK;
}

// ---- the DFSPH step ---------------------------------------------------------------------------------------
static int team_dfsph_sorted(Team &t) {
    int rc = team_each(t, fsa_sort_particles);
    if (!rc) rc = team_each(t, fsa_compute_density_alpha);
    if (!rc) rc = team_after_sort(t);
    if (!rc) rc = team_refresh(t, {"density", "DF_factor"});
    return rc;
}

// one correction loop (solver/sph_solver.cu:160-185 / :226-250): rounds of solve*Stiff, the test on the maximum of
// error[] over all owned particles, applyPStiff; *applied = applyPStiff passes done
static int team_dfsph_loop(Team &t, int (*solve)(fsa_solver *), double start, int *applied) {
    *applied = 0;
    for (int iter = 0; iter < t.maxiter; iter++) {
        int rc = team_each(t, solve);
        if (!rc) rc = team_refresh(t, {"pstiff"});
        if (rc) return rc;
        double worst = start;
        for (TeamMember &mb : t.m) {
            FSA_HIP(hipSetDevice(mb.s->device));
            double part = start;
            rc = launch_owned_max(mb.s, mb.own[0], mb.own[1], mb.s->b.error, start, &part);  // waits
            if (rc) return rc;
            worst = std::max(worst, part);
        }
        if (worst < t.ethres) break;
        rc = team_each(t, fsa_apply_pstiff);
        if (!rc) rc = team_refresh(t, {"v_star"});
        if (rc) return rc;
        (*applied)++;
    }
    return FSA_OK;
}

static int team_step_dfsph(Team &t) {
    int rc = team_each(t, fsa_compute_non_pressure_force);
    if (!rc) rc = team_refresh(t, {"v_star"});
    int dens = 0, div = 0;
    if (!rc) rc = team_dfsph_loop(t, fsa_solve_density_stiff, -9999.0, &dens);
    if (!rc) rc = team_each(t, fsa_update_position);
    if (!rc) rc = team_exchange(t);
    if (!rc) rc = team_dfsph_sorted(t);
    if (!rc) rc = team_each(t, fsa_solve_divergence_stiff);  // the extra first launch of solver/sph_solver.cu:226
    if (!rc) rc = team_refresh(t, {"pstiff"});
    if (!rc) rc = team_dfsph_loop(t, fsa_solve_divergence_stiff, 0.0, &div);
    if (!rc) rc = team_each(t, fsa_update_velocities);
    if (rc) return rc;
    if (t.rounds.size() >= 8192) {
        t.rounds.erase(t.rounds.begin(), t.rounds.begin() + 4096);
        t.rounds2.erase(t.rounds2.begin(), t.rounds2.begin() + 4096);
    }
    t.rounds.push_back(dens);
    t.rounds2.push_back(div);
    t.steps++;
    return FSA_OK;
}

static int team_create(int mode, int world, fsa_solver **solvers, const int *cuts, int halo, int record_cap, int n_global,
                       Team **out) {
    if (world < 1 || !solvers || !cuts || halo < 1 || record_cap < 0 || n_global < 1) return fail(FSA_ERR_INVALID, "bad argument");
    for (int r = 0; r < world; r++)
        if (cuts[r + 1] - cuts[r] < halo + 1)
            return fail(FSA_ERR_INVALID, "slab %d has %d layers, fewer than halo + 1 = %d", r, cuts[r + 1] - cuts[r], halo + 1);
    Team *t = new (std::nothrow) Team();
    if (!t) return fail(FSA_ERR_NOMEM, "out of host memory");
    t->mode = mode;
    t->carry_w = mode == 1 ? 3 : 1;
    t->world = world;
    t->halo = halo;
    t->record_cap = record_cap;
    t->n_global = n_global;
    t->cuts.assign(cuts, cuts + world + 1);
    Solver *s0 = reinterpret_cast<Solver *>(solvers[0]);
    t->rb = s0 ? slab_record_bytes(s0) : 48;
    t->real = s0 ? s0->real_size() : 4;
    t->m.resize(world);
    for (int r = 0; r < world; r++) {
        Solver *s = reinterpret_cast<Solver *>(solvers[r]);
        int rc = s && s->precision != s0->precision ? fail(FSA_ERR_INVALID, "slab %d: mixed precisions", r)
                                                    : team_member_setup(*t, t->m[r], s, r);
        if (rc) {
            team_free(*t);
            delete t;
            return rc;
        }
    }
    *out = t;
    return FSA_OK;
}

}  // namespace fsa

using namespace fsa;

extern "C" {

int fsa_slab_iisph_create_local(int world, fsa_solver **solvers, const int *cuts, int halo, int record_cap,
                                int n_global, fsa_slab_iisph **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    Team *t = nullptr;
    int rc = team_create(0, world, solvers, cuts, halo, record_cap, n_global, &t);
    if (rc) return rc;
    *out = reinterpret_cast<fsa_slab_iisph *>(t);
    return FSA_OK;
}

int fsa_slab_iisph_setup(fsa_slab_iisph *ht) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return fail(FSA_ERR_INVALID, "team is NULL");
    if (t->mode != 0) return fail(FSA_ERR_INVALID, "not a team of the IISPH step");
    int rc = team_exchange(*t);  // hands out the halo copies
    if (!rc) rc = team_each(*t, fsa_setup_iisph);
    if (!rc) rc = team_after_sort(*t);
    if (!rc) rc = team_refresh(*t, {"density", "dii", "restDensity"});
    if (rc) return rc;
    t->set_up = true;
    return FSA_OK;
}

int fsa_slab_iisph_step(fsa_slab_iisph *ht, int nsteps) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return fail(FSA_ERR_INVALID, "team is NULL");
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    if (t->mode != 0) return fail(FSA_ERR_INVALID, "not a team of the IISPH step");
    if (!t->set_up) return fail(FSA_ERR_INVALID, "fsa_slab_iisph_step needs fsa_slab_iisph_setup first");
    for (int k = 0; k < nsteps; k++) {
        int rc = team_step(*t);
        if (rc) return rc;
    }
    return FSA_OK;
}

int fsa_slab_iisph_stats(fsa_slab_iisph *ht, long long *steps, long long *lost, long long *records_sent, int *rounds,
                         int rounds_cap) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return fail(FSA_ERR_INVALID, "team is NULL");
    for (TeamMember &mb : t->m) {
        FSA_HIP(hipSetDevice(mb.s->device));
        FSA_HIP(counted_stream_sync(mb.s->stream));
    }
    if (steps) *steps = t->steps;
    if (lost) *lost = t->lost;
    if (records_sent) *records_sent = t->sent;
    if (rounds)  // of the last rounds_cap steps, oldest first
        for (int k = 0; k < rounds_cap; k++) {
            const long long i = (long long)t->rounds.size() - rounds_cap + k;
            rounds[k] = i >= 0 ? t->rounds[(size_t)i] : 0;
        }
    return FSA_OK;
}

int fsa_slab_dfsph_create_local(int world, fsa_solver **solvers, const int *cuts, int halo, int record_cap, int maxiter,
                                double ethres, fsa_slab_dfsph **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (maxiter < 0) return fail(FSA_ERR_INVALID, "maxiter < 0");
    Team *t = nullptr;
    int rc = team_create(1, world, solvers, cuts, halo, record_cap, 1, &t);
    if (rc) return rc;
    t->maxiter = maxiter;
    t->ethres = ethres;
    *out = reinterpret_cast<fsa_slab_dfsph *>(t);
    return FSA_OK;
}

int fsa_slab_dfsph_setup(fsa_slab_dfsph *ht) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return fail(FSA_ERR_INVALID, "team is NULL");
    if (t->mode != 1) return fail(FSA_ERR_INVALID, "not a team of the DFSPH step");
    int rc = team_exchange(*t);  // hands out the halo copies
    if (!rc) rc = team_dfsph_sorted(*t);  // SetupDFSPH, solver/sph_solver.cpp:102-106
    if (rc) return rc;
    t->set_up = true;
    return FSA_OK;
}

int fsa_slab_dfsph_step(fsa_slab_dfsph *ht, int nsteps) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return fail(FSA_ERR_INVALID, "team is NULL");
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    if (t->mode != 1) return fail(FSA_ERR_INVALID, "not a team of the DFSPH step");
    if (!t->set_up) return fail(FSA_ERR_INVALID, "fsa_slab_dfsph_step needs fsa_slab_dfsph_setup first");
    for (int k = 0; k < nsteps; k++) {
        int rc = team_step_dfsph(*t);
        if (rc) return rc;
    }
    return FSA_OK;
}

int fsa_slab_dfsph_stats(fsa_slab_dfsph *ht, long long *steps, long long *lost, long long *records_sent, int *density_rounds,
                         int *divergence_rounds, int rounds_cap) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return fail(FSA_ERR_INVALID, "team is NULL");
    for (TeamMember &mb : t->m) {
        FSA_HIP(hipSetDevice(mb.s->device));
        FSA_HIP(counted_stream_sync(mb.s->stream));
    }
    if (steps) *steps = t->steps;
    if (lost) *lost = t->lost;
    if (records_sent) *records_sent = t->sent;
    for (int k = 0; k < rounds_cap; k++) {  // of the last rounds_cap steps, oldest first
        const long long i = (long long)t->rounds.size() - rounds_cap + k;
        if (density_rounds) density_rounds[k] = i >= 0 ? t->rounds[(size_t)i] : 0;
        if (divergence_rounds) divergence_rounds[k] = i >= 0 && (size_t)i < t->rounds2.size() ? t->rounds2[(size_t)i] : 0;
    }
    return FSA_OK;
}

int fsa_slab_dfsph_destroy(fsa_slab_dfsph *ht) { return fsa_slab_iisph_destroy(reinterpret_cast<fsa_slab_iisph *>(ht)); }

int fsa_slab_iisph_destroy(fsa_slab_iisph *ht) {
    Team *t = reinterpret_cast<Team *>(ht);
    if (!t) return FSA_OK;
    team_free(*t);
    delete t;
    return FSA_OK;
}

}  // extern "C"
