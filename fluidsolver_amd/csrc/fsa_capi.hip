// C ABI of the particle-solver step (see include/fluidsolver_amd.h): handle management, upload /
// read-back, call-order checks and the per-step orchestration.  No kernel lives in this file.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <new>

#include "fsa_solver.h"
#include "fsa_tile.h"

namespace fsa {

static std::atomic<long long> g_host_waits{0};
hipError_t counted_stream_sync(hipStream_t st) {
    g_host_waits++;
    return hipStreamSynchronize(st);
}
hipError_t counted_device_sync() {
    g_host_waits++;
    return hipDeviceSynchronize();
}
// An event of some steps ago (slab groups): waiting for it only keeps the host from running further ahead of the
// device, which still has the steps in between queued; counted apart.
static std::atomic<long long> g_event_waits{0};
hipError_t counted_event_sync(hipEvent_t ev) {
    if (hipEventQuery(ev) == hipSuccess) return hipSuccess;
    g_event_waits++;
    return hipEventSynchronize(ev);
}

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- timing ------------------------------------------------------------------------------------
static int kernel_id(Solver *s, const char *name) {
    for (size_t k = 0; k < s->kernel_names.size(); k++)
        if (s->kernel_names[k] == name) return (int)k;
    s->kernel_names.push_back(name);
    return (int)s->kernel_names.size() - 1;
}

static hipEvent_t take_event(Solver *s) {
    if (!s->event_pool.empty()) {
        hipEvent_t e = s->event_pool.back();
        s->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ScopedTimer::ScopedTimer(Solver *solver, const char *name) : s(solver) {
    if (!s->timing) return;
    TimedLaunch t;
    t.kernel_id = kernel_id(s, name);
    t.start = take_event(s);
    t.stop = take_event(s);
    if (!t.start || !t.stop) return;
    (void)hipEventRecord(t.start, s->stream);
    stop = t.stop;
    s->launches.push_back(t);
}

ScopedTimer::~ScopedTimer() {
    if (stop) (void)hipEventRecord(stop, s->stream);
}

// ---- parameters --------------------------------------------------------------------------------
template <typename T> static void to_dev_params(const fsa_params &h, DevParams<T> &d) {
    d.dx = (T)h.dx;
    for (int a = 0; a < 3; a++) {
        d.gmin[a] = (T)h.gridxmin[a];
        d.gmax[a] = (T)h.gridxmax[a];
        d.gres[a] = h.gridres[a];
        d.gravity[a] = (T)h.gravity[a];
    }
    d.dt = (T)h.dt;
    d.viscosity = (T)h.viscosity;
    d.sr = (T)h.smoothradius;
    d.kspikydiff = (T)h.kspikydiff;
    d.kernel_cubic = (T)h.kernel_cubic;
    d.kernel_cubic_gradient = (T)h.kernel_cubic_gradient;
    d.kpoly6 = (T)h.kpoly6;
    d.surface_tension = (T)h.surface_tension;
    d.boundary_friction = (T)h.boundary_friction;
    d.pressureK = (T)h.pressureK;
    d.boundary_visc = (T)h.boundary_visc;
    d.accel_limit = (T)h.acceleration_limit;
    d.boundary_mode = h.boundary_mode;
    d.z0 = h.slab_nz > 0 ? h.slab_z0 : 0;
    d.nz = h.slab_nz > 0 ? h.slab_nz : h.gridres[2];
    d.num_cells = h.gridres[0] * h.gridres[1] * d.nz;
    d.own_lo = -(1 << 30);
    d.own_hi = 1 << 30;
    d.edge[0] = d.edge[1] = d.edge[2] = d.edge[3] = 0;
    d.rows = nullptr;
    d.big = nullptr;
    d.count = nullptr;
    d.halt = nullptr;
}

void make_dev_params(const fsa_params &h, DevParams<float> &d) { to_dev_params(h, d); }

int check_params(const fsa_params *p) {
    if (!p) return fail(FSA_ERR_INVALID, "params is NULL");
    for (int a = 0; a < 3; a++)
        if (p->gridres[a] <= 0) return fail(FSA_ERR_INVALID, "gridres[%d] = %d must be positive", a, p->gridres[a]);
    if (!(p->dx > 0)) return fail(FSA_ERR_INVALID, "dx must be positive");
    if (!(p->smoothradius > 0)) return fail(FSA_ERR_INVALID, "smoothradius must be positive");
    // the 27-cell walk covers the support only if a cell is at least one smoothing radius wide
    if (p->dx < p->smoothradius) return fail(FSA_ERR_INVALID, "dx (%g) < smoothradius (%g)", p->dx, p->smoothradius);
    if (p->slab_nz < 0 || p->slab_z0 < 0 || (p->slab_nz > 0 && p->slab_z0 + p->slab_nz > p->gridres[2]))
        return fail(FSA_ERR_INVALID, "slab [%d, %d) outside the grid (%d layers)", p->slab_z0,
                    p->slab_z0 + p->slab_nz, p->gridres[2]);
    const int nz = p->slab_nz > 0 ? p->slab_nz : p->gridres[2];
    const double cells = (double)p->gridres[0] * p->gridres[1] * nz;
    if (cells >= (double)FSA_GRID_UNDEF)
        return fail(FSA_ERR_INVALID, "%.0f cells: the cell id must stay below GRID_UNDEF (%u)", cells, FSA_GRID_UNDEF);
    if (p->boundary_mode != FSA_BOUNDARY_VERBATIM && p->boundary_mode != FSA_BOUNDARY_LEGACY_CLAMP)
        return fail(FSA_ERR_INVALID, "unknown boundary_mode %d", p->boundary_mode);
    return FSA_OK;
}

static void free_all(Solver *s) {
    Buffers &b = s->b;
    void *ptrs[] = {b.posType[0], b.posType[1], b.velMass[0], b.velMass[1], b.restDensity[0], b.restDensity[1],
                    b.uniqueId[0], b.uniqueId[1], b.density, b.pressure, b.force, b.v_star, b.v_star_alt, b.DF_factor, b.pstiff,
                    b.error, b.reduce_tmp, b.aux4, b.pd, b.pressure_alt, b.dii, b.dijpjl, b.pressureForce, b.aii,
                    b.density_star, b.groupSorted, b.vFrac, b.densArr, b.hash, b.slot,
                    b.tmpIndex, b.tmpHash, b.sortedIndex, b.sortedHash, b.cellCount, b.cellStart, b.gridCellStart,
                    b.gridCellEnd, b.blockRows, b.bigBlocks, b.blockLo, b.blockHi, b.nbList, b.nbCount, b.hitWords, b.hitCount, b.wallCount, b.wallList, b.wallForce, b.applyRec, b.scanTmp, b.undefBlockCount, b.undefBlockStart, b.color, b.normal, b.group,
                    b.indexTable, b.orderKey[0], b.orderKey[1], b.slabCounters, b.slabVars};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &t : s->launches) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto e : s->event_pool) (void)hipEventDestroy(e);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
}

template <typename P> static int dalloc(P *&ptr, size_t bytes, bool zero) {
    void *p = nullptr;
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(FSA_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
        e = hipMemset(p, 0, bytes);
        if (e != hipSuccess) return fail(FSA_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    ptr = (P *)p;
    return FSA_OK;
}

static int activate(Solver *s) {
    FSA_HIP(hipSetDevice(s->device));
    return FSA_OK;
}

// scratch device buffer for host<->device staging of one array
struct Staging {
    void *d = nullptr;
    ~Staging() {
        if (d) (void)hipFree(d);
    }
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 256;
        hipError_t e = hipMalloc(&d, bytes);
        if (e != hipSuccess) return fail(FSA_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return FSA_OK;
    }
};

static void invalidate_order(Solver *s) {
    s->hash_valid = s->hash_ahead = s->perm_valid = s->sorted_valid = false;
    s->index_table_valid = s->ref_cells_valid = false;
    s->pd_valid = false;
    s->lists_valid = false;
    s->hits_valid = false;
}

}  // namespace fsa

using namespace fsa;

// A solver in a slab group keeps its particle count on the device; an entry point that needs it on the host
// waits for the queued steps first.
static int refresh_count(Solver *s) {
    if (!s->count_stale) return FSA_OK;
    FSA_HIP(counted_stream_sync(s->stream));
    int vars[kVarCount];
    FSA_HIP(hipMemcpy(vars, s->b.slabVars, sizeof(vars), hipMemcpyDeviceToHost));
    s->n = vars[kVarN];
    s->count_stale = false;
    if (vars[kVarError])
        return fail(FSA_ERR_CAPACITY, "slab group: a count outgrew its buffer (bits %d); the state is invalid", vars[kVarError]);
    return FSA_OK;
}

#define NEED(s)                                                   \
    do {                                                          \
        if (!(s)) return fail(FSA_ERR_INVALID, "solver is NULL"); \
        int _rc = activate(s);                                    \
        if (!_rc) _rc = refresh_count(s);                         \
        if (_rc) return _rc;                                      \
    } while (0)
#define TRY(expr)          \
    do {                   \
        int _rc = (expr);  \
        if (_rc) return _rc; \
    } while (0)

extern "C" {

const char *fsa_last_error(void) { return g_err; }
const char *fsa_version(void) { return "fluidsolver_amd 0.1 (gfx950)"; }

int fsa_derive_params(fsa_params *p, int precision, double spacing, double smoothratio, const double vmin[3],
                      const double vmax[3]) {
    if (!p || !vmin || !vmax) return fail(FSA_ERR_INVALID, "NULL argument");
    if (precision == FSA_PRECISION_F32) {
        // MultiphaseSPHSolver.cpp:381,452-469 with `float` variables: pow(float, float) is the float
        // overload, pow(float, int) promotes to double, the pi literals are double (3.141593f is not).
        const float sp = (float)spacing;
        const float sr = sp * (float)smoothratio;
        p->spacing = sp;
        p->smoothradius = sr;
        p->dx = sr;
        for (int a = 0; a < 3; a++) {
            const float lo = (float)vmin[a], hi = (float)vmax[a];
            p->gridxmin[a] = lo;
            p->gridxmax[a] = hi;
            p->gridres[a] = (int)roundf((hi - lo) / sr);
        }
        p->kspikydiff = (float)(-45.0f / (3.141592 * (double)powf(sr, 6.0f)));
        p->kernel_cubic = (float)(1 / 3.141593 / pow((double)(sr / 2), 3.0));
        p->kernel_cubic_gradient = (float)(1.5 / (double)3.141593f / pow((double)(sr / 2), 4.0));
        p->kpoly6 = (float)(315.0f / (64.0f * 3.141592 * (double)powf(sr, 9.0f)));
    } else if (precision == FSA_PRECISION_F64) {
        const double sr = spacing * smoothratio;
        p->spacing = spacing;
        p->smoothradius = sr;
        p->dx = sr;
        for (int a = 0; a < 3; a++) {
            p->gridxmin[a] = vmin[a];
            p->gridxmax[a] = vmax[a];
            p->gridres[a] = (int)round((vmax[a] - vmin[a]) / sr);
        }
        p->kspikydiff = -45.0f / (3.141592 * pow(sr, 6.0));
        p->kernel_cubic = 1 / 3.141593 / pow(sr / 2, 3.0);
        p->kernel_cubic_gradient = 1.5 / (double)3.141593f / pow(sr / 2, 4.0);
        p->kpoly6 = 315.0f / (64.0f * 3.141592 * pow(sr, 9.0));
    } else {
        return fail(FSA_ERR_INVALID, "unknown precision %d", precision);
    }
    return FSA_OK;
}

int fsa_create(const fsa_params *p, int precision, int capacity, int device, fsa_solver **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    TRY(check_params(p));
    if (precision != FSA_PRECISION_F32 && precision != FSA_PRECISION_F64)
        return fail(FSA_ERR_INVALID, "unknown precision %d", precision);
    if (capacity < 0) return fail(FSA_ERR_INVALID, "capacity %d < 0", capacity);
    int ndev = 0;
    FSA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(FSA_ERR_INVALID, "device %d of %d", device, ndev);

    Solver *s = new (std::nothrow) Solver();
    if (!s) return fail(FSA_ERR_NOMEM, "out of host memory");
    s->precision = precision;
    s->device = device;
    s->capacity = capacity;
    s->hp = *p;
    to_dev_params(*p, s->pf);
    to_dev_params(*p, s->pd);
    s->num_cells = s->pf.num_cells;
    s->slab = p->slab_nz > 0;
    // experiment switches, read here once and never in a launch path
    if (const char *e = getenv("FSA_XCD_RUN_LOG")) s->xcd_run_log = std::min(12, std::max(0, atoi(e)));
#ifdef FSA_ABLATE
    if (const char *e = getenv("FSA_FAST_DBG")) s->dbg_fast = atoi(e);
#endif
    if (const char *e = getenv("FSA_HIT_REPLAY")) s->use_replay = atoi(e) != 0;
    if (const char *e = getenv("FSA_WALL_PASS")) s->wall_mode = atoi(e) == 0 ? 1 : 2;

    int rc = activate(s);
    if (rc == FSA_OK && hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(FSA_ERR_HIP, "hipStreamCreate failed");
    s->stream = s->own_stream;

    const size_t N = (size_t)capacity, R = s->real_size(), C = (size_t)s->num_cells;
    const size_t nblocks = (N + kBlock - 1) / kBlock;
    const size_t scan_tiles = (C + 1) / (kBlock * 8) + (nblocks + 1) / (kBlock * 8) + 8;  // both prefix sums at once
    Buffers &b = s->b;
#define ALLOC(field, bytes, zero) \
    if (rc == FSA_OK) rc = dalloc(field, (bytes), (zero))
    for (int k = 0; k < 2; k++) {
        ALLOC(b.posType[k], N * 4 * R, true);
        ALLOC(b.velMass[k], N * 4 * R, true);
        ALLOC(b.restDensity[k], N * R, true);
        ALLOC(b.uniqueId[k], N * sizeof(int), true);
    }
    // the reference leaves these uninitialised (cudaMalloc); zero is this implementation's choice
    ALLOC(b.density, N * R, true);
    ALLOC(b.pressure, N * R, true);  // MultiphaseSPHSolver.cpp:975 does clear this one
    ALLOC(b.force, N * 3 * R, true);
    ALLOC(b.v_star, N * 3 * R, true);
    ALLOC(b.v_star_alt, N * 3 * R, true);
    ALLOC(b.DF_factor, N * R, true);
    ALLOC(b.pstiff, N * R, true);
    ALLOC(b.error, N * R, true);
    ALLOC(b.reduce_tmp, 2048 * sizeof(double), true);
    ALLOC(b.aux4, N * 4 * R, true);
    ALLOC(b.pd, N * 2 * R, true);
    ALLOC(b.hash, N * sizeof(int), true);
    ALLOC(b.slot, N * sizeof(int), true);
    ALLOC(b.tmpIndex, N * sizeof(int), true);
    ALLOC(b.tmpHash, N * sizeof(int), true);
    ALLOC(b.sortedIndex, N * sizeof(int), true);
    ALLOC(b.sortedHash, N * sizeof(int), true);
    ALLOC(b.cellCount, (C + 8) * sizeof(int), true);
    ALLOC(b.cellStart, (C + 8) * sizeof(int), true);
    ALLOC(b.gridCellStart, C * sizeof(int), true);
    ALLOC(b.gridCellEnd, C * sizeof(int), true);
    ALLOC(b.blockRows, (nblocks + 8) * 20 * sizeof(int), true);
    ALLOC(b.bigBlocks, (nblocks + 8) * sizeof(int), true);
    ALLOC(b.blockLo, (nblocks + 8) * sizeof(int), true);
    ALLOC(b.blockHi, (nblocks + 8) * sizeof(int), true);
    if (precision == FSA_PRECISION_F32) {  // recorded hits of the WCSPH passes (fsa_sph_fast.hip)
        ALLOC(b.hitWords, (nblocks + 8) * kHitSlots * kTunedBS * 8, false);
        ALLOC(b.hitCount, (nblocks + 8) * kTunedBS, true);
        ALLOC(b.wallCount, (nblocks + 8) * kTunedBS, true);
        ALLOC(b.wallList, (N + 8) * sizeof(int), true);
        ALLOC(b.wallForce, (N + 8) * 4 * sizeof(float), false);
        ALLOC(b.applyRec, (N + 8) * 4 * sizeof(float), false);
    }
    ALLOC(b.scanTmp, scan_tiles * sizeof(int), true);
    ALLOC(b.undefBlockCount, (nblocks + 8) * sizeof(int), true);
    ALLOC(b.undefBlockStart, (nblocks + 8) * sizeof(int), true);
    if (!s->slab) {  // indexed by uniqueId, which is a global id in a slab run
        ALLOC(b.color, N * 4 * R, true);
        ALLOC(b.normal, N * 3 * R, true);
        ALLOC(b.group, N * sizeof(int), true);
        ALLOC(b.indexTable, N * sizeof(int), true);
        // IISPH step (not available in slab solvers)
        ALLOC(b.pressure_alt, N * R, true);
        ALLOC(b.dii, N * 3 * R, true);
        ALLOC(b.dijpjl, N * 3 * R, true);
        ALLOC(b.pressureForce, N * 3 * R, true);
        ALLOC(b.aii, N * R, true);
        ALLOC(b.density_star, N * R, true);
        ALLOC(b.groupSorted, N * sizeof(int), true);
    } else {
        ALLOC(b.orderKey[0], N * sizeof(long long), true);
        ALLOC(b.orderKey[1], N * sizeof(long long), true);
        ALLOC(b.slabCounters, 8 * sizeof(int), true);
        ALLOC(b.slabVars, 16 * sizeof(int), true);
    }
#undef ALLOC
    if (rc != FSA_OK) {
        free_all(s);
        delete s;
        return rc;
    }
    s->pf.rows = s->b.blockRows;
    s->pf.big = s->b.bigBlocks;
    // the zero fills above ran on the NULL stream, which the solver's non-blocking stream does not wait for
    if (counted_device_sync() != hipSuccess) {
        free_all(s);
        delete s;
        return fail(FSA_ERR_HIP, "hipDeviceSynchronize failed after the allocations");
    }
    *out = reinterpret_cast<fsa_solver *>(s);
    return FSA_OK;
}

int fsa_destroy(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s) return FSA_OK;
    (void)hipSetDevice(s->device);
    (void)counted_stream_sync(s->stream);
    free_all(s);
    delete s;
    return FSA_OK;
}

int fsa_copy_param_to_device(fsa_solver *hs, const fsa_params *p) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    TRY(check_params(p));
    DevParams<float> nf;
    to_dev_params(*p, nf);
    if (nf.num_cells != s->num_cells || nf.gres[0] != s->pf.gres[0] || nf.gres[1] != s->pf.gres[1] ||
        nf.gres[2] != s->pf.gres[2] || nf.z0 != s->pf.z0 || nf.nz != s->pf.nz)
        return fail(FSA_ERR_INVALID, "grid resolution / slab differ from the ones the solver was created with");
    if (s->pf.count) return fail(FSA_ERR_INVALID, "the solver is a member of a slab group: destroy the group first");
    const int own_lo = s->pf.own_lo, own_hi = s->pf.own_hi;
    s->hp = *p;
    to_dev_params(*p, s->pf);
    to_dev_params(*p, s->pd);
    s->pf.own_lo = s->pd.own_lo = own_lo;
    s->pf.own_hi = s->pd.own_hi = own_hi;
    s->pf.rows = s->b.blockRows;
    s->pf.big = s->b.bigBlocks;
    invalidate_order(s);
    return FSA_OK;
}

int fsa_set_stream(fsa_solver *hs, void *hip_stream) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    FSA_HIP(counted_stream_sync(s->stream));
    s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
    return FSA_OK;
}

int fsa_synchronize(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    FSA_HIP(counted_stream_sync(s->stream));
    return FSA_OK;
}

int fsa_upload(fsa_solver *hs, int n, const void *pos3, const void *vel3, const void *color4, const void *normal3,
               const int *type, const int *group, const int *uniqueId, const void *mass, const void *restDensity) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (n < 0 || n > s->capacity) return fail(FSA_ERR_CAPACITY, "n = %d exceeds capacity %d", n, s->capacity);
    if (n > 0 && (!pos3 || !vel3 || !type || !mass || !restDensity))
        return fail(FSA_ERR_INVALID, "pos, vel, type, mass and restDensity are required");
    const size_t N = (size_t)n, R = s->real_size();
    if (s->slab && (color4 || normal3 || group))
        return fail(FSA_ERR_INVALID, "a slab solver does not store color / normal / group: pass NULL");
    if (uniqueId && !s->slab) {
        std::vector<unsigned char> seen(N, 0);
        for (size_t i = 0; i < N; i++) {
            const int u = uniqueId[i];
            if (u < 0 || (size_t)u >= N || seen[u])
                return fail(FSA_ERR_INVALID, "uniqueId must be a permutation of 0..n-1 (entry %zu = %d)", i, u);
            seen[u] = 1;
        }
    }
    FSA_HIP(counted_stream_sync(s->stream));
    s->has_rigid = false;
    for (size_t i = 0; i < N; i++)
        if (type[i] == FSA_TYPE_RIGID) s->has_rigid = true;
    s->fluid_mass_uniform = false;
    if (R == sizeof(float)) {  // one mass for all fluid particles? (bit patterns compared)
        const unsigned *m = (const unsigned *)mass;
        bool any = false, same = true;
        unsigned first = 0;
        for (size_t i = 0; i < N && same; i++) {
            if (type[i] != FSA_TYPE_FLUID) continue;
            if (!any) {
                first = m[i];
                any = true;
            } else if (m[i] != first) {
                same = false;
            }
        }
        s->fluid_mass_uniform = any && same;
    }
    s->n = n;
    s->cur = 0;
    invalidate_order(s);
    s->count_valid = false;
    s->owned_from_sort = false;
    s->iisph = false;  // the per-position copy of group[] is rebuilt by the next IISPH call
    if (s->typenum) {  // volume fractions belonged to the particles uploaded before
        (void)hipFree(s->b.vFrac);
        (void)hipFree(s->b.densArr);
        s->b.vFrac = s->b.densArr = nullptr;
        s->typenum = 0;
    }
    if (n == 0) return FSA_OK;

    Staging sp, sv, st, sm;
    TRY(sp.alloc(N * 3 * R));
    TRY(sv.alloc(N * 3 * R));
    TRY(st.alloc(N * sizeof(int)));
    TRY(sm.alloc(N * R));
    FSA_HIP(hipMemcpy(sp.d, pos3, N * 3 * R, hipMemcpyHostToDevice));
    FSA_HIP(hipMemcpy(sv.d, vel3, N * 3 * R, hipMemcpyHostToDevice));
    FSA_HIP(hipMemcpy(st.d, type, N * sizeof(int), hipMemcpyHostToDevice));
    FSA_HIP(hipMemcpy(sm.d, mass, N * R, hipMemcpyHostToDevice));
    TRY(launch_pack(s, n, sp.d, sv.d, (const int *)st.d, sm.d));
    FSA_HIP(hipMemcpy(s->b.restDensity[0], restDensity, N * R, hipMemcpyHostToDevice));

    std::vector<int> uid(N);
    for (size_t i = 0; i < N; i++) uid[i] = uniqueId ? uniqueId[i] : (int)i;
    FSA_HIP(hipMemcpy(s->b.uniqueId[0], uid.data(), N * sizeof(int), hipMemcpyHostToDevice));

    if (s->slab) {
        TRY(launch_slab_init_keys(s));
    }
    // colour, normal and group never change on this path: keep them in uniqueId order
    auto store_static = [&](void *dst, const void *src, size_t elem_bytes) -> int {
        if (!src) {
            FSA_HIP(hipMemsetAsync(dst, 0, N * elem_bytes, s->stream));  // waited for at the end of fsa_upload
            return FSA_OK;
        }
        std::vector<unsigned char> tmp(N * elem_bytes);
        const unsigned char *in = (const unsigned char *)src;
        for (size_t i = 0; i < N; i++) memcpy(&tmp[(size_t)uid[i] * elem_bytes], in + i * elem_bytes, elem_bytes);
        FSA_HIP(hipMemcpy(dst, tmp.data(), N * elem_bytes, hipMemcpyHostToDevice));
        return FSA_OK;
    };
    if (!s->slab) {
        TRY(store_static(s->b.color, color4, 4 * R));
        TRY(store_static(s->b.normal, normal3, 3 * R));
        TRY(store_static(s->b.group, group, sizeof(int)));
    }
    // start from the zero state the constructor gives (on the solver's own stream, which does not wait for the
    // NULL stream)
    FSA_HIP(hipMemsetAsync(s->b.force, 0, N * 3 * R, s->stream));
    FSA_HIP(hipMemsetAsync(s->b.v_star, 0, N * 3 * R, s->stream));
    FSA_HIP(hipMemsetAsync(s->b.density, 0, N * R, s->stream));
    FSA_HIP(hipMemsetAsync(s->b.pressure, 0, N * R, s->stream));
    FSA_HIP(counted_stream_sync(s->stream));
    return FSA_OK;
}

int fsa_num_particles(fsa_solver *hs, int *n) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s || !n) return fail(FSA_ERR_INVALID, "NULL argument");
    NEED(s);  // in a slab group the count comes from the device
    *n = s->n;
    return FSA_OK;
}

int fsa_num_undef(fsa_solver *hs, int *n) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!n) return fail(FSA_ERR_INVALID, "n is NULL");
    if (!s->count_valid) return fail(FSA_ERR_INVALID, "no sort has run since the last upload");
    int in_grid = 0;
    FSA_HIP(counted_stream_sync(s->stream));
    FSA_HIP(hipMemcpy(&in_grid, s->b.cellStart + s->num_cells, sizeof(int), hipMemcpyDeviceToHost));
    *n = s->n - in_grid;
    return FSA_OK;
}

// ---- the operator seam -------------------------------------------------------------------------
int fsa_calc_hash(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->hash_ahead) TRY(launch_hash_count(s));  // else the last step's integration has done it on the way
    s->hash_ahead = false;
    s->hash_valid = true;
    s->perm_valid = false;
    s->sorted_valid = false;
    s->ref_cells_valid = false;
    return FSA_OK;
}

int fsa_sort_particle(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->hash_valid) return fail(FSA_ERR_INVALID, "fsa_sort_particle needs fsa_calc_hash first");
    TRY(launch_scan_cells(s));
    TRY(launch_scatter(s));
    TRY(launch_rank(s));
    s->perm_valid = true;
    s->count_valid = true;
    return FSA_OK;
}

int fsa_reorder_data_and_find_cell_start(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->perm_valid) return fail(FSA_ERR_INVALID, "reorder needs fsa_sort_particle first");
    TRY(launch_reorder(s));
    s->cur = 1 - s->cur;
    s->sorted_valid = true;
    s->owned_from_sort = true;
    s->pd_valid = false;
    s->lists_valid = false;
    s->hits_valid = false;
    s->hash_valid = s->hash_ahead = false;  // hash[] described the previous order
    s->perm_valid = false;
    s->index_table_valid = false;
    s->ref_cells_valid = false;
    return FSA_OK;
}

int fsa_sort_particles(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    // the three stages in five launches (see fsa_step); same arrays, same bits (z-slab solvers: ties inside a cell
    // by order key, and the new keys written, in the same rank pass)
    if (!s->hash_valid && !s->hash_ahead) TRY(launch_hash_fused(s, false));
    TRY(launch_sort_fused(s, true));
    s->cur = 1 - s->cur;
    s->hash_valid = s->hash_ahead = false;  // hash[] described the previous order
    s->sorted_valid = s->owned_from_sort = s->count_valid = true;
    s->pd_valid = s->lists_valid = s->hits_valid = false;
    s->perm_valid = s->index_table_valid = s->ref_cells_valid = false;
    return FSA_OK;
}

int fsa_compute_pressure(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "ComputePressure needs a sorted state");
    return launch_density(s);
}

int fsa_compute_force(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "computeForce needs a sorted state");
    return launch_force(s);
}

int fsa_advect(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    TRY(launch_advect(s));
    // positions moved: the cell ranges no longer describe them
    s->sorted_valid = false;
    s->hash_valid = s->hash_ahead = false;
    s->perm_valid = false;
    return FSA_OK;
}

int fsa_rigid_particle_volume(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "RigidParticleVolume needs a sorted state");
    return launch_rigid_volume(s);
}

int fsa_step(fsa_solver *hs, int nsteps) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    if (s->slab) {  // the particle set changes between integration and sort (exchange): no hash ahead of time
        for (int k = 0; k < nsteps; k++) {
            TRY(fsa_sort_particles(hs));
            TRY(launch_density(s));
            TRY(launch_force(s));
            TRY(fsa_advect(hs));
        }
        return FSA_OK;
    }
    // Eight launches per step (seven plus the wall pass): the integration computes the next step's hash and counts on the way, the two
    // prefix sums share two launches, the rank pass moves the particles and writes the row table.
    for (int k = 0; k < nsteps; k++) {
        if (!s->hash_valid && !s->hash_ahead) TRY(launch_hash_fused(s, false));
        TRY(launch_sort_fused(s, false));
        s->hash_valid = s->hash_ahead = false;  // hash[] described the previous order
        s->cur = 1 - s->cur;
        s->sorted_valid = true;
        s->owned_from_sort = true;
        s->count_valid = true;
        s->pd_valid = s->lists_valid = s->hits_valid = false;
        s->perm_valid = s->index_table_valid = s->ref_cells_valid = false;
        TRY(launch_density(s));
        TRY(launch_force(s));
        TRY(launch_hash_fused(s, true));
        s->sorted_valid = false;  // positions moved: the cell ranges no longer describe them ...
        s->hash_ahead = true;     // ... but hash / slot / cellCount already do
    }
    return FSA_OK;
}

// ---- read-back ---------------------------------------------------------------------------------
int fsa_readback_pos_color(fsa_solver *hs, void *pos3, void *color4, int *n) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (n) *n = s->n;
    const size_t N = (size_t)s->n, R = s->real_size();
    if (N == 0) return FSA_OK;
    if (pos3) {
        Staging st;
        TRY(st.alloc(N * 3 * R));
        TRY(launch_unpack_vec3(s, s->b.posType[s->cur], st.d, s->n));
        FSA_HIP(counted_stream_sync(s->stream));
        FSA_HIP(hipMemcpy(pos3, st.d, N * 3 * R, hipMemcpyDeviceToHost));
    }
    if (color4 && s->slab) return fail(FSA_ERR_INVALID, "a slab solver does not store color");
    if (color4) {
        Staging st;
        TRY(st.alloc(N * 4 * R));
        TRY(launch_gather_by_uid(s, s->b.color, st.d, s->n, (int)(4 * R / 4)));
        FSA_HIP(counted_stream_sync(s->stream));
        FSA_HIP(hipMemcpy(color4, st.d, N * 4 * R, hipMemcpyDeviceToHost));
    }
    return FSA_OK;
}

static int copy_out(Solver *s, const void *dsrc, void *dst, size_t bytes) {
    FSA_HIP(counted_stream_sync(s->stream));
    if (bytes) FSA_HIP(hipMemcpy(dst, dsrc, bytes, hipMemcpyDeviceToHost));
    return FSA_OK;
}

int fsa_get_buffer(fsa_solver *hs, const char *name, void *dst, size_t bytes) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!name || (!dst && bytes)) return fail(FSA_ERR_INVALID, "NULL argument");
    const size_t N = (size_t)s->n, R = s->real_size(), C = (size_t)s->num_cells;
    const std::string k(name);
    auto want = [&](size_t expect) -> int {
        if (bytes != expect)
            return fail(FSA_ERR_INVALID, "buffer '%s' is %zu bytes, caller passed %zu", name, expect, bytes);
        return FSA_OK;
    };
    const int c = s->cur;
    if (k == "pos" || k == "vel") {
        TRY(want(N * 3 * R));
        Staging st;
        TRY(st.alloc(bytes));
        TRY(launch_unpack_vec3(s, k == "pos" ? s->b.posType[c] : s->b.velMass[c], st.d, s->n));
        return copy_out(s, st.d, dst, bytes);
    }
    if (k == "type" || k == "mass") {
        TRY(want(k == "type" ? N * sizeof(int) : N * R));
        Staging st;
        TRY(st.alloc(bytes));
        TRY(launch_unpack_w(s, k == "type" ? s->b.posType[c] : s->b.velMass[c], st.d, s->n, k == "type"));
        return copy_out(s, st.d, dst, bytes);
    }
    if (s->slab && (k == "color" || k == "normal" || k == "group" || k == "indexTable"))
        return fail(FSA_ERR_INVALID, "a slab solver does not store '%s' (uniqueId is a global id there)", name);
    if (k == "color" || k == "normal" || k == "group") {
        const size_t words = k == "color" ? 4 * R / 4 : k == "normal" ? 3 * R / 4 : 1;
        TRY(want(N * words * 4));
        Staging st;
        TRY(st.alloc(bytes));
        const void *table = k == "color" ? s->b.color : k == "normal" ? s->b.normal : (const void *)s->b.group;
        TRY(launch_gather_by_uid(s, table, st.d, s->n, (int)words));
        return copy_out(s, st.d, dst, bytes);
    }
    if (k == "neighbourCount") {  // entries of each particle's recorded list (builds the lists if stale); -1 = walks the cells
        TRY(want(N * sizeof(int)));
        if (!s->sorted_valid || s->precision != FSA_PRECISION_F32 || !s->use_lists)
            return fail(FSA_ERR_INVALID, "neighbourCount needs a sorted fp32 state with lists enabled");
        TRY(ensure_lists(s));
        std::vector<unsigned short> h(N);
        FSA_HIP(counted_stream_sync(s->stream));
        if (N) FSA_HIP(hipMemcpy(h.data(), s->b.nbCount, N * sizeof(unsigned short), hipMemcpyDeviceToHost));
        int *out = (int *)dst;
        for (size_t i = 0; i < N; i++) out[i] = h[i] == 0xffff ? -1 : (int)h[i];
        return FSA_OK;
    }
    if (k == "hitCount") {  // extra words of each particle's recorded hits of the last variant-3 pass; -1 = walks the cells
        TRY(want(N * sizeof(int)));
        if (!s->hits_valid) return fail(FSA_ERR_INVALID, "hitCount needs the record of a variant-3 density or force pass");
        std::vector<unsigned char> h(N);
        FSA_HIP(counted_stream_sync(s->stream));
        if (N) FSA_HIP(hipMemcpy(h.data(), s->b.hitCount, N, hipMemcpyDeviceToHost));
        int *out = (int *)dst;
        for (size_t i = 0; i < N; i++) out[i] = h[i] == 255 ? -1 : (int)h[i];
        return FSA_OK;
    }
    if (k == "force") { TRY(want(N * 3 * R)); return copy_out(s, s->b.force, dst, bytes); }
    if (k == "v_star") { TRY(want(N * 3 * R)); return copy_out(s, s->b.v_star, dst, bytes); }
    if (k == "density") { TRY(want(N * R)); return copy_out(s, s->b.density, dst, bytes); }
    if (k == "pressure") { TRY(want(N * R)); return copy_out(s, s->b.pressure, dst, bytes); }
    if (k == "DF_factor") { TRY(want(N * R)); return copy_out(s, s->b.DF_factor, dst, bytes); }
    if (k == "pstiff") { TRY(want(N * R)); return copy_out(s, s->b.pstiff, dst, bytes); }
    if (k == "error") { TRY(want(N * R)); return copy_out(s, s->b.error, dst, bytes); }
    if (k == "dii" || k == "dijpjl" || k == "pressureForce" || k == "aii" || k == "density_star") {
        if (s->slab && !s->slab_carry) return fail(FSA_ERR_INVALID, "a slab solver has no IISPH buffers");
        const bool vec = k == "dii" || k == "dijpjl" || k == "pressureForce";
        TRY(want(N * (vec ? 3 : 1) * R));
        const void *p = k == "dii" ? s->b.dii : k == "dijpjl" ? s->b.dijpjl : k == "pressureForce" ? s->b.pressureForce
                        : k == "aii" ? s->b.aii : s->b.density_star;
        return copy_out(s, p, dst, bytes);
    }
    if (k == "restDensity") { TRY(want(N * R)); return copy_out(s, s->b.restDensity[c], dst, bytes); }
    if (k == "uniqueId") { TRY(want(N * sizeof(int))); return copy_out(s, s->b.uniqueId[c], dst, bytes); }
    if (k == "orderKey") {
        if (!s->slab) return fail(FSA_ERR_INVALID, "orderKey exists in slab solvers only");
        TRY(want(N * sizeof(long long)));
        return copy_out(s, s->b.orderKey[c], dst, bytes);
    }
    if (k == "indexTable") {
        TRY(want(N * sizeof(int)));
        if (!s->index_table_valid) {
            TRY(launch_index_table(s));
            s->index_table_valid = true;
        }
        return copy_out(s, s->b.indexTable, dst, bytes);
    }
    if (k == "particleHash") {
        TRY(want(N * sizeof(int)));
        if (s->perm_valid) return copy_out(s, s->b.sortedHash, dst, bytes);
        if (s->hash_valid) return copy_out(s, s->b.hash, dst, bytes);
        if (s->sorted_valid) return copy_out(s, s->b.sortedHash, dst, bytes);
        // after the integration: still the cells of the last sort, as in the reference, where Advect
        // leaves particleHash alone
        if (s->owned_from_sort) return copy_out(s, s->b.sortedHash, dst, bytes);
        return fail(FSA_ERR_INVALID, "particleHash is not defined for the current state");
    }
    if (k == "particleIndex") {
        TRY(want(N * sizeof(int)));
        if (s->perm_valid || s->sorted_valid) return copy_out(s, s->b.sortedIndex, dst, bytes);
        if (s->hash_valid) {  // calcHashD writes the identity, solver/sph_kernel_impl.cuh:110
            int *o = (int *)dst;
            for (size_t i = 0; i < N; i++) o[i] = (int)i;
            return FSA_OK;
        }
        return fail(FSA_ERR_INVALID, "particleIndex is not defined for the current state");
    }
    if (k == "gridCellStart" || k == "gridCellEnd") {
        TRY(want(C * sizeof(int)));
        if (!s->sorted_valid && !s->perm_valid) return fail(FSA_ERR_INVALID, "no sort has run on the current state");
        if (!s->ref_cells_valid) {
            TRY(launch_ref_cells(s));
            s->ref_cells_valid = true;
        }
        return copy_out(s, k == "gridCellStart" ? s->b.gridCellStart : s->b.gridCellEnd, dst, bytes);
    }
    if (k == "cellStartDense") {
        TRY(want((C + 1) * sizeof(int)));
        return copy_out(s, s->b.cellStart, dst, bytes);
    }
    return fail(FSA_ERR_UNKNOWN_NAME, "unknown buffer '%s'", name);
}

int fsa_set_buffer(fsa_solver *hs, const char *name, const void *src, size_t bytes) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!name || (!src && bytes)) return fail(FSA_ERR_INVALID, "NULL argument");
    const size_t N = (size_t)s->n, R = s->real_size();
    const std::string k(name);
    auto want = [&](size_t expect) -> int {
        if (bytes != expect)
            return fail(FSA_ERR_INVALID, "buffer '%s' is %zu bytes, caller passed %zu", name, expect, bytes);
        return FSA_OK;
    };
    const int c = s->cur;
    FSA_HIP(counted_stream_sync(s->stream));
    s->pd_valid = false;  // density, pressure, mass or viscosity may change under it
    s->lists_valid = false;
    s->hits_valid = false;
    // a member of a slab group: its next step is already prepared from the old values (records on their way to
    // the neighbours); the group notices by this flag and refuses to go on
    if (s->pf.count) s->hash_ahead = false;
    auto plain = [&](void *ddst, size_t expect) -> int {
        TRY(want(expect));
        if (bytes) FSA_HIP(hipMemcpy(ddst, src, bytes, hipMemcpyHostToDevice));
        return FSA_OK;
    };
    if (k == "force") return plain(s->b.force, N * 3 * R);
    if (k == "v_star") return plain(s->b.v_star, N * 3 * R);
    if (k == "density") return plain(s->b.density, N * R);
    if (k == "pressure") return plain(s->b.pressure, N * R);
    if (k == "DF_factor") return plain(s->b.DF_factor, N * R);
    if (k == "pstiff") return plain(s->b.pstiff, N * R);
    if (k == "error") return plain(s->b.error, N * R);
    if (!s->slab || s->slab_carry) {
        if (k == "dii") return plain(s->b.dii, N * 3 * R);
        if (k == "dijpjl") return plain(s->b.dijpjl, N * 3 * R);
        if (k == "pressureForce") return plain(s->b.pressureForce, N * 3 * R);
        if (k == "aii") return plain(s->b.aii, N * R);
        if (k == "density_star") return plain(s->b.density_star, N * R);
    }
    if (k == "restDensity") return plain(s->b.restDensity[c], N * R);
    if (k == "pos" || k == "vel") {
        TRY(want(N * 3 * R));
        Staging st;
        TRY(st.alloc(bytes));
        if (bytes) FSA_HIP(hipMemcpy(st.d, src, bytes, hipMemcpyHostToDevice));
        TRY(launch_set_xyz(s, k == "pos" ? s->b.posType[c] : s->b.velMass[c], st.d, s->n));
        FSA_HIP(counted_stream_sync(s->stream));
        if (k == "pos") invalidate_order(s);
        return FSA_OK;
    }
    if (k == "type" || k == "mass") {
        TRY(want(k == "type" ? N * sizeof(int) : N * R));
        Staging st;
        TRY(st.alloc(bytes));
        if (bytes) FSA_HIP(hipMemcpy(st.d, src, bytes, hipMemcpyHostToDevice));
        // the one-mass flavour of applyPStiff holds only for the masses and types handed over by fsa_upload: any
        // later change of either goes back to the general kernels (not looked at again)
        s->fluid_mass_uniform = false;
        if (k == "type") {
            s->has_rigid = false;
            for (size_t i = 0; i < N; i++)
                if (((const int *)src)[i] == FSA_TYPE_RIGID) s->has_rigid = true;
        }
        TRY(launch_set_w(s, k == "type" ? s->b.posType[c] : s->b.velMass[c], st.d, s->n, k == "type"));
        FSA_HIP(counted_stream_sync(s->stream));
        return FSA_OK;
    }
    return fail(FSA_ERR_UNKNOWN_NAME, "buffer '%s' cannot be set", name);
}

// ---- measurement -------------------------------------------------------------------------------
int fsa_timing_enable(fsa_solver *hs, int on) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    s->timing = on != 0;
    return FSA_OK;
}

int fsa_timing_reset(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    FSA_HIP(counted_stream_sync(s->stream));
    for (auto &t : s->launches) {
        s->event_pool.push_back(t.start);
        s->event_pool.push_back(t.stop);
    }
    s->launches.clear();
    return FSA_OK;
}

int fsa_timing_read(fsa_solver *hs, fsa_kernel_time *out, int cap, int *count) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!count || (cap > 0 && !out)) return fail(FSA_ERR_INVALID, "NULL argument");
    FSA_HIP(counted_stream_sync(s->stream));
    const int K = (int)s->kernel_names.size();
    std::vector<double> ms(K, 0.0);
    std::vector<int> cnt(K, 0);
    for (auto &t : s->launches) {
        float e = 0.f;
        FSA_HIP(hipEventElapsedTime(&e, t.start, t.stop));
        ms[t.kernel_id] += e;
        cnt[t.kernel_id]++;
    }
    int w = 0;
    for (int k = 0; k < K; k++) {
        if (cnt[k] == 0) continue;
        if (w < cap) {
            memset(&out[w], 0, sizeof(out[w]));
            strncpy(out[w].name, s->kernel_names[k].c_str(), FSA_KERNEL_NAME_MAX - 1);
            out[w].launches = cnt[k];
            out[w].total_ms = ms[k];
        }
        w++;
    }
    *count = w;
    return FSA_OK;
}

// ---- DFSPH -------------------------------------------------------------------------------------
#define DFSPH_ENTRY(name, launcher, needs_sort)                                                      \
    int name(fsa_solver *hs) {                                                                       \
        Solver *s = reinterpret_cast<Solver *>(hs);                                                  \
        NEED(s);                                                                                     \
        if (s->slab && !s->slab_carry_vstar)                                                         \
            return fail(FSA_ERR_INVALID, #name ": a slab solver runs the DFSPH step only after fsa_slab_enable_dfsph"); \
        if (needs_sort && !s->sorted_valid) return fail(FSA_ERR_INVALID, #name " needs a sorted state"); \
        return launcher(s);                                                                          \
    }
DFSPH_ENTRY(fsa_compute_density_alpha, launch_density_alpha, true)
DFSPH_ENTRY(fsa_compute_non_pressure_force, launch_non_pressure_force, true)
DFSPH_ENTRY(fsa_solve_density_stiff, launch_solve_density_stiff, true)
DFSPH_ENTRY(fsa_solve_divergence_stiff, launch_solve_divergence_stiff, true)
DFSPH_ENTRY(fsa_apply_pstiff, launch_apply_pstiff, true)
DFSPH_ENTRY(fsa_update_velocities, launch_update_velocities, false)
#undef DFSPH_ENTRY

int fsa_update_position(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (s->slab && !s->slab_carry_vstar)
        return fail(FSA_ERR_INVALID, "fsa_update_position: a slab solver runs the DFSPH step only after fsa_slab_enable_dfsph");
    TRY(launch_update_position(s));
    // positions moved: the cell ranges no longer describe them
    s->sorted_valid = false;
    s->hash_valid = s->hash_ahead = false;
    s->perm_valid = false;
    return FSA_OK;
}

// Both loops queue all their rounds at once; the maximum of error[] and the test against the threshold run on the
// device after each density-change pass, and the passes of the rounds after the stop return at once.  The host waits
// only if the caller asks for the iteration count or the last error (one read-back).
int fsa_correct_density_error(fsa_solver *hs, int maxiter, double ethres, int *iterations, double *last_error) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    TRY(launch_dfsph_loop_begin(s));
    int rc = FSA_OK;
    s->dfsph_chain = 1;  // the passes hand their packed records on to each other
    for (int iter = 0; iter < maxiter && !rc; iter++) {
        rc = fsa_solve_density_stiff(hs);
        if (!rc) rc = launch_dfsph_round_end(s, -9999.0, ethres);  // solver/sph_solver.cu:176-180
        if (!rc) rc = fsa_apply_pstiff(hs);
    }
    s->dfsph_chain = 0;
    dfsph_loop_end(s);
    if (rc) return rc;
    TRY(fsa_update_position(hs));
    if (iterations || last_error) TRY(launch_dfsph_loop_result(s, -9999.0, iterations, last_error));
    return FSA_OK;
}

int fsa_correct_divergence_error(fsa_solver *hs, int maxiter, double ethres, int *iterations, double *last_error) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    TRY(launch_dfsph_loop_begin(s));
    s->dfsph_chain = 1;  // the passes hand their packed records on to each other
    int rc = fsa_solve_divergence_stiff(hs);  // the extra first launch of solver/sph_solver.cu:226
    for (int iter = 0; iter < maxiter && !rc; iter++) {
        rc = fsa_solve_divergence_stiff(hs);
        if (!rc) rc = launch_dfsph_round_end(s, 0.0, ethres);
        if (!rc) rc = fsa_apply_pstiff(hs);
    }
    s->dfsph_chain = 0;
    dfsph_loop_end(s);
    if (rc) return rc;
    TRY(fsa_update_velocities(hs));
    if (iterations || last_error) TRY(launch_dfsph_loop_result(s, 0.0, iterations, last_error));
    return FSA_OK;
}

int fsa_setup_dfsph(fsa_solver *hs) {
    TRY(fsa_sort_particles(hs));
    return fsa_compute_density_alpha(hs);
}

int fsa_step_dfsph(fsa_solver *hs, int nsteps) {
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    for (int k = 0; k < nsteps; k++) {
        TRY(fsa_compute_non_pressure_force(hs));
        TRY(fsa_correct_density_error(hs, 5, 1.0, nullptr, nullptr));
        TRY(fsa_sort_particles(hs));
        TRY(fsa_compute_density_alpha(hs));
        TRY(fsa_correct_divergence_error(hs, 5, 1.0, nullptr, nullptr));
    }
    return FSA_OK;
}

// ---- IISPH -------------------------------------------------------------------------------------
// Marks the solver as running the IISPH step: from then on every sort also carries pressure[] and keeps
// the per-position copy of group[] current.  The first call brings that copy up to date.
static int enter_iisph(Solver *s, const char *who) {
    if (s->slab && !s->slab_carry)
        return fail(FSA_ERR_INVALID, "%s: a slab solver runs the IISPH step only after fsa_slab_enable_iisph", who);
    if (s->iisph) return FSA_OK;
    s->iisph = true;
    if (s->n == 0) return FSA_OK;
    return launch_gather_by_uid(s, s->b.group, s->b.groupSorted, s->n, 1);
}

#define IISPH_ENTRY(name, launcher)                                                                  \
    int name(fsa_solver *hs) {                                                                       \
        Solver *s = reinterpret_cast<Solver *>(hs);                                                  \
        NEED(s);                                                                                     \
        TRY(enter_iisph(s, #name));                                                                  \
        if (!s->sorted_valid) return fail(FSA_ERR_INVALID, #name " needs a sorted state");           \
        return launcher(s);                                                                          \
    }
IISPH_ENTRY(fsa_non_pressure_force_multiphase, launch_npf_multiphase)
IISPH_ENTRY(fsa_iisph_factor, launch_iisph_factor)
IISPH_ENTRY(fsa_iisph_predict_density, launch_iisph_predict_density)
IISPH_ENTRY(fsa_iisph_calc_dijpjl, launch_iisph_dijpjl)
IISPH_ENTRY(fsa_iisph_calc_new_pressure, launch_iisph_new_pressure)
IISPH_ENTRY(fsa_iisph_calc_pressure_force, launch_iisph_pressure_force)
#undef IISPH_ENTRY

int fsa_set_volume_fractions(fsa_solver *hs, int typenum, const double *dens_arr, const void *vfrac) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (s->slab) return fail(FSA_ERR_INVALID, "fsa_set_volume_fractions is not available in a slab solver");
    if (typenum < 1 || typenum > 10) return fail(FSA_ERR_INVALID, "typenum %d outside 1..10", typenum);
    if (!dens_arr || (s->n > 0 && !vfrac)) return fail(FSA_ERR_INVALID, "NULL argument");
    FSA_HIP(counted_stream_sync(s->stream));
    const size_t N = (size_t)s->n, R = s->real_size(), K = (size_t)typenum;
    if (typenum != s->typenum) {  // the one allocation after fsa_create, sized for the solver's capacity
        if (s->b.vFrac) (void)hipFree(s->b.vFrac);
        if (s->b.densArr) (void)hipFree(s->b.densArr);
        s->b.vFrac = s->b.densArr = nullptr;
        s->typenum = 0;
        TRY(dalloc(s->b.vFrac, (size_t)s->capacity * K * R, true));
        TRY(dalloc(s->b.densArr, K * R, true));
        s->typenum = typenum;
    }
    // rows arrive in the current device order; they are kept in uniqueId order
    std::vector<int> uid(N);
    if (N) FSA_HIP(hipMemcpy(uid.data(), s->b.uniqueId[s->cur], N * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<unsigned char> table(N * K * R), dens(K * R);
    for (size_t i = 0; i < N; i++)
        memcpy(&table[(size_t)uid[i] * K * R], (const unsigned char *)vfrac + i * K * R, K * R);
    for (size_t k = 0; k < K; k++) {
        if (R == sizeof(float)) {
            const float v = (float)dens_arr[k];
            memcpy(&dens[k * R], &v, R);
        } else {
            memcpy(&dens[k * R], &dens_arr[k], R);
        }
    }
    if (N) FSA_HIP(hipMemcpy(s->b.vFrac, table.data(), table.size(), hipMemcpyHostToDevice));
    FSA_HIP(hipMemcpy(s->b.densArr, dens.data(), dens.size(), hipMemcpyHostToDevice));
    return FSA_OK;
}

int fsa_effective_mass(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (s->slab) return fail(FSA_ERR_INVALID, "fsa_effective_mass is not available in a slab solver");
    if (s->typenum == 0) return fail(FSA_ERR_INVALID, "fsa_effective_mass needs fsa_set_volume_fractions first");
    s->pd_valid = false;
    s->fluid_mass_uniform = false;  // the masses become per-particle values
    return launch_effective_mass(s);
}

int fsa_iisph_solve_pressure(fsa_solver *hs, int maxiter, int *rounds, double *last_mean) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (maxiter < 0) return fail(FSA_ERR_INVALID, "maxiter < 0");
    // All rounds are queued at once; the stop test of solver/sph_solver.cu:1440 runs on the device after each
    // round's mean, and once it has fired the remaining rounds' kernels return at their first instruction.  One
    // read-back at the end tells how many rounds ran (each swaps the two pressure arrays).
    TRY(launch_iisph_rounds_begin(s));
    int rc = FSA_OK;
    for (int iter = 0; iter < maxiter && !rc; iter++) {
        s->iisph_chain = iter == 0 ? 1 : 2;  // the passes hand their packed records on to each other
        rc = fsa_iisph_calc_dijpjl(hs);
        if (!rc) rc = fsa_iisph_calc_new_pressure(hs);
        if (!rc) rc = launch_iisph_round_end(s);
    }
    s->iisph_chain = 0;
    int done = 0;
    double mean = 0.0;
    const int rc2 = launch_iisph_rounds_finish(s, maxiter, &done, &mean);
    if (rc) return rc;
    if (rc2) return rc2;
    if (rounds) *rounds = done;
    if (last_mean) *last_mean = mean;
    return FSA_OK;
}

int fsa_iisph_update(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    TRY(fsa_iisph_calc_pressure_force(hs));
    TRY(launch_iisph_integrate(s));
    // positions moved: the cell ranges no longer describe them
    s->sorted_valid = false;
    s->hash_valid = s->hash_ahead = false;
    s->perm_valid = false;
    s->pd_valid = false;
    return FSA_OK;
}

int fsa_setup_iisph(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    TRY(enter_iisph(s, "fsa_setup_iisph"));
    TRY(fsa_sort_particles(hs));
    TRY(fsa_iisph_factor(hs));
    return fsa_rigid_particle_volume(hs);
}

int fsa_step_iisph(fsa_solver *hs, int nsteps) {
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    for (int k = 0; k < nsteps; k++) {
        TRY(fsa_non_pressure_force_multiphase(hs));
        TRY(fsa_iisph_predict_density(hs));
        TRY(fsa_iisph_solve_pressure(hs, 10, nullptr, nullptr));  // "temp 10", solver/sph_solver.cu:1416
        TRY(fsa_iisph_update(hs));
        TRY(fsa_sort_particles(hs));
        TRY(fsa_iisph_factor(hs));
    }
    return FSA_OK;
}

// ---- z-slab decomposition ----------------------------------------------------------------------
int fsa_slab_record_bytes(fsa_solver *hs, int *bytes) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s || !bytes) return fail(FSA_ERR_INVALID, "NULL argument");
    *bytes = slab_record_bytes(s);
    return FSA_OK;
}

int fsa_slab_pack(fsa_solver *hs, const int own[2], const int keep[2], const int left[2], const int right[2],
                  void *d_send_left, int cap_left, void *d_send_right, int cap_right, int counts[4]) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_pack needs a solver created with slab_nz > 0");
    if (!own || !keep || !left || !right || !counts) return fail(FSA_ERR_INVALID, "NULL argument");
    if (cap_left < 0 || cap_right < 0 || (cap_left > 0 && !d_send_left) || (cap_right > 0 && !d_send_right))
        return fail(FSA_ERR_INVALID, "send buffer missing");
    // a particle outside [slab_z0, slab_z0 + slab_nz) would hash to GRID_UNDEF at the next sort
    if (keep[0] < s->pf.z0 || keep[1] > s->pf.z0 + s->pf.nz)
        return fail(FSA_ERR_INVALID, "keep range [%d, %d) exceeds the layers [%d, %d) this solver holds", keep[0],
                    keep[1], s->pf.z0, s->pf.z0 + s->pf.nz);
    TRY(launch_slab_pack(s, own, keep, left, right, d_send_left, cap_left, d_send_right, cap_right, counts));
    if (counts[0] > s->capacity)
        return fail(FSA_ERR_CAPACITY, "%d particles stay, capacity %d", counts[0], s->capacity);
    if (counts[1] > cap_left || counts[2] > cap_right)
        return fail(FSA_ERR_CAPACITY, "send buffers too small: %d / %d records for %d / %d slots", counts[1],
                    counts[2], cap_left, cap_right);
    s->n = counts[0];
    s->cur = 1 - s->cur;
    invalidate_order(s);
    s->count_valid = false;
    s->owned_from_sort = false;
    return FSA_OK;
}

int fsa_slab_export_owned(fsa_solver *hs, const int own[2], void *d_records, int cap, int *count) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_export_owned needs a solver created with slab_nz > 0");
    if (!own || !count || cap < 0 || (cap > 0 && !d_records)) return fail(FSA_ERR_INVALID, "bad argument");
    TRY(launch_slab_export(s, own, d_records, cap, count));
    if (*count > cap) return fail(FSA_ERR_CAPACITY, "%d owned particles for %d record slots", *count, cap);
    return FSA_OK;
}

int fsa_slab_unpack(fsa_solver *hs, const void *d_records, int count) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_unpack needs a solver created with slab_nz > 0");
    if (count < 0 || (count > 0 && !d_records)) return fail(FSA_ERR_INVALID, "bad record buffer");
    if ((long long)s->n + count > s->capacity)
        return fail(FSA_ERR_CAPACITY, "%d + %d particles exceed capacity %d", s->n, count, s->capacity);
    TRY(launch_slab_unpack(s, d_records, count));
    s->n += count;
    invalidate_order(s);
    s->count_valid = false;
    s->owned_from_sort = false;
    return FSA_OK;
}

// ---- the IISPH step on slab solvers ------------------------------------------------------------
int fsa_slab_enable_iisph(fsa_solver *hs, int n_global, const int *group, const void *normal3) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_enable_iisph needs a solver created with slab_nz > 0");
    if (n_global <= 0 || !group || !normal3) return fail(FSA_ERR_INVALID, "bad argument");
    if (s->slab_carry_vstar) return fail(FSA_ERR_INVALID, "the solver is set up for the DFSPH step on slabs");
    FSA_HIP(counted_stream_sync(s->stream));
    const size_t N = (size_t)s->capacity, R = s->real_size(), G = (size_t)n_global;
    Buffers &b = s->b;
    if (!s->slab_carry) {
        TRY(dalloc(b.pressure_alt, N * R, true));
        TRY(dalloc(b.dii, N * 3 * R, true));
        TRY(dalloc(b.dijpjl, N * 3 * R, true));
        TRY(dalloc(b.pressureForce, N * 3 * R, true));
        TRY(dalloc(b.aii, N * R, true));
        TRY(dalloc(b.density_star, N * R, true));
        TRY(dalloc(b.groupSorted, N * sizeof(int), true));
    }
    if (s->table_n != n_global) {
        if (b.group) (void)hipFree(b.group);
        if (b.normal) (void)hipFree(b.normal);
        b.group = nullptr;
        b.normal = nullptr;
        s->table_n = 0;
        TRY(dalloc(b.group, G * sizeof(int), false));
        TRY(dalloc(b.normal, G * 3 * R, false));
        s->table_n = n_global;
    }
    FSA_HIP(hipMemcpy(b.group, group, G * sizeof(int), hipMemcpyHostToDevice));
    FSA_HIP(hipMemcpy(b.normal, normal3, G * 3 * R, hipMemcpyHostToDevice));
    s->slab_carry = true;
    s->iisph = false;  // the per-position copy of group[] is rebuilt by the next IISPH call
    return FSA_OK;
}

// ---- the DFSPH step on slab solvers ------------------------------------------------------------
int fsa_slab_enable_dfsph(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_enable_dfsph needs a solver created with slab_nz > 0");
    if (s->slab_carry) return fail(FSA_ERR_INVALID, "the solver is set up for the IISPH step on slabs");
    if (s->pf.count) return fail(FSA_ERR_INVALID, "the solver is a member of a slab group");
    s->slab_carry_vstar = true;
    return FSA_OK;
}

int fsa_slab_pack_carry(fsa_solver *hs, const int own[2], const int keep[2], const int left[2], const int right[2],
                        void *d_send_left, void *d_extra_left, int cap_left, void *d_send_right, void *d_extra_right,
                        int cap_right, int counts[4]) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab || !(s->slab_carry || s->slab_carry_vstar))
        return fail(FSA_ERR_INVALID, "fsa_slab_pack_carry needs fsa_slab_enable_iisph or fsa_slab_enable_dfsph first");
    if (!own || !keep || !left || !right || !counts) return fail(FSA_ERR_INVALID, "NULL argument");
    if (cap_left < 0 || cap_right < 0 || (cap_left > 0 && (!d_send_left || !d_extra_left)) ||
        (cap_right > 0 && (!d_send_right || !d_extra_right)))
        return fail(FSA_ERR_INVALID, "send buffer missing");
    if (keep[0] < s->pf.z0 || keep[1] > s->pf.z0 + s->pf.nz)
        return fail(FSA_ERR_INVALID, "keep range [%d, %d) exceeds the layers [%d, %d) this solver holds", keep[0],
                    keep[1], s->pf.z0, s->pf.z0 + s->pf.nz);
    TRY(launch_slab_pack(s, own, keep, left, right, d_send_left, cap_left, d_send_right, cap_right, counts,
                         d_extra_left, d_extra_right));
    if (counts[0] > s->capacity)
        return fail(FSA_ERR_CAPACITY, "%d particles stay, capacity %d", counts[0], s->capacity);
    if (counts[1] > cap_left || counts[2] > cap_right)
        return fail(FSA_ERR_CAPACITY, "send buffers too small: %d / %d records for %d / %d slots", counts[1],
                    counts[2], cap_left, cap_right);
    s->n = counts[0];
    s->cur = 1 - s->cur;
    invalidate_order(s);
    s->count_valid = false;
    s->owned_from_sort = false;
    return FSA_OK;
}

int fsa_slab_unpack_carry(fsa_solver *hs, const void *d_records, const void *d_extras, int count) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab || !(s->slab_carry || s->slab_carry_vstar))
        return fail(FSA_ERR_INVALID, "fsa_slab_unpack_carry needs fsa_slab_enable_iisph or fsa_slab_enable_dfsph first");
    if (count < 0 || (count > 0 && (!d_records || !d_extras))) return fail(FSA_ERR_INVALID, "bad record buffer");
    if ((long long)s->n + count > s->capacity)
        return fail(FSA_ERR_CAPACITY, "%d + %d particles exceed capacity %d", s->n, count, s->capacity);
    TRY(launch_slab_unpack(s, d_records, count, d_extras));
    s->n += count;
    invalidate_order(s);
    s->count_valid = false;
    s->owned_from_sort = false;
    return FSA_OK;
}

int fsa_slab_layer_indices(fsa_solver *hs, int layer_lo, int layer_hi, int *d_idx, int cap, int *count) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_layer_indices needs a solver created with slab_nz > 0");
    if (!count || cap < 0 || (cap > 0 && !d_idx)) return fail(FSA_ERR_INVALID, "bad argument");
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "fsa_slab_layer_indices needs a sorted state");
    if (layer_lo < s->pf.z0 || layer_hi > s->pf.z0 + s->pf.nz || layer_lo > layer_hi)
        return fail(FSA_ERR_INVALID, "layers [%d, %d) outside the layers held [%d, %d)", layer_lo, layer_hi, s->pf.z0,
                    s->pf.z0 + s->pf.nz);
    return launch_slab_layer_indices(s, layer_lo, layer_hi, d_idx, cap, count);
}

// the per-particle fields that the neighbour passes of the IISPH step read of a neighbour: device array and reals per particle
static int slab_field(Solver *s, const char *name, void **ptr, int *reals) {
    if (!name) return fail(FSA_ERR_INVALID, "NULL argument");
    const std::string k(name);
    *ptr = nullptr;
    *reals = 1;
    if (k == "density") *ptr = s->b.density;
    else if (k == "pressure") *ptr = s->b.pressure;
    else if (k == "restDensity") *ptr = s->b.restDensity[s->cur];
    else if (k == "error") *ptr = s->b.error;
    else if (k == "pstiff") *ptr = s->b.pstiff;
    else if (k == "DF_factor") *ptr = s->b.DF_factor;
    else if (k == "v_star") { *ptr = s->b.v_star; *reals = 3; }
    else if (k == "dii") { *ptr = s->b.dii; *reals = 3; }
    else if (k == "dijpjl") { *ptr = s->b.dijpjl; *reals = 3; }
    if (!*ptr) return fail(FSA_ERR_UNKNOWN_NAME, "field '%s' cannot be moved (or is not allocated in this solver)", name);
    return FSA_OK;
}

static int slab_move_field(fsa_solver *hs, bool scatter, const char *name, const int *d_idx, int count, void *d_buf) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (count < 0 || (count > 0 && (!d_idx || !d_buf))) return fail(FSA_ERR_INVALID, "bad argument");
    void *field = nullptr;
    int reals = 1;
    TRY(slab_field(s, name, &field, &reals));
    if (scatter) s->pd_valid = false;
    return launch_field_move(s, scatter, field, reals * (int)(s->real_size() / 4), d_idx, count, d_buf);
}

int fsa_slab_gather_field(fsa_solver *hs, const char *name, const int *d_idx, int count, void *d_out) {
    return slab_move_field(hs, false, name, d_idx, count, d_out);
}

int fsa_slab_scatter_field(fsa_solver *hs, const char *name, const int *d_idx, int count, const void *d_in) {
    return slab_move_field(hs, true, name, d_idx, count, const_cast<void *>(d_in));
}

int fsa_slab_owned_sum(fsa_solver *hs, const char *name, const int own[2], double *sum) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_owned_sum needs a solver created with slab_nz > 0");
    if (!own || !sum) return fail(FSA_ERR_INVALID, "NULL argument");
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "fsa_slab_owned_sum needs a sorted state");
    void *field = nullptr;
    int reals = 1;
    TRY(slab_field(s, name, &field, &reals));
    if (reals != 1) return fail(FSA_ERR_INVALID, "'%s' is not a scalar field", name);
    return launch_owned_sum(s, own[0], own[1], field, sum);
}

int fsa_slab_owned_max(fsa_solver *hs, const char *name, const int own[2], double start, double *max) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_owned_max needs a solver created with slab_nz > 0");
    if (!own || !max) return fail(FSA_ERR_INVALID, "NULL argument");
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "fsa_slab_owned_max needs a sorted state");
    void *field = nullptr;
    int reals = 1;
    TRY(slab_field(s, name, &field, &reals));
    if (reals != 1) return fail(FSA_ERR_INVALID, "'%s' is not a scalar field", name);
    return launch_owned_max(s, own[0], own[1], field, start, max);
}

int fsa_slab_set_owned(fsa_solver *hs, const int own[2]) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s || !own) return fail(FSA_ERR_INVALID, "NULL argument");
    if (!s->slab) return fail(FSA_ERR_INVALID, "fsa_slab_set_owned needs a solver created with slab_nz > 0");
    if (own[0] < s->pf.z0 || own[1] > s->pf.z0 + s->pf.nz || own[0] > own[1])
        return fail(FSA_ERR_INVALID, "owned layers [%d, %d) outside the layers held [%d, %d)", own[0], own[1], s->pf.z0,
                    s->pf.z0 + s->pf.nz);
    s->pf.own_lo = s->pd.own_lo = own[0];
    s->pf.own_hi = s->pd.own_hi = own[1];
    return FSA_OK;
}

int fsa_set_scene_has_rigid(fsa_solver *hs, int has_rigid) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s) return fail(FSA_ERR_INVALID, "solver is NULL");
    s->has_rigid = has_rigid != 0;
    return FSA_OK;
}

// Not part of the public header: how often the library has made the host wait for the device so far.
long long fsa_debug_host_waits(void) { return g_host_waits.load(); }
// ... and how often it has waited for an event recorded some steps earlier (the device was not idle meanwhile)
long long fsa_debug_event_waits(void) { return g_event_waits.load(); }

// 1 when applyPStiff would take the one-mass flavour on this solver's present state (tests)
int fsa_debug_one_mass_flavour(fsa_solver *hs) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    return s && s->fluid_mass_uniform ? 1 : 0;
}

// Not part of the public header: kernel-ablation timing used by tools/ablate.py.
int fsa_debug_time_pass(fsa_solver *hs, int which, int ablate, int reps) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    if (!s->sorted_valid) return fail(FSA_ERR_INVALID, "needs a sorted state");
    return debug_time_pass(s, which, ablate, reps);
}

// Not part of the public header: the density pass and the force replay in one launch (tools/overlap_probe.py).
int fsa_debug_fused_probe(fsa_solver *hs, int mode) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    NEED(s);
    return launch_fused_probe(s, mode);
}

int fsa_set_variant(fsa_solver *hs, int variant) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s) return fail(FSA_ERR_INVALID, "solver is NULL");
    if (variant < 0 || variant > 3) return fail(FSA_ERR_INVALID, "unknown variant %d", variant);
    s->variant = variant;
    return FSA_OK;
}

int fsa_set_neighbour_lists(fsa_solver *hs, int enabled) {
    Solver *s = reinterpret_cast<Solver *>(hs);
    if (!s) return fail(FSA_ERR_INVALID, "solver is NULL");
    if (enabled < 0 || enabled > 2) return fail(FSA_ERR_INVALID, "fsa_set_neighbour_lists: %d is neither 0, 1 nor 2", enabled);
    s->use_lists = enabled != 0;
    s->use_replay = enabled == 1;
    s->lists_valid = false;
    return FSA_OK;
}

}  // extern "C"
