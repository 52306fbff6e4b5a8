// z-slab groups: the step of a scene cut into z-slabs (fsa_slab.hip), run for all slabs of this process -- or for
// this process's slab, with the neighbours in other processes -- without a host round trip per step.
//
// The reference steps one device from one host thread (MultiphaseSPHSolver::Step,
// solver/MultiphaseSPHSolver.cpp:199-217 under /root/reference/particle_solver).  What a slab step adds is the
// exchange between integration and sort, and the host must not wait for it: the particle count of every slab
// lives in device memory (DevParams::count), the pack and unpack kernels keep it and the message counts up to
// date there, and launch grids are sized for the solver's capacity.
//
//   pack       two passes (fsa_slab.hip).  The first integrates the particles of the layers next to the cuts and
//              writes those that are now in a neighbour's layers as records; a one-thread kernel turns the atomic
//              counters into the message headers.  The second integrates the rest, decides who stays and who is
//              dropped (the halo copies of the step before) and computes the hash and cell counts of those that
//              stay; nothing is moved, the sort gathers what is left.  The pack of step k + 1 thus runs at the end
//              of step k, and so does its transport, on a stream of its own: the force pass handles the blocks next
//              to the cuts first, then come the first pack pass and the transport, and the force pass of all other
//              blocks and the second pack pass run while the records travel;
//   transport  LOCAL: all slabs belong to solvers of this process (on one GPU or several): the send buffer is
//              copied into the neighbour's receive buffer on the sender's stream, events order it against the
//              neighbour's unpack of this step and of the step before;
//              RCCL: one slab per process: ncclSend / ncclRecv to both z-neighbours in one group call on the
//              solver's stream (xGMI between the GPUs of a node);
//   unpack     (start of the next step, once the transport is through) records of both receive buffers are appended
//              and hashed, the number of slots is raised;
//   then scan -> scatter -> rank + reorder (the particle count becomes what the sort kept) -> density -> force.
// FSA_SLAB_OVERLAP=0 (read when a group is made) keeps everything on the solver's stream, for comparison.
//
// Message sizes.  A send or copy needs its size on the host when it is queued, and the host does not know this
// step's counts.  Both ends therefore size the message of step k from the count of step k - kLag, which each of
// them has by then without waiting (the sender's own count, and the header the receiver got, both mirrored into
// pinned host memory by the one-thread kernels and fenced by an event), plus head room; the actual count rides
// in the buffer's header slot.  The first kLag steps send whole buffers.  A count that outgrows its message is cut
// and flagged on the device; fsa_slab_group_sync reports it as an error (the run is then invalid: re-create the
// group with more head room).
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "fsa_solver.h"

namespace fsa {

constexpr int kLag = 3;    // steps between a count and the message size derived from it
constexpr int kRing = 8;   // mirror slots (> kLag)
constexpr int kMirrorInts = 8;

struct Member {
    Solver *s = nullptr;
    int rank = 0;
    int own[2] = {0, 0}, held[2] = {0, 0}, left[2] = {0, 0}, right[2] = {0, 0};
    bool has_left = false, has_right = false;
    char *send[2] = {nullptr, nullptr};  // [0] to the left neighbour, [1] to the right; header slot first
    char *recv[2] = {nullptr, nullptr};  // [0] from the left neighbour, [1] from the right
    int *mirror = nullptr;               // pinned host memory: kRing x kMirrorInts
    hipEvent_t ev_pack[kRing] = {}, ev_unpack[kRing] = {}, ev_sent[2] = {};
    hipEvent_t ev_arrived = nullptr;     // RCCL: this step's messages are in the receive buffers
    hipStream_t comm = nullptr;          // the transport's stream (the solver's own without overlap)
    int size_out[2] = {0, 0};            // records the messages take that were packed last
    int bound_in[2] = {0, 0};            // records the messages can hold that are on their way here
    bool ahead = false;                  // the next step's messages are packed and on their way, the hash is ahead
};

struct Group {
    int kind = 0;  // 0 LOCAL, 1 RCCL, 2 HOSTED (as RCCL, the bytes through the caller's functions and host memory)
    fsa_host_transport via{};
    char *stage = nullptr;  // HOSTED: pinned host memory for the messages of a step
    size_t stage_bytes = 0;
    bool overlap = true;
    int world = 0, halo = 0, record_cap = 0, rb = 0, slack = 4096;
    std::vector<int> cuts;
    std::vector<Member> m;
    ncclComm_t comm = nullptr;
    long long step = 0;  // steps queued since the group was made
    long long sent = 0;  // records sent by this process's members, as far as the host has seen the counts
};

static bool overlap_wanted() {
    const char *e = getenv("FSA_SLAB_OVERLAP");
    return !(e && e[0] == '0');
}

// Head room of a message over the count it is sized from, in records beyond the 12.5 %.  FSA_SLAB_MSG_SLACK (read when
// a group is made; may be negative, which makes messages too small on purpose: tests of the cut counts).
static int slack_wanted() {
    const char *e = getenv("FSA_SLAB_MSG_SLACK");
    return e ? atoi(e) : 4096;
}

static int next_size(int count, int cap, int slack) {
    const long long v = (long long)count + count / 8 + slack;
    return (int)std::max<long long>(0, std::min<long long>(v, cap));
}

#define FSA_NCCL(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess)                                                                               \
            return ::fsa::fail(FSA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, \
                               __LINE__);                                                                    \
    } while (0)

static void held_of(const Group &g, int r, int out[2]) {
    if (r < 0 || r >= g.world) {
        out[0] = out[1] = 0;
        return;
    }
    const int gz = g.cuts.back();
    out[0] = std::max(g.cuts[r] - g.halo, 0);
    out[1] = std::min(g.cuts[r + 1] + g.halo, gz);
}

static void free_member(Member &mb) {
    if (mb.s) {
        (void)hipSetDevice(mb.s->device);
        (void)counted_stream_sync(mb.s->stream);
        // the solver goes back to a host-side count
        if (mb.s->pf.count) {
            int n = mb.s->n;
            if (hipMemcpy(&n, mb.s->b.slabVars + kVarN, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) mb.s->n = n;
            mb.s->pf.count = mb.s->pd.count = nullptr;
            mb.s->slab_slots = nullptr;
            mb.s->force_zone = 0;
            for (int e = 0; e < 4; e++) mb.s->pf.edge[e] = mb.s->pd.edge[e] = 0;
            mb.s->count_stale = false;
            // hash[] may hold the marks of a pack whose message never went out
            mb.s->hash_valid = mb.s->hash_ahead = false;
        }
    }
    for (int d = 0; d < 2; d++) {
        if (mb.send[d]) (void)hipFree(mb.send[d]);
        if (mb.recv[d]) (void)hipFree(mb.recv[d]);
        if (mb.ev_sent[d]) (void)hipEventDestroy(mb.ev_sent[d]);
    }
    for (int k = 0; k < kRing; k++) {
        if (mb.ev_pack[k]) (void)hipEventDestroy(mb.ev_pack[k]);
        if (mb.ev_unpack[k]) (void)hipEventDestroy(mb.ev_unpack[k]);
    }
    if (mb.ev_arrived) (void)hipEventDestroy(mb.ev_arrived);
    if (mb.comm && mb.s && mb.comm != mb.s->stream) {
        (void)hipStreamSynchronize(mb.comm);
        (void)hipStreamDestroy(mb.comm);
    }
    if (mb.mirror) (void)hipHostFree(mb.mirror);
}

static int setup_member(Group &g, Member &mb, Solver *s, int rank) {
    if (!s) return fail(FSA_ERR_INVALID, "solver is NULL");
    if (!s->slab) return fail(FSA_ERR_INVALID, "a slab group needs solvers created with slab_nz > 0");
    if (s->precision != FSA_PRECISION_F32 || s->variant != 3 || s->has_rigid)
        return fail(FSA_ERR_INVALID, "a slab group needs fp32 solvers on kernel variant 3 and a scene without TYPE_RIGID "
                                     "particles (the kernels that take the particle count from device memory)");
    mb.s = s;
    mb.rank = rank;
    mb.own[0] = g.cuts[rank];
    mb.own[1] = g.cuts[rank + 1];
    held_of(g, rank, mb.held);
    held_of(g, rank - 1, mb.left);
    held_of(g, rank + 1, mb.right);
    mb.has_left = rank > 0;
    mb.has_right = rank + 1 < g.world;
    if (mb.held[0] != s->pf.z0 || mb.held[1] != s->pf.z0 + s->pf.nz)
        return fail(FSA_ERR_INVALID, "rank %d: the solver holds layers [%d, %d), the cuts ask for [%d, %d)", rank, s->pf.z0,
                    s->pf.z0 + s->pf.nz, mb.held[0], mb.held[1]);
    FSA_HIP(hipSetDevice(s->device));
    const size_t bytes = (size_t)(g.record_cap + 1) * g.rb;
    for (int d = 0; d < 2; d++) {
        if (d == 0 ? !mb.has_left : !mb.has_right) continue;
        FSA_HIP(hipMalloc((void **)&mb.send[d], bytes));
        FSA_HIP(hipMalloc((void **)&mb.recv[d], bytes));
        FSA_HIP(hipMemset(mb.send[d], 0, g.rb));
        FSA_HIP(hipMemset(mb.recv[d], 0, g.rb));
        FSA_HIP(hipEventCreateWithFlags(&mb.ev_sent[d], hipEventDisableTiming));
    }
    for (int k = 0; k < kRing; k++) {
        FSA_HIP(hipEventCreateWithFlags(&mb.ev_pack[k], hipEventDisableTiming));
        FSA_HIP(hipEventCreateWithFlags(&mb.ev_unpack[k], hipEventDisableTiming));
    }
    FSA_HIP(hipEventCreateWithFlags(&mb.ev_arrived, hipEventDisableTiming));
    mb.comm = s->stream;
    if (g.overlap) FSA_HIP(hipStreamCreateWithFlags(&mb.comm, hipStreamNonBlocking));
    FSA_HIP(hipHostMalloc((void **)&mb.mirror, sizeof(int) * kRing * kMirrorInts, hipHostMallocDefault));
    std::memset(mb.mirror, 0, sizeof(int) * kRing * kMirrorInts);
    // from here on the count lives on the device
    FSA_HIP(counted_stream_sync(s->stream));
    int vars[16] = {0};
    vars[kVarN] = s->n;
    FSA_HIP(hipMemcpy(s->b.slabVars, vars, sizeof(vars), hipMemcpyHostToDevice));
    FSA_HIP(hipMemset(s->b.slabCounters, 0, 8 * sizeof(int)));
    FSA_HIP(counted_device_sync());
    s->pf.count = s->pd.count = s->b.slabVars + kVarN;
    s->slab_slots = s->b.slabVars + kVarSlots;
    s->hash_valid = s->hash_ahead = false;
    s->pf.own_lo = s->pd.own_lo = mb.own[0];
    s->pf.own_hi = s->pd.own_hi = mb.own[1];
    // the layers a particle can leave for a neighbour from in one step (from anywhere else it would have to cross
    // a whole cell layer, two particle spacings, to get there; the pack flags that)
    const int reach = g.halo + 1;
    int edge[4] = {0, 0, 0, 0};
    if (mb.has_left) { edge[0] = mb.own[0]; edge[1] = std::min(mb.own[0] + reach, mb.own[1]); }
    if (mb.has_right) { edge[2] = std::max(mb.own[1] - reach, mb.own[0]); edge[3] = mb.own[1]; }
    for (int e = 0; e < 4; e++) s->pf.edge[e] = s->pd.edge[e] = edge[e];
    return FSA_OK;
}

// Sizes of a member's two outgoing messages of step `msg`, from its own counts of kLag steps before.
static int outgoing_sizes(Group &g, Member &mb, long long msg) {
    for (int d = 0; d < 2; d++) mb.size_out[d] = g.record_cap;
    if (msg >= kLag) {
        const int slot = (int)((msg - kLag) % kRing);
        FSA_HIP(counted_event_sync(mb.ev_pack[slot]));  // long done: no wait in steady state
        const int *mi = mb.mirror + slot * kMirrorInts;
        for (int d = 0; d < 2; d++) mb.size_out[d] = next_size(mi[d], g.record_cap, g.slack);
        g.sent += (long long)mi[0] + mi[1];
    }
    return FSA_OK;
}

// The sizes of the two incoming messages of step `msg`, from the headers received kLag steps before: what the
// RCCL transport posts its receives with.  (The local transport copies what the sender sized and uses this only to
// check that both ends arrive at the same number, which between processes nothing else would notice before the
// send and the receive disagree.)
static int incoming_sizes(Group &g, Member &mb, long long msg, int size_in[2]) {
    size_in[0] = size_in[1] = g.record_cap;
    if (msg >= kLag) {
        const int slot = (int)((msg - kLag) % kRing);
        FSA_HIP(counted_event_sync(mb.ev_unpack[slot]));
        const int *mi = mb.mirror + slot * kMirrorInts;
        for (int d = 0; d < 2; d++) size_in[d] = next_size(mi[4 + d], g.record_cap, g.slack);
    }
    return FSA_OK;
}

// First pack pass of message `msg` for a member (with the integration of the step before when `with_advect`).
static int pack_edges(Group &g, Member &mb, long long msg, bool with_advect) {
    Solver *s = mb.s;
    int rc = outgoing_sizes(g, mb, msg);
    if (rc) return rc;
    const int slot = (int)(msg % kRing);
    rc = launch_slab_edge_send(s, with_advect, mb.own, mb.left, mb.right, mb.send[0], mb.size_out[0], mb.send[1],
                               mb.size_out[1]);
    if (rc) return rc;
    // (the headers on the solver's stream too: a kernel on the transport's stream was tried and, with several
    // members per GPU, costs more in stream switching than the launch it takes off this one)
    rc = launch_slab_pack_tail(s, s->stream, mb.send[0], mb.size_out[0], mb.send[1], mb.size_out[1],
                               mb.mirror + slot * kMirrorInts);
    if (rc) return rc;
    FSA_HIP(hipEventRecord(mb.ev_pack[slot], s->stream));
    s->count_stale = true;
    return FSA_OK;
}

// Transport of message `msg`, queued on the members' transport streams behind their first pack pass.
static int transport(Group &g, long long msg) {
    const int slot = (int)(msg % kRing), prev = (int)((msg + kRing - 1) % kRing);
    if (g.kind == 0) {
        for (Member &mb : g.m) {
            Solver *s = mb.s;
            FSA_HIP(hipSetDevice(s->device));
            hipStream_t main = s->stream;
            if (mb.comm != main) FSA_HIP(hipStreamWaitEvent(mb.comm, mb.ev_pack[slot], 0));
            s->stream = mb.comm;  // (the timer brackets the stream the copies run on)
            hipError_t err = hipSuccess;
            auto keep = [&](hipError_t e) {
                if (err == hipSuccess) err = e;
            };
            {
                ScopedTimer t(s, "exchange");
                for (int d = 0; d < 2; d++) {
                    if (!mb.send[d]) continue;
                    Member &peer = g.m[mb.rank + (d == 0 ? -1 : 1)];
                    // the neighbour must be through with what it received in the step before
                    if (msg > 0) keep(hipStreamWaitEvent(mb.comm, peer.ev_unpack[prev], 0));
                    const size_t bytes = (size_t)(1 + mb.size_out[d]) * g.rb;
                    char *dst = peer.recv[1 - d];
                    if (peer.s->device == s->device)
                        keep(hipMemcpyAsync(dst, mb.send[d], bytes, hipMemcpyDeviceToDevice, mb.comm));
                    else
                        keep(hipMemcpyPeerAsync(dst, peer.s->device, mb.send[d], s->device, bytes, mb.comm));
                    keep(hipEventRecord(mb.ev_sent[d], mb.comm));
                    peer.bound_in[1 - d] = mb.size_out[d];
                }
            }
            s->stream = main;  // (before any return)
            FSA_HIP(err);
        }
        // the receiving ends derive their sizes as they would under RCCL: they must be the senders' sizes
        for (Member &mb : g.m) {
            int size_in[2] = {0, 0};
            int rc = incoming_sizes(g, mb, msg, size_in);
            if (rc) return rc;
            for (int d = 0; d < 2; d++) {
                if (!mb.recv[d]) continue;
                const Member &peer = g.m[mb.rank + (d == 0 ? -1 : 1)];
                if (size_in[d] != peer.size_out[1 - d])
                    return fail(FSA_ERR_INVALID, "slab group: message %lld from slab %d to slab %d is sized %d records by "
                                                 "the sender and %d by the receiver", msg, peer.rank, mb.rank,
                                peer.size_out[1 - d], size_in[d]);
            }
        }
    } else if (g.kind == 2) {
        Member &mb = g.m[0];
        Solver *s = mb.s;
        int size_in[2] = {0, 0};
        int rc = incoming_sizes(g, mb, msg, size_in);
        if (rc) return rc;
        size_t out_bytes[2] = {0, 0}, in_bytes[2] = {0, 0}, need = 0;
        for (int d = 0; d < 2; d++) {
            if (!mb.send[d]) continue;
            out_bytes[d] = (size_t)(1 + mb.size_out[d]) * g.rb;
            in_bytes[d] = (size_t)(1 + size_in[d]) * g.rb;
            need += out_bytes[d] + in_bytes[d];
        }
        if (need > g.stage_bytes) {
            if (g.stage) (void)hipHostFree(g.stage);
            g.stage = nullptr;
            g.stage_bytes = 0;
            FSA_HIP(hipHostMalloc((void **)&g.stage, need, hipHostMallocDefault));
            g.stage_bytes = need;
        }
        char *out[2] = {g.stage, g.stage + out_bytes[0]};
        char *in[2] = {out[1] + out_bytes[1], out[1] + out_bytes[1] + in_bytes[0]};
        if (mb.comm != s->stream) FSA_HIP(hipStreamWaitEvent(mb.comm, mb.ev_pack[slot], 0));
        for (int d = 0; d < 2; d++)
            if (mb.send[d]) FSA_HIP(hipMemcpyAsync(out[d], mb.send[d], out_bytes[d], hipMemcpyDeviceToHost, mb.comm));
        FSA_HIP(counted_stream_sync(mb.comm));
        // the slab below first on every rank: rank r's call meets rank r - 1's second one
        for (int d = 0; d < 2; d++)
            if (mb.send[d] && g.via.sendrecv(g.via.ctx, mb.rank + (d == 0 ? -1 : 1), out[d], out_bytes[d], in[d], in_bytes[d]) != 0)
                return fail(FSA_ERR_HIP, "slab group: the caller's sendrecv failed");
        for (int d = 0; d < 2; d++) {
            if (!mb.send[d]) continue;
            FSA_HIP(hipMemcpyAsync(mb.recv[d], in[d], in_bytes[d], hipMemcpyHostToDevice, mb.comm));
            mb.bound_in[d] = size_in[d];
        }
        FSA_HIP(counted_stream_sync(mb.comm));  // (the staging memory is used again by the next message)
        FSA_HIP(hipEventRecord(mb.ev_arrived, mb.comm));
    } else {
        Member &mb = g.m[0];
        Solver *s = mb.s;
        int size_in[2] = {0, 0};
        int rc = incoming_sizes(g, mb, msg, size_in);
        if (rc) return rc;
        hipStream_t main = s->stream;
        if (mb.comm != main) FSA_HIP(hipStreamWaitEvent(mb.comm, mb.ev_pack[slot], 0));
        s->stream = mb.comm;
        ncclResult_t nr = ncclSuccess;
        {
            ScopedTimer t(s, "exchange");
            nr = ncclGroupStart();
            for (int d = 0; d < 2 && nr == ncclSuccess; d++) {
                if (!mb.send[d]) continue;
                const int peer = mb.rank + (d == 0 ? -1 : 1);
                nr = ncclSend(mb.send[d], (size_t)(1 + mb.size_out[d]) * g.rb, ncclChar, peer, g.comm, mb.comm);
                if (nr == ncclSuccess)
                    nr = ncclRecv(mb.recv[d], (size_t)(1 + size_in[d]) * g.rb, ncclChar, peer, g.comm, mb.comm);
                mb.bound_in[d] = size_in[d];
            }
            const ncclResult_t ne = ncclGroupEnd();
            if (nr == ncclSuccess) nr = ne;
        }
        s->stream = main;
        if (nr != ncclSuccess) return fail(FSA_ERR_HIP, "RCCL send / receive failed: %s", ncclGetErrorString(nr));
        FSA_HIP(hipEventRecord(mb.ev_arrived, mb.comm));
    }
    return FSA_OK;
}

// Second pack pass of a member.
static int pack_rest(Member &mb, bool with_advect) {
    Solver *s = mb.s;
    int rc = launch_slab_keep_hash(s, with_advect, mb.own, mb.held, mb.left, mb.right);
    if (rc) return rc;
    if (with_advect) s->sorted_valid = s->lists_valid = s->hits_valid = false;  // positions moved
    s->hash_valid = s->perm_valid = false;
    s->hash_ahead = true;  // hash / slot / cellCount describe the slots of the current arrays
    mb.ahead = true;
    return FSA_OK;
}

static int group_step(Group &g) {
    const long long msg = g.step;
    const int slot = (int)(msg % kRing);
    // ---- pack and transport of this step's messages, where the step before has not left them done --------------
    bool any = false, all = true;
    for (Member &mb : g.m) {
        const bool ahead = mb.ahead && mb.s->hash_ahead;
        any |= ahead;
        all &= ahead;
    }
    if (any && !all)
        return fail(FSA_ERR_INVALID, "the particles of a member of the slab group were changed between two steps of the "
                                     "group: destroy the group and make a new one");
    if (g.kind != 0 && msg > 0 && !all)
        return fail(FSA_ERR_INVALID, "the particles of a member of the slab group were changed between two steps of the "
                                     "group (its messages are already with the neighbours): destroy the group and make a "
                                     "new one");
    if (!all) {
        for (Member &mb : g.m) {
            FSA_HIP(hipSetDevice(mb.s->device));
            int rc = pack_edges(g, mb, msg, false);
            if (rc) return rc;
        }
        int rc = transport(g, msg);
        if (rc) return rc;
        for (Member &mb : g.m) {
            FSA_HIP(hipSetDevice(mb.s->device));
            rc = pack_rest(mb, false);
            if (rc) return rc;
        }
    }
    // ---- unpack, the solver's own step up to the forces next to the cuts, first pack pass of the next message ----
    for (Member &mb : g.m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        if (g.kind == 0) {
            for (int d = 0; d < 2; d++) {
                if (!mb.recv[d]) continue;
                FSA_HIP(hipStreamWaitEvent(s->stream, g.m[mb.rank + (d == 0 ? -1 : 1)].ev_sent[1 - d], 0));
                // and this member's own copies have left its send buffers, which the next pack overwrites
                if (mb.comm != s->stream) FSA_HIP(hipStreamWaitEvent(s->stream, mb.ev_sent[d], 0));
            }
        } else if (mb.comm != s->stream) {
            FSA_HIP(hipStreamWaitEvent(s->stream, mb.ev_arrived, 0));
        }
        int rc = launch_slab_unpack_hash(s, mb.recv[0], mb.bound_in[0], mb.recv[1], mb.bound_in[1],
                                         mb.mirror + slot * kMirrorInts);
        if (rc) return rc;
        FSA_HIP(hipEventRecord(mb.ev_unpack[slot], s->stream));
        mb.ahead = false;
        rc = launch_sort_fused(s, false);
        if (rc) return rc;
        s->cur = 1 - s->cur;
        s->hash_valid = s->hash_ahead = false;
        s->sorted_valid = s->owned_from_sort = s->count_valid = true;
        s->pd_valid = s->lists_valid = s->hits_valid = false;
        s->perm_valid = s->index_table_valid = s->ref_cells_valid = false;
        rc = launch_density(s);
        if (rc) return rc;
        // (a member without neighbours has no edges: its forces come in one launch)
        s->force_zone = (mb.has_left || mb.has_right) ? 1 : 0;
        rc = launch_force(s);
        s->force_zone = 0;
        if (rc) return rc;
        rc = pack_edges(g, mb, msg + 1, true);
        if (rc) return rc;
    }
    // ---- the records travel; meanwhile the other forces and the second pack pass ----------------------------------
    int rc = transport(g, msg + 1);
    if (rc) return rc;
    for (Member &mb : g.m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        if (mb.has_left || mb.has_right) {
            s->force_zone = 2;
            rc = launch_force(s);
            s->force_zone = 0;
            if (rc) return rc;
        }
        rc = pack_rest(mb, true);
        if (rc) return rc;
    }
    g.step++;
    return FSA_OK;
}

}  // namespace fsa

using namespace fsa;

extern "C" {

int fsa_slab_group_create_local(int world, fsa_solver **solvers, const int *cuts, int halo, int record_cap,
                                fsa_slab_group **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (world < 1 || !solvers || !cuts || halo < 1 || record_cap < 0) return fail(FSA_ERR_INVALID, "bad argument");
    for (int r = 0; r < world; r++)
        if (cuts[r + 1] - cuts[r] < halo + 1)
            return fail(FSA_ERR_INVALID, "slab %d has %d layers, fewer than halo + 1 = %d", r, cuts[r + 1] - cuts[r], halo + 1);
    Group *g = new (std::nothrow) Group();
    if (!g) return fail(FSA_ERR_NOMEM, "out of host memory");
    g->kind = 0;
    g->overlap = overlap_wanted();
    g->slack = slack_wanted();
    g->world = world;
    g->halo = halo;
    g->record_cap = record_cap;
    g->cuts.assign(cuts, cuts + world + 1);
    g->rb = solvers[0] ? slab_record_bytes(reinterpret_cast<Solver *>(solvers[0])) : 48;
    g->m.resize(world);
    for (int r = 0; r < world; r++) {
        int rc = setup_member(*g, g->m[r], reinterpret_cast<Solver *>(solvers[r]), r);
        if (rc) {
            for (Member &mb : g->m) free_member(mb);
            delete g;
            return rc;
        }
    }
    *out = reinterpret_cast<fsa_slab_group *>(g);
    return FSA_OK;
}

int fsa_slab_group_unique_id(char id[FSA_SLAB_GROUP_ID_BYTES]) {
    if (!id) return fail(FSA_ERR_INVALID, "id is NULL");
    static_assert(sizeof(ncclUniqueId) <= FSA_SLAB_GROUP_ID_BYTES, "id buffer too small");
    ncclUniqueId u;
    FSA_NCCL(ncclGetUniqueId(&u));
    std::memset(id, 0, FSA_SLAB_GROUP_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return FSA_OK;
}

int fsa_slab_group_create_rccl(fsa_solver *solver, int rank, int world, const char id[FSA_SLAB_GROUP_ID_BYTES],
                               const int *cuts, int halo, int record_cap, fsa_slab_group **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !solver || !cuts || !id || halo < 1 || record_cap < 0)
        return fail(FSA_ERR_INVALID, "bad argument");
    for (int r = 0; r < world; r++)
        if (cuts[r + 1] - cuts[r] < halo + 1)
            return fail(FSA_ERR_INVALID, "slab %d has %d layers, fewer than halo + 1 = %d", r, cuts[r + 1] - cuts[r], halo + 1);
    Group *g = new (std::nothrow) Group();
    if (!g) return fail(FSA_ERR_NOMEM, "out of host memory");
    g->kind = 1;
    g->overlap = overlap_wanted();
    g->slack = slack_wanted();
    g->world = world;
    g->halo = halo;
    g->record_cap = record_cap;
    g->cuts.assign(cuts, cuts + world + 1);
    Solver *s = reinterpret_cast<Solver *>(solver);
    g->rb = slab_record_bytes(s);
    g->m.resize(1);
    int rc = setup_member(*g, g->m[0], s, rank);
    if (!rc) {
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof(u));
        ncclResult_t r = ncclCommInitRank(&g->comm, world, u, rank);
        if (r != ncclSuccess) rc = fail(FSA_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    if (rc) {
        free_member(g->m[0]);
        delete g;
        return rc;
    }
    *out = reinterpret_cast<fsa_slab_group *>(g);
    return FSA_OK;
}

int fsa_slab_group_create_hosted(fsa_solver *solver, int rank, int world, const fsa_host_transport *via, const int *cuts,
                                 int halo, int record_cap, fsa_slab_group **out) {
    if (!out) return fail(FSA_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !solver || !cuts || !via || !via->sendrecv || halo < 1 || record_cap < 0)
        return fail(FSA_ERR_INVALID, "bad argument");
    for (int r = 0; r < world; r++)
        if (cuts[r + 1] - cuts[r] < halo + 1)
            return fail(FSA_ERR_INVALID, "slab %d has %d layers, fewer than halo + 1 = %d", r, cuts[r + 1] - cuts[r], halo + 1);
    Group *g = new (std::nothrow) Group();
    if (!g) return fail(FSA_ERR_NOMEM, "out of host memory");
    g->kind = 2;
    g->via = *via;
    g->overlap = overlap_wanted();
    g->slack = slack_wanted();
    g->world = world;
    g->halo = halo;
    g->record_cap = record_cap;
    g->cuts.assign(cuts, cuts + world + 1);
    Solver *s = reinterpret_cast<Solver *>(solver);
    g->rb = slab_record_bytes(s);
    g->m.resize(1);
    const int rc = setup_member(*g, g->m[0], s, rank);
    if (rc) {
        free_member(g->m[0]);
        delete g;
        return rc;
    }
    *out = reinterpret_cast<fsa_slab_group *>(g);
    return FSA_OK;
}

int fsa_slab_group_step(fsa_slab_group *hg, int nsteps) {
    Group *g = reinterpret_cast<Group *>(hg);
    if (!g) return fail(FSA_ERR_INVALID, "group is NULL");
    if (nsteps < 0) return fail(FSA_ERR_INVALID, "nsteps < 0");
    for (int k = 0; k < nsteps; k++) {
        int rc = group_step(*g);
        if (rc) return rc;
    }
    return FSA_OK;
}

int fsa_slab_group_sync(fsa_slab_group *hg, fsa_slab_group_stats *st) {
    Group *g = reinterpret_cast<Group *>(hg);
    if (!g) return fail(FSA_ERR_INVALID, "group is NULL");
    long long lost = 0, particles = 0, max_particles = 0;
    int error = 0;
    for (Member &mb : g->m) {
        Solver *s = mb.s;
        FSA_HIP(hipSetDevice(s->device));
        FSA_HIP(counted_stream_sync(s->stream));
        if (mb.comm != s->stream) FSA_HIP(counted_stream_sync(mb.comm));
        int vars[8];
        FSA_HIP(hipMemcpy(vars, s->b.slabVars, sizeof(vars), hipMemcpyDeviceToHost));
        s->n = vars[kVarN];
        s->count_stale = false;
        lost += vars[kVarLost];
        error |= vars[kVarError];
        particles += vars[kVarN];
        max_particles = std::max<long long>(max_particles, vars[kVarN]);
    }
    if (st) {
        st->steps = g->step;
        st->particles = particles;
        st->max_particles = max_particles;
        st->lost = lost;
        st->records_sent = g->sent;
        st->error_bits = error;
    }
    if (error)
        return fail(FSA_ERR_CAPACITY, "slab group: a count outgrew its buffer (bits %d: 1 solver capacity, 2 / 4 message to the "
                                      "left / right, 8 / 16 message from the left / right, 32 a particle crossed a whole cell "
                                      "layer in one step); the state is invalid", error);
    return FSA_OK;
}

int fsa_slab_group_rccl_loopback(fsa_slab_group *hg, const void *d_src, void *d_dst, size_t bytes) {
    Group *g = reinterpret_cast<Group *>(hg);
    if (!g || g->kind != 1 || !g->comm) return fail(FSA_ERR_INVALID, "needs a group made by fsa_slab_group_create_rccl");
    if (bytes && (!d_src || !d_dst)) return fail(FSA_ERR_INVALID, "NULL buffer");
    Member &mb = g->m[0];
    FSA_HIP(hipSetDevice(mb.s->device));
    FSA_NCCL(ncclGroupStart());
    FSA_NCCL(ncclSend(d_src, bytes, ncclChar, mb.rank, g->comm, mb.s->stream));
    FSA_NCCL(ncclRecv(d_dst, bytes, ncclChar, mb.rank, g->comm, mb.s->stream));
    FSA_NCCL(ncclGroupEnd());
    FSA_HIP(counted_stream_sync(mb.s->stream));
    return FSA_OK;
}

int fsa_slab_group_destroy(fsa_slab_group *hg) {
    Group *g = reinterpret_cast<Group *>(hg);
    if (!g) return FSA_OK;
    for (Member &mb : g->m) free_member(mb);
    if (g->comm) (void)ncclCommDestroy(g->comm);
    if (g->stage) (void)hipHostFree(g->stage);
    delete g;
    return FSA_OK;
}

}  // extern "C"
