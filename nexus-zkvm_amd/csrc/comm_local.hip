// The in-process transport of a row-sharded prove: ONE process, one thread per GPU (or several contexts on one GPU), no RCCL.
// Every collective of nx_comm is a rendezvous of the W threads on a shared board (mutex + condition variable) plus device-to-device
// copies that each rank PULLS from its peers' buffers on a stream of its own — hipMemcpyAsync between two devices of one process goes
// over xGMI peer to peer once peer access is enabled, which nx_comm_local_create does for the devices of the group.  This is the
// transport a single-process prover farm uses on one node, and what the GPU test-suite runs the 2 / 4 / 8-rank proofs on (the Python
// ThreadGroup of sharded.py is the same protocol with the interpreter's lock in every collective: tools/thread_ranks_bench.py measures
// both).  A rank that fails calls abort(): the board is marked broken and every waiting or later rendezvous returns an error; waits are
// also bounded by the context option "comm.timeout_ms".
#include "internal.h"
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string.h>

struct nx_comm_group {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    struct Slot { const void* p = nullptr; const size_t* off = nullptr; const size_t* cnt = nullptr; size_t n = 0; };
    std::vector<Slot> slot;
    std::vector<int> device;          // device of every rank's context (peer access is enabled pairwise at create)
    std::vector<std::pair<const uint32_t*, size_t>> mail;      // point-to-point board, index src * world + dst
    std::vector<char> mail_busy;      // the receiver is copying out of this mailbox's buffer right now: the sender must not return (and free it)
    int inflight = 0;                 // ranks that are pulling from their peers' buffers right now (between the two rendezvous of a collective)
    std::vector<std::vector<char>> reach;   // reach[rank][device]: rank's device reads that device's memory directly (itself, or peer access enabled at create)
};

namespace nx {
namespace {

struct LocalComm {
    nx_comm iface;            // FIRST member: nx_comm* <-> LocalComm*
    nx_ctx* ctx = nullptr;
    nx_comm_group* g = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0;
};

// all ranks arrive, all leave; 0 ok, 1 when the group is broken or the wait timed out (the group is broken then)
static int rendezvous(LocalComm* c) {
    nx_comm_group* g = c->g;
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return 1;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->world) { g->arrived = 0; g->generation++; g->cv.notify_all(); return 0; }
    const int limit_ms = c->ctx->opt.comm_timeout_ms;
    auto done = [&] { return g->generation != gen || g->broken; };
    if (limit_ms > 0) {
        if (!g->cv.wait_for(lk, std::chrono::milliseconds(limit_ms), done)) { g->broken = true; g->cv.notify_all(); }
    } else g->cv.wait(lk, done);
    return g->generation != gen && !g->broken ? 0 : 1;
}
// every failure of one rank breaks the group: its peers' waits return at once instead of running into the timeout
static int fail(LocalComm* c, const char* what) {
    {
        std::lock_guard<std::mutex> lk(c->g->mu);
        c->g->broken = true;
        c->g->cv.notify_all();
    }
    (void)set_err(c->ctx, NX_ERR_HIP, std::string("local transport: ") + what);
    return 1;
}
// a wait on the board's condition variable, bounded by "comm.timeout_ms" like the rendezvous; false = timed out (the caller fails -> group broken)
template <class Pred>
static bool wait_bounded(LocalComm* c, std::unique_lock<std::mutex>& lk, Pred done) {
    const int limit_ms = c->ctx->opt.comm_timeout_ms;
    if (limit_ms > 0) return c->g->cv.wait_for(lk, std::chrono::milliseconds(limit_ms), done);
    c->g->cv.wait(lk, done);
    return true;
}
// A rank that pulls from its peers' buffers is "in flight" from the first rendezvous until its copies have completed.  A rank that
// leaves a collective through a failure (a timeout of the closing rendezvous, a broken group) first waits — unbounded: copies are
// finite — until nobody is in flight, so that no peer is still reading the buffers its caller is about to free.
static void flight_begin(LocalComm* c) { std::lock_guard<std::mutex> lk(c->g->mu); c->g->inflight++; }
static void flight_end(LocalComm* c) { std::lock_guard<std::mutex> lk(c->g->mu); c->g->inflight--; c->g->cv.notify_all(); }
static void drain_flights(LocalComm* c) { std::unique_lock<std::mutex> lk(c->g->mu); c->g->cv.wait(lk, [&] { return c->g->inflight <= 0; }); }
struct Flight { LocalComm* c; explicit Flight(LocalComm* x) : c(x) { flight_begin(c); } ~Flight() { flight_end(c); } };
#define L_MEET(c) do { if (rendezvous(c)) return fail(c, "a peer failed or did not arrive in time (group broken)"); } while (0)
// the closing rendezvous of a collective: on failure the peers may still be reading this rank's send buffer
#define L_MEET_CLOSE(c) do { if (rendezvous(c)) { drain_flights(c); return fail(c, "a peer failed or did not arrive in time (group broken)"); } } while (0)
#define L_HIP(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(c, hipGetErrorString(e__)); } while (0)
// Inside a Flight scope: pulls already queued on c->stream may still be READING the peers' send buffers, and ~Flight tells the peers this
// rank is done with them.  Drain the stream before failing, so that a peer whose closing rendezvous fails frees nothing under a copy.
#define L_HIP_FLIGHT(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (void)hipStreamSynchronize((c)->stream); return fail(c, hipGetErrorString(e__)); } } while (0)

static int cb_allgather(void* user, const void* h_send, size_t bytes, void* h_recv) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    g->slot[c->rank].p = h_send;
    L_MEET(c);
    { Flight fl(c); for (int r = 0; r < g->world; r++) memcpy((uint8_t*)h_recv + (size_t)r * bytes, g->slot[r].p, bytes); }
    L_MEET_CLOSE(c);                             // nobody reuses its send buffer before everyone has read it
    return 0;
}
static int cb_broadcast(void* user, void* h_buf, size_t bytes, int32_t root) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    if (c->rank == root) g->slot[root].p = h_buf;
    L_MEET(c);
    { Flight fl(c); if (c->rank != root) memcpy(h_buf, g->slot[root].p, bytes); }
    L_MEET_CLOSE(c);
    return 0;
}
static int cb_allgather_dev(void* user, const uint32_t* d_send, size_t n_words, uint32_t* d_recv) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    DeviceGuard dg(c->ctx);
    g->slot[c->rank].p = d_send;
    L_MEET(c);
    {
        Flight fl(c);
        for (int k = 0; k < g->world && n_words; k++) {
            const int r = (c->rank + k) % g->world;          // every rank starts with a different peer: the pulls spread over the links
            L_HIP_FLIGHT(c, hipMemcpyAsync(d_recv + (size_t)r * n_words, g->slot[r].p, n_words * 4, hipMemcpyDeviceToDevice, c->stream));
        }
        L_HIP(c, hipStreamSynchronize(c->stream));
    }
    L_MEET_CLOSE(c);
    return 0;
}
static int cb_alltoallv(void* user, const uint32_t* d_send, const size_t* soff, const size_t* scnt, uint32_t* d_recv, const size_t* roff, const size_t* rcnt) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    DeviceGuard dg(c->ctx);
    nx_comm_group::Slot& mine = g->slot[c->rank];
    mine.p = d_send; mine.off = soff; mine.cnt = scnt;
    L_MEET(c);
    {
        Flight fl(c);
        for (int k = 0; k < g->world; k++) {
            const int r = (c->rank + k) % g->world;
            const nx_comm_group::Slot& s = g->slot[r];
            if (s.cnt[c->rank] != rcnt[r]) { (void)hipStreamSynchronize(c->stream); return fail(c, "all-to-all: a peer sends a different count than this rank expects"); }
            if (rcnt[r]) L_HIP_FLIGHT(c, hipMemcpyAsync(d_recv + roff[r], (const uint32_t*)s.p + s.off[c->rank], rcnt[r] * 4, hipMemcpyDeviceToDevice, c->stream));
        }
        L_HIP(c, hipStreamSynchronize(c->stream));
    }
    L_MEET_CLOSE(c);
    return 0;
}
// point to point (the ring commit protocol): the receiver pulls
static int cb_send(void* user, int32_t dst, const uint32_t* d_buf, size_t n_words) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    if (dst < 0 || dst >= g->world || dst == c->rank) return fail(c, "send: bad destination rank");
    if (nx_sync(c->ctx) != NX_OK) return fail(c, "send: the context's stream failed");
    bool in_time, broken;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        auto& m = g->mail[(size_t)c->rank * g->world + dst];
        m = {d_buf, n_words};
        g->cv.notify_all();
        in_time = wait_bounded(c, lk, [&] { return m.first == nullptr || g->broken; });       // the receiver copied it
        broken = g->broken;
        // a receiver that has taken the pointer is copying out of d_buf: the caller frees it once this returns — wait the copy out
        char& busy = g->mail_busy[(size_t)c->rank * g->world + dst];
        if (busy) g->cv.wait(lk, [&] { return !busy; });
        if (!in_time || broken) m = {nullptr, 0};
    }
    if (!in_time) return fail(c, "send: the receiver did not arrive in time");
    return broken ? fail(c, "group broken") : 0;
}
static int cb_recv(void* user, int32_t src, uint32_t* d_buf, size_t n_words) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    DeviceGuard dg(c->ctx);
    if (src < 0 || src >= g->world || src == c->rank) return fail(c, "recv: bad source rank");
    const uint32_t* from = nullptr;
    const char* why = nullptr;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        auto& m = g->mail[(size_t)src * g->world + c->rank];
        if (!wait_bounded(c, lk, [&] { return m.first != nullptr || g->broken; })) why = "recv: the sender did not arrive in time";
        else if (g->broken) why = "group broken";
        else if (m.second != n_words) why = "recv: the sender announced a different length";
        else { from = m.first; g->mail_busy[(size_t)src * g->world + c->rank] = 1; }      // taken: the sender stays until the copy is done
    }
    if (why) return fail(c, why);
    hipError_t e = hipMemcpyAsync(d_buf, from, n_words * 4, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->mail_busy[(size_t)src * g->world + c->rank] = 0;
        g->mail[(size_t)src * g->world + c->rank] = {nullptr, 0};
        g->cv.notify_all();
    }
    if (e != hipSuccess) return fail(c, hipGetErrorString(e));
    return 0;
}
// sum mod p over the ranks: everyone snapshots its buffer, then PULLS each peer's snapshot into a local scratch buffer (a copy works with
// or without peer access: the runtime stages it; a kernel reading another device's memory would fault without it — ADVICE r4) and adds it
static int cb_allreduce_m31(void* user, uint32_t* d_buf, size_t n_words) {
    LocalComm* c = (LocalComm*)user; nx_comm_group* g = c->g;
    DeviceGuard dg(c->ctx);
    uint32_t* snap = nullptr; uint32_t* pulled = nullptr;
    struct Free { nx_ctx* ctx; uint32_t** a; uint32_t** b; ~Free() { if (*a) (void)nx_free(ctx, *a); if (*b) (void)nx_free(ctx, *b); } } guard{c->ctx, &snap, &pulled};
    if (nx_alloc(c->ctx, std::max<size_t>(n_words, 1), &snap) != NX_OK || nx_alloc(c->ctx, std::max<size_t>(n_words, 1), &pulled) != NX_OK) return fail(c, "allreduce: out of device memory");
    if (nx_copy(c->ctx, snap, d_buf, n_words) != NX_OK || nx_sync(c->ctx) != NX_OK) return fail(c, "allreduce: the snapshot failed");
    g->slot[c->rank].p = snap;
    L_MEET(c);
    {
        Flight fl(c);
        for (int k = 1; k < g->world && n_words; k++) {
            const int r = (c->rank + k) % g->world;
            L_HIP_FLIGHT(c, hipMemcpyAsync(pulled, g->slot[r].p, n_words * 4, hipMemcpyDeviceToDevice, c->stream));
            L_HIP(c, hipStreamSynchronize(c->stream));
            if (nx_m31_add_into(c->ctx, d_buf, pulled, n_words) != NX_OK || nx_sync(c->ctx) != NX_OK) return fail(c, "allreduce: the modular add failed");
        }
    }
    L_MEET_CLOSE(c);                                 // the snapshots are freed only when everyone has pulled them
    return 0;
}
static void cb_abort(void* user) {
    LocalComm* c = (LocalComm*)user;
    std::lock_guard<std::mutex> lk(c->g->mu);
    c->g->broken = true;
    c->g->cv.notify_all();
}

}  // namespace
}  // namespace nx

using namespace nx;

extern "C" {

int nx_comm_group_create(int32_t world, nx_comm_group** out) {
    if (!out || world < 1 || world > 64) return set_err(nullptr, NX_ERR_ARG, "nx_comm_group_create: 1 <= world <= 64 required");
    nx_comm_group* g = new nx_comm_group();
    g->world = world; g->slot.resize(world); g->device.assign(world, -1); g->mail.assign((size_t)world * world, {nullptr, 0}); g->mail_busy.assign((size_t)world * world, 0); g->reach.assign((size_t)world, {});
    *out = g;
    return NX_OK;
}
void nx_comm_group_destroy(nx_comm_group* g) { delete g; }
// Re-arm a group that a failure (abort, a timeout) left broken.  Only when EVERY rank has left its collective — the caller's
// threads have all returned from their prove calls: a long-lived prover farm calls it between proofs after an asymmetric failure.
int nx_comm_group_reset(nx_comm_group* g) {
    if (!g) return set_err(nullptr, NX_ERR_ARG, "nx_comm_group_reset: NULL group");
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->inflight > 0) return set_err(nullptr, NX_ERR_ARG, "nx_comm_group_reset: a rank is still inside a collective");
    g->broken = false; g->arrived = 0; g->generation++;
    for (auto& m : g->mail) m = {nullptr, 0};
    for (auto& b : g->mail_busy) b = 0;
    g->cv.notify_all();
    return NX_OK;
}
int nx_comm_group_broken(const nx_comm_group* g) { return g && g->broken ? 1 : 0; }
// 1: copies from to_rank's device into from_rank's go peer to peer (or both share a device); 0: the runtime stages them; -1: a rank is not created yet
int nx_comm_group_peer_access(nx_comm_group* g, int32_t from_rank, int32_t to_rank) {
    if (!g || from_rank < 0 || to_rank < 0 || from_rank >= g->world || to_rank >= g->world) return -1;
    std::lock_guard<std::mutex> lk(g->mu);
    const int d = g->device[to_rank];
    if (d < 0 || g->device[from_rank] < 0 || (size_t)d >= g->reach[from_rank].size()) return -1;
    return g->reach[from_rank][d] ? 1 : 0;
}

int nx_comm_local_create(nx_comm_group* g, nx_ctx* ctx, int32_t rank, nx_comm** out) {
    NX_GUARD(ctx);
    if (!g || !ctx || !out || rank < 0 || rank >= g->world) return set_err(ctx, NX_ERR_ARG, "nx_comm_local_create: bad argument");
    LocalComm* c = new LocalComm();
    c->ctx = ctx; c->g = g; c->rank = rank;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_err(ctx, NX_ERR_HIP, "nx_comm_local_create: hipStreamCreate failed"); }
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->device[rank] = ctx->device;
    }
    {   // pulls read the peers' memory: peer access from this context's device to every other GPU of the node lets a copy go over xGMI
        // directly; without it (not reachable) the runtime stages the copy — every collective here moves data by hipMemcpyAsync only,
        // never by a kernel dereferencing a peer's pointer, so both work.  The outcome per device is kept for the record.
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); n_dev = 0; }
        std::vector<char> reach((size_t)std::max(n_dev, 1), 0);
        for (int d = 0; d < n_dev; d++) {
            if (d == ctx->device) { reach[d] = 1; continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(d, 0);
            reach[d] = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            if (e != hipSuccess) (void)hipGetLastError();
        }
        std::lock_guard<std::mutex> lk(g->mu);
        g->reach[rank] = reach;
    }
    memset(&c->iface, 0, sizeof c->iface);
    c->iface.rank = rank; c->iface.world = g->world; c->iface.user = c;
    c->iface.send = cb_send; c->iface.recv = cb_recv; c->iface.allreduce_m31 = cb_allreduce_m31;
    c->iface.allgather = cb_allgather; c->iface.broadcast = cb_broadcast;
    c->iface.alltoallv = cb_alltoallv; c->iface.allgather_dev = cb_allgather_dev; c->iface.abort = cb_abort;
    *out = &c->iface;
    return NX_OK;
}
void nx_comm_local_destroy(nx_comm* comm) {
    if (!comm) return;
    LocalComm* c = (LocalComm*)comm->user;
    DeviceGuard g(c->ctx);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C"
