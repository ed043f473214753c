// One-shot all-reduce of the flat gradient across the GPUs of one node, fused with the
// weight update (replaces the host-staged broadcast / slab gather / master update of
// hyper_version_mGPU_MBGD.cu:962-978, 1093-1171).
//
// The message is small (46 853 .. 108 853 values), so the exchange is latency-bound and a
// ring's 2(n-1) hops are the wrong shape.  Every rank owns an exchange buffer that its peers
// map (IPC handle between processes, or the pointer itself inside one process); in ONE kernel
// a workgroup takes a 4 KB chunk of the vector and
//   1. writes its chunk of the local gradient into slot [rank] of every peer's buffer over
//      xGMI (all links busy at once, one hop),
//   2. waits for its writes to complete and raises flag [rank][chunk] at every peer,
//   3. waits until its own flags [0..n)[chunk] show this call's sequence number,
//   4. adds the n slots in rank order - the same order on every rank, so all ranks compute
//      bit-identical sums - and applies the update rule to its chunk of the parameters.
// Chunks are independent, so there is no grid-wide barrier.  Slots and flags are double-buffered
// by call parity: a rank can only start call k+2 after every peer raised its flags of call k+1,
// which each peer did after its kernel of call k (the last reader of that parity) had finished.
// The buffers are uncached (fine-grained) device memory: peers' writes bypass the local L2,
// so must the local reads.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <cstring>
#include <new>
#include <random>

#include "comm.h"
#include "profile.h"

namespace ccnn {

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

struct RulePlain {   // buf[e] = sum
    template <typename T>
    __device__ void operator()(T *p, long e, T sum) const { p[e] = sum; }
};
template <typename T>
struct RuleSgd {     // params[e] -= lr * sum / divisor, elements in [lo, hi) untouched
    T lr;
    T divisor;
    long lo, hi;
    __device__ void operator()(T *p, long e, T sum) const
    {
        if (e >= lo && e < hi) return;
        p[e] = p[e] - lr * sum / divisor;
    }
};
template <>
struct RuleSgd<float> {   // fp32 variant: plain sum, no division (gcn_mbgd_fp32.cu:1031-1059)
    float lr;
    float divisor;
    long lo, hi;
    __device__ void operator()(float *p, long e, float sum) const
    {
        if (e >= lo && e < hi) return;
        p[e] = p[e] - lr * sum;
    }
};

template <typename T, typename Rule>
__global__ __launch_bounds__(256) void allreduce_apply(T *target, const T *src, long n,   // may alias (in-place sum)
                                                       CommDev c, unsigned step, Rule rule)
{
    constexpr int VW = 16 / sizeof(T);
    constexpr int CH = COMM_CHUNK_BYTES / sizeof(T);
    typedef T Vec __attribute__((ext_vector_type(16 / sizeof(T))));
    const int tid = threadIdx.x, chunk = blockIdx.x;
    const int par = step & 1;
    const long e0 = (long)chunk * CH + (long)tid * VW;
    const long slot_off = (long)chunk * COMM_CHUNK_BYTES + (long)tid * 16;

    // 1. this rank's chunk -> slot [par][rank] of every peer (clamped, unconditional load; tail lanes
    //    carry copies that nobody reads back)
    Vec v;
#pragma unroll
    for (int j = 0; j < VW; j++) v[j] = src[min(e0 + j, n - 1)];
    for (int k = 0; k < c.world; k++) {
        const int r = (c.rank + 1 + k) % c.world;   // start at the next rank: spreads the first writes over the links
        comm_store16(comm_slot(c, r, par, c.rank) + slot_off, &v);
    }
    // 2. publish, 3. wait for every rank's chunk
    comm_publish(c, par, step, chunk, tid);
    // a wait that ran out leaves `target` as it was (the status word tells the host): sums of
    // stale slots are never applied
    if (!comm_wait(c, par, step, chunk, tid)) return;
    // 4. sum in rank order, apply
    T sum[VW];
#pragma unroll
    for (int j = 0; j < VW; j++) sum[j] = 0;
    for (int r = 0; r < c.world; r++) {
        Vec u;
        comm_load16(&u, comm_slot(c, c.rank, par, r) + slot_off);
#pragma unroll
        for (int j = 0; j < VW; j++) sum[j] = sum[j] + u[j];
    }
#pragma unroll
    for (int j = 0; j < VW; j++)
        if (e0 + j < n) rule(target, e0 + j, sum[j]);
}

template <typename T, typename Rule>
int allreduce_launch(ccnn_comm *cm, T *target, const T *src, long n, Rule rule, void *stream)
{
    if (!cm || !target || !src || n <= 0) return CCNN_E_INVALID;
    if ((int)sizeof(T) != cm->elem || n > cm->max_elems) return CCNN_E_RANGE;
    for (int r = 0; r < cm->dev.world; r++)
        if (!cm->dev.peer[r]) return CCNN_E_INVALID;   // not connected
    constexpr int CH = COMM_CHUNK_BYTES / sizeof(T);
    const int chunks = (int)((n + CH - 1) / CH);
    const unsigned step = cm->next_step();
    launch("allreduce_apply", allreduce_apply<T, Rule>, dim3(chunks), dim3(256), 0, (hipStream_t)stream, target, src,
           n, cm->dev, step, rule);
    CCNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace ccnn

using namespace ccnn;

namespace {
// CCNN_COMM_DEBUG=1: one line per buffer event on stderr (owner address, IPC handle bytes, the
// address a peer's handle was mapped at, return codes of unmap and free).
bool comm_debug()
{
    static const bool on = getenv("CCNN_COMM_DEBUG") != nullptr;
    return on;
}
// Identity of the physical device behind the current HIP device (ranks may address one GPU under
// different ordinals): PCI domain, bus and device number.
unsigned long long device_identity(int device)
{
    int dom = 0, bus = 0, dv = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess) dom = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess) bus = 0;
    if (hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, device) != hipSuccess) dv = 0;
    (void)hipGetLastError();
    return ((unsigned long long)(unsigned)dom << 32) | ((unsigned long long)(bus & 0xffff) << 16) | (unsigned)(dv & 0xffff) | (1ull << 63);
}
void log_handle(const char *what, int rank, int peer, const void *ptr, const void *handle, int rc)
{
    if (!comm_debug()) return;
    char hex[2 * CCNN_COMM_HANDLE_BYTES + 1] = {0};
    if (handle)
        for (int i = 0; i < CCNN_COMM_HANDLE_BYTES; i++)
            snprintf(hex + 2 * i, 3, "%02x", static_cast<const unsigned char *>(handle)[i]);
    fprintf(stderr, "[ccnn-comm rank %d] %s peer=%d ptr=%p rc=%d handle=%s\n", rank, what, peer, ptr, rc, hex);
}

// Exchange buffers live as long as the process: ccnn_comm_destroy hands a buffer back to this
// list and a later ccnn_comm_create on the same device takes it again (zeroed).  Uncached memory
// (hipDeviceMallocUncached) must not go back to the allocator while the process runs: the next
// ordinary allocation that lands on the freed range computes wrong results.  Measured with two ranks
// on one GPU (tools/comm_reuse_probe.py, CCNN_COMM_POOL=free): after hipFree of a first set of
// buffers the owner's next large allocation - a training workspace on the freed address range -
// gives LOCAL gradients (no exchange involved) wrong by up to 20 %, call after call, with the inputs
// intact and no wait timed out, in 4 of 6 rank-runs; the same gradient is exact in any other block,
// and in the same block once a fill has overwritten all of it, so nothing keeps writing there (no
// late peer store, no tagged words).  With the same buffers allocated as ordinary device memory
// (CCNN_COMM_ALLOC=plain) and freed: 0 of 6; with the buffers kept: 0 of 6.  The fault follows the
// page attribute, not the peers' mappings; keeping the buffers avoids it, and so would ordinary
// memory - which peers could not write coherently over xGMI.
struct PooledBuffer {
    void *ptr;
    size_t bytes;
    int device;
    bool in_use;
};
std::mutex g_pool_mu;
std::vector<PooledBuffer> g_pool;

hipError_t pool_acquire(int device, size_t bytes, void **out)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (PooledBuffer &b : g_pool) {
        if (!b.in_use && b.device == device && b.bytes >= bytes) {
            b.in_use = true;
            *out = b.ptr;
            return hipSuccess;
        }
    }
    void *p = nullptr;
    // uncached: peers write it over xGMI behind the local L2's back
    // (CCNN_COMM_ALLOC=plain, diagnostics only: ordinary device memory - tells "the uncached attribute
    // outlives the allocation" from "a peer's mapping outlives its close" in tools/comm_reuse_probe.py)
    static const bool plain = getenv("CCNN_COMM_ALLOC") && !strcmp(getenv("CCNN_COMM_ALLOC"), "plain");
    const hipError_t e = plain ? hipMalloc(&p, bytes) : hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) return e;
    g_pool.push_back({p, bytes, device, true});
    *out = p;
    return hipSuccess;
}

void pool_release(void *ptr)
{
    // CCNN_COMM_POOL=free (diagnostics only): give the buffer back to the allocator, the behaviour the
    // pool exists to avoid - tools/comm_reuse_probe.py uses it to reproduce what then goes wrong
    static const bool do_free = getenv("CCNN_COMM_POOL") && !strcmp(getenv("CCNN_COMM_POOL"), "free");
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++) {
        if (g_pool[i].ptr != ptr) continue;
        if (do_free) {
            const hipError_t e = hipFree(ptr);
            if (comm_debug()) fprintf(stderr, "[ccnn-comm] free ptr=%p bytes=%zu rc=%d\n", ptr, g_pool[i].bytes, (int)e);
            g_pool.erase(g_pool.begin() + i);
        } else {
            g_pool[i].in_use = false;
        }
        return;
    }
}

}  // namespace

extern "C" {

int ccnn_comm_create(int world, int rank, long max_elems, int elem_size, ccnn_comm_t **comm, void *handle)
{
    if (!comm || !handle || world < 1 || world > COMM_MAX_RANKS || rank < 0 || rank >= world || max_elems <= 0 ||
        (elem_size != 4 && elem_size != 8))
        return CCNN_E_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == CCNN_COMM_HANDLE_BYTES, "handle size");
    ccnn_comm *cm = new (std::nothrow) ccnn_comm();
    if (!cm) return (int)hipErrorOutOfMemory;
    cm->max_elems = max_elems;
    cm->elem = elem_size;
    cm->step = 0;
    CommDev &d = cm->dev;
    d.world = world;
    d.rank = rank;
    d.spin_ticks = COMM_SPIN_TICKS;
    // In-step exchange: every rank writes to every rank (one hop).  The other form - a workgroup's
    // sums go to one owner rank, who adds them and sends the total round: a quarter (8 ranks) of the
    // bytes per link for one more hop - has only ever run with ranks sharing one GPU, so it is
    // opt-in (CCNN_COMM_EXCHANGE=owner, or ccnn_comm_set_exchange) until a run over xGMI shows it
    // faster; bench.py times both during its warm-up and keeps the faster one.
    d.via_owner = 0;
    if (const char *ex = getenv("CCNN_COMM_EXCHANGE")) {
        if (!strcmp(ex, "direct")) d.via_owner = 0;
        if (!strcmp(ex, "owner")) d.via_owner = 1;
    }
    if (const char *ms = getenv("CCNN_COMM_WAIT_MS")) {   // tests shorten the limit
        const long v = atol(ms);
        if (v > 0) d.spin_ticks = (unsigned long long)v * 100000ull;
    }
    // 2.5 x the vector: the fused minibatch step ships its sums in launch geometry (padded tiles,
    // a quarter more) as sequence-tagged words (twice the bytes)
    const size_t cap_bytes = (size_t)max_elems * elem_size * 5 / 2 + 131072;
    const int nchunks = (int)((cap_bytes + COMM_CHUNK_BYTES - 1) / COMM_CHUNK_BYTES);
    d.nflags = nchunks > COMM_MIN_FLAGS ? nchunks : COMM_MIN_FLAGS;
    d.slot_bytes = (long)nchunks * COMM_CHUNK_BYTES;
    d.flags_off = 2L * world * d.slot_bytes;
    d.status_off = d.flags_off + (long)align_up((size_t)2 * world * d.nflags * sizeof(unsigned));
    cm->total_bytes = (size_t)d.status_off + 256;
    for (int r = 0; r < COMM_MAX_RANKS; r++) {
        d.peer[r] = nullptr;
        cm->opened[r] = false;
    }
    hipError_t e = hipGetDevice(&cm->device);
    void *buf = nullptr;
    cm->shared_device = cm->any_shared = false;
    if (e == hipSuccess) e = pool_acquire(cm->device, cm->total_bytes, &buf);
    if (e == hipSuccess) e = hipMemset(buf, 0, cm->total_bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) {
        std::random_device rd;
        cm->nonce = rd() ^ (unsigned)std::chrono::steady_clock::now().time_since_epoch().count() ^
                    (unsigned)(reinterpret_cast<uintptr_t>(cm) >> 4);
        e = hipMemcpy(static_cast<char *>(buf) + d.status_off + 8, &cm->nonce, sizeof(unsigned), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        cm->identity = device_identity(cm->device);
        e = hipMemcpy(static_cast<char *>(buf) + d.status_off + 16, &cm->identity, sizeof(cm->identity), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(handle), buf);
    log_handle("create", rank, rank, buf, handle, (int)e);
    if (e != hipSuccess) {
        if (buf) pool_release(buf);
        delete cm;
        return (int)e;
    }
    d.peer[rank] = static_cast<char *>(buf);
    *comm = cm;
    return 0;
}

int ccnn_comm_connect(ccnn_comm_t *cm, const void *handles)
{
    if (!cm || !handles) return CCNN_E_INVALID;
    const hipIpcMemHandle_t *h = static_cast<const hipIpcMemHandle_t *>(handles);
    for (int r = 0; r < cm->dev.world; r++) {
        if (r == cm->dev.rank || cm->dev.peer[r]) continue;
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h[r], hipIpcMemLazyEnablePeerAccess);
        log_handle("open", cm->dev.rank, r, p, &h[r], (int)e);
        if (e != hipSuccess) return (int)e;
        cm->dev.peer[r] = static_cast<char *>(p);
        cm->opened[r] = true;
    }
    // every rank wrote its nonce and its device's identity before it published its handle:
    // combine the nonces into the first sequence number (the same on every rank), and note
    // whether a peer sits on this rank's own GPU
    unsigned seed = 0;
    unsigned long long ids[COMM_MAX_RANKS];
    for (int r = 0; r < cm->dev.world; r++) {
        unsigned v = 0;
        unsigned long long id = 0;
        hipError_t e = hipMemcpy(&v, cm->dev.peer[r] + cm->dev.status_off + 8, sizeof(v), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&id, cm->dev.peer[r] + cm->dev.status_off + 16, sizeof(id), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return (int)e;
        seed = seed * 2654435761u + v;
        if (r != cm->dev.rank && id == cm->identity) cm->shared_device = true;
        ids[r] = id;
        for (int q = 0; q < r; q++)
            if (ids[q] == id) cm->any_shared = true;
    }
    cm->step = seed | 1u;
    return 0;
}

int ccnn_comm_connect_local(ccnn_comm_t *cm, ccnn_comm_t *const *peers)
{
    if (!cm || !peers) return CCNN_E_INVALID;
    for (int r = 0; r < cm->dev.world; r++) {
        if (r == cm->dev.rank) continue;
        if (!peers[r] || peers[r]->dev.rank != r || peers[r]->dev.world != cm->dev.world ||
            peers[r]->total_bytes != cm->total_bytes)
            return CCNN_E_INVALID;
        if (peers[r]->device != cm->device) {
            hipError_t e = hipDeviceEnablePeerAccess(peers[r]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return (int)e;
            (void)hipGetLastError();
        }
        cm->dev.peer[r] = peers[r]->dev.peer[r];
        if (peers[r]->identity == cm->identity) cm->shared_device = true;
    }
    for (int r = 0; r < cm->dev.world; r++)
        for (int q = 0; q < r; q++)
            if ((r == cm->dev.rank ? cm : peers[r])->identity == (q == cm->dev.rank ? cm : peers[q])->identity) cm->any_shared = true;
    unsigned seed = 0;
    for (int r = 0; r < cm->dev.world; r++) seed = seed * 2654435761u + (r == cm->dev.rank ? cm->nonce : peers[r]->nonce);
    cm->step = seed | 1u;
    return 0;
}

int ccnn_comm_shares_device(ccnn_comm_t *cm, int *shared)
{
    if (!cm || !shared) return CCNN_E_INVALID;
    *shared = cm->shared_device ? 1 : 0;
    return 0;
}

int ccnn_comm_set_exchange(ccnn_comm_t *cm, int via_owner)
{
    if (!cm || (via_owner != 0 && via_owner != 1)) return CCNN_E_INVALID;
    cm->dev.via_owner = via_owner;
    return 0;
}

int ccnn_comm_status(ccnn_comm_t *cm, int *status)
{
    if (!cm || !status) return CCNN_E_INVALID;
    unsigned v = 0;
    hipError_t e = hipMemcpy(&v, cm->dev.peer[cm->dev.rank] + cm->dev.status_off, sizeof(v), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    *status = (int)v;
    return 0;
}

int ccnn_comm_disconnect(ccnn_comm_t *cm)
{
    if (!cm) return CCNN_E_INVALID;
    int first_error = 0;
    for (int r = 0; r < cm->dev.world; r++) {
        if (r == cm->dev.rank) continue;
        if (cm->opened[r]) {
            const hipError_t e = hipIpcCloseMemHandle(cm->dev.peer[r]);
            log_handle("close", cm->dev.rank, r, cm->dev.peer[r], nullptr, (int)e);
            if (e != hipSuccess && !first_error) first_error = (int)e;   // a mapping that stays open must not go unnoticed
        }
        cm->opened[r] = false;
        cm->dev.peer[r] = nullptr;
    }
    return first_error;
}

int ccnn_comm_destroy(ccnn_comm_t *cm)
{
    if (!cm) return 0;
    const int rc = ccnn_comm_disconnect(cm);
    if (cm->dev.peer[cm->dev.rank]) {
        log_handle("keep", cm->dev.rank, cm->dev.rank, cm->dev.peer[cm->dev.rank], nullptr, 0);
        pool_release(cm->dev.peer[cm->dev.rank]);   // kept for the next comm of this process, never freed
    }
    delete cm;
    return rc;
}

int ccnn_allreduce_f32(ccnn_comm_t *cm, float *buf, long n, void *stream)
{
    return allreduce_launch<float>(cm, buf, buf, n, RulePlain(), stream);
}

int ccnn_allreduce_f64(ccnn_comm_t *cm, double *buf, long n, void *stream)
{
    return allreduce_launch<double>(cm, buf, buf, n, RulePlain(), stream);
}

int ccnn_allreduce_update_f32(ccnn_comm_t *cm, float *params, const float *grad, float lr, int z,
                              int freeze_conv_bias, void *stream)
{
    ccnn_dims_t d;
    if (make_dims(z, &d)) return CCNN_E_INVALID;
    RuleSgd<float> rule{lr, 1.0f, freeze_conv_bias ? (long)d.off_b0 : 0L, freeze_conv_bias ? (long)d.off_w1 : 0L};
    return allreduce_launch<float>(cm, params, grad, d.nparams, rule, stream);
}

int ccnn_allreduce_update_f64(ccnn_comm_t *cm, double *params, const double *grad, double lr, int divisor, long n,
                              void *stream)
{
    if (divisor <= 0) return CCNN_E_INVALID;
    RuleSgd<double> rule{lr, (double)divisor, 0L, 0L};
    return allreduce_launch<double>(cm, params, grad, n, rule, stream);
}

}  // extern "C"
