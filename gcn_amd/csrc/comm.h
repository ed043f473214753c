// Node-local exchange buffers shared by the stand-alone all-reduce (comm.hip) and the
// minibatch step that exchanges inside its reduction kernel (mbgd.hip).
//
// Every rank owns one uncached buffer, mapped by all peers:
//   data   [2 parities][world ranks][slot_bytes]   slot [p][r] = rank r's contribution of a call
//   flags  [2 parities][world ranks][nflags] u32   flag [p][r][i] = sequence number of the call whose
//                                                  part i (a chunk / a workgroup's outputs) rank r has
//                                                  finished writing into this buffer
//   status u32                                     != 0: some workgroup gave up waiting
// A call alternates the parity of the previous one.  A rank can only start call k+2 after every
// peer raised its flags of call k+1, which each peer did after its kernel of call k (the last
// reader of that parity) had finished, so two parities suffice.
#pragma once

#include "ccnn_common.h"

namespace ccnn {

constexpr int COMM_MAX_RANKS = 16;
constexpr int COMM_CHUNK_BYTES = 4096;      // granularity of the stand-alone all-reduce (256 threads x 16 B)
constexpr unsigned long long COMM_SPIN_TICKS = 1000000000ull;   // default wait limit: 10 s of the 100 MHz wall clock
constexpr int COMM_MIN_FLAGS = 1024;        // room for one flag per workgroup of the fused minibatch step

struct CommDev {            // passed to kernels by value; world == 0 means "no exchange"
    char *peer[COMM_MAX_RANKS];   // every rank's exchange buffer as mapped here ([rank] = the local one)
    int world, rank;
    long slot_bytes;              // bytes of one [parity][rank] slot
    long flags_off, status_off;   // byte offsets inside a buffer (the word after status holds the rank's nonce)
    int nflags;
    unsigned long long spin_ticks;   // how long a workgroup waits for a peer (100 MHz ticks)
    int via_owner;                // in-step exchange: 0 = every rank writes to every rank; 1 = reduce at an owner, who sends the sum round
};

#ifdef __HIPCC__
__device__ __forceinline__ char *comm_slot(const CommDev &c, int buffer_of, int par, int from)
{
    return c.peer[buffer_of] + ((long)par * c.world + from) * c.slot_bytes;
}

// Payload accesses: relaxed SYSTEM-scope atomics (sc0 sc1: performed at the system coherence
// point, never served from or parked in a local cache).
template <typename T>
__device__ __forceinline__ void comm_store(T *dst, T v)
{
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename T>
__device__ __forceinline__ T comm_load(const T *src)
{
    return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// 16 bytes (dst 16-byte aligned) as two 8-byte accesses
__device__ __forceinline__ void comm_store16(void *dst, const void *src)
{
    const unsigned long long *s = static_cast<const unsigned long long *>(src);
    comm_store(static_cast<unsigned long long *>(dst), s[0]);
    comm_store(static_cast<unsigned long long *>(dst) + 1, s[1]);
}
__device__ __forceinline__ void comm_load16(void *dst, const void *src)
{
    unsigned long long *d = static_cast<unsigned long long *>(dst);
    d[0] = comm_load(static_cast<const unsigned long long *>(src));
    d[1] = comm_load(static_cast<const unsigned long long *>(src) + 1);
}

// Flag-in-word transfer: a 32-bit payload and the call's sequence number travel in ONE 8-byte
// word, so a reader knows a word is current from the word itself: no fence, no separate flag,
// no ordering between stores.  Costs twice the bytes; right for latency-bound messages.
__device__ __forceinline__ unsigned long long comm_tag(unsigned bits, unsigned step)
{
    return (unsigned long long)bits | ((unsigned long long)step << 32);
}
// Stores n (<= 4) tagged words to dst (16-byte aligned), two per 16-byte access where possible.
__device__ __forceinline__ void comm_store_tagged(unsigned long long *dst, const unsigned *bits, int n, unsigned step)
{
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < n) comm_store(dst + j, comm_tag(bits[j], step));
}
// Polls n (<= 4) tagged words until all carry `step`; false if the wait limit ran out.
__device__ __forceinline__ bool comm_load_tagged(const unsigned long long *src, unsigned *bits, int n, unsigned step,
                                                 const CommDev &c)
{
    unsigned long long w[4] = {0, 0, 0, 0};
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (true) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < n) {
                w[j] = comm_load(src + j);
                all = all && (unsigned)(w[j] >> 32) == step;
            }
        }
        if (all) break;
        if ((++spins & 1023) == 0) {
            unsigned *status = reinterpret_cast<unsigned *>(c.peer[c.rank] + c.status_off);
            if (wall_clock64() - t0 > c.spin_ticks || comm_load(status)) {
                comm_store(status, 1u);
                return false;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < n) bits[j] = (unsigned)w[j];
    return true;
}

// After a workgroup has written its part into slot [par][rank] of every peer: make the writes
// visible system-wide, then raise flag [par][rank][part] at every peer.  All threads call it.
// The fence is what the memory model offers for "payload before flag" (a wait for the stores'
// acknowledgements alone is not a documented guarantee).  It writes back the whole L2, so it is
// cheap in a kernel that follows small ones and costs >10 us right behind a kernel that left
// megabytes of dirty lines; kernels in that position use the flag-in-word transfer instead.
__device__ __forceinline__ void comm_publish(const CommDev &c, int par, unsigned step, int part, int tid)
{
    __threadfence_system();
    __syncthreads();
    if (tid < c.world) {
        unsigned *flag = reinterpret_cast<unsigned *>(c.peer[tid] + c.flags_off) +
                         ((long)par * c.world + c.rank) * c.nflags + part;
        comm_store(flag, step);
    }
}

// Waits until every rank's flag [par][r][part] in the local buffer shows `step`.  All threads
// call it; afterwards the slots may be read with comm_load.  Returns false -- to every thread of
// the workgroup alike -- when a wait ran out (status is then set): the slots hold whatever was
// there before and must not be used.
__device__ __forceinline__ bool comm_wait(const CommDev &c, int par, unsigned step, int part, int tid)
{
    char *mine = c.peer[c.rank];
    int gave_up = 0;
    if (tid < c.world) {
        unsigned *status = reinterpret_cast<unsigned *>(mine + c.status_off);
        const unsigned *flag = reinterpret_cast<const unsigned *>(mine + c.flags_off) +
                               ((long)par * c.world + tid) * c.nflags + part;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (comm_load(flag) != step) {
            // all ranks run the same call, so the wait ends; the limit bounds the damage if a
            // peer process died
            if ((++spins & 1023) == 0 &&
                (wall_clock64() - t0 > c.spin_ticks ||
                 __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                gave_up = 1;
                break;
            }
        }
    }
    const bool ok = !__syncthreads_or(gave_up);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: nothing below is served from a stale cache
    return ok;
}
#endif

}  // namespace ccnn

struct ccnn_comm {
    ccnn::CommDev dev;
    int device;
    long max_elems;
    int elem;
    size_t total_bytes;   // bytes of the exchange buffer
    unsigned long long identity;   // PCI identity of this rank's GPU
    bool shared_device;   // some peer rank runs on this rank's own GPU (known after connect)
    bool any_shared;      // some two ranks of the group share a GPU: the same answer on every rank
    bool opened[ccnn::COMM_MAX_RANKS];
    unsigned step;
    unsigned nonce;   // drawn at creation; the ranks' nonces together seed the sequence numbers
    // Sequence number and parity of the next call (all ranks issue the same calls in the same
    // order).  Numbering starts at a value derived from every rank's nonce, so the numbers of one
    // set of buffers are not those of the previous set: a later allocation may get the same
    // memory back, and a leftover word that carried a number that is expected again would be
    // taken for current if the clearing of the new buffer were ever reordered after a peer's store.
    unsigned next_step()
    {
        step++;
        if (step == 0) step = 2;   // 0 is the flags' initial value; keep the parity alternating
        return step;
    }
};
