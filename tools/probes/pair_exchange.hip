// Probe: what it costs two workgroups on one XCD to trade a row of 100 (or 200) tagged 8-byte words
// inside a launch - the exchange a forward pass split over the two workgroups of a sample would need
// for its FC1 partial sums.  Pairs are blocks i and i + 8 (consecutive blocks go round the eight
// XCDs, so a pair shares an L2), as sample_fwd_bwd pairs them.
//
// Every round: thread n < words stores {payload, round tag} to its workgroup's outbox with a relaxed
// agent-scope atomic store and polls the partner's outbox word n until it carries the round's tag;
// a workgroup barrier ends the round.  Outboxes are double-buffered by round parity (a workgroup
// reaches round r + 2 only after its partner has published r + 1, i.e. finished reading r).  `work`
// dependent FMAs per thread between rounds stand for the phases between two exchanges; the host
// reports the launch time per round with and without the exchange.  Waits are bounded (50 ms).
//   hipcc --offload-arch=gfx950 -O3 pair_exchange.hip -o pair_exchange
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(512) void pair_loop(unsigned long long *box, int words, int rounds, int work, int exchange,
                                                 unsigned *abort_flag, unsigned *bad, float *sink)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    const int half = (b >> 3) & 1, row = (b & 7) + 8 * (b >> 4);
    unsigned long long *mine = box + ((size_t)row * 2 + half) * 2 * words;
    const unsigned long long *theirs = box + ((size_t)row * 2 + (half ^ 1)) * 2 * words;
    float x = (float)tid;
    unsigned wrong = 0;
    for (int r = 1; r <= rounds; r++) {
        for (int i = 0; i < work; i++) x = fmaf(x, 1.0000001f, 0.5f);
        if (exchange && tid < words) {
            const int par = r & 1;
            const unsigned payload = (unsigned)(r * 131 + row * 7 + half * 3 + tid);
            __hip_atomic_store(mine + par * words + tid, ((unsigned long long)payload << 32) | (unsigned)r, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall();
            unsigned long long v;
            while ((unsigned)(v = __hip_atomic_load(theirs + par * words + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != (unsigned)r) {
                if (wall() - t0 > 5000000ull || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if ((unsigned)(v >> 32) != (unsigned)(r * 131 + row * 7 + (half ^ 1) * 3 + tid)) wrong++;
            x += (float)(v >> 32) * 1e-9f;
        }
        __syncthreads();
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (wrong) atomicAdd(bad, wrong);
    sink[(size_t)b * 512 + tid] = x;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    unsigned long long *box;
    unsigned *flags;
    float *sink;
    const int nwg = 256;
    CK(hipMalloc(&box, sizeof(unsigned long long) * (nwg / 2) * 2 * 2 * 256));
    CK(hipMalloc(&flags, 8));
    CK(hipMalloc(&sink, sizeof(float) * nwg * 512));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int words : {100, 200}) {
        for (int work : {0, 2000}) {
            float us[2] = {0, 0};
            unsigned h[2] = {0, 0};
            for (int exchange = 0; exchange < 2; exchange++) {
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(box, 0, sizeof(unsigned long long) * (nwg / 2) * 2 * 2 * 256));
                    CK(hipMemset(flags, 0, 8));
                    CK(hipEventRecord(e0));
                    pair_loop<<<nwg, 512>>>(box, words, rounds, work, exchange, flags, flags + 1, sink);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    us[exchange] = ms * 1e3f / rounds;
                }
                CK(hipMemcpy(h, flags, 8, hipMemcpyDeviceToHost));
                if (h[0] || h[1]) printf("words %d work %d exchange %d: abort %u wrong %u\n", words, work, exchange, h[0], h[1]);
            }
            printf("%d words, %d FMAs between rounds: %.2f us per round without, %.2f us with the exchange (+%.2f us)\n", words, work,
                   us[0], us[1], us[1] - us[0]);
        }
    }
    return 0;
}
