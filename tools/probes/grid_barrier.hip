// Probe: what a grid-wide barrier costs INSIDE one launch on this part when no cache is written
// back or invalidated by hand - the hand-over data is stored write-through and read past the
// caches (agent-scope relaxed atomics: sc1 stores / loads on gfx950), so the barrier itself is one
// counter and a poll.  Decides whether the two launches of a training step (per-sample chain,
// batch sums + update) could become phases of one persistent launch.
//
// Every workgroup: [optionally store ROW bytes write-through] -> arrive (atomic add, one lane) ->
// poll the counter until all have arrived -> [optionally read ROW bytes that a workgroup of ANOTHER
// XCD stored, past the caches, and check them].  Repeated `iters` times inside one launch; the
// host times the launch (HIP events) and workgroup 0 stamps the 100 MHz clock every iteration.
// Waits are bounded: a workgroup that waits longer than ~50 ms raises a flag and everybody leaves.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>

#include <algorithm>
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

// mode 0: barrier only.  mode 1: + hand-over through write-through stores and cache-bypassing loads
// (words floats per thread).  mode 2: hand-over through ordinary stores and loads with a release
// fence before the arrive and an acquire fence after the wait (what __threadfence-based code pays:
// the L2 is written back and invalidated).
__global__ __launch_bounds__(512) void barrier_loop(unsigned *counter, unsigned *abort_flag, float *rows, int words,
                                                    int iters, int mode, unsigned long long *stamps, unsigned *bad)
{
    const int tid = threadIdx.x, nt = blockDim.x, nwg = gridDim.x, wg = blockIdx.x;
    // a partner on another XCD (consecutive workgroup ids go round the eight XCDs)
    const int partner = (wg + 8 * 5 + 3) % nwg;
    float *mine = rows + (size_t)wg * nt * words, *theirs = rows + (size_t)partner * nt * words;
    unsigned wrong = 0;
    for (int it = 0; it < iters; it++) {
        if (wg == 0 && tid == 0) stamps[it] = wall();
        if (mode == 1) {
            for (int w = 0; w < words; w++)
                __hip_atomic_store(mine + w * nt + tid, (float)(it * 1000 + wg + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (mode == 2) {
            for (int w = 0; w < words; w++) mine[w * nt + tid] = (float)(it * 1000 + wg + w);
            __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default in HIP device code: L2 write-back
        }
        __syncthreads();   // (mode 1: the write-through stores of the whole workgroup have been issued ...
        if (tid == 0) {
            if (mode == 1) __builtin_amdgcn_s_waitcnt(0);   // ... and acknowledged before the arrive)
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * nwg;
            const unsigned long long t0 = wall();
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (wall() - t0 > 5000000ull) {   // 50 ms
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        if (mode == 1) {
            for (int w = 0; w < words; w++) {
                const float v = __hip_atomic_load(theirs + w * nt + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wrong += v != (float)(it * 1000 + partner + w);
            }
        } else if (mode == 2) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            for (int w = 0; w < words; w++) wrong += theirs[w * nt + tid] != (float)(it * 1000 + partner + w);
        }
        // (the rows are rewritten next iteration: a second barrier would be needed to protect readers
        // that are still reading; the values carry the iteration so a torn hand-over shows as `wrong`.
        // A real two-phase step has two barriers per step, which is what 2 x this cost stands for.)
        if (mode) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counter + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)(it + 1) * nwg;
                const unsigned long long t0 = wall();
                while (__hip_atomic_load(counter + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    if (wall() - t0 > 5000000ull) {
                        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        }
    }
    if (wg == 0 && tid == 0) stamps[iters] = wall();
    if (wrong) atomicAdd(bad, wrong);
}

// Two-level barrier: groups of G workgroups (workgroup w belongs to group w % (nwg / G), so a group
// stays on one XCD when nwg / G is a multiple of 8) arrive at their own counter, 256 bytes apart;
// the last arriver of a group arrives at the top counter.  flags = 0: everybody polls the top
// counter; flags = 1: the last arriver at the top writes the generation into one release flag per
// group and a group polls its own flag.
__global__ __launch_bounds__(512) void barrier_tree(unsigned *mem, int G, int flags, int iters, int sleep, unsigned long long *stamps)
{
    const int tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const int ng = nwg / G, g = wg % ng;
    unsigned *abort_flag = mem, *top = mem + 64, *grp = mem + 128 + 64 * g, *rel = mem + 128 + 64 * (ng + g);
    for (int it = 0; it < iters; it++) {
        if (wg == 0 && tid == 0) stamps[it] = wall();
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1) % G == 0) {
                const unsigned t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (flags && t + 1 == (unsigned)(it + 1) * ng)
                    for (int k = 0; k < ng; k++)
                        __hip_atomic_store(mem + 128 + 64 * (ng + k), (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned *watch = flags ? rel : top;
            const unsigned want = flags ? (unsigned)(it + 1) : (unsigned)(it + 1) * ng;
            const unsigned long long t0 = wall();
            while (__hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (wall() - t0 > 5000000ull) {
                    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (sleep) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
    if (wg == 0 && tid == 0) stamps[iters] = wall();
}

int main()
{
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned *counter, *abort_flag, *bad;
    unsigned long long *stamps;
    float *rows;
    const int iters = 200, max_words = 8;
    CK(hipMalloc((void **)&counter, 256));
    CK(hipMalloc((void **)&abort_flag, 4));
    CK(hipMalloc((void **)&bad, 4));
    CK(hipMalloc((void **)&stamps, sizeof(unsigned long long) * (iters + 1)));
    CK(hipMalloc((void **)&rows, sizeof(float) * 512 * max_words * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%d compute units; one launch = %d iterations; times per iteration (mode 0: one barrier; modes 1, 2: hand-over + two barriers)\n",
           cus, iters);
    for (int mode : {0, 1, 2}) {
        for (int nwg : {cus / 2, cus, 2 * cus}) {
            for (int nt : {64, 512}) {
                for (int words : {1, 8}) {
                    if (mode == 0 && words != 1) continue;
                    int occ = 0;
                    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, barrier_loop, nt, 0));
                    if ((long)occ * cus < nwg) continue;   // every workgroup must be resident: they wait for each other
                    for (int rep = 0; rep < 2; rep++) {    // the second launch is the measurement
                        CK(hipMemset(counter, 0, 256));
                        CK(hipMemset(abort_flag, 0, 4));
                        CK(hipMemset(bad, 0, 4));
                        CK(hipEventRecord(e0));
                        hipLaunchKernelGGL(barrier_loop, dim3(nwg), dim3(nt), 0, 0, counter, abort_flag, rows, words, iters, mode,
                                           stamps, bad);
                        CK(hipGetLastError());
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                    }
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned ab = 0, wrong = 0;
                    std::vector<unsigned long long> st(iters + 1);
                    CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(&wrong, bad, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * (iters + 1), hipMemcpyDeviceToHost));
                    std::vector<double> per;
                    for (int i = 20; i < iters; i++) per.push_back((st[i + 1] - st[i]) * 0.01);
                    std::sort(per.begin(), per.end());
                    printf("mode %d  %4d workgroups x %3d threads, %5d B per workgroup: launch %.1f us = %.2f us/iter; in-kernel median %.2f "
                           "p90 %.2f us%s%s\n",
                           mode, nwg, nt, mode ? words * nt * 4 : 0, ms * 1e3, ms * 1e3 / iters, per[per.size() / 2],
                           per[per.size() * 9 / 10], ab ? "  TIMED OUT" : "", wrong ? "  WRONG DATA" : "");
                    if (ab) return 2;
                }
            }
        }
    }
    {
        unsigned *mem;
        CK(hipMalloc((void **)&mem, 256 * 1024));
        for (int nwg : {cus / 2, cus}) {
            for (int G : {4, 8, 16, 32}) {
                for (int flags : {0, 1}) {
                    for (int sleep : {1, 0}) {
                        for (int rep = 0; rep < 2; rep++) {
                            CK(hipMemset(mem, 0, 256 * 1024));
                            CK(hipEventRecord(e0));
                            hipLaunchKernelGGL(barrier_tree, dim3(nwg), dim3(512), 0, 0, mem, G, flags, iters, sleep, stamps);
                            CK(hipGetLastError());
                            CK(hipEventRecord(e1));
                            CK(hipEventSynchronize(e1));
                        }
                        float ms = 0;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        unsigned ab = 0;
                        CK(hipMemcpy(&ab, mem, 4, hipMemcpyDeviceToHost));
                        std::vector<unsigned long long> st(iters + 1);
                        CK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * (iters + 1), hipMemcpyDeviceToHost));
                        std::vector<double> per;
                        for (int i = 20; i < iters; i++) per.push_back((st[i + 1] - st[i]) * 0.01);
                        std::sort(per.begin(), per.end());
                        printf("tree  %4d workgroups x 512 threads, groups of %2d, %s, %s: %.2f us per barrier (launch / iters), in-kernel median %.2f p90 %.2f%s\n",
                               nwg, G, flags ? "release flag per group" : "all poll the top counter", sleep ? "s_sleep between polls" : "tight poll",
                               ms * 1e3 / iters, per[per.size() / 2], per[per.size() * 9 / 10], ab ? "  TIMED OUT" : "");
                        if (ab) return 2;
                    }
                }
            }
        }
    }
    return 0;
}
