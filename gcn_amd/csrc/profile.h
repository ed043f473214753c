// Kernel launch helper with optional per-kernel timing (measurement harness only).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace ccnn {

extern bool g_profile_on;
// A fresh (or recycled) event pair; profile_push files it under `name` for ccnn_profile_report.
void profile_events(hipEvent_t *start, hipEvent_t *stop);
void profile_push(const char *name, hipEvent_t start, hipEvent_t stop);

// Launches kernel<<<grid, block, smem, st>>>(args...).  While timing is enabled the launch
// carries a start/stop event pair attached to the dispatch itself (hipExtLaunchKernelGGL), so
// the elapsed time is the kernel's own begin-to-end interval, the same interval a kernel
// trace reports; events recorded around the launch would add the dispatch latency.
template <typename... KArgs, typename... Args>
inline void launch(const char *name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t st,
                   Args... args)
{
    if (!g_profile_on) {
        hipLaunchKernelGGL(kernel, grid, block, smem, st, static_cast<KArgs>(args)...);
        return;
    }
    hipEvent_t e0, e1;
    profile_events(&e0, &e1);
    hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)smem, st, e0, e1, 0, static_cast<KArgs>(args)...);
    profile_push(name, e0, e1);
}

}  // namespace ccnn
