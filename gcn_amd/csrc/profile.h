// Optional per-kernel timing with HIP events (measurement harness only).
#pragma once
#include <hip/hip_runtime.h>

namespace ccnn {

extern bool g_profile_on;
void profile_begin(const char *name, hipStream_t st);
void profile_end(hipStream_t st);

// Brackets one kernel launch with a start/stop event pair when timing is enabled.
struct ProfileScope {
    hipStream_t st;
    bool on;
    ProfileScope(const char *name, hipStream_t s) : st(s), on(g_profile_on)
    {
        if (on) profile_begin(name, st);
    }
    ~ProfileScope()
    {
        if (on) profile_end(st);
    }
};

}  // namespace ccnn
