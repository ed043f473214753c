// Per-kernel timing for bench.py's roofline figure: every launch made while
// timing is enabled carries a HIP event pair on its own stream;
// ccnn_profile_report() sums the elapsed times per kernel name.
#include "profile.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ccnn.h"

namespace ccnn {

bool g_profile_on = false;

namespace {
struct Record {
    const char *name;
    hipEvent_t start, stop;
};
std::mutex g_mu;
std::vector<Record> g_records;
std::vector<hipEvent_t> g_free;

hipEvent_t get_event()
{
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void profile_events(hipEvent_t *start, hipEvent_t *stop)
{
    std::lock_guard<std::mutex> lk(g_mu);
    *start = get_event();
    *stop = get_event();
}

void profile_push(const char *name, hipEvent_t start, hipEvent_t stop)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_records.push_back({name, start, stop});
}

}  // namespace ccnn

using namespace ccnn;

extern "C" {

int ccnn_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_profile_on = on != 0;
    return 0;
}

int ccnn_profile_report(int index, char *name, int name_len, double *total_ms, long *launches)
{
    static std::vector<std::pair<std::string, std::pair<double, long>>> table;
    std::lock_guard<std::mutex> lk(g_mu);
    if (index == 0) {
        // fold all pending records into the table (waits for their events)
        std::map<std::string, std::pair<double, long>> acc;
        for (auto &r : g_records) {
            float ms = 0;
            if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
                auto &a = acc[r.name];
                a.first += ms;
                a.second += 1;
            }
            g_free.push_back(r.start);
            g_free.push_back(r.stop);
        }
        g_records.clear();
        table.assign(acc.begin(), acc.end());
    }
    if (index < 0 || index >= (int)table.size() || !name || name_len <= 0 || !total_ms || !launches) return 0;
    std::strncpy(name, table[index].first.c_str(), name_len - 1);
    name[name_len - 1] = 0;
    *total_ms = table[index].second.first;
    *launches = table[index].second.second;
    return 1;
}

}  // extern "C"
