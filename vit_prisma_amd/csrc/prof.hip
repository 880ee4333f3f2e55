#include "prof.hpp"

#include <vector>

namespace {
struct Pool {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
    double flops = 0, bytes = 0;
};
Pool g_pool[PV_PROF__COUNT];
bool g_on = false;
uint32_t g_mask = 0xffffffffu;
constexpr size_t kMaxEvents = 16384;
}  // namespace

bool pv_prof_on() { return g_on; }
bool pv_prof_on(int kind) { return g_on && ((g_mask >> kind) & 1u); }

int pv_prof_begin(int kind, hipStream_t stream, double flops, double bytes) {
    Pool& p = g_pool[kind];
    if (p.used >= kMaxEvents) return -1;
    if (p.used >= p.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        p.start.push_back(a);
        p.stop.push_back(b);
    }
    const int tok = (int)p.used++;
    p.flops += flops;
    p.bytes += bytes;
    (void)hipEventRecord(p.start[tok], stream);
    return tok;
}

bool pv_prof_events(int kind, double flops, double bytes, hipEvent_t* start, hipEvent_t* stop) {
    if (!pv_prof_on(kind)) return false;
    Pool& p = g_pool[kind];
    if (p.used >= kMaxEvents) return false;
    if (p.used >= p.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
        p.start.push_back(a);
        p.stop.push_back(b);
    }
    const size_t tok = p.used++;
    p.flops += flops;
    p.bytes += bytes;
    *start = p.start[tok];
    *stop = p.stop[tok];
    return true;
}

void pv_prof_end(int kind, int token, hipStream_t stream) { (void)hipEventRecord(g_pool[kind].stop[token], stream); }

extern "C" int pv_prof_enable(int32_t on) {
    g_on = (on & 1) != 0;
    g_mask = (on >> 8) ? (uint32_t)(on >> 8) : 0xffffffffu;      // bits 8.. = kernel families to time (0 = all)
    return PV_OK;
}

extern "C" int pv_prof_reset(void) {
    for (auto& p : g_pool) {
        p.used = 0;
        p.flops = 0;
        p.bytes = 0;
    }
    return PV_OK;
}

// Synchronises the recorded events of `kind` and returns launches, summed kernel time (ms) and the
// summed algorithmic flops / bytes the launchers declared.
extern "C" int pv_prof_read(int32_t kind, int64_t* launches, double* total_ms, double* flops, double* bytes) {
    PV_REQUIRE(kind >= 0 && kind < PV_PROF__COUNT, "profile kind");
    Pool& p = g_pool[kind];
    double ms = 0;
    for (size_t i = 0; i < p.used; ++i) {
        PV_HIP_CHECK(hipEventSynchronize(p.stop[i]));
        float t = 0;
        PV_HIP_CHECK(hipEventElapsedTime(&t, p.start[i], p.stop[i]));
        ms += t;
    }
    if (launches) *launches = (int64_t)p.used;
    if (total_ms) *total_ms = ms;
    if (flops) *flops = p.flops;
    if (bytes) *bytes = p.bytes;
    return PV_OK;
}
