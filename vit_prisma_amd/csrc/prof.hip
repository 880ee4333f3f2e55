#include "prof.hpp"

#include <vector>

namespace {
struct Pool {
    std::vector<hipEvent_t> start, stop;
    std::vector<double> ev_flops, ev_bytes;      // per event: what its launcher declared
    std::vector<int> ev_tag;                     // per event: the instance tag current at its launch (pv_prof_set_tag)
    size_t used = 0;
    double flops = 0, bytes = 0;
    void note(size_t tok, double fl, double by, int tag) {
        if (ev_flops.size() <= tok) { ev_flops.resize(tok + 1); ev_bytes.resize(tok + 1); ev_tag.resize(tok + 1); }
        ev_flops[tok] = fl; ev_bytes[tok] = by; ev_tag[tok] = tag;
    }
};
Pool g_pool[PV_PROF__COUNT];
bool g_on = false;
uint32_t g_mask = 0xffffffffu;
constexpr size_t kMaxEvents = 16384;
int g_tag = 0;
}  // namespace

void pv_prof_set_tag(int tag) { g_tag = tag; }

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
    p.note((size_t)tok, flops, bytes, g_tag);
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
    p.note(tok, flops, bytes, g_tag);
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

// The same for the launches of `kind` that carried instance tag `tag` (GEMMs: 1 QKV, 2 O-projection, 3 MLP-1, 4 MLP-2, 0 others --
// pv_launch_gemm tags its launch by epilogue and shape): the per-instance roofline fractions of bench.py.
extern "C" int pv_prof_read_tag(int32_t kind, int32_t tag, int64_t* launches, double* total_ms, double* flops, double* bytes) {
    PV_REQUIRE(kind >= 0 && kind < PV_PROF__COUNT, "profile kind");
    Pool& p = g_pool[kind];
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    for (size_t i = 0; i < p.used; ++i) {
        if (p.ev_tag[i] != tag) continue;
        PV_HIP_CHECK(hipEventSynchronize(p.stop[i]));
        float t = 0;
        PV_HIP_CHECK(hipEventElapsedTime(&t, p.start[i], p.stop[i]));
        ms += t; fl += p.ev_flops[i]; by += p.ev_bytes[i]; ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return PV_OK;
}

// ---------------------------------------------------------------------------------------------------
// kernel-choice overrides (tests, A/B measurements).  Process-global, never read from the environment.
// ---------------------------------------------------------------------------------------------------
#include <string.h>

PvTuning g_pv_tuning;

namespace {
int* tuning_field(const char* key) {
    if (!key) return nullptr;
    if (!strcmp(key, "gemm_tile")) return &g_pv_tuning.gemm_tile;
    if (!strcmp(key, "gemm_v1")) return &g_pv_tuning.gemm_v1;
    if (!strcmp(key, "gemm_v1patch")) return &g_pv_tuning.gemm_v1patch;
    if (!strcmp(key, "attn_wg")) return &g_pv_tuning.attn_wg;
    if (!strcmp(key, "attn_direct")) return &g_pv_tuning.attn_direct;
    if (!strcmp(key, "prof_markers")) return &g_pv_tuning.prof_markers;
    if (!strcmp(key, "sae_exact")) return &g_pv_tuning.sae_exact;
    if (!strcmp(key, "sae_fold")) return &g_pv_tuning.sae_fold;
    if (!strcmp(key, "enc_rounds")) return &g_pv_tuning.enc_rounds;
    if (!strcmp(key, "enc_tm256")) return &g_pv_tuning.enc_tm256;
    if (!strcmp(key, "sae_inline_fb")) return &g_pv_tuning.sae_inline_fb;
    if (!strcmp(key, "dense_group")) return &g_pv_tuning.dense_group;
    if (!strcmp(key, "gemm_dbg")) return &g_pv_tuning.gemm_dbg;
    if (!strcmp(key, "gemm_loop")) return &g_pv_tuning.gemm_loop;
    if (!strcmp(key, "gemm_persist")) return &g_pv_tuning.gemm_persist;
    if (!strcmp(key, "gemm_stagger")) return &g_pv_tuning.gemm_stagger;
    if (!strcmp(key, "gemm_cus")) return &g_pv_tuning.gemm_cus;
    if (!strcmp(key, "dense_fp32")) return &g_pv_tuning.dense_fp32;
    return nullptr;
}
}  // namespace

extern "C" int pv_debug_set_tuning(const char* key, int32_t value) {
    if (key && !strcmp(key, "reset")) {
        g_pv_tuning = PvTuning();
        return PV_OK;
    }
    int* f = tuning_field(key);
    PV_REQUIRE(f != nullptr, "unknown tuning key");
#ifndef PV_TUNING
    PV_REQUIRE(f != &g_pv_tuning.gemm_dbg || value == 0, "gemm_dbg ablations exist only in -DPV_TUNING builds");
#endif
    *f = value;
    return PV_OK;
}

// value of one key; key "any" = 1 when any field differs from its default (what bench.py asserts to be 0)
extern "C" int pv_debug_get_tuning(const char* key, int32_t* value) {
    PV_REQUIRE(key && value, "null argument");
    if (!strcmp(key, "any")) {
        const PvTuning d;
        const PvTuning& t = g_pv_tuning;
        *value = (t.gemm_tile != d.gemm_tile || t.gemm_v1 != d.gemm_v1 || t.gemm_v1patch != d.gemm_v1patch || t.attn_wg != d.attn_wg || t.attn_direct != d.attn_direct ||
                  t.prof_markers != d.prof_markers || t.sae_exact != d.sae_exact || t.sae_fold != d.sae_fold || t.enc_rounds != d.enc_rounds || t.enc_tm256 != d.enc_tm256 || t.sae_inline_fb != d.sae_inline_fb || t.dense_group != d.dense_group || t.gemm_dbg != d.gemm_dbg ||
                  t.gemm_loop != d.gemm_loop || t.gemm_persist != d.gemm_persist || t.gemm_stagger != d.gemm_stagger || t.gemm_cus != d.gemm_cus || t.dense_fp32 != d.dense_fp32) ? 1 : 0;
        return PV_OK;
    }
    const int* f = tuning_field(key);
    PV_REQUIRE(f != nullptr, "unknown tuning key");
    *value = *f;
    return PV_OK;
}
