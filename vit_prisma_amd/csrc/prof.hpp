// Opt-in per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
// Disabled by default: launchers pay one predictable branch.
#pragma once
#include "pv_common.hpp"

enum {
    PV_PROF_GEMM = 0,        // MFMA GEMM (all epilogues)
    PV_PROF_ATTN = 1,        // attention core
    PV_PROF_LN = 2,          // layernorm / embed assembly
    PV_PROF_SAE_ENC = 3,     // SAE encoder GEMM + top-k
    PV_PROF_SAE_BWD = 4,     // SAE sparse backward
    PV_PROF_SAE_APPLY = 5,   // SAE clip + project + Adam
    PV_PROF_MISC = 6,
    PV_PROF__COUNT = 7
};

// instance tag attached to the events recorded from now on (pv_prof_read_tag sums one tag's launches); 0 = untagged
void pv_prof_set_tag(int tag);
bool pv_prof_on();
bool pv_prof_on(int kind);      // enabled AND this kernel family selected (pv_prof_enable's mask)
// Records a start event on `stream`; returns a token (< 0 when disabled / pool exhausted).
int pv_prof_begin(int kind, hipStream_t stream, double flops, double bytes);
void pv_prof_end(int kind, int token, hipStream_t stream);

// Event pair for a launch that carries its own timestamps (hipExtLaunchKernelGGL attaches the start / stop events to
// the kernel's dispatch packet: no separate marker packets on the stream, so timing a launch costs no stream time).
// Returns false when this kernel family is not being timed.
bool pv_prof_events(int kind, double flops, double bytes, hipEvent_t* start, hipEvent_t* stop);

struct ProfScope {
    int kind, token;
    hipStream_t stream;
    ProfScope(int k, hipStream_t s, double flops, double bytes) : kind(k), token(-1), stream(s) {
        if (k < PV_PROF__COUNT && pv_prof_on(k)) token = pv_prof_begin(k, s, flops, bytes);     // (k == PV_PROF__COUNT: inert scope)
    }
    ~ProfScope() {
        if (token >= 0) pv_prof_end(kind, token, stream);
    }
};
