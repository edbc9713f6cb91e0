// pm_common.hpp — context, error plumbing, scratch arena and per-kernel event timing shared by
// the translation units of libpm_hip.so.  gfx950 only; no CUDA-compat layer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pm.h"

namespace pm {

void set_error(const char* fmt, ...);

#define PM_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::pm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
            return PM_E_HIP;                                                            \
        }                                                                               \
    } while (0)

#define PM_REQUIRE(cond, status, msg)                                                   \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::pm::set_error("%s: %s", __func__, msg);                                   \
            return (status);                                                            \
        }                                                                               \
    } while (0)

struct KernelTimer {
    double total_ms = 0.0;
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace pm

struct pm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // grow-only device scratch; carved by a bump pointer that every API call resets
    char* arena = nullptr;
    size_t arena_cap = 0;
    size_t arena_off = 0;
    // pinned host staging for small results
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // timing
    bool timing = false;
    std::map<std::string, pm::KernelTimer> timers;
    std::vector<hipEvent_t> event_pool;
    int n_cu = 256;
    // kNN side-band: [0..1] epoch-tagged 64-bit maxima (never cleared), diag words (re-scan count,
    // non-finite flag) filled only while knn_diag is on
    unsigned long long* knn_stats = nullptr;
    unsigned* knn_diag_words = nullptr;
    unsigned knn_epoch = 0;
    bool knn_diag = false;
    // stable compaction: epoch-tagged per-block survivor counts
    unsigned* fg_counts = nullptr;
    unsigned fg_epoch = 0;
    // debugging aid: last candidate buffer of the matcher (device pointer inside the arena)
    const void* dbg_ptr = nullptr;
    size_t dbg_bytes = 0;
};

namespace pm {

int arena_reserve(pm_ctx* ctx, size_t bytes);          // ensure capacity (may sync + realloc)
void arena_reset(pm_ctx* ctx);
void* arena_take(pm_ctx* ctx, size_t bytes);            // 256-B aligned carve; nullptr if over cap
int pinned_reserve(pm_ctx* ctx, size_t bytes);

// RAII event bracket: records start/stop on ctx->stream when timing is on.
struct ScopedKernelTime {
    pm_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    ScopedKernelTime(pm_ctx* c, const char* n);
    ~ScopedKernelTime();
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace pm
