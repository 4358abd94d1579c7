// common.h -- context, workspace arena, error plumbing and launch helpers of libimcom_hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/imcom_hip.h"

namespace imcom {

constexpr int NB = 128;  // block size of the blocked factorisation / solve (MFMA tile = NB x NB)
constexpr int MAX_INC = 24;  // diagonal increments per stamp (kappa nodes + repair add/restore pairs)
constexpr int MAX_INC_HOST = MAX_INC;

void set_error(const char *fmt, ...);

#define IMCOM_HIP_CHECK(expr)                                                                  \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            imcom::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                             __LINE__);                                                        \
            return IMCOM_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

#define IMCOM_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            imcom::set_error(__VA_ARGS__); \
            return IMCOM_ERR_ARG;         \
        }                                 \
    } while (0)

#define IMCOM_TRY(expr)             \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != IMCOM_OK) return rc__; \
    } while (0)

struct ProfileSlot {
    double ms = 0.0;
    long launches = 0;
};

struct PendingEvent {
    std::string family;
    hipEvent_t start, stop;
    long launches;
};

}  // namespace imcom

struct imcom_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;  // second queue for work that overlaps the main stream (eigensolver rotations)
    std::vector<hipEvent_t> sync_events;  // plain (no timing) events for cross-stream ordering
    std::vector<hipStream_t> sub_streams;  // streams of the sub-batches of one call (eigen.hip)
    std::vector<hipStream_t> part_streams; // the same with a share of the CUs each (IMCOM_SPLIT_CUS=1; eigen.hip)
    int *flag_pin = nullptr;               // page-locked read-back of per-stamp flags (an async copy into pageable memory would block the host)
    size_t flag_pin_count = 0;
    long deferred_flags = 0;               // imcom_solve_chol_resident_begin: flags on their way to flag_pin (0: no begin outstanding)
    hipEvent_t stream_event = nullptr;    // orders a newly bound stream behind the previous one (imcom_ctx_set_stream)
    // bump-allocated device workspace; reset at the start of every API call that uses it
    char *ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_used = 0;
    size_t ws_limit = 0;       // != 0: ws_take hands out nothing beyond this offset (a sub-batch's share of the workspace, eigen.hip)
    bool ws_external = false;  // imcom_ctx_set_workspace: the caller owns the workspace memory, the library never allocates device memory
    size_t ws_need = 0;        // what the last ws_reserve asked for (imcom_ctx_workspace_needed)
    // Cholesky repair (lakernel.py:262-279): the caller's estimate of max |w[0]| over the stamps of the next calls (0: none;
    // imcom_ctx_set_repair_hint) and what the last call's repaired stamps had (imcom_ctx_last_repair)
    double repair_hint = 0.0;
    bool repair_expect = false;  // imcom_ctx_set_repair_expect: the host-entry Cholesky calls go straight to the repair
    int last_repair_count = 0;
    double last_w0_min = 0.0, last_w0_max = 0.0;
    // pinned host staging for small per-stamp arrays
    char *pin = nullptr;
    size_t pin_bytes = 0;
    size_t pin_used = 0;
    bool profile = false;
    bool profile_fine = false;  // imcom_ctx_profile_enable(ctx, 2): also the per-launch scopes inside long stages (symv4)
    std::map<std::string, imcom::ProfileSlot> prof;
    std::vector<imcom::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    int cu_count = 256;
    // imcom_solve_iter_stats: what the last imcom_solve_iter call did at its last kappa node
    double iter_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // patches, sum up^2 steps, sum steps, sum up^2, largest union, 1 = blocked solver, bytes streamed, 1 = half storage
    std::vector<int> iter_steps;                // CG steps per output pixel [batch][m]
};

namespace imcom {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Reserve `bytes` in the context workspace.  The workspace only grows between calls: ws_reserve()
// is called once per API call with the total, so pointers handed out by ws_take() stay valid.
int ws_reserve(imcom_ctx *ctx, size_t bytes);
void *ws_take(imcom_ctx *ctx, size_t bytes);
int pin_reserve(imcom_ctx *ctx, size_t bytes);
void *pin_take(imcom_ctx *ctx, size_t bytes);

// Small host arrays go through a pinned ring so the async copy never reads caller memory after the
// call returned; the stream is drained when the ring wraps.
template <typename T>
inline int upload(imcom_ctx *ctx, T *dst, const T *src_host, size_t count)
{
    if (count == 0) return IMCOM_OK;
    const size_t bytes = count * sizeof(T);
    if (!ctx->pin) IMCOM_TRY(pin_reserve(ctx, (size_t)4 << 20));
    if (bytes > ctx->pin_bytes / 4) {  // large: plain synchronous copy
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        IMCOM_HIP_CHECK(hipMemcpy(dst, src_host, bytes, hipMemcpyHostToDevice));
        return IMCOM_OK;
    }
    size_t off = align_up(ctx->pin_used, 64);
    if (off + bytes > ctx->pin_bytes) {
        // the ring wraps: every copy queued out of it must have run -- on the current stream and on the streams the context's
        // sub-batches run on (eigen.hip: their copies out of the ring may still be pending)
        IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->own_stream && ctx->own_stream != ctx->stream) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->own_stream));
        if (ctx->aux_stream) IMCOM_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
        for (auto s_ : ctx->sub_streams) IMCOM_HIP_CHECK(hipStreamSynchronize(s_));
        for (auto s_ : ctx->part_streams) IMCOM_HIP_CHECK(hipStreamSynchronize(s_));
        off = 0;
    }
    memcpy(ctx->pin + off, src_host, bytes);
    ctx->pin_used = off + bytes;
    IMCOM_HIP_CHECK(hipMemcpyAsync(dst, ctx->pin + off, bytes, hipMemcpyHostToDevice, ctx->stream));
    return IMCOM_OK;
}

// HIP-event bracket around a group of launches of one kernel family (no-op unless profiling is on).
struct ProfScope {
    imcom_ctx *ctx;
    hipEvent_t start = nullptr, stop = nullptr;
    const char *family;
    long launches;
    ProfScope(imcom_ctx *c, const char *fam, long n = 1, bool fine = false);  // fine: only at profile level 2
    ~ProfScope();
};

int profile_collect(imcom_ctx *ctx);
int ensure_aux(imcom_ctx *ctx);  // creates ctx->aux_stream on first need (api.hip)

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return IMCOM_ERR_HIP;
    }
    return IMCOM_OK;
}

}  // namespace imcom
