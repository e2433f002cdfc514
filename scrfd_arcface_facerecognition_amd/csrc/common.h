// Shared internals of libfaceid.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/faceid.h"

namespace fid {

void set_error(const char *fmt, ...);

#define FID_HIP(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fid::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return FID_E_HIP;                                                            \
        }                                                                                \
    } while (0)

#define FID_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            fid::set_error(__VA_ARGS__);       \
            return FID_E_INVALID;              \
        }                                      \
    } while (0)

#define FID_TRY(expr)                \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != FID_OK) return rc_; \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Division of a non-negative int (< 2^31) by a launch-time constant as multiply-high + shift: the kernels decode
// (item -> image, tile row, tile column, cout block) several times per step, and a runtime integer division costs
// ~40 instructions each.  q = x / d  ==  d == 1 ? x : umulhi(x, mul) >> shr.
struct FastDiv {
    unsigned mul, shr, d;
};
inline FastDiv fastdiv_make(int d) {
    FastDiv f{0u, 0u, (unsigned)d};
    if (d > 1) {
        unsigned lg = 0;
        while ((1ull << lg) < (unsigned long long)d) lg++;      // ceil(log2(d))
        const unsigned p = 31 + lg;
        f.mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
        f.shr = p - 32;
    }
    return f;
}
#ifdef __HIPCC__
// Workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2): position of workgroup b in XCD-major order, so
// that workgroups which share an L2 get CONSECUTIVE work items (neighbouring tiles share halo pixels, the cout blocks of one
// tile share its whole patch).  A bijection on [0, g); speed only, never correctness.
__device__ __forceinline__ int xcd_major_id(int b, int g) {
    const int q = g >> 3, r = g & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__device__ __forceinline__ int fastdiv(int x, const FastDiv &f) { return f.d == 1 ? x : (int)(__umulhi((unsigned)x, f.mul) >> f.shr); }
#endif
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace fid

struct fid_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::mutex mu;
    // growable scratch arenas (never shrink; no allocation on the steady-state path)
    void *scratch[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // 0 post-process, 1 split-K partials, 2 / 3 match, 4 the autotuner's cache-flush buffer
    size_t scratch_bytes[5] = {0, 0, 0, 0, 0};
    hipEvent_t events[FID_MAX_EVENTS] = {};
    hipStream_t copy_stream = nullptr;   // H2D uploads that overlap compute (video front-end)
    hipEvent_t copy_done = nullptr, compute_done = nullptr;
    hipEvent_t slot_uploaded[FID_UPLOAD_SLOTS] = {}, slot_released[FID_UPLOAD_SLOTS] = {};   // per staging buffer
    bool slot_has_upload[FID_UPLOAD_SLOTS] = {}, slot_has_release[FID_UPLOAD_SLOTS] = {};
    // SCRFD post-process state
    int cand_cap = 4096;
    int32_t *status_dev = nullptr;  // [0] max candidates seen, [1] max survivors seen
    int last_out_cap = 0;
};

namespace fid {
// scratch arena `slot` with at least `bytes` bytes (grows by reallocating; contents undefined)
int get_scratch(fid_ctx *ctx, int slot, size_t bytes, void **out);
// give a scratch arena back (synchronises the stream first): the autotuner's 320 MB flush buffer does not outlive a tuning run
int release_scratch(fid_ctx *ctx, int slot);
// hipFuncAttributeMaxDynamicSharedMemorySize >= bytes for kernel `func` on the context's device.  The attribute is per device, so the
// largest value set so far is remembered per (kernel, device) -- under a mutex: contexts on several devices and several host
// threads launch the same kernels (a function-local static flag covered neither).
int ensure_dyn_lds(fid_ctx *ctx, const void *func, int bytes);
}  // namespace fid
