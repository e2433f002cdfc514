// Context, device memory, events.  Plain HIP runtime calls; no torch anywhere in this library.
#include "common.h"

#include <map>

namespace fid {
static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
int get_scratch(fid_ctx *ctx, int slot, size_t bytes, void **out) {
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            FID_HIP(hipStreamSynchronize(ctx->stream));
            FID_HIP(hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        size_t want = bytes + bytes / 4 + 4096;
        FID_HIP(hipMalloc(&ctx->scratch[slot], want));
        ctx->scratch_bytes[slot] = want;
    }
    *out = ctx->scratch[slot];
    return FID_OK;
}
int release_scratch(fid_ctx *ctx, int slot) {
    if (ctx->scratch[slot]) {
        FID_HIP(hipStreamSynchronize(ctx->stream));
        FID_HIP(hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
    }
    return FID_OK;
}
// hipFuncSetAttribute acts on the thread's CURRENT device: set the context's first, or a thread that drives contexts on two devices
// records the attribute as done for one device while having set it on the other
int ensure_dyn_lds(fid_ctx *ctx, const void *func, int bytes) {
    // FID_KLOG=1 (tools/klog_map.py): name the kernel every launcher is about to launch -- with net.hip's "[klog] op" lines the map op -> kernel
    // that turns a profiler's per-kernel MFMA instruction counts into executed / algorithmic work per kernel
    static const bool klog = getenv("FID_KLOG") != nullptr;
    if (klog) fprintf(stderr, "[klog] kernel %s\n", hipKernelNameRefByPtr(func, ctx->stream));
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> set_bytes;
    std::lock_guard<std::mutex> lk(mu);
    int &cur = set_bytes[std::make_pair(func, ctx->device)];
    if (bytes > cur) {
        FID_HIP(hipSetDevice(ctx->device));
        FID_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        cur = bytes;
    }
    return FID_OK;
}
}  // namespace fid

extern "C" {

int fid_abi_version(void) { return FID_ABI_VERSION; }
const char *fid_last_error(void) { return fid::g_err.c_str(); }

int fid_device_count(int *count) {
    FID_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *count = n;
    return FID_OK;
}

int fid_ctx_create(int device, void *stream, fid_ctx **out) {
    FID_REQUIRE(out, "out is NULL");
    int n = 0;
    FID_HIP(hipGetDeviceCount(&n));
    FID_REQUIRE(device >= 0 && device < n, "device %d out of range (have %d)", device, n);
    FID_HIP(hipSetDevice(device));
    fid_ctx *c = new fid_ctx();
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        FID_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    FID_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    FID_HIP(hipMalloc((void **)&c->status_dev, 64));
    FID_HIP(hipMemsetAsync(c->status_dev, 0, 64, c->stream));
    *out = c;
    return FID_OK;
}

int fid_ctx_destroy(fid_ctx *ctx) {
    if (!ctx) return FID_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 4; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    for (int i = 0; i < FID_MAX_EVENTS; i++)
        if (ctx->events[i]) (void)hipEventDestroy(ctx->events[i]);
    if (ctx->status_dev) (void)hipFree(ctx->status_dev);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->copy_done) (void)hipEventDestroy(ctx->copy_done);
    if (ctx->compute_done) (void)hipEventDestroy(ctx->compute_done);
    for (int i = 0; i < FID_UPLOAD_SLOTS; i++) {
        if (ctx->slot_uploaded[i]) (void)hipEventDestroy(ctx->slot_uploaded[i]);
        if (ctx->slot_released[i]) (void)hipEventDestroy(ctx->slot_released[i]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FID_OK;
}

int fid_sync(fid_ctx *ctx) {
    FID_REQUIRE(ctx, "ctx is NULL");
    FID_HIP(hipStreamSynchronize(ctx->stream));
    return FID_OK;
}

int fid_device_name(fid_ctx *ctx, char *buf, int buflen) {
    FID_REQUIRE(ctx && buf && buflen > 0, "bad args");
    hipDeviceProp_t prop;
    FID_HIP(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, buflen, "%s|%s|cus=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return FID_OK;
}

int fid_malloc(fid_ctx *ctx, size_t bytes, void **dptr) {
    FID_REQUIRE(ctx && dptr, "bad args");
    FID_HIP(hipSetDevice(ctx->device));
    FID_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return FID_OK;
}

int fid_free(fid_ctx *ctx, void *dptr) {
    FID_REQUIRE(ctx, "ctx is NULL");
    if (!dptr) return FID_OK;
    FID_HIP(hipStreamSynchronize(ctx->stream));
    FID_HIP(hipFree(dptr));
    return FID_OK;
}

int fid_memcpy_h2d(fid_ctx *ctx, void *dst, const void *src, size_t bytes) {
    FID_REQUIRE(ctx && (bytes == 0 || (dst && src)), "bad args");
    if (bytes == 0) return FID_OK;
    // pageable source: hipMemcpyAsync stages it and returns once the source may be reused
    FID_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return FID_OK;
}

int fid_memcpy_d2h(fid_ctx *ctx, void *dst, const void *src, size_t bytes) {
    FID_REQUIRE(ctx && (bytes == 0 || (dst && src)), "bad args");
    if (bytes == 0) return FID_OK;
    FID_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FID_HIP(hipStreamSynchronize(ctx->stream));
    return FID_OK;
}

int fid_memset(fid_ctx *ctx, void *dst, int value, size_t bytes) {
    FID_REQUIRE(ctx && (bytes == 0 || dst), "bad args");
    if (bytes == 0) return FID_OK;
    FID_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return FID_OK;
}

// ---- pinned host memory + an upload stream: the video front-end's double-buffered H2D (reference main.py:174-184
// reads one frame at a time; here the next batch is uploaded while the current one is processed) ----
int fid_pinned_alloc(fid_ctx *ctx, size_t bytes, void **hptr) {
    FID_REQUIRE(ctx && hptr && bytes > 0, "bad args");
    FID_HIP(hipSetDevice(ctx->device));
    FID_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return FID_OK;
}

int fid_pinned_free(fid_ctx *ctx, void *hptr) {
    FID_REQUIRE(ctx, "ctx is NULL");
    if (hptr) FID_HIP(hipHostFree(hptr));
    return FID_OK;
}

static int ensure_copy_stream(fid_ctx *ctx) {
    if (!ctx->copy_stream) {
        FID_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        FID_HIP(hipEventCreateWithFlags(&ctx->copy_done, hipEventDisableTiming));
        FID_HIP(hipEventCreateWithFlags(&ctx->compute_done, hipEventDisableTiming));
    }
    return FID_OK;
}

// enqueue a pinned-host -> device copy on the context's upload stream; it starts only after everything enqueued
// so far on the compute stream has finished (the conservative form: the destination may still be read by ANY earlier kernel;
// for double buffering use the slot form below, which waits only for the last reader of that buffer)
int fid_upload_async(fid_ctx *ctx, void *dst_dev, const void *src_pinned, size_t bytes) {
    FID_REQUIRE(ctx && dst_dev && src_pinned && bytes > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_TRY(ensure_copy_stream(ctx));
    FID_HIP(hipEventRecord(ctx->compute_done, ctx->stream));
    FID_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->compute_done, 0));
    FID_HIP(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    FID_HIP(hipEventRecord(ctx->copy_done, ctx->copy_stream));
    return FID_OK;
}

// make the compute stream wait for the uploads enqueued so far (no host synchronisation)
int fid_upload_wait(fid_ctx *ctx) {
    FID_REQUIRE(ctx, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    if (ctx->copy_done) FID_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_done, 0));
    return FID_OK;
}

// ---- per-buffer ordering for double / triple buffering: slot k names one device staging buffer.
//   fid_upload_release(k)      "everything enqueued so far on the compute stream is the last reader of buffer k"
//   fid_upload_async_slot(k)   copy into buffer k; waits ONLY for buffer k's release (not for the step that is running on
//                              another buffer), so the upload of batch i+1 really overlaps the compute of batch i
//   fid_upload_wait_slot(k)    the compute stream waits for buffer k's upload
static int slot_events(fid_ctx *ctx, int slot) {
    FID_REQUIRE(slot >= 0 && slot < FID_UPLOAD_SLOTS, "upload slot %d outside [0, %d)", slot, FID_UPLOAD_SLOTS);
    FID_TRY(ensure_copy_stream(ctx));
    if (!ctx->slot_uploaded[slot]) {
        FID_HIP(hipEventCreateWithFlags(&ctx->slot_uploaded[slot], hipEventDisableTiming));
        FID_HIP(hipEventCreateWithFlags(&ctx->slot_released[slot], hipEventDisableTiming));
    }
    return FID_OK;
}

int fid_upload_release(fid_ctx *ctx, int slot) {
    FID_REQUIRE(ctx, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_TRY(slot_events(ctx, slot));
    FID_HIP(hipEventRecord(ctx->slot_released[slot], ctx->stream));
    ctx->slot_has_release[slot] = true;
    return FID_OK;
}

int fid_upload_async_slot(fid_ctx *ctx, int slot, void *dst_dev, const void *src_pinned, size_t bytes) {
    FID_REQUIRE(ctx && dst_dev && src_pinned && bytes > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_TRY(slot_events(ctx, slot));
    if (ctx->slot_has_release[slot]) FID_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->slot_released[slot], 0));
    FID_HIP(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    FID_HIP(hipEventRecord(ctx->slot_uploaded[slot], ctx->copy_stream));
    ctx->slot_has_upload[slot] = true;
    return FID_OK;
}

int fid_upload_wait_slot(fid_ctx *ctx, int slot) {
    FID_REQUIRE(ctx, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_TRY(slot_events(ctx, slot));
    if (ctx->slot_has_upload[slot]) FID_HIP(hipStreamWaitEvent(ctx->stream, ctx->slot_uploaded[slot], 0));
    return FID_OK;
}

int fid_event_record(fid_ctx *ctx, int slot) {
    FID_REQUIRE(ctx && slot >= 0 && slot < FID_MAX_EVENTS, "bad event slot %d", slot);
    if (!ctx->events[slot]) FID_HIP(hipEventCreate(&ctx->events[slot]));
    FID_HIP(hipEventRecord(ctx->events[slot], ctx->stream));
    return FID_OK;
}

int fid_event_elapsed_ms(fid_ctx *ctx, int a, int b, float *ms) {
    FID_REQUIRE(ctx && ms && a >= 0 && b >= 0 && a < FID_MAX_EVENTS && b < FID_MAX_EVENTS, "bad args");
    FID_REQUIRE(ctx->events[a] && ctx->events[b], "event slot not recorded");
    FID_HIP(hipEventSynchronize(ctx->events[b]));
    FID_HIP(hipEventElapsedTime(ms, ctx->events[a], ctx->events[b]));
    return FID_OK;
}

}  // extern "C"
