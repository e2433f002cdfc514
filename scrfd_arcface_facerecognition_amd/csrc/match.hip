// L2 normalisation and the gallery match.
//   fid_l2_normalize_f16 / fid_gallery_create : the two np.linalg.norm divisions of
//       compute_similarity (reference utils/helpers.py:120-123), hoisted: every vector is scaled to
//       unit length ONCE (gallery at build time, queries once per batch) and stored as fp16.
//   fid_match : the per-target python loop of reference main.py:136-142 as ONE MFMA GEMM
//       [n x dim] x [dim x G] (conv.hip, 1x1 "conv" whose weights are the gallery) with a fused
//       arg-max epilogue -- the n x G score matrix is never written.  First maximum wins ties and a
//       match needs score > max(0, thresh), exactly like the strict '>' chain of the reference.
#include "conv.h"

struct fid_gallery {
    int G = 0, Gp = 0, dim = 0;
    void *unit_f16 = nullptr;  // [Gp][dim], rows >= G are zero
};

namespace fid {
namespace {

// one wavefront per row
__global__ void __launch_bounds__(256) l2norm_rows(const float *__restrict__ x, int n, int dim, _Float16 *__restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float *r = x + (size_t)row * dim;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) ss = fmaf(r[i], r[i], ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    for (int i = lane; i < dim; i += 64) out[(size_t)row * dim + i] = (_Float16)(r[i] / nrm);
}

__global__ void __launch_bounds__(256) match_finalize(const unsigned long long *amax, int n, int G, float thresh, int *idx,
                                                      float *score) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = amax[i];
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    const float s = __uint_as_float(u);
    const int j = (int)(~(unsigned)key);
    const bool ok = key != 0ull && j < G && s > 0.f && s > thresh;
    idx[i] = ok ? j : -1;
    score[i] = ok ? s : 0.f;
}

int gemm_vs_gallery(fid_ctx *ctx, fid_gallery *g, const void *q, int n, int flags, void *out, unsigned long long *amax) {
    ConvArgs a{};
    a.in = q;
    a.w = g->unit_f16;
    a.out = out;
    a.amax = amax;
    a.H = a.W = a.Ho = a.Wo = 1;
    a.Cin_p = g->dim;
    a.Cout_p = g->Gp;
    a.w_rows = g->Gp;
    a.kh = a.kw = 1; a.stride = 1; a.pad = 0;
    a.M = n;
    a.act = ACT_NONE;
    a.flags = flags;
    a.in_bytes = (unsigned)((size_t)n * g->dim * 2);
    a.w_bytes = (unsigned)((size_t)g->Gp * g->dim * 2);
    ConvPlan plan = conv_plan(a, ctx->num_cus, false);
    return conv_launch(ctx, a, plan);
}

}  // namespace
}  // namespace fid

extern "C" {

int fid_l2_normalize_f16(fid_ctx *ctx, const float *emb_dev, int n, int dim, void *out_f16_dev) {
    FID_REQUIRE(ctx && emb_dev && out_f16_dev && n > 0 && dim > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipLaunchKernelGGL(fid::l2norm_rows, dim3(fid::cdiv(n, 4)), dim3(256), 0, ctx->stream, emb_dev, n, dim, (_Float16 *)out_f16_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

int fid_gallery_create(fid_ctx *ctx, const float *gallery, int G, int dim, fid_gallery **out) {
    FID_REQUIRE(ctx && gallery && out && G > 0, "bad args");
    FID_REQUIRE(dim > 0 && dim % 32 == 0, "embedding dim %d must be a multiple of 32", dim);
    FID_REQUIRE((size_t)((G + 31) / 32 * 32) * dim * 2 <= 0x7FFFFFF0ull, "gallery of %d x %d exceeds 2 GiB in fp16", G, dim);
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));
    fid_gallery *g = new fid_gallery();
    g->G = G; g->Gp = (G + 31) / 32 * 32; g->dim = dim;
    void *tmp = nullptr;
    FID_HIP(hipMalloc(&g->unit_f16, (size_t)g->Gp * dim * 2 + 256));
    FID_HIP(hipMalloc(&tmp, (size_t)G * dim * 4));
    FID_HIP(hipMemsetAsync(g->unit_f16, 0, (size_t)g->Gp * dim * 2, ctx->stream));
    FID_HIP(hipMemcpyAsync(tmp, gallery, (size_t)G * dim * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(fid::l2norm_rows, dim3(fid::cdiv(G, 4)), dim3(256), 0, ctx->stream, (const float *)tmp, G, dim, (_Float16 *)g->unit_f16);
    FID_HIP(hipStreamSynchronize(ctx->stream));
    FID_HIP(hipFree(tmp));
    *out = g;
    return FID_OK;
}

int fid_gallery_destroy(fid_ctx *ctx, fid_gallery *g) {
    if (!g) return FID_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (g->unit_f16) (void)hipFree(g->unit_f16);
    delete g;
    return FID_OK;
}

int fid_gallery_info(fid_gallery *g, int *G, int *G_padded, int *dim) {
    FID_REQUIRE(g, "gallery is NULL");
    if (G) *G = g->G;
    if (G_padded) *G_padded = g->Gp;
    if (dim) *dim = g->dim;
    return FID_OK;
}

int fid_match(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, float thresh, int32_t *idx_dev, float *score_dev) {
    FID_REQUIRE(ctx && g && query_f16_dev && idx_dev && score_dev && n > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    void *ws;
    FID_TRY(fid::get_scratch(ctx, 2, (size_t)n * 8, &ws));
    FID_HIP(hipMemsetAsync(ws, 0, (size_t)n * 8, ctx->stream));
    FID_TRY(fid::gemm_vs_gallery(ctx, g, query_f16_dev, n, fid::CF_ARGMAX, nullptr, (unsigned long long *)ws));
    hipLaunchKernelGGL(fid::match_finalize, dim3(fid::cdiv(n, 256)), dim3(256), 0, ctx->stream, (const unsigned long long *)ws, n, g->G,
                       thresh, idx_dev, score_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

int fid_cosine_matrix(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, float *out_dev) {
    FID_REQUIRE(ctx && g && query_f16_dev && out_dev && n > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return fid::gemm_vs_gallery(ctx, g, query_f16_dev, n, fid::CF_OUT_F32, out_dev, nullptr);
}

}  // extern "C"
