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
// counts != NULL: row = face slot (frame row / F, face row % F); a slot at or past its frame's face count holds no face (its crop was
// zero-filled by fid_align_crops) and is written as a ZERO row -- the gathered query matrix then carries the face counts itself
// (a rank that receives it can tell faces from empty slots: SURVEY.md 8e; reference main.py:132 iterates detected faces only)
__global__ void __launch_bounds__(256) l2norm_rows(const float *__restrict__ x, int n, int dim, _Float16 *__restrict__ out,
                                                   const int *__restrict__ counts = nullptr, int F = 1) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    if (counts) {
        const int b = row / F, f = row - b * F;
        if (f >= counts[b]) {
            // an EMPTY slot is a zero row whose first element is -0.0 (bit pattern 0x8000): numerically the zero row it always was (score 0,
            // never a match), but distinguishable from the all +0.0 row a DEGENERATE face of the valid prefix gets below -- so the
            // gathered matrix carries counts[] exactly (pipeline.gathered_face_counts; reference main.py:132 iterates every detected face)
            for (int i = lane; i < dim; i += 64)
                out[(size_t)row * dim + i] = i == 0 ? __builtin_bit_cast(_Float16, (unsigned short)0x8000) : (_Float16)0.f;
            return;
        }
    }
    const float *r = x + (size_t)row * dim;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) ss = fmaf(r[i], r[i], ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    // an all-zero (or non-finite) row becomes a ZERO row, never NaN: a zero row scores 0 against everything and a match
    // needs a score > 0, which is what the reference's `nan > x == False` chain does with such a target (main.py:139-140)
    const bool ok = nrm > 0.f && nrm < __builtin_inff();
    for (int i = lane; i < dim; i += 64) out[(size_t)row * dim + i] = ok ? (_Float16)(r[i] / nrm) : (_Float16)0.f;
}

// top-k per query row of a [n, ld] fp32 score matrix (k <= 8): one wavefront per row, each lane keeps the
// best k of its strided share in registers (insertion), then k rounds of wave arg-max merge them.
// Order: score descending, index ascending on ties; only scores > max(0, thresh) are reported.
template <int K>
__global__ void __launch_bounds__(256) topk_rows(const float *__restrict__ scores, int n, int G, int ld, float thresh,
                                                 int *__restrict__ idx_out, float *__restrict__ score_out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float *r = scores + (size_t)row * ld;
    float bs[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; j++) { bs[j] = -2.f; bi[j] = 0x7FFFFFFF; }
    for (int i = lane; i < G; i += 64) {
        float s = r[i];
        int id = i;
        if (s > bs[K - 1]) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const bool better = s > bs[j] || (s == bs[j] && id < bi[j]);
                const float ts = bs[j]; const int ti = bi[j];
                if (better) { bs[j] = s; bi[j] = id; s = ts; id = ti; }
            }
        }
    }
    const float floor_ = thresh > 0.f ? thresh : 0.f;
    int head = 0;   // this lane's next unconsumed candidate is bs[head] (static indexing via select chain)
    for (int t = 0; t < K; t++) {
        float s = -2.f; int id = 0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < K; j++) if (j == head) { s = bs[j]; id = bi[j]; }
        float ws = s; int wi = id;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(ws, o); const int oi = __shfl_xor(wi, o);
            if (os > ws || (os == ws && oi < wi)) { ws = os; wi = oi; }
        }
        if (wi == id && ws == s) head++;            // the winning lane advances
        if (lane == 0) {
            const bool ok = ws > floor_;
            idx_out[(size_t)row * K + t] = ok ? wi : -1;
            score_out[(size_t)row * K + t] = ok ? ws : 0.f;
        }
    }
}

// normalise `n` host-provided rows and write them to arbitrary gallery rows (upsert); zero rows = deleted
__global__ void __launch_bounds__(256) set_rows(const float *__restrict__ x, const int *__restrict__ rows, int n, int dim, int cap,
                                                _Float16 *__restrict__ gal) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const int dst = rows[r];
    if (dst < 0 || dst >= cap) return;
    const float *src = x + (size_t)r * dim;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) ss = fmaf(src[i], src[i], ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    const bool ok = nrm > 0.f && nrm < __builtin_inff();
    for (int i = lane; i < dim; i += 64) gal[(size_t)dst * dim + i] = ok ? (_Float16)(src[i] / nrm) : (_Float16)0.f;
}

// arg-max over `parts` key arrays (one per gallery shard; keys carry GLOBAL column indices, so the maximum key is the best
// score and, among equal scores, the lowest gallery index -- the strict-'>' scan of main.py:139-140) + threshold
__global__ void __launch_bounds__(256) match_finalize(const unsigned long long *amax, int parts, int n, int G, float thresh, int *idx,
                                                      float *score) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = amax[i];
    for (int p = 1; p < parts; p++) {
        const unsigned long long k = amax[(size_t)p * n + i];
        key = k > key ? k : key;
    }
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    const float s = __uint_as_float(u);
    const int j = (int)(~(unsigned)key);
    const bool ok = key != 0ull && j >= 0 && j < G && s > 0.f && s > thresh;
    idx[i] = ok ? j : -1;
    score[i] = ok ? s : 0.f;
}

int gemm_vs_gallery(fid_ctx *ctx, fid_gallery *g, const void *q, int n, int flags, void *out, unsigned long long *amax, int col0 = 0) {
    if (flags == CF_ARGMAX && match_scan256_applicable(n, g->Gp, g->dim, ctx->num_cus))
        return match_scan256_launch(ctx, q, g->unit_f16, n, g->Gp, g->dim, col0, amax);
    ConvArgs a{};
    a.in = q;
    a.w = g->unit_f16;
    a.out = out;
    a.amax = amax;
    a.amax_col0 = col0;
    a.H = a.W = a.Ho = a.Wo = 1;
    a.Cin_p = g->dim;
    a.Cout_p = g->Gp;
    a.w_rows = g->Gp;
    a.kh = a.kw = 1; a.stride = 1; a.pad = 0;
    a.M = n;
    a.act = ACT_NONE;
    a.flags = flags;
    a.tm_fast = getenv("FID_MATCH_TN_FAST") ? 0 : 1;   // the query tiles of one gallery tile run together: the gallery streams from HBM once
    a.in_bytes = (unsigned)((size_t)n * g->dim * 2);
    a.w_bytes = (unsigned)((size_t)g->Gp * g->dim * 2);
    ConvPlan plan = conv_plan(a, ctx->num_cus, false);
    // large galleries: the register-staged 128x128x64 kernel (measured 480-510 TFLOP/s against 350-390 for the LDS-DMA ring:
    // with >= 2 tiles per CU resident its loads of the next K-step overlap the other workgroup's MFMAs, and the ring's
    // fill rate -- one 1-KB piece per ~70 cycles and CU -- is what bounds a 128x128 tile)
    if ((long long)cdiv(n, 128) * cdiv(g->Gp, 128) >= 2LL * ctx->num_cus && g->dim % 64 == 0 && !getenv("FID_MATCH_DMA")) {
        plan.gen = 1; plan.bm = 128; plan.bn = 128; plan.bk = 64; plan.ksplit = 1; plan.partial_bytes = 0;
    }
    return conv_launch(ctx, a, plan);
}

}  // namespace
}  // namespace fid

extern "C" {

int fid_l2_normalize_f16(fid_ctx *ctx, const float *emb_dev, int n, int dim, void *out_f16_dev) {
    FID_REQUIRE(ctx && emb_dev && out_f16_dev && n > 0 && dim > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    hipLaunchKernelGGL(fid::l2norm_rows, dim3(fid::cdiv(n, 4)), dim3(256), 0, ctx->stream, emb_dev, n, dim, (_Float16 *)out_f16_dev, (const int *)nullptr, 1);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

int fid_l2_normalize_f16_slots(fid_ctx *ctx, const float *emb_dev, int n, int dim, const int32_t *counts_dev, int faces_per_frame,
                               void *out_f16_dev) {
    FID_REQUIRE(ctx && emb_dev && out_f16_dev && counts_dev && n > 0 && dim > 0 && faces_per_frame > 0 && n % faces_per_frame == 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    hipLaunchKernelGGL(fid::l2norm_rows, dim3(fid::cdiv(n, 4)), dim3(256), 0, ctx->stream, emb_dev, n, dim, (_Float16 *)out_f16_dev,
                       (const int *)counts_dev, faces_per_frame);
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
    hipLaunchKernelGGL(fid::l2norm_rows, dim3(fid::cdiv(G, 4)), dim3(256), 0, ctx->stream, (const float *)tmp, G, dim, (_Float16 *)g->unit_f16, (const int *)nullptr, 1);
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

int fid_gallery_data(fid_gallery *g, void **unit_rows_dev) {
    FID_REQUIRE(g && unit_rows_dev, "bad args");
    *unit_rows_dev = g->unit_f16;
    return FID_OK;
}

int fid_match(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, float thresh, int32_t *idx_dev, float *score_dev) {
    FID_REQUIRE(ctx && g && query_f16_dev && idx_dev && score_dev && n > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    void *ws;
    FID_TRY(fid::get_scratch(ctx, 2, (size_t)n * 8, &ws));
    FID_HIP(hipMemsetAsync(ws, 0, (size_t)n * 8, ctx->stream));
    FID_TRY(fid::gemm_vs_gallery(ctx, g, query_f16_dev, n, fid::CF_ARGMAX, nullptr, (unsigned long long *)ws));
    hipLaunchKernelGGL(fid::match_finalize, dim3(fid::cdiv(n, 256)), dim3(256), 0, ctx->stream, (const unsigned long long *)ws, 1, n, g->G,
                       thresh, idx_dev, score_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

// Gallery sharded over ranks (the 1 M-entry variant): every rank scans ITS rows for all queries and emits one
// packed key per query, (sortable(score) << 32) | ~(first_row + local index); the keys of all ranks are exchanged by one tiny
// all-gather (8 bytes per query and rank) and fid_match_merge takes the maximum -- the same result as one scan of the whole
// gallery, first index winning ties, because the shards are contiguous row blocks.
int fid_match_keys(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, int first_row, uint64_t *keys_dev) {
    FID_REQUIRE(ctx && g && query_f16_dev && keys_dev && n > 0 && first_row >= 0, "bad args");
    FID_REQUIRE((long long)first_row + g->Gp < 0x7FFFFFFFll, "global gallery index overflows 31 bits");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_HIP(hipMemsetAsync(keys_dev, 0, (size_t)n * 8, ctx->stream));
    return fid::gemm_vs_gallery(ctx, g, query_f16_dev, n, fid::CF_ARGMAX, nullptr, (unsigned long long *)keys_dev, first_row);
}

int fid_match_merge(fid_ctx *ctx, const uint64_t *keys_dev, int parts, int n, int G_total, float thresh, int32_t *idx_dev,
                    float *score_dev) {
    FID_REQUIRE(ctx && keys_dev && idx_dev && score_dev && parts > 0 && n > 0 && G_total > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    hipLaunchKernelGGL(fid::match_finalize, dim3(fid::cdiv(n, 256)), dim3(256), 0, ctx->stream, (const unsigned long long *)keys_dev, parts, n,
                       G_total, thresh, idx_dev, score_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

// Vector-store style use of the gallery (the product layer's QdrantManager.search_similar / add_embedding /
// delete, reference qdrant_manager.py:91-212): top-k search with a score threshold, and in-place upserts.
int fid_gallery_topk(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, int k, float thresh, int32_t *idx_dev,
                     float *score_dev) {
    FID_REQUIRE(ctx && g && query_f16_dev && idx_dev && score_dev && n > 0, "bad args");
    FID_REQUIRE(k == 1 || k == 2 || k == 4 || k == 5 || k == 8, "k must be one of 1, 2, 4, 5, 8");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    // the score matrix is materialised per chunk of queries (<= 256 MiB) -- top-k needs every score once
    const int chunk = std::max(1, (int)std::min<long long>(n, (256ll << 20) / ((long long)g->Gp * 4)));
    void *ws;
    FID_TRY(fid::get_scratch(ctx, 3, (size_t)chunk * g->Gp * 4, &ws));
    for (int q0 = 0; q0 < n; q0 += chunk) {
        const int m = std::min(chunk, n - q0);
        const char *q = (const char *)query_f16_dev + (size_t)q0 * g->dim * 2;
        FID_TRY(fid::gemm_vs_gallery(ctx, g, q, m, fid::CF_OUT_F32, ws, nullptr));
        dim3 grid(fid::cdiv(m, 4));
        int32_t *io = idx_dev + (size_t)q0 * k;
        float *so = score_dev + (size_t)q0 * k;
#define TOPK(KK) hipLaunchKernelGGL(fid::topk_rows<KK>, grid, dim3(256), 0, ctx->stream, (const float *)ws, m, g->G, g->Gp, thresh, io, so)
        switch (k) { case 1: TOPK(1); break; case 2: TOPK(2); break; case 4: TOPK(4); break; case 5: TOPK(5); break; default: TOPK(8); }
#undef TOPK
    }
    FID_HIP(hipGetLastError());
    return FID_OK;
}

// rows_host[i] in [0, G): overwrite gallery row rows_host[i] with the unit vector of emb_host[i] (upsert);
// an all-zero embedding deletes the row (a zero row can never match: a match needs a score > 0)
int fid_gallery_set_rows(fid_ctx *ctx, fid_gallery *g, const int32_t *rows_host, const float *emb_host, int n) {
    FID_REQUIRE(ctx && g && rows_host && emb_host && n > 0, "bad args");
    for (int i = 0; i < n; i++) FID_REQUIRE(rows_host[i] >= 0 && rows_host[i] < g->G, "row %d outside the gallery (%d rows)", rows_host[i], g->G);
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    void *ws;
    const size_t eb = (size_t)n * g->dim * 4, rb = ((size_t)n * 4 + 255) & ~(size_t)255;
    FID_TRY(fid::get_scratch(ctx, 3, eb + rb, &ws));
    FID_HIP(hipMemcpyAsync(ws, rows_host, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    FID_HIP(hipMemcpyAsync((char *)ws + rb, emb_host, eb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(fid::set_rows, dim3(fid::cdiv(n, 4)), dim3(256), 0, ctx->stream, (const float *)((char *)ws + rb), (const int *)ws, n,
                       g->dim, g->G, (_Float16 *)g->unit_f16);
    FID_HIP(hipStreamSynchronize(ctx->stream));   // the host buffers may be reused on return
    return FID_OK;
}

int fid_cosine_matrix(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, float *out_dev) {
    FID_REQUIRE(ctx && g && query_f16_dev && out_dev && n > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    return fid::gemm_vs_gallery(ctx, g, query_f16_dev, n, fid::CF_OUT_F32, out_dev, nullptr);
}

}  // extern "C"
