// SCRFD post-process on the GPU: threshold + decode + compact, exact sort, greedy NMS, max_num.
// Replaces the numpy/python code of reference models/scrfd.py:89-178 (forward's decode loop and
// detect's sort / NMS / max_num) and :180-207 (nms), utils/helpers.py:62-107 (distance2bbox/kps).
//
// Bit-exactness contract: every fp32 operation below is a single correctly
// rounded IEEE operation in the same order numpy performs it (no FMA contraction: the explicit
// __f*_rn intrinsics forbid it), so given identical head tensors the survivors, their order and
// their coordinates are identical to the reference's.
//
// HBM-bound integer/fp32 work: one pass over the 16800 x 15 head values per frame (1.0 MB),
// everything after the threshold touches only the few candidates.
#include "common.h"

namespace {

constexpr int CAND_W = 16;  // floats per candidate record: x1 y1 x2 y2 score kps[10] orig_index

struct HeadViews {
    const float *ptr[9];
    int pix_stride[9];
    int anc_stride[9];
    long long batch_stride[9];
};

__device__ __forceinline__ unsigned sortable(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One thread per anchor.  scrfd.py:89-119 + :145-148 (the division by det_scale).
__global__ void __launch_bounds__(256) decode_compact(HeadViews hv, int in_h, int in_w, int A, float thr,
                                                      float det_scale, int by_index, int cand_cap, int *cand_count,
                                                      unsigned long long *cand_key, float *cand_data) {
    const int b = blockIdx.y;
    const int n8 = (in_h / 8) * (in_w / 8) * A, n16 = (in_h / 16) * (in_w / 16) * A,
              n32 = (in_h / 32) * (in_w / 32) * A;
    int g = blockIdx.x * blockDim.x + threadIdx.x;  // flat anchor index over the 3 levels
    if (g >= n8 + n16 + n32) return;
    int lvl, i, stride;
    if (g < n8) { lvl = 0; i = g; stride = 8; }
    else if (g < n8 + n16) { lvl = 1; i = g - n8; stride = 16; }
    else { lvl = 2; i = g - n8 - n16; stride = 32; }
    const int pix = i / A, a = i - pix * A;
    const float score = hv.ptr[lvl][b * hv.batch_stride[lvl] + (long long)pix * hv.pix_stride[lvl] + a * hv.anc_stride[lvl]];
    if (!(score >= thr)) return;
    const int slot = atomicAdd(&cand_count[b], 1);
    if (slot >= cand_cap) return;  // overflow is reported through the count itself
    const int fw = in_w / stride;
    const float cx = (float)((pix % fw) * stride), cy = (float)((pix / fw) * stride);
    const float fs = (float)stride;
    const float *bp = hv.ptr[3 + lvl] + b * hv.batch_stride[3 + lvl] + (long long)pix * hv.pix_stride[3 + lvl] + a * hv.anc_stride[3 + lvl];
    const float *kp = hv.ptr[6 + lvl] + b * hv.batch_stride[6 + lvl] + (long long)pix * hv.pix_stride[6 + lvl] + a * hv.anc_stride[6 + lvl];
    float *o = cand_data + ((size_t)b * cand_cap + slot) * CAND_W;
    // distance2bbox on (pred * stride), then / det_scale
    o[0] = __fdiv_rn(__fsub_rn(cx, __fmul_rn(bp[0], fs)), det_scale);
    o[1] = __fdiv_rn(__fsub_rn(cy, __fmul_rn(bp[1], fs)), det_scale);
    o[2] = __fdiv_rn(__fadd_rn(cx, __fmul_rn(bp[2], fs)), det_scale);
    o[3] = __fdiv_rn(__fadd_rn(cy, __fmul_rn(bp[3], fs)), det_scale);
    o[4] = score;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        o[5 + 2 * k] = __fdiv_rn(__fadd_rn(cx, __fmul_rn(kp[2 * k], fs)), det_scale);
        o[6 + 2 * k] = __fdiv_rn(__fadd_rn(cy, __fmul_rn(kp[2 * k + 1], fs)), det_scale);
    }
    o[15] = __int_as_float(g);
    // order: score descending, then flat anchor index ascending (what a stable argsort()[::-1]
    // followed by nms()'s own stable argsort()[::-1] produces; identical to any order when tie-free)
    cand_key[(size_t)b * cand_cap + slot] =
        by_index ? ((unsigned long long)(~(unsigned)g) << 32) : (((unsigned long long)sortable(score) << 32) | (unsigned)(~(unsigned)g));
}

// distance2bbox / distance2kps on plain arrays (utils/helpers.py:62-107): ncol = 4 -> bbox, else kps pairs
__global__ void decode_points(const float *points, const float *dist, int n, int ncol, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ncol) return;
    const int r = i / ncol, c = i - r * ncol;
    const float p = points[r * 2 + (c & 1)];
    out[i] = (ncol == 4 && c < 2) ? __fsub_rn(p, dist[i]) : __fadd_rn(p, dist[i]);
}

// fid_nms entry: records from a plain [K,5] det array.  Order: score desc, index DESC
// (scores.argsort()[::-1] with a stable sort, scrfd.py:188).
__global__ void load_dets(const float *dets, int K, int *cand_count, unsigned long long *cand_key, float *cand_data) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cand_count[0] = K;
    if (i >= K) return;
    float *o = cand_data + (size_t)i * CAND_W;
    for (int k = 0; k < 5; k++) o[k] = dets[i * 5 + k];
    for (int k = 5; k < 15; k++) o[k] = 0.f;
    o[15] = __int_as_float(i);
    cand_key[i] = ((unsigned long long)sortable(dets[i * 5 + 4]) << 32) | (unsigned)i;
}

// Exact sort by counting: rank = number of keys that come before mine (keys are unique).
// O(K^2) per frame but K is tens..hundreds after the 0.5 threshold; no capacity or pow2 limits.
__global__ void __launch_bounds__(256) rank_scatter(const int *cand_count, const unsigned long long *cand_key,
                                                    const float *cand_data, float *sorted, int cand_cap) {
    const int b = blockIdx.y;
    const int K = min(cand_count[b], cand_cap);
    if ((int)(blockIdx.x * 256) >= K) return;
    __shared__ unsigned long long tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long *keys = cand_key + (size_t)b * cand_cap;
    const unsigned long long mine = i < K ? keys[i] : 0ull;
    int rank = 0;
    for (int t0 = 0; t0 < K; t0 += 256) {
        __syncthreads();
        tile[threadIdx.x] = (t0 + (int)threadIdx.x < K) ? keys[t0 + threadIdx.x] : 0ull;
        __syncthreads();
        const int n = min(256, K - t0);
        for (int j = 0; j < n; j++) rank += tile[j] > mine;
    }
    if (i < K) {
        const float4 *src = (const float4 *)(cand_data + ((size_t)b * cand_cap + i) * CAND_W);
        float4 *dst = (float4 *)(sorted + ((size_t)b * cand_cap + rank) * CAND_W);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    }
}

constexpr int NMS_MASK_K = 512;   // candidate lists up to this length take the bit-mask path of nms_select (32 KB of masks)

// Greedy NMS (scrfd.py:187-205) + max_num selection (scrfd.py:159-177).  One workgroup per frame.
// LDS: removed[cand_cap] bytes | keep[cand_cap] ints | value[cand_cap] floats | box[n_box] float4 | sup[NMS_MASK_K][NMS_MASK_K/32] masks.
// The boxes of the n_box best candidates are staged in LDS: the greedy loop is one iteration + barrier per surviving candidate, and
// reading candidate i's box from global memory put a trip to L2 (~0.3 us) on every iteration (29 us per 64-frame step).
__global__ void __launch_bounds__(512) nms_select(const int *cand_count, const float *sorted, int cand_cap, float iou_thr,
                                                  int max_num, int metric, int img_h, int img_w, float *det_out,
                                                  float *kps_out, int *counts_out, int out_cap, int *keep_idx_out,
                                                  int *status, int n_box, int mask_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int Kraw = cand_count[b];
    const int K = min(Kraw, cand_cap);
    unsigned char *removed = smem;
    int *keep = (int *)(smem + ((cand_cap + 15) & ~15));
    float *value = (float *)(keep + cand_cap);
    const float *rec = sorted + (size_t)b * cand_cap * CAND_W;
    float4 *box = (float4 *)(smem + ((((cand_cap + 15) & ~15) + (size_t)cand_cap * 8 + 15) & ~(size_t)15));
    const int KB = min(K, n_box);
    for (int j = threadIdx.x; j < K; j += blockDim.x) removed[j] = 0;
    for (int j = threadIdx.x; j < KB; j += blockDim.x) box[j] = *(const float4 *)(rec + j * CAND_W);
    __syncthreads();
    int nkeep = 0;
    if (mask_ok && K <= NMS_MASK_K && K <= n_box) {
        // Small candidate lists (the usual case: tens to a few hundred per frame): all pairwise decisions first, in parallel, as bit
        // masks -- sup[i] bit j = "i suppresses j" = !(ovr(i, j) <= thr), j > i -- then ONE wave walks the list in score order and ORs
        // the masks of the candidates it keeps.  Same arithmetic, same decisions as the loop below (every operation of the overlap is
        // commutative in (i, j)); the loop's barrier per surviving candidate (8 waves, ~0.25 us each) is gone.
        constexpr int MW = NMS_MASK_K / 64;                      // 64-bit words per row
        unsigned long long *sup = (unsigned long long *)(box + n_box);   // [K][MW]
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, nwv = blockDim.x >> 6;
        const int KW = (K + 63) >> 6;                            // words in use
        for (int i = wv; i < K; i += nwv) {                      // a wave per row i, a lane per column j, one ballot per 64 columns
            const float4 bi = box[i];
            const float area_i = __fmul_rn(__fadd_rn(__fsub_rn(bi.z, bi.x), 1.f), __fadd_rn(__fsub_rn(bi.w, bi.y), 1.f));
            for (int w = 0; w < KW; w++) {
                const int j = w * 64 + ln;
                bool hit = false;
                if (w * 64 + 63 > i && j > i && j < K) {
                    const float4 bj = box[j];
                    const float area_j = __fmul_rn(__fadd_rn(__fsub_rn(bj.z, bj.x), 1.f), __fadd_rn(__fsub_rn(bj.w, bj.y), 1.f));
                    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y), xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
                    const float w_ = fmaxf(0.f, __fadd_rn(__fsub_rn(xx2, xx1), 1.f));
                    const float h_ = fmaxf(0.f, __fadd_rn(__fsub_rn(yy2, yy1), 1.f));
                    const float inter = __fmul_rn(w_, h_);
                    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, area_j), inter));
                    hit = !(ovr <= iou_thr);
                }
                const unsigned long long m = __ballot(hit);
                if (ln == 0) sup[i * MW + w] = m;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                  // one wave: lane l owns word l of the removed mask (MW <= 64)
            const int l = threadIdx.x;
            constexpr int MW2 = NMS_MASK_K / 64;
            const unsigned long long *sup2 = (const unsigned long long *)(box + n_box);
            unsigned long long gone = 0;
            for (int i = 0; i < K; i++) {
                const unsigned long long wi = __shfl(gone, i >> 6);   // (uniform across the wave)
                if ((wi >> (i & 63)) & 1ull) continue;
                if (l == 0) keep[nkeep] = i;
                nkeep++;
                if (l < MW2) gone |= sup2[i * MW2 + l];
            }
            if (l == 0) value[0] = __int_as_float(nkeep);        // hand the count to the other waves
        }
        __syncthreads();
        nkeep = __float_as_int(value[0]);
        __syncthreads();
    } else
    for (int i = 0; i < K; i++) {
        if (removed[i]) continue;  // uniform: written only before the previous barrier
        if (threadIdx.x == 0) keep[nkeep] = i;
        nkeep++;
        const float4 bi = i < KB ? box[i] : *(const float4 *)(rec + i * CAND_W);
        const float x1 = bi.x, y1 = bi.y, x2 = bi.z, y2 = bi.w;
        const float area_i = __fmul_rn(__fadd_rn(__fsub_rn(x2, x1), 1.f), __fadd_rn(__fsub_rn(y2, y1), 1.f));
        for (int j = i + 1 + threadIdx.x; j < K; j += blockDim.x) {
            if (removed[j]) continue;
            const float4 bj = j < KB ? box[j] : *(const float4 *)(rec + j * CAND_W);
            const float area_j = __fmul_rn(__fadd_rn(__fsub_rn(bj.z, bj.x), 1.f), __fadd_rn(__fsub_rn(bj.w, bj.y), 1.f));
            const float xx1 = fmaxf(x1, bj.x), yy1 = fmaxf(y1, bj.y), xx2 = fminf(x2, bj.z), yy2 = fminf(y2, bj.w);
            const float w = fmaxf(0.f, __fadd_rn(__fsub_rn(xx2, xx1), 1.f));
            const float h = fmaxf(0.f, __fadd_rn(__fsub_rn(yy2, yy1), 1.f));
            const float inter = __fmul_rn(w, h);
            const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, area_j), inter));
            if (!(ovr <= iou_thr)) removed[j] = 1;  // np.where(ovr <= thr) keeps; NaN is dropped
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- selection / output ----
    const bool select = (max_num > 0) && (max_num < nkeep);
    if (select) {
        const float cxi = (float)(img_w / 2), cyi = (float)(img_h / 2);  // image_center = (H//2, W//2)
        for (int k = threadIdx.x; k < nkeep; k += blockDim.x) {
            const float *r = rec + keep[k] * CAND_W;
            const float area = __fmul_rn(__fsub_rn(r[2], r[0]), __fsub_rn(r[3], r[1]));
            float v = area;
            if (metric != 0) {
                const float ox = __fsub_rn(__fdiv_rn(__fadd_rn(r[0], r[2]), 2.f), cxi);
                const float oy = __fsub_rn(__fdiv_rn(__fadd_rn(r[1], r[3]), 2.f), cyi);
                const float d2 = __fadd_rn(__fmul_rn(ox, ox), __fmul_rn(oy, oy));
                v = __fsub_rn(area, __fmul_rn(d2, 2.f));
            }
            value[k] = v;
        }
        __syncthreads();
    }
    const int n_out = select ? max_num : nkeep;
    for (int k = threadIdx.x; k < nkeep; k += blockDim.x) {
        int p = k;
        if (select) {  // np.argsort(values)[::-1][:max_num], stable: larger value first, ties -> larger index first
            const float v = value[k];
            int rank = 0;
            for (int j = 0; j < nkeep; j++) rank += (value[j] > v) || (value[j] == v && j > k);
            p = rank < max_num ? rank : -1;
        }
        if (p < 0 || p >= out_cap) continue;
        const float *r = rec + keep[k] * CAND_W;
        if (det_out) {
            float *d = det_out + ((size_t)b * out_cap + p) * 5;
            for (int c = 0; c < 5; c++) d[c] = r[c];
        }
        if (kps_out) {
            float *q = kps_out + ((size_t)b * out_cap + p) * 10;
            for (int c = 0; c < 10; c++) q[c] = r[5 + c];
        }
        if (keep_idx_out) keep_idx_out[(size_t)b * out_cap + p] = __float_as_int(r[15]);
    }
    if (threadIdx.x == 0) {
        counts_out[b] = min(n_out, out_cap);
        atomicMax(&status[0], Kraw);
        atomicMax(&status[1], n_out);
    }
}

}  // namespace

namespace fid {
// dynamic LDS of nms_select for a candidate capacity, and how many boxes fit behind the per-candidate arrays (at most 1024)
static size_t nms_lds_bytes(int cand_cap, int *n_box) {
    const size_t base = ((((size_t)cand_cap + 15) & ~(size_t)15) + (size_t)cand_cap * 8 + 15) & ~(size_t)15;
    const size_t masks = (size_t)NMS_MASK_K * (NMS_MASK_K / 64) * 8;          // 32 KB, only when they fit beside >= NMS_MASK_K boxes
    const size_t room = base < 160 * 1024 ? (160 * 1024 - base) / 16 : 0;
    int nb = (int)std::min<size_t>(1024, room);
    const bool with_masks = base + (size_t)NMS_MASK_K * 16 + masks <= 160 * 1024;
    if (with_masks) nb = (int)std::min<size_t>(1024, (160 * 1024 - base - masks) / 16);
    else nb = std::min(nb, NMS_MASK_K - 1);                                    // fewer boxes than NMS_MASK_K: the kernel never takes the mask path
    if (n_box) *n_box = with_masks ? nb : -nb;                                 // (negative: no mask area behind the boxes)
    return base + (size_t)nb * 16 + (with_masks ? masks : 0);
}
// shared by fid_scrfd_postprocess and the fused pipeline
int scrfd_postprocess_launch(fid_ctx *ctx, const HeadViews &hv, int B, int in_h, int in_w, int A, int img_h, int img_w,
                             float conf, float iou, int max_num, int metric, float *det_dev, float *kps_dev,
                             int32_t *counts_dev, int cap) {
    const int cc = ctx->cand_cap;
    // workspace: counts | keys | records | sorted records
    const size_t off_keys = ((size_t)B * 4 + 255) & ~(size_t)255;
    const size_t off_data = off_keys + (size_t)B * cc * 8;
    const size_t off_sorted = off_data + (size_t)B * cc * CAND_W * 4;
    const size_t total = off_sorted + (size_t)B * cc * CAND_W * 4;
    void *ws;
    FID_TRY(get_scratch(ctx, 0, total, &ws));
    int *cand_count = (int *)ws;
    auto *keys = (unsigned long long *)((char *)ws + off_keys);
    float *data = (float *)((char *)ws + off_data);
    float *sorted = (float *)((char *)ws + off_sorted);
    // letterbox geometry in double, exactly as python does it (scrfd.py:123-134)
    const double im_ratio = (double)img_h / (double)img_w;
    const double model_ratio = (double)in_h / (double)in_w;
    int new_h;
    if (im_ratio > model_ratio) new_h = in_h;
    else new_h = (int)((double)in_w * im_ratio);
    const float det_scale = (float)((double)new_h / (double)img_h);
    FID_HIP(hipMemsetAsync(cand_count, 0, (size_t)B * 4, ctx->stream));
    const int total_anchors = ((in_h / 8) * (in_w / 8) + (in_h / 16) * (in_w / 16) + (in_h / 32) * (in_w / 32)) * A;
    dim3 g1(cdiv(total_anchors, 256), B);
    hipLaunchKernelGGL(decode_compact, g1, dim3(256), 0, ctx->stream, hv, in_h, in_w, A, conf, det_scale, 0, cc, cand_count,
                       keys, data);
    dim3 g2(cdiv(cc, 256), B);
    hipLaunchKernelGGL(rank_scatter, g2, dim3(256), 0, ctx->stream, cand_count, keys, data, sorted, cc);
    int n_box = 0;
    const size_t lds = nms_lds_bytes(cc, &n_box);
    hipLaunchKernelGGL(nms_select, dim3(B), dim3(512), lds, ctx->stream, cand_count, sorted, cc, iou, max_num, metric, img_h,
                       img_w, det_dev, kps_dev, counts_dev, cap, (int *)nullptr, ctx->status_dev, n_box < 0 ? -n_box : n_box, n_box > 0);
    FID_HIP(hipGetLastError());
    ctx->last_out_cap = cap;
    return FID_OK;
}
}  // namespace fid

extern "C" {

int fid_scrfd_set_candidate_capacity(fid_ctx *ctx, int cand_cap) {
    FID_REQUIRE(ctx, "ctx is NULL");
    // LDS of nms_select: ~9 bytes per candidate; 16800 anchors -> 151 KB of the CU's 160 KB
    FID_REQUIRE(cand_cap >= 16 && cand_cap <= 16800, "cand_cap %d outside [16, 16800]", cand_cap);
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    ctx->cand_cap = cand_cap;
    FID_TRY(fid::ensure_dyn_lds(ctx, (const void *)nms_select, (int)fid::nms_lds_bytes(cand_cap, nullptr)));
    return FID_OK;
}

int fid_scrfd_postprocess(fid_ctx *ctx, const float *const head_dev[9], const int32_t pix_stride[9],
                          const int32_t anc_stride[9], const int64_t batch_stride[9], int B, int in_h, int in_w,
                          int num_anchors, int img_h, int img_w, float conf_thres, float iou_thres, int max_num,
                          int metric, float *det_dev, float *kps_dev, int32_t *counts_dev, int cap) {
    FID_REQUIRE(ctx && head_dev && pix_stride && anc_stride && batch_stride, "NULL argument");
    FID_REQUIRE(B > 0 && cap > 0 && det_dev && kps_dev && counts_dev, "bad batch/capacity/output");
    FID_REQUIRE(in_h > 0 && in_w > 0 && in_h % 32 == 0 && in_w % 32 == 0, "input size %dx%d not a multiple of 32", in_w, in_h);
    FID_REQUIRE(img_h > 0 && img_w > 0 && num_anchors > 0, "bad image size / anchors");
    FID_REQUIRE(metric == 0 || metric == 1, "metric must be 0 (max) or 1 (default)");
    HeadViews hv;
    for (int k = 0; k < 9; k++) {
        FID_REQUIRE(head_dev[k], "head %d is NULL", k);
        hv.ptr[k] = head_dev[k];
        hv.pix_stride[k] = pix_stride[k];
        hv.anc_stride[k] = anc_stride[k];
        hv.batch_stride[k] = batch_stride[k];
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    FID_TRY(fid::ensure_dyn_lds(ctx, (const void *)nms_select, (int)fid::nms_lds_bytes(ctx->cand_cap, nullptr)));   // (above 64 KB from ~2000 candidates on)
    return fid::scrfd_postprocess_launch(ctx, hv, B, in_h, in_w, num_anchors, img_h, img_w, conf_thres, iou_thres, max_num,
                                         metric, det_dev, kps_dev, counts_dev, cap);
}

// SCRFD.forward's decode loop alone (scrfd.py:89-119): candidates >= threshold in anchor order
// (levels 8,16,32 concatenated), NOT divided by det_scale.  rec_dev: [B, cand_cap, 16] floats
// (x1 y1 x2 y2 score kps[10] flat_anchor_index-as-int), counts_dev [B].
int fid_scrfd_decode(fid_ctx *ctx, const float *const head_dev[9], const int32_t pix_stride[9], const int32_t anc_stride[9],
                     const int64_t batch_stride[9], int B, int in_h, int in_w, int num_anchors, float conf_thres,
                     float *rec_dev, int32_t *counts_dev) {
    FID_REQUIRE(ctx && head_dev && pix_stride && anc_stride && batch_stride && rec_dev && counts_dev, "NULL argument");
    FID_REQUIRE(B > 0 && in_h > 0 && in_w > 0 && in_h % 32 == 0 && in_w % 32 == 0 && num_anchors > 0, "bad sizes");
    HeadViews hv;
    for (int k = 0; k < 9; k++) {
        FID_REQUIRE(head_dev[k], "head %d is NULL", k);
        hv.ptr[k] = head_dev[k]; hv.pix_stride[k] = pix_stride[k]; hv.anc_stride[k] = anc_stride[k]; hv.batch_stride[k] = batch_stride[k];
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    const int cc = ctx->cand_cap;
    const size_t off_keys = ((size_t)B * 4 + 255) & ~(size_t)255, off_data = off_keys + (size_t)B * cc * 8;
    void *ws;
    FID_TRY(fid::get_scratch(ctx, 0, off_data + (size_t)B * cc * CAND_W * 4, &ws));
    int *cand_count = (int *)ws;
    auto *keys = (unsigned long long *)((char *)ws + off_keys);
    float *data = (float *)((char *)ws + off_data);
    FID_HIP(hipMemsetAsync(cand_count, 0, (size_t)B * 4, ctx->stream));
    const int total = ((in_h / 8) * (in_w / 8) + (in_h / 16) * (in_w / 16) + (in_h / 32) * (in_w / 32)) * num_anchors;
    hipLaunchKernelGGL(decode_compact, dim3(fid::cdiv(total, 256), B), dim3(256), 0, ctx->stream, hv, in_h, in_w, num_anchors,
                       conf_thres, 1.0f, 1, cc, cand_count, keys, data);
    hipLaunchKernelGGL(rank_scatter, dim3(fid::cdiv(cc, 256), B), dim3(256), 0, ctx->stream, cand_count, keys, data, rec_dev, cc);
    FID_HIP(hipMemcpyAsync(counts_dev, cand_count, (size_t)B * 4, hipMemcpyDeviceToDevice, ctx->stream));
    FID_HIP(hipGetLastError());
    return FID_OK;
}

// utils/helpers.py:62-83 / :86-107 on device arrays: points [n,2], distance [n,ncol] -> out [n,ncol]
int fid_distance2bbox(fid_ctx *ctx, const float *points_dev, const float *dist_dev, int n, float *out_dev) {
    FID_REQUIRE(ctx && points_dev && dist_dev && out_dev && n > 0, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    hipLaunchKernelGGL(decode_points, dim3(fid::cdiv(n * 4, 256)), dim3(256), 0, ctx->stream, points_dev, dist_dev, n, 4, out_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}
int fid_distance2kps(fid_ctx *ctx, const float *points_dev, const float *dist_dev, int n, int ncol, float *out_dev) {
    FID_REQUIRE(ctx && points_dev && dist_dev && out_dev && n > 0 && ncol > 0 && ncol % 2 == 0 && ncol != 4, "bad args");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    hipLaunchKernelGGL(decode_points, dim3(fid::cdiv(n * ncol, 256)), dim3(256), 0, ctx->stream, points_dev, dist_dev, n, ncol, out_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

int fid_scrfd_check(fid_ctx *ctx, int *max_candidates) {
    FID_REQUIRE(ctx, "ctx is NULL");
    int st[2] = {0, 0};
    FID_HIP(hipMemcpyAsync(st, ctx->status_dev, 8, hipMemcpyDeviceToHost, ctx->stream));
    FID_HIP(hipStreamSynchronize(ctx->stream));
    FID_HIP(hipMemsetAsync(ctx->status_dev, 0, 8, ctx->stream));
    if (max_candidates) *max_candidates = st[0];
    if (st[0] > ctx->cand_cap) {
        fid::set_error("a frame produced %d candidates >= conf_thres but the workspace holds %d "
                       "(fid_scrfd_set_candidate_capacity)", st[0], ctx->cand_cap);
        return FID_E_CAPACITY;
    }
    if (st[1] > ctx->last_out_cap) {
        fid::set_error("a frame kept %d detections but the output capacity is %d", st[1], ctx->last_out_cap);
        return FID_E_CAPACITY;
    }
    return FID_OK;
}

// SCRFD.nms(dets, iou_thres) (scrfd.py:180-207) on a plain [K,5] array: keep indices into dets.
int fid_nms(fid_ctx *ctx, const float *dets_dev, int K, float iou_thres, int32_t *keep_dev, int32_t *count_dev) {
    FID_REQUIRE(ctx && keep_dev && count_dev && K >= 0, "bad args");
    FID_REQUIRE(K <= 16800, "K=%d exceeds 16800", K);
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    if (K == 0) { FID_HIP(hipMemsetAsync(count_dev, 0, 4, ctx->stream)); return FID_OK; }
    FID_REQUIRE(dets_dev, "dets is NULL");
    const int cc = K;
    const size_t off_keys = 256, off_data = off_keys + (size_t)cc * 8, off_sorted = off_data + (size_t)cc * CAND_W * 4;
    void *ws;
    FID_TRY(fid::get_scratch(ctx, 0, off_sorted + (size_t)cc * CAND_W * 4, &ws));
    int *cand_count = (int *)ws;
    auto *keys = (unsigned long long *)((char *)ws + off_keys);
    float *data = (float *)((char *)ws + off_data);
    float *sorted = (float *)((char *)ws + off_sorted);
    hipLaunchKernelGGL(load_dets, dim3(fid::cdiv(K, 256)), dim3(256), 0, ctx->stream, dets_dev, K, cand_count, keys, data);
    hipLaunchKernelGGL(rank_scatter, dim3(fid::cdiv(cc, 256), 1), dim3(256), 0, ctx->stream, cand_count, keys, data, sorted, cc);
    int n_box = 0;
    const size_t lds = fid::nms_lds_bytes(cc, &n_box);
    FID_TRY(fid::ensure_dyn_lds(ctx, (const void *)nms_select, (int)std::max<size_t>(lds, fid::nms_lds_bytes(ctx->cand_cap, nullptr))));
    hipLaunchKernelGGL(nms_select, dim3(1), dim3(512), lds, ctx->stream, cand_count, sorted, cc, iou_thres, 0, 0, 1, 1,
                       (float *)nullptr, (float *)nullptr, count_dev, K, keep_dev, ctx->status_dev + 4, n_box < 0 ? -n_box : n_box, n_box > 0);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // extern "C"
