// Conv-net executor: runs a layer table (lower.py) as a fixed sequence of HIP kernel launches on the
// context's stream.  This is what replaces onnxruntime.InferenceSession.run on the hot path
// (reference models/scrfd.py:83, models/arcface.py:51) together with the blob conversion that
// precedes it (cv2.dnn.blobFromImage(s), scrfd.py:76-82 / arcface.py:44-50), which is fused into
// the first convolution: the stem reads the uint8 BGR pixels directly.
//
// Memory: activations live in a few slot buffers (liveness-planned by lower.py) sized for
// max_batch; weights are one packed blob.  Nothing is allocated on the steady-state path.
#include "net.h"

#include <map>
#include <set>

struct fid_net {
    int n_ops = 0, n_tensors = 0, in_h = 0, in_w = 0, max_batch = 0;
    std::vector<int32_t> ops, tensors;
    void *blob = nullptr;
    size_t blob_bytes = 0;
    std::vector<void *> slots;
    std::vector<size_t> slot_bytes_per_image;
    double macs_per_image = 0;
    int sub_batch = 0;   // images per depth-first pass (0 = whole batch)
    // autotuned kernel choice per conv op and batch size: tuned[op][batch]
    std::vector<std::map<int, fid::ConvPlan>> tuned;
    std::vector<char> tdir;              // per tensor: 1 = last written from the last work item to the first (ConvArgs::rev)
    int alternate = 1;                   // FID_NO_REV=1: every layer walks forward
    std::vector<std::set<int>> plan_ok;   // batches whose tuned[op] entry has been checked against this library's candidates (or was tuned here)
    int autotune = 1;
    bool tuned_now = false;              // a candidate was timed during the current run_all: the flush buffer is given back at its end
    size_t partial_cap = 0;
    hipEvent_t *prof_events = nullptr;
    int n_prof_events = 0;
    // hipGraph replay of the launch sequence: one executable graph per (frame buffer, batch) the caller keeps coming back with
    struct Replay { hipGraphExec_t exec = nullptr; int seen = 0; unsigned long long last_use = 0; const void *partial = nullptr; };
    std::map<std::pair<const void *, int>, Replay> replays;
    unsigned long long run_counter = 0;
    int graphs = 0;      // FID_GRAPH=1 turns the replay on (measured: no gain on this stack)
    // persisted kernel plans (FID_PLAN=file / fid_net_plan_load): keyed by device name + a hash of the layer table, so that
    // two boxes run the SAME kernels and fp32 summation orders and return bit-identical heads / embeddings
    unsigned long long table_hash = 0;
    std::string device_key, plan_path;
    // repacked weight copies per (op, kind), built on first use (repack.hip); freed with the net
    std::map<std::pair<int, int>, void *> alt_w;
};

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---- first conv: uint8 BGR image -> fp16 NHWC, 3x3, stride 1|2, pad 1, fused normalisation --------
// blob = (pixel - 127.5) * scale with BGR->RGB swap (A.4).  The kernel works on the exact integers
// (2*pixel - 255); scale/2, the channel swap and the following BatchNorm are folded into the fp32
// weights by lower.py.  Zero padding pads the *normalised* blob, i.e. contributes exactly 0.
// VALU kernel (K = 27 is too short for the matrix cores to matter: 0.3 % / 0.6 % of the net's MACs);
// weights are wave-uniform -> scalar loads, one FMA per (tap, channel, cout).
template <int CPW, int SPLIT = 4>   // SPLIT waves share a group of 64 pixels, each computing CPW of the couts
__global__ void __launch_bounds__(256) stem_conv3x3(const uint8_t *__restrict__ img, const float *__restrict__ w,
                                                    const float *__restrict__ bias, const float *__restrict__ slope,
                                                    _Float16 *__restrict__ out, int H, int W, int Ho, int Wo, int Cout_p,
                                                    int stride, int act, long long total_pix) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long pix = ((long long)blockIdx.x * (4 / SPLIT) + wv / SPLIT) * 64 + (threadIdx.x & 63);
    const int cg = wv % SPLIT;
    if (pix >= total_pix) return;
    const int hw = Ho * Wo;
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long long)n * hw);
    const int oy = r / Wo, ox = r - oy * Wo;
    const int iy0 = oy * stride - 1, ix0 = ox * stride - 1;
    float x[27];
    const uint8_t *base = img + (size_t)n * H * W * 3;
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int iy = iy0 + dy, ix = ix0 + dx;
            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const uint8_t *p = base + ((size_t)(in ? iy : 0) * W + (in ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) x[(dy * 3 + dx) * 3 + c] = in ? (float)(2 * (int)p[c] - 255) : 0.f;
        }
    }
    _Float16 res[CPW];
#pragma unroll
    for (int c = 0; c < CPW; c++) {
        const int co = cg * CPW + c;
        const float *wr = w + co * 27;
        float acc = bias[co];
#pragma unroll
        for (int k = 0; k < 27; k++) acc = fmaf(x[k], wr[k], acc);
        if (act == ACT_RELU) acc = fmaxf(acc, 0.f);
        else if (act == ACT_PRELU) acc = acc > 0.f ? acc : acc * slope[co];
        res[c] = (_Float16)acc;
    }
    _Float16 *o = out + (size_t)pix * Cout_p + cg * CPW;
#pragma unroll
    for (int c = 0; c < CPW; c += 8) *(half8 *)(o + c) = *(half8 *)(res + c);
}

// ---- first conv of a net on the matrix cores: uint8 frame -> 3x3 conv (K = 27 padded to 32) -> bias/act -> fp16 NHWC ----
// The VALU kernel above does 27 FMAs per output value (31 TFLOP/s on the IResNet stem: 90 us for 64 faces, 5 % of the net).
// Here a workgroup walks 16x16-pixel output tiles: the haloed uint8 patch is fetched as aligned dwords of the frame rows,
// stored as exact integers 2p-255 in fp16 (outside the frame = 0, the blob's zero padding), each wave gathers the K = 27
// operand of 16 pixels at a time (k = dy*9 + dx*3 + c: nine consecutive elements per patch row) for one
// v_mfma_f32_16x16x32_f16 per 16 output channels; results leave through a per-wave LDS transpose as 16-byte stores.
// Weights: the op's fp32 [Cout_p][27] rows (scale and BatchNorm folded by lower.py), rounded to fp16 once per workgroup.
typedef _Float16 sm_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 sm_half4 __attribute__((ext_vector_type(4)));
typedef float sm_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sm_u32x4 __attribute__((ext_vector_type(4)));

template <int NI, int STRIDE>
__global__ void __launch_bounds__(256) stem_conv_mfma(const uint8_t *__restrict__ img, const float *__restrict__ w,
                                                      const float *__restrict__ bias, const float *__restrict__ slope,
                                                      _Float16 *__restrict__ out, int H, int W, int Ho, int Wo, int act,
                                                      int tiles_x, int tiles_y, int n_tiles, int ablate) {
    constexpr int CP = NI * 16, TS = 16;                     // output channels (padded), tile edge
    constexpr int PR = (TS - 1) * STRIDE + 3;                // patch rows = patch pixel columns (18 / 33)
    constexpr int ND = (PR * 3 + 1 + 3) / 4;                 // dwords per patch row: 1 byte of lead-in (the window is dword aligned)
    constexpr int RSH = ND * 4;                              // halfs per patch row in LDS
    constexpr int OROWB = CP * 2, CPP = OROWB / 16;          // bytes / 16-byte chunks per output pixel
    __shared__ __attribute__((aligned(16))) _Float16 sIn[PR * RSH + 8];
    __shared__ __attribute__((aligned(16))) _Float16 sWt[CP * 32];
    __shared__ __attribute__((aligned(16))) char sSt[4 * 4 * TS * OROWB];         // per wave: its four output rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    for (int i = tid; i < CP * 32; i += 256) {
        const int co = i >> 5, k = i & 31;
        sWt[i] = k < 27 ? (_Float16)w[co * 27 + k] : (_Float16)0.f;
    }
    if (tid < 8) sIn[PR * RSH + tid] = (_Float16)0.f;         // the "k >= 27" element
    __syncthreads();
    sm_half8 wf[NI];
    sm_f32x4 bv[NI], sl[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        wf[ni] = *(const sm_half8 *)(sWt + (ni * 16 + frow) * 32 + fq * 8);
        bv[ni] = bias ? *(const sm_f32x4 *)(bias + ni * 16 + fq * 4) : sm_f32x4{0.f, 0.f, 0.f, 0.f};
        sl[ni] = act == ACT_PRELU ? *(const sm_f32x4 *)(slope + ni * 16 + fq * 4) : sm_f32x4{1.f, 1.f, 1.f, 1.f};
    }
    int koff[8];                                              // k = fq*8 + j -> element offset from the pixel's patch base
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = fq * 8 + j, dy = k / 9, rem = k - dy * 9;
        koff[j] = k < 27 ? dy * RSH + rem : -1;
    }
    const int tiles_per_img = tiles_x * tiles_y, rowbytes = W * 3;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n = tile / tiles_per_img, r = tile - n * tiles_per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int iy0 = ty * TS * STRIDE - 1, ix0 = tx * TS * STRIDE - 1;
        const int bx0 = ix0 * 3 - 1;                          // dword-aligned window start (48*STRIDE*tx - 4)
        const uint8_t *base = img + (size_t)n * H * rowbytes;
        // window element e (= byte of the dword window) is channel (e-1)%3 of patch column (e-1)/3: inside the frame for e in [elo, ehi)
        const int elo = 1 + 3 * max(0, -ix0), ehi = 1 + 3 * min(PR, W - ix0);
        for (int d = tid; d < PR * ND; d += 256) {
            const int pr = d / ND, dc = d - pr * ND;
            const int iy = iy0 + pr, bx = bx0 + dc * 4;
            const bool rin = (unsigned)iy < (unsigned)H;
            unsigned v4 = 0u;
            if (rin && bx >= 0 && bx + 4 <= rowbytes && !(ablate & 2)) v4 = *(const unsigned *)(base + (size_t)iy * rowbytes + bx);
            sm_half4 h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = dc * 4 + j, v = (v4 >> (8 * j)) & 0xFF;
                h[j] = (rin && e >= elo && e < ehi) ? (_Float16)(float)(2 * v - 255) : (_Float16)0.f;
            }
            *(sm_half4 *)(sIn + pr * RSH + dc * 4) = h;
        }
        __syncthreads();
        char *sS = sSt + wave * (4 * TS * OROWB);
        // the wave's four output rows: all K gathers first, then the products, then the staged rows leave as 16-byte stores.  A pixel's
        // 16-byte chunks are rotated by the pixel index in the staging rows (row pitch = all 32 banks: unrotated, 16 pixels hit one bank).
        sm_half8 pf[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int pbase = ((wave * 4 + rr) * STRIDE) * RSH + 1 + frow * STRIDE * 3;
#pragma unroll
            for (int j = 0; j < 8; j++) pf[rr][j] = sIn[koff[j] >= 0 ? pbase + koff[j] : PR * RSH];
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                sm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[rr], acc, 0, 0, 0);
                sm_f32x4 v = acc + bv[ni];
                if (act == ACT_RELU) {
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.f);
                } else if (act == ACT_PRELU) {
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = v[i] > 0.f ? v[i] : v[i] * sl[ni][i];
                }
                const int ch = ni * 2 + (fq >> 1);              // 16-byte chunk of the pixel row
                *(sm_half4 *)(sS + (rr * TS + frow) * OROWB + (((ch + frow) % CPP) << 4) + (fq & 1) * 8) = __builtin_convertvector(v, sm_half4);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = lane; g < 4 * TS * CPP; g += 64) {
            const int p = g / CPP, c = g - p * CPP;             // p = rr * 16 + pixel
            const sm_u32x4 o = *(const sm_u32x4 *)(sS + p * OROWB + (((c + (p & 15)) % CPP) << 4));
            const int oy = ty * TS + wave * 4 + (p >> 4), ox = tx * TS + (p & 15);
            if (oy < Ho && ox < Wo && !(ablate & 1)) *(sm_u32x4 *)((char *)out + (((size_t)n * Ho + oy) * Wo + ox) * OROWB + c * 16) = o;
        }
        __syncthreads();                                      // everyone is done with the patch
    }
}

// ---- max pool k x k / stride / pad on NHWC fp16, 8 channels (16 B) per thread ----------------------
__global__ void __launch_bounds__(256) maxpool_nhwc(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int H, int W,
                                                    int Ho, int Wo, int Cp, int k, int stride, int pad, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = Cp >> 3;
    const int cg = (int)(idx % c8);
    const long long pix = idx / c8;
    const int hw = Ho * Wo;
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long long)n * hw);
    const int oy = r / Wo, ox = r - oy * Wo;
    half8 m;
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = (_Float16)(-65504.f);
    for (int dy = 0; dy < k; dy++) {
        const int iy = oy * stride - pad + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int dx = 0; dx < k; dx++) {
            const int ix = ox * stride - pad + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const half8 v = *(const half8 *)(in + ((size_t)(n * H + iy) * W + ix) * Cp + cg * 8);
#pragma unroll
            for (int i = 0; i < 8; i++) m[i] = v[i] > m[i] ? v[i] : m[i];
        }
    }
    *(half8 *)(out + (size_t)pix * Cp + cg * 8) = m;
}

// ---- depthwise k x k conv on NHWC fp16 (SCRFD-500M / MobileFaceNet), 8 channels per thread ---------
// weights fp32 [k*k][Cp] (BatchNorm folded), bias fp32 [Cp]; HBM-bound: k*k MACs per loaded element.
__global__ void __launch_bounds__(256) dwconv_nhwc(const _Float16 *__restrict__ in, const float *__restrict__ w,
                                                   const float *__restrict__ bias, const float *__restrict__ slope,
                                                   _Float16 *__restrict__ out, int H, int W, int Ho, int Wo, int Cp, int k,
                                                   int stride, int pad, int act, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c8 = Cp >> 3;
    const int cg = (int)(idx % c8);
    const long long pix = idx / c8;
    const int hw = Ho * Wo;
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long long)n * hw);
    const int oy = r / Wo, ox = r - oy * Wo;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = bias[cg * 8 + i];
    for (int dy = 0; dy < k; dy++) {
        const int iy = oy * stride - pad + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int dx = 0; dx < k; dx++) {
            const int ix = ox * stride - pad + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const half8 v = *(const half8 *)(in + ((size_t)(n * H + iy) * W + ix) * Cp + cg * 8);
            const float *wr = w + (size_t)(dy * k + dx) * Cp + cg * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = fmaf((float)v[i], wr[i], acc[i]);
        }
    }
    half8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float a = acc[i];
        if (act == ACT_RELU) a = fmaxf(a, 0.f);
        else if (act == ACT_PRELU) a = a > 0.f ? a : a * slope[cg * 8 + i];
        o[i] = (_Float16)a;
    }
    *(half8 *)(out + (size_t)pix * Cp + cg * 8) = o;
}

// ---- depthwise 3x3 / stride 1 / pad 1 with the input tile in LDS ("LDS halo tiles, half8 reads"): a workgroup takes a tile of
// 14 rows x 16 columns x 64 channels; its haloed 16 x 18 input patch (37 KB) is fetched once (16-byte loads, every input byte once per tile
// instead of nine times through L1), thread = (8-channel group, column, row half) walks its 7 output rows with a three-row window in
// registers (three 16-byte LDS reads per output instead of nine global ones).  fp32 fmaf chain in dwconv_nhwc's (dy, dx) order from the
// bias, taps outside the image SKIPPED: bit-identical to dwconv_nhwc.  MobileFaceNet's conv2_dw (128 x 56 x 56, 32 faces): 41 -> 27.6 us.
constexpr int DW_TR = 14, DW_TC = 16, DW_PR = DW_TR + 2, DW_PC = DW_TC + 2;
__global__ void __launch_bounds__(256) dwconv3x3_lds(const _Float16 *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                     const float *__restrict__ slope, _Float16 *__restrict__ out, int H, int W, int Cp, int act,
                                                     int tiles_x, int tiles_y, int cblks) {
    __shared__ __attribute__((aligned(16))) char sP[DW_PR * DW_PC * 128];
    const int tid = threadIdx.x;
    int item = blockIdx.x;
    const int cb = item % cblks; item /= cblks;
    const int tx = item % tiles_x; item /= tiles_x;
    const int ty = item % tiles_y, n = item / tiles_y;
    const int y0 = ty * DW_TR - 1, x0 = tx * DW_TC - 1, c0 = cb * 64;
    // patch: 16 x 18 pixels x 8 chunks of 16 bytes; pixels outside the image are never read back (their taps are skipped): left as they are
    for (int i = tid; i < DW_PR * DW_PC * 8; i += 256) {
        const int p = i >> 3, c = i & 7;
        const int py = p / DW_PC, px = p - py * DW_PC;
        const int iy = y0 + py, ix = x0 + px;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            *(uint4 *)(sP + p * 128 + c * 16) = *(const uint4 *)(in + ((size_t)(n * H + iy) * W + ix) * Cp + c0 + c * 8);
    }
    const int cg = tid & 7, col = (tid >> 3) & 15, rh = tid >> 7;          // 8-channel group, tile column, row half (7 rows each)
    float wv[9][8], bv[8], sv[8];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int i = 0; i < 8; i++) wv[t][i] = w[(size_t)t * Cp + c0 + cg * 8 + i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        bv[i] = bias[c0 + cg * 8 + i];
        sv[i] = act == ACT_PRELU ? slope[c0 + cg * 8 + i] : 1.f;
    }
    __syncthreads();
    const int ox = tx * DW_TC + col;
    const bool okc[3] = {(unsigned)(ox - 1) < (unsigned)W, ox < W, (unsigned)(ox + 1) < (unsigned)W};
    if (ox >= W) return;
    half8 win[3][3];                                            // [patch row (mod 3)][dx]
    const char *base = sP + col * 128 + cg * 16;
    auto load_row = [&](int pr, int slot) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++) win[slot][dx] = *(const half8 *)(base + (pr * DW_PC + dx) * 128);
    };
    const int r0 = rh * (DW_TR / 2);
    load_row(r0, 0); load_row(r0 + 1, 1);
#pragma unroll
    for (int r = 0; r < DW_TR / 2; r++) {
        load_row(r0 + r + 2, (r + 2) % 3);
        const int oy = ty * DW_TR + r0 + r;
        if (oy >= H) break;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = bv[i];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            if ((unsigned)(oy - 1 + dy) >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                if (!okc[dx]) continue;
                const half8 v = win[(r + dy) % 3][dx];
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = fmaf((float)v[i], wv[dy * 3 + dx][i], acc[i]);
            }
        }
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float a = acc[i];
            if (act == ACT_RELU) a = fmaxf(a, 0.f);
            else if (act == ACT_PRELU) a = a > 0.f ? a : a * sv[i];
            o[i] = (_Float16)a;
        }
        *(half8 *)(out + ((size_t)(n * H + oy) * W + ox) * Cp + c0 + cg * 8) = o;
    }
}

// ---- global depthwise conv (k x k "valid" on a k x k map -> 1 x 1: MobileFaceNet's GDC, 7 x 7 x 512): eight lanes per (image, 8-channel group),
// lane r sums tap row r (k taps, fp32 fmaf chain from 0), lane 0 then adds bias + the rows in row order (a fixed order: deterministic; NOT
// dwconv_nhwc's single 49-tap chain, which ran as 2048 threads of 49 dependent steps = 20 us for 0.1 MFLOP).
__global__ void __launch_bounds__(256) gdc_rows(const _Float16 *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                const float *__restrict__ slope, _Float16 *__restrict__ out, int k, int Cp, int act, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = idx >> 3, r = idx & 7;                        // (image, channel group), tap row
    const int c8 = Cp >> 3;
    const bool live = g < total;
    const int n = live ? g / c8 : 0, cg = live ? g - n * c8 : 0;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.f;
    if (live && r < k) {
        for (int dx = 0; dx < k; dx++) {
            const half8 v = *(const half8 *)(in + ((size_t)(n * k + r) * k + dx) * Cp + cg * 8);
            const float *wr = w + (size_t)(r * k + dx) * Cp + cg * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = fmaf((float)v[i], wr[i], acc[i]);
        }
    }
    float sum[8];
#pragma unroll
    for (int i = 0; i < 8; i++) sum[i] = live ? bias[cg * 8 + i] : 0.f;
    for (int rr = 0; rr < k; rr++)                              // rows in order, read from the lane that holds them (k <= 8)
#pragma unroll
        for (int i = 0; i < 8; i++) sum[i] += __shfl(acc[i], (threadIdx.x & 63 & ~7) + rr, 64);
    if (live && r == 0) {
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float a = sum[i];
            if (act == ACT_RELU) a = fmaxf(a, 0.f);
            else if (act == ACT_PRELU) a = a > 0.f ? a : a * slope[cg * 8 + i];
            o[i] = (_Float16)a;
        }
        *(half8 *)(out + (size_t)n * Cp + cg * 8) = o;
    }
}

// reads `n16` 16-byte words (tuning only: brings a layer's input back into L2 / the Infinity Cache after the cache flush, where the
// producing layer would have left it)
__global__ void __launch_bounds__(256) touch_kernel(const uint4 *__restrict__ p, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u && sink) *sink = acc;     // (practically never)
}

struct TensorView {
    void *ptr;
    int C, Cp, H, W, dtype;
};

TensorView view(const fid_net *net, int id, int first = 0) {
    const int32_t *t = &net->tensors[(size_t)id * FID_TENSOR_WORDS];
    const size_t per_image = (size_t)t[T_H] * t[T_W] * t[T_CP] * (t[T_DTYPE] == 1 ? 4 : 2);
    return TensorView{(char *)net->slots[t[T_SLOT]] + per_image * first, t[T_C], t[T_CP], t[T_H], t[T_W], t[T_DTYPE]};
}

// bump when conv_candidates() / the kernels' tile meanings change: it is hashed into the plan key
constexpr int FID_PLAN_REV = 5;

unsigned long long fnv1a(const void *p, size_t n, unsigned long long h = 1469598103934665603ull) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// one line per pick:  <device>|<table hash>|<op>|<batch>|gen bm bn bk ksplit ns partial_bytes
void plan_line(const fid_net *net, int oi, int batch, const ConvPlan &c, char *buf, size_t cap) {
    snprintf(buf, cap, "%s|%016llx|%d|%d|%d %d %d %d %d %d %zu\n", net->device_key.c_str(), net->table_hash, oi, batch, c.gen, c.bm, c.bn,
             c.bk, c.ksplit, c.ns, c.partial_bytes);
}

int plan_load(fid_net *net, const char *path, int *n_loaded) {
    FILE *f = fopen(path, "r");
    int n = 0;
    if (f) {
        char line[512];
        while (fgets(line, sizeof(line), f)) {
            char dev[256];
            unsigned long long h = 0;
            int oi = 0, batch = 0;
            ConvPlan c{};
            if (sscanf(line, "%255[^|]|%llx|%d|%d|%d %d %d %d %d %d %zu", dev, &h, &oi, &batch, &c.gen, &c.bm, &c.bn, &c.bk, &c.ksplit, &c.ns,
                       &c.partial_bytes) != 11)
                continue;
            if (h != net->table_hash || net->device_key != dev || oi < 0 || oi >= net->n_ops || batch <= 0) continue;
            net->tuned[oi][batch] = c;      // later lines win (a re-tuned pick is appended); checked against the candidates at first use
            net->plan_ok[oi].erase(batch);
            n++;
        }
        fclose(f);
    }
    if (n_loaded) *n_loaded = n;
    return FID_OK;
}

bool same_kernel(const ConvPlan &x, const ConvPlan &y) {
    return x.gen == y.gen && x.bm == y.bm && x.bn == y.bn && x.bk == y.bk && x.ksplit == y.ksplit && x.ns == y.ns;
}

// repacked weight copies of op `oi` that no tuned plan of the op uses any more (the autotuner builds one per packing it times)
void drop_unused_alt_weights(fid_ctx *ctx, fid_net *net, int oi) {
    bool used[4] = {false, false, false, false};
    for (const auto &kv : net->tuned[oi]) used[plan_alt_kind(kv.second) & 3] = true;
    bool synced = false;
    for (auto it = net->alt_w.begin(); it != net->alt_w.end();) {
        if (it->first.first == oi && !used[it->first.second & 3]) {
            if (!synced) { (void)hipStreamSynchronize(ctx->stream); synced = true; }
            (void)hipFree(it->second);
            it = net->alt_w.erase(it);
        } else ++it;
    }
}

// ConvArgs::w_alt for `plan` (built on the context's stream the first time an op needs that packing)
int set_alt_weights(fid_ctx *ctx, fid_net *net, int oi, ConvArgs &a, const ConvPlan &plan) {
    const int kind = plan_alt_kind(plan);
    a.w_alt = nullptr;
    if (kind == 0) return FID_OK;
    auto key = std::make_pair(oi, kind);
    auto it = net->alt_w.find(key);
    if (it == net->alt_w.end()) {
        void *p = nullptr;
        FID_HIP(hipMalloc(&p, repack_bytes(kind, a.w_rows, a.Cin_p, a.kh * a.kw) + 256));      // (w_rows = Cout_p except for the fused shortcut + conv op)
        const int rc = repack_weights(ctx, kind, a.w, p, a.w_rows, a.Cin_p, a.kh * a.kw);
        if (rc != FID_OK) { (void)hipFree(p); return rc; }
        it = net->alt_w.emplace(key, p).first;
    }
    a.w_alt = it->second;
    return FID_OK;
}

// one op on images [first, first + batch) of the resident batch
int run_op(fid_ctx *ctx, fid_net *net, int oi, const uint8_t *images, int first, int batch, void *partial_ws) {
    const int32_t *op = &net->ops[(size_t)oi * FID_OP_WORDS];
    const char *blob = (const char *)net->blob;
    static const bool klog = getenv("FID_KLOG") != nullptr;
    if (klog) fprintf(stderr, "[klog] op %d\n", oi);
    const TensorView dst = view(net, op[W_DST], first);
    images += (size_t)first * net->in_h * net->in_w * 3;
    const float *bias = op[W_BOFF] >= 0 ? (const float *)(blob + op[W_BOFF]) : nullptr;
    const float *slope = op[W_SOFF] >= 0 ? (const float *)(blob + op[W_SOFF]) : nullptr;
    if (op[W_TYPE] != OP_CONV && op[W_TYPE] != OP_BBLOCK) net->tdir[op[W_DST]] = 0;
    switch (op[W_TYPE]) {
        case OP_STEM: {
            const long long total = (long long)batch * dst.H * dst.W;
            const int cpw = dst.Cp / 4;
            const float *w = (const float *)(blob + op[W_WOFF]);
            dim3 grid((unsigned)cdiv64(total, 64));
            static const bool valu = getenv("FID_STEM_VALU") != nullptr;   // the VALU kernel stays for comparison
            const int st = op[W_STRIDE];
            if (!valu && net->in_w % 4 == 0 && (st == 1 || st == 2) && (dst.Cp == 16 || dst.Cp == 32 || dst.Cp == 64 || dst.Cp == 128)) {
                const int tx = cdiv(dst.W, 16), ty = cdiv(dst.H, 16), nt = batch * tx * ty;
                static const int abl = getenv("FID_STEMM_ABLATE") ? atoi(getenv("FID_STEMM_ABLATE")) : 0;      // timing experiments (wrong results): 1 no stores, 2 no frame loads
                // workgroups per CU = what fits (64 couts: 146 VGPRs, three waves per SIMD; measured on IResNet's stem at 64 faces: 3 -> 33.9 us, 4 -> 39.8, 6 -> 39.3;
                // without any load or store 29.4: the K gather and the fp32 epilogue are the time, not the 103 MB of output)
                const int g = std::min(nt, ctx->num_cus * (dst.Cp == 128 ? 2 : (dst.Cp == 64 ? 3 : 4)));
#define STEMM(NI, ST) hipLaunchKernelGGL((stem_conv_mfma<NI, ST>), dim3(g), dim3(256), 0, ctx->stream, images, w, bias, slope, (_Float16 *)dst.ptr, net->in_h, net->in_w, dst.H, dst.W, op[W_ACT], tx, ty, nt, abl)
                if (dst.Cp == 128) { if (st == 1) STEMM(8, 1); else STEMM(8, 2); }            // (MobileFaceNet's first conv: 3 -> 128 channels, stride 2)
                else if (dst.Cp == 64) { if (st == 1) STEMM(4, 1); else STEMM(4, 2); }
                else if (dst.Cp == 32) { if (st == 1) STEMM(2, 1); else STEMM(2, 2); }
                else { if (st == 1) STEMM(1, 1); else STEMM(1, 2); }
#undef STEMM
                break;
            }
            static const int split = getenv("FID_STEM_SPLIT") ? atoi(getenv("FID_STEM_SPLIT")) : 2;   // measured on IResNet stem: 4 -> 107 us, 2 -> 91 us, 1 -> 102 us
#define STEM(CPW) hipLaunchKernelGGL(stem_conv3x3<CPW>, grid, dim3(256), 0, ctx->stream, images, w, bias, slope, (_Float16 *)dst.ptr, net->in_h, net->in_w, dst.H, dst.W, dst.Cp, op[W_STRIDE], op[W_ACT], total)
#define STEMS(CPW, SP) hipLaunchKernelGGL((stem_conv3x3<CPW, SP>), dim3((unsigned)cdiv64(total, 64 * (4 / SP))), dim3(256), 0, ctx->stream, images, w, bias, slope, (_Float16 *)dst.ptr, net->in_h, net->in_w, dst.H, dst.W, dst.Cp, op[W_STRIDE], op[W_ACT], total)
            if (split == 1 && dst.Cp == 64) STEMS(64, 1);
            else if (split == 2 && dst.Cp == 64) STEMS(32, 2);
            else if (split == 1 && dst.Cp == 32) STEMS(32, 1);
            else if (cpw == 8) STEM(8);
            else if (cpw == 16) STEM(16);
            else if (cpw == 32) STEM(32);
            else { set_error("stem: Cout_p=%d unsupported", dst.Cp); return FID_E_INVALID; }
#undef STEM
#undef STEMS
            break;
        }
        case OP_CONV: {
            if (op[W_X_SRC2] == 0 && op[W_X_W2OFF] > 0) {        // a block's shortcut conv that its consumer may absorb (lower.py): word 29 = that conv's op index + 1
                const int cons = op[W_X_W2OFF] - 1;
                auto itc = net->tuned[cons].find(batch);
                if (itc != net->tuned[cons].end() && itc->second.gen == 12 && net->plan_ok[cons].count(batch)) break;   // absorbed at this batch size: nothing to do
            }
            const TensorView src = view(net, op[W_SRC], first);
            ConvArgs a{}, af{};
            bool has_sc = false;
            a.in = src.ptr;
            a.w = blob + op[W_WOFF];
            a.bias = bias;
            a.slope = slope;
            a.out = dst.ptr;
            if (op[W_X_DST2] > 0) {                              // fused shortcut + stride-2 conv: second output, same shape
                const TensorView d2 = view(net, op[W_X_DST2] - 1, first);
                FID_REQUIRE(d2.H == dst.H && d2.W == dst.W && d2.Cp == dst.Cp && d2.dtype == dst.dtype && op[W_X_ACT2] == ACT_NONE &&
                            op[W_X_COUT1P] == dst.Cp, "op %d: bad fused shortcut record", oi);
                a.out2 = d2.ptr;
            }
            a.H = src.H; a.W = src.W; a.Cin_p = src.Cp;
            a.Ho = dst.H; a.Wo = dst.W; a.Cout_p = dst.Cp;
            a.w_rows = op[W_WROWS];
            a.kh = op[W_KH]; a.kw = op[W_KW]; a.stride = op[W_STRIDE]; a.pad = op[W_PAD];
            a.M = batch * dst.H * dst.W;
            a.act = op[W_ACT]; a.flags = op[W_FLAGS]; a.nsig = op[W_NSIG];
            if (dst.dtype == 1) a.flags |= CF_OUT_F32;
            if (op[W_RES] >= 0) {
                const TensorView r = view(net, op[W_RES], first);
                a.res = r.ptr; a.res_H = r.H; a.res_W = r.W; a.res_Cp = r.Cp;
            }
            a.in_bytes = (unsigned)((size_t)batch * src.H * src.W * src.Cp * 2);
            a.w_bytes = (unsigned)op[W_WBYTES];
            if (op[W_X_SRC2] > 0) {                              // the block's shortcut conv as extra K-steps on the block input
                const TensorView x2 = view(net, op[W_X_SRC2] - 1, first);
                FID_REQUIRE(x2.dtype == 0 && op[W_X_T2] >= 1 && op[W_X_KW2] >= 1 && op[W_X_S2] >= 1 && op[W_X_DST2] == 0 &&
                            (dst.H - 1) * op[W_X_S2] + (op[W_X_T2] / op[W_X_KW2] - 1) < x2.H && (dst.W - 1) * op[W_X_S2] + op[W_X_KW2] - 1 < x2.W,
                            "op %d: bad fused shortcut-conv record", oi);
                FID_REQUIRE(op[W_X_W2OFF] > 0 && op[W_X_B2OFF] > 0 && op[W_X_SCOP] > 0 && op[W_X_SCOP] <= oi && op[W_RES] >= 0, "op %d: fused shortcut-conv record without its second image", oi);
                // the fused form (a generation-12 pick): second weight image [kh*kw * Cin_p | T2 * Cin2_p], summed bias, no residual
                af = a;
                af.in2 = x2.ptr; af.H2 = x2.H; af.W2 = x2.W; af.Cin2_p = x2.Cp; af.T2 = op[W_X_T2]; af.kw2 = op[W_X_KW2]; af.s2 = op[W_X_S2];
                af.in2_bytes = (unsigned)((size_t)batch * x2.H * x2.W * x2.Cp * 2);
                af.w = blob + op[W_X_W2OFF];
                af.bias = (const float *)(blob + op[W_X_B2OFF]);
                af.w_bytes = (unsigned)((size_t)op[W_WROWS] * (op[W_KH] * op[W_KW] * src.Cp + op[W_X_T2] * x2.Cp) * 2);
                af.res = nullptr;
                has_sc = true;
            }
            a.partial = af.partial = (float *)partial_ws;
            a.rev = af.rev = net->alternate && op[W_SRC] >= 0 && !net->tdir[op[W_SRC]];
            // every candidate of this op: the plain conv's, and (generation 12) generation 2's with the shortcut as extra K-steps
            auto all_candidates = [&]() {
                std::vector<ConvPlan> v = conv_candidates(a, ctx->num_cus, true);
                static const bool no_sc = getenv("FID_NO_SC_RUNTIME") != nullptr;
                if (has_sc && !no_sc)
                    for (ConvPlan c : conv_candidates(af, ctx->num_cus, true)) { c.gen = 12; v.push_back(c); }   // (ns = 10: conv3x3_s2's form, else generation 2's)
                return v;
            };
            auto launch_plan = [&](const ConvPlan &c) -> int {
                if (c.gen != 12) return conv_launch(ctx, a, c);
                ConvPlan c2 = c;
                c2.gen = c.ns == 10 ? 10 : 2;
                af.w_alt = a.w_alt;                              // (the 3x3 part in fragment order: set_alt_weights built it from the plain first image)
                return conv_launch(ctx, af, c2);
            };
            ConvPlan plan;
            auto &cache = net->tuned[oi];
            auto it = cache.find(batch);
            if (it != cache.end() && !net->plan_ok[oi].count(batch)) {
                // a pick that came from a plan file (FID_PLAN / fid_net_plan_load) is text: before its first use it must name one of
                // THIS library's candidates for the op (another revision's candidate set, a hand-edited line) and its split-K
                // workspace -- recomputed here, not taken from the file -- must fit the scratch this run sized
                bool ok = false;
                for (const ConvPlan &c : all_candidates())
                    if (same_kernel(c, it->second) && c.partial_bytes <= net->partial_cap && (c.ksplit == 1 || partial_ws)) { it->second = c; ok = true; break; }
                if (!ok && it->second.ksplit == 1) {             // (conv_plan's heuristic pick is not always in the candidate list)
                    const ConvPlan h = conv_plan(a, ctx->num_cus, false);
                    if (same_kernel(h, it->second)) { it->second = h; ok = true; }
                }
                if (ok) net->plan_ok[oi].insert(batch);
                else { cache.erase(it); it = cache.end(); }
            }
            if (it != cache.end()) {
                plan = it->second;
            } else if (net->autotune && partial_ws) {
                // first time this op runs at this batch size: time every candidate kernel on the real
                // operands (the op is idempotent) and keep the fastest; a split-K plan must win by 10 %
                // to be preferred (its summation order differs from the unsplit kernels)
                std::vector<ConvPlan> cands = all_candidates();
                if (const char *fg = getenv("FID_FORCE_GEN")) {      // tests: exercise one kernel family wherever it applies
                    std::vector<ConvPlan> only;
                    for (const ConvPlan &c : cands)
                        if (c.gen == atoi(fg)) only.push_back(c);
                    if (!only.empty()) cands = only;
                }
                if (const char *fn = getenv("FID_FORCE_NS")) {       // tests: one ring variant of generation 2 / 5
                    std::vector<ConvPlan> only;
                    for (const ConvPlan &c : cands)
                        if (((c.gen == 2 || c.gen == 5 || c.gen == 12) && c.ns == atoi(fn)) || (c.gen == 9 && (c.ns == 1 ? 3 : (c.ns == 4 ? 4 : (c.ns == 6 ? (c.bm == 512 ? 7 : 6) : (c.ns >= 7 ? c.ns + 1 + (c.ns == 8 && c.bm == 512 ? 20 : 0) + (c.ns == 8 && c.bm == 256 && c.bn == 128 ? 30 : 0) : c.bm / 256)))) == atoi(fn))) only.push_back(c);
                    if (!only.empty()) cands = only;
                }
                const float pc2_bias = getenv("FID_PC2_BIAS") ? (float)atof(getenv("FID_PC2_BIAS")) : 1.f;
                hipEvent_t e0, e1;
                FID_HIP(hipEventCreate(&e0));
                FID_HIP(hipEventCreate(&e1));
                // Candidates are timed COLD: between two uses of a layer's weights a whole step (1.8 GB of activations of both nets) passes
                // through L2 and the 256 MB Infinity Cache, so in production every launch finds its weights in HBM -- while repetitions of one
                // op find them in L2, which flattered the kernels that re-stream the weights per work item (conv_gw on the 224-channel 20x20
                // layers: 36 us back to back, 55 us in the net).  A 320 MB fill before every timed repetition restores the production state.
                static const bool cold_tune = !getenv("FID_TUNE_HOT");
                void *flush = nullptr;
                constexpr size_t FLUSH_BYTES = 320ull << 20;
                if (cold_tune) { FID_TRY(get_scratch(ctx, 4, FLUSH_BYTES, &flush)); net->tuned_now = true; }   // (slot 4: released when this run ends, run_all)
                float best = 1e30f;
                if (cands.empty()) { set_error("op %d: no kernel candidate", oi); return FID_E_STATE; }
                plan = cands[0];
                // a plain pick also costs the shortcut conv's launch (it ran a moment ago with its own pick): timed here under the SAME cold
                // protocol as the candidates (fill, re-touch of its input = the block input, first repetition dropped) and added to the plain
                // candidates (three back-to-back hot runs made the shortcut look cheaper than it is inside the net)
                float t_sc = 0.f;
                if (has_sc) {
                    t_sc = 1e30f;
                    for (int rep = 0; rep < 4; rep++) {
                        if (flush) {
                            FID_HIP(hipMemsetAsync(flush, rep & 1, FLUSH_BYTES, ctx->stream));
                            if (af.in2_bytes <= (128u << 20))
                                hipLaunchKernelGGL(touch_kernel, dim3(ctx->num_cus * 4), dim3(256), 0, ctx->stream, (const uint4 *)af.in2, (size_t)af.in2_bytes / 16, (unsigned *)nullptr);
                        }
                        FID_HIP(hipEventRecord(e0, ctx->stream));
                        FID_TRY(run_op(ctx, net, op[W_X_SCOP] - 1, images - (size_t)first * net->in_h * net->in_w * 3, first, batch, partial_ws));
                        FID_HIP(hipEventRecord(e1, ctx->stream));
                        FID_HIP(hipEventSynchronize(e1));
                        float ms = 0;
                        FID_HIP(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0 || !flush) t_sc = std::min(t_sc, ms);
                    }
                }
                for (const ConvPlan &c : cands) {
                    if (c.partial_bytes > net->partial_cap) continue;
                    float tmin = 1e30f;
                    static const int tune_reps = getenv("FID_TUNE_REPS") ? std::max(2, atoi(getenv("FID_TUNE_REPS"))) : 5;
                    for (int rep = 0; rep < tune_reps; rep++) {
                        FID_TRY(set_alt_weights(ctx, net, oi, a, c));
                        if (flush) {
                            FID_HIP(hipMemsetAsync(flush, rep & 1, FLUSH_BYTES, ctx->stream));
                            if (a.in_bytes <= (128u << 20))      // the input a producer wrote a moment ago is still on chip unless it is huge
                                hipLaunchKernelGGL(touch_kernel, dim3(ctx->num_cus * 4), dim3(256), 0, ctx->stream, (const uint4 *)a.in, (size_t)a.in_bytes / 16, (unsigned *)nullptr);
                        }
                        FID_HIP(hipEventRecord(e0, ctx->stream));
                        FID_TRY(launch_plan(c));
                        FID_HIP(hipEventRecord(e1, ctx->stream));
                        FID_HIP(hipEventSynchronize(e1));
                        float ms = 0;
                        FID_HIP(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0) tmin = std::min(tmin, ms);
                    }
                    float score = c.ksplit > 1 ? tmin * 1.1f : tmin;
                    if (has_sc && c.gen != 12) score += t_sc;
                    // FID_TUNE_SHARE = s (0 .. 1; plans for two-lane deployments, tools/make_plan.sh): a launch on a fraction f of the CUs scores
                    // t (1 - s (1 - f)) -- the CUs it leaves free run the other lane's kernels.  Measured on IResNet's 7x7 layers at 64 faces:
                    // conv_ks MOSAIC (128 workgroups, 38 us) against generation 2 (392 workgroups, 36 us): the step is 1.3 % shorter with MOSAIC.
                    static const float tune_share = getenv("FID_TUNE_SHARE") ? (float)atof(getenv("FID_TUNE_SHARE")) : 0.f;
                    if (tune_share > 0.f) score *= 1.f - tune_share * (1.f - conv_plan_cu_share(a, c, ctx->num_cus));
                    if (c.gen == 8) score *= pc2_bias;          // experiments: FID_PC2_BIAS < 1 prefers the two-tile kernel although it is slower alone
                    static const int tune_verbose = getenv("FID_TUNE_LOG") ? atoi(getenv("FID_TUNE_LOG")) : 0;
                    if (tune_verbose >= 2)
                        fprintf(stderr, "[cand] op %d gen %d tile %dx%dx%d ns %d split %d: %.1f us\n", oi, c.gen, c.bm, c.bn, c.bk, c.ns, c.ksplit, tmin * 1e3f);
                    if (score < best) { best = score; plan = c; }
                }
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                cache[batch] = plan;
                net->plan_ok[oi].insert(batch);
                drop_unused_alt_weights(ctx, net, oi);
                if (!net->plan_path.empty()) {                      // persist the pick (one O_APPEND line)
                    if (FILE *pf = fopen(net->plan_path.c_str(), "a")) {
                        char line[512];
                        plan_line(net, oi, batch, plan, line, sizeof(line));
                        fputs(line, pf);
                        fclose(pf);
                    }
                }
                if (getenv("FID_TUNE_LOG"))
                    fprintf(stderr, "[tune] op %d M=%d Cin_p=%d Cout_p=%d k=%d s=%d -> gen %d tile %dx%dx%d ns %d split %d (%.1f us)\n", oi, a.M,
                            a.Cin_p, a.Cout_p, a.kh, a.stride, plan.gen, plan.bm, plan.bn, plan.bk, plan.ns, plan.ksplit, best * 1e3f);
            } else {
                plan = conv_direct_applicable(a) ? ConvPlan{} : conv_plan(a, ctx->num_cus, partial_ws != nullptr);
                if (conv_direct_applicable(a)) { plan.gen = 0; plan.ksplit = 1; }
            }
            FID_TRY(set_alt_weights(ctx, net, oi, a, plan));
            FID_TRY(launch_plan(plan));
            net->tdir[op[W_DST]] = (char)(a.rev && conv_walks_reverse(plan));
            if (op[W_X_DST2] > 0) net->tdir[op[W_X_DST2] - 1] = net->tdir[op[W_DST]];
            break;
        }
        case OP_STEMFUSED: {
            const bool old_stem = getenv("FID_STEM_OLD") != nullptr;   // the flattened-fragment kernel stays for comparison (read per launch: the tests switch it)
            FID_TRY((old_stem ? stem_fused_launch : stem_rows_launch)(ctx, images, batch, net->in_h, net->in_w, blob + op[W_F_W0], (const float *)(blob + op[W_F_B0]),
                                      blob + op[W_F_W1], (const float *)(blob + op[W_F_B1]), blob + op[W_F_W2],
                                      (const float *)(blob + op[W_F_B2]), dst.ptr, dst.Cp));
            break;
        }
        case OP_STEMBLOCK: {
            FID_REQUIRE(dst.Cp == 64 && dst.dtype == 0 && dst.H == net->in_h && dst.W == net->in_w && stem_block_applicable(dst.H, dst.W), "op %d: bad fused stem-block record", oi);
            void *xe = nullptr;
            if (op[W_S_DST2] > 0) {
                const TensorView d2 = view(net, op[W_S_DST2] - 1, first);
                FID_REQUIRE(d2.Cp == 64 && d2.dtype == 0 && d2.H == (dst.H + 1) / 2 && d2.W == (dst.W + 1) / 2, "op %d: second output of the fused stem block", oi);
                xe = d2.ptr;
            }
            FID_TRY(stem_block_launch(ctx, images, batch, dst.H, dst.W, (const float *)(blob + op[W_WOFF]), bias, slope, op[W_S_ACT0], blob + op[W_S_W1],
                                      (const float *)(blob + op[W_S_B1]), (op[W_FLAGS] & CF_BORDER) ? 9 : 1,
                                      op[W_S_S1] >= 0 ? (const float *)(blob + op[W_S_S1]) : nullptr, op[W_ACT], dst.ptr, xe));
            break;
        }
        case OP_LATFPN: {
            const TensorView src = view(net, op[W_SRC], first);
            const void *res = nullptr;
            int rH = 0, rW = 0;
            if (op[W_RES] >= 0) {
                const TensorView r = view(net, op[W_RES], first);
                FID_REQUIRE(r.Cp == 64 && r.dtype == 0, "op %d: coarser lateral", oi);
                res = r.ptr; rH = r.H; rW = r.W;
            }
            FID_REQUIRE(src.dtype == 0 && dst.dtype == 0 && dst.Cp == 64 && src.H == dst.H && src.W == dst.W && lat_fpn_applicable(src.Cp, src.H, src.W, rH, rW, res != nullptr),
                        "op %d: bad fused lateral + fpn record", oi);
            void *lat = nullptr;
            if (op[W_L_LAT] > 0) {
                const TensorView l = view(net, op[W_L_LAT] - 1, first);
                FID_REQUIRE(l.Cp == 64 && l.dtype == 0 && l.H == dst.H && l.W == dst.W, "op %d: lateral output of the fused lateral + fpn op", oi);
                lat = l.ptr;
            }
            FID_TRY(lat_fpn_launch(ctx, src.ptr, batch, src.H, src.W, src.Cp, blob + op[W_L_W0], (const float *)(blob + op[W_L_B0]), res, rH, rW, blob + op[W_WOFF], bias, dst.ptr, lat));
            break;
        }
        case OP_BBLOCK: {
            const TensorView src = view(net, op[W_SRC], first);
            FID_REQUIRE((src.Cp == 64 || src.Cp == 32) && dst.Cp == src.Cp && src.H == dst.H && src.W == dst.W && src.dtype == 0 && dst.dtype == 0, "op %d: bad fused block record", oi);
            const int rev = net->alternate && !net->tdir[op[W_SRC]];
            FID_TRY(conv_bb_launch(ctx, src.ptr, blob + op[W_B_W1], (const float *)(blob + op[W_B_B1]), (op[W_FLAGS] & CF_BORDER) ? 9 : 1, op[W_B_ACT1],
                                   op[W_B_S1] >= 0 ? (const float *)(blob + op[W_B_S1]) : nullptr, blob + op[W_B_W2], (const float *)(blob + op[W_B_B2]), dst.ptr,
                                   batch, dst.H, dst.W, op[W_ACT], rev, src.Cp));
            net->tdir[op[W_DST]] = (char)rev;
            break;
        }
        case OP_DWPW: {
            const TensorView src = view(net, op[W_SRC], first);
            const void *res = nullptr;
            if (op[W_RES] >= 0) {
                const TensorView r = view(net, op[W_RES], first);
                FID_REQUIRE(r.H == dst.H && r.W == dst.W && r.Cp == dst.Cp && r.dtype == 0, "op %d: residual shape", oi);
                res = r.ptr;
            }
            FID_REQUIRE(src.dtype == 0 && dst.dtype == 0 && op[W_D_WOFF] >= 0 && op[W_D_BOFF] >= 0 && op[W_WOFF] >= 0, "op %d: bad depthwise + pointwise record", oi);
            FID_TRY(dwpw_launch(ctx, src.ptr, (const float *)(blob + op[W_D_WOFF]), (const float *)(blob + op[W_D_BOFF]),
                                op[W_D_SOFF] >= 0 ? (const float *)(blob + op[W_D_SOFF]) : nullptr, op[W_D_ACT], blob + op[W_WOFF], bias, slope, op[W_ACT], res,
                                dst.ptr, batch, src.H, src.W, dst.H, dst.W, src.Cp, dst.Cp, op[W_STRIDE]));
            break;
        }
        case OP_MBBLOCK: {
            const TensorView src = view(net, op[W_SRC], first);
            FID_REQUIRE(src.dtype == 0 && dst.dtype == 0 && op[W_M_W1] >= 0 && op[W_M_B1] >= 0 && op[W_M_DW] >= 0 && op[W_M_DWB] >= 0 && op[W_WOFF] >= 0 &&
                        op[W_M_GP] > 0 && (op[W_RES] < 0 || op[W_RES] == op[W_SRC]), "op %d: bad bottleneck record", oi);
            FID_REQUIRE(dst.H == (src.H - 1) / op[W_STRIDE] + 1 && dst.W == (src.W - 1) / op[W_STRIDE] + 1, "op %d: bottleneck output shape", oi);
            FID_TRY(mbf_block_launch(ctx, src.ptr, blob + op[W_M_W1], (const float *)(blob + op[W_M_B1]),
                                     op[W_M_S1] >= 0 ? (const float *)(blob + op[W_M_S1]) : nullptr, op[W_M_ACT1], (const float *)(blob + op[W_M_DW]),
                                     (const float *)(blob + op[W_M_DWB]), op[W_M_DWS] >= 0 ? (const float *)(blob + op[W_M_DWS]) : nullptr, op[W_M_DWACT],
                                     blob + op[W_WOFF], bias, slope, op[W_ACT], op[W_RES] >= 0, dst.ptr, batch, src.H, src.W, src.Cp, op[W_M_GP], dst.Cp,
                                     op[W_STRIDE]));
            break;
        }
        case OP_MAXPOOL: {
            const TensorView src = view(net, op[W_SRC], first);
            const long long total = (long long)batch * dst.H * dst.W * (dst.Cp / 8);
            hipLaunchKernelGGL(maxpool_nhwc, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, ctx->stream, (const _Float16 *)src.ptr,
                               (_Float16 *)dst.ptr, src.H, src.W, dst.H, dst.W, dst.Cp, op[W_KH], op[W_STRIDE], op[W_PAD], total);
            break;
        }
        case OP_DWCONV: {
            const TensorView src = view(net, op[W_SRC], first);
            const long long total = (long long)batch * dst.H * dst.W * (dst.Cp / 8);
            static const bool gdc_serial = getenv("FID_GDC_SERIAL") != nullptr;     // A/B: the one-chain kernel
            if (op[W_KH] == op[W_KW] && op[W_KH] <= 8 && op[W_PAD] == 0 && src.H == op[W_KH] && src.W == op[W_KW] && dst.H == 1 && dst.W == 1 && !gdc_serial) {
                const int groups = batch * (dst.Cp / 8);            // global depthwise conv: one output pixel per image
                hipLaunchKernelGGL(gdc_rows, dim3((unsigned)cdiv(groups * 8, 256)), dim3(256), 0, ctx->stream, (const _Float16 *)src.ptr,
                                   (const float *)(blob + op[W_WOFF]), bias, slope, (_Float16 *)dst.ptr, op[W_KH], dst.Cp, op[W_ACT], groups);
                break;
            }
            static const bool dw_global = getenv("FID_DW_GLOBAL") != nullptr;      // A/B: the kernel without the LDS tile everywhere
            // launches with enough pixels take the LDS-tiled kernel (below ~50 k pixels a launch is latency-bound either way: SCRFD-500M on ONE frame measured
            // 1.4 % slower with it, on 32 frames 1.3 % faster; MobileFaceNet's 56 x 56 x 128 layer at 32 faces 41 -> 27.6 us)
            if (!dw_global && op[W_KH] == 3 && op[W_KW] == 3 && op[W_STRIDE] == 1 && op[W_PAD] == 1 && dst.Cp % 64 == 0 && src.H >= 12 && src.W >= 12 &&
                (long long)batch * src.H * src.W >= 50000) {
                const int tx = cdiv(src.W, DW_TC), ty = cdiv(src.H, DW_TR), cbs = dst.Cp / 64;
                hipLaunchKernelGGL(dwconv3x3_lds, dim3((unsigned)(batch * ty * tx * cbs)), dim3(256), 0, ctx->stream, (const _Float16 *)src.ptr,
                                   (const float *)(blob + op[W_WOFF]), bias, slope, (_Float16 *)dst.ptr, src.H, src.W, dst.Cp, op[W_ACT], tx, ty, cbs);
                break;
            }
            hipLaunchKernelGGL(dwconv_nhwc, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, ctx->stream, (const _Float16 *)src.ptr,
                               (const float *)(blob + op[W_WOFF]), bias, slope, (_Float16 *)dst.ptr, src.H, src.W, dst.H, dst.W,
                               dst.Cp, op[W_KH], op[W_STRIDE], op[W_PAD], op[W_ACT], total);
            break;
        }
        default:
            set_error("net: unknown op type %d at %d", op[W_TYPE], oi);
            return FID_E_INVALID;
    }
    return FID_OK;
}

// worst-case split-K workspace over all ops at this batch
size_t partial_need(fid_ctx *ctx, fid_net *net, int batch) {
    size_t need = 0;
    for (int oi = 0; oi < net->n_ops; oi++) {
        const int32_t *op = &net->ops[(size_t)oi * FID_OP_WORDS];
        if (op[W_TYPE] != OP_CONV || op[W_X_DST2] > 0) continue;   // (the fused shortcut + conv op never splits K)
        const TensorView src = view(net, op[W_SRC]), dst = view(net, op[W_DST]);
        ConvArgs a{};
        a.Cin_p = src.Cp; a.Cout_p = dst.Cp; a.kh = op[W_KH]; a.kw = op[W_KW];
        a.M = batch * dst.H * dst.W; a.flags = op[W_FLAGS];
        need = std::max(need, conv_plan(a, ctx->num_cus, true).partial_bytes);
        // every candidate's workspace, whether or not this net autotunes: a loaded plan may name any of them
        a.H = src.H; a.W = src.W; a.Ho = dst.H; a.Wo = dst.W; a.stride = op[W_STRIDE]; a.pad = op[W_PAD]; a.w_rows = op[W_WROWS];
        for (const ConvPlan &c : conv_candidates(a, ctx->num_cus, true)) need = std::max(need, c.partial_bytes);
    }
    return std::max<size_t>(need, 256);
}

static int run_all_impl(fid_ctx *ctx, fid_net *net, const uint8_t *images, int batch, float *op_ms);

// (the failure path too gives the tuner's flush arena back and clears the flag: a run_op error during tuning used to leave 320 MB with the
// context until the next successful run -- ADVICE r4)
int run_all(fid_ctx *ctx, fid_net *net, const uint8_t *images, int batch, float *op_ms) {
    const int rc = run_all_impl(ctx, net, images, batch, op_ms);
    if (rc != FID_OK && ctx && net && net->tuned_now) {
        net->tuned_now = false;
        (void)release_scratch(ctx, 4);
    }
    return rc;
}

static int run_all_impl(fid_ctx *ctx, fid_net *net, const uint8_t *images, int batch, float *op_ms) {
    FID_REQUIRE(ctx && net && images, "NULL argument");
    FID_REQUIRE(batch > 0 && batch <= net->max_batch, "batch %d outside [1, %d]", batch, net->max_batch);
    // r01's recorded SIGSEGV (gpurun_out/gpu_tests_13.log): a working tree whose fid_net_create did not size `tuned` yet indexed
    // tuned[op] of an empty vector at the first conv of the first net that ran.  The table is sized in fid_net_create; keep it checked.
    FID_REQUIRE((int)net->tuned.size() == net->n_ops, "net: plan table has %zu entries for %d ops", net->tuned.size(), net->n_ops);
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    void *partial_ws = nullptr;
    const int sbq = (!op_ms && net->sub_batch > 0) ? std::min(net->sub_batch, batch) : batch;
    const size_t need = std::max(partial_need(ctx, net, sbq), partial_need(ctx, net, batch % sbq ? batch % sbq : sbq));
    if (need) FID_TRY(get_scratch(ctx, 1, need, &partial_ws));
    net->partial_cap = need;
    if (op_ms) {
        if (net->n_prof_events < net->n_ops + 1) {
            net->prof_events = new hipEvent_t[net->n_ops + 1];
            for (int i = 0; i <= net->n_ops; i++) FID_HIP(hipEventCreate(&net->prof_events[i]));
            net->n_prof_events = net->n_ops + 1;
        }
        FID_HIP(hipEventRecord(net->prof_events[0], ctx->stream));
    }
    if (op_ms) {   // per-op timing: whole batch per op (layer by layer)
        for (int oi = 0; oi < net->n_ops; oi++) {
            FID_TRY(run_op(ctx, net, oi, images, 0, batch, need ? partial_ws : nullptr));
            FID_HIP(hipEventRecord(net->prof_events[oi + 1], ctx->stream));
        }
    } else {
        // depth-first over sub-batches: every layer's input was written a moment ago by the previous
        // layer of the SAME sub-batch and is still in L2 / Infinity Cache instead of coming back from HBM
        const int sb = net->sub_batch > 0 ? net->sub_batch : batch;
        auto issue = [&]() -> int {
            for (int first = 0; first < batch; first += sb) {
                const int nb = std::min(sb, batch - first);
                for (int oi = 0; oi < net->n_ops; oi++) FID_TRY(run_op(ctx, net, oi, images, first, nb, need ? partial_ws : nullptr));
            }
            return FID_OK;
        };
        // The sequence is fixed once every op has its tuned plan, so a caller that comes back with the same frame buffer and
        // batch (the pipeline's resident buffers) gets ONE graph launch instead of n_ops kernel launches.  The first two runs
        // of a key are issued eagerly (the autotuner synchronises, launchers set function attributes, scratch may grow: none
        // of that may happen inside a capture); at most MAX_REPLAYS graphs are kept, least recently used first out.
        constexpr int MAX_REPLAYS = 8;
        net->run_counter++;
        fid_net::Replay *rp = nullptr;
        if (net->graphs) {
            auto key = std::make_pair((const void *)images, batch);
            auto it = net->replays.find(key);
            if (it == net->replays.end()) {
                if ((int)net->replays.size() >= MAX_REPLAYS) {
                    auto victim = net->replays.begin();
                    for (auto j = net->replays.begin(); j != net->replays.end(); ++j)
                        if (j->second.last_use < victim->second.last_use) victim = j;
                    if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
                    net->replays.erase(victim);
                }
                it = net->replays.emplace(key, fid_net::Replay{}).first;
            }
            rp = &it->second;
            rp->last_use = net->run_counter;
            rp->seen++;
            if (rp->exec && rp->partial != partial_ws) {      // the context's split-K scratch was re-allocated (another net grew it)
                (void)hipGraphExecDestroy(rp->exec);
                rp->exec = nullptr;
            }
        }
        if (rp && rp->exec) {
            FID_HIP(hipGraphLaunch(rp->exec, ctx->stream));
        } else if (rp && rp->seen > 2) {
            hipGraph_t g = nullptr;
            FID_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
            const int rc = issue();
            const hipError_t ce = hipStreamEndCapture(ctx->stream, &g);
            if (rc != FID_OK) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            FID_HIP(ce);
            FID_HIP(hipGraphInstantiate(&rp->exec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            rp->partial = partial_ws;
            FID_HIP(hipGraphLaunch(rp->exec, ctx->stream));
        } else {
            FID_TRY(issue());
        }
    }
    FID_HIP(hipGetLastError());
    if (op_ms) {
        FID_HIP(hipEventSynchronize(net->prof_events[net->n_ops]));
        for (int oi = 0; oi < net->n_ops; oi++) FID_HIP(hipEventElapsedTime(&op_ms[oi], net->prof_events[oi], net->prof_events[oi + 1]));
    }
    if (net->tuned_now) {                    // the tuner's 320 MB cache-flush buffer does not stay with the context
        net->tuned_now = false;
        FID_TRY(release_scratch(ctx, 4));
    }
    return FID_OK;
}

}  // namespace
}  // namespace fid

extern "C" {

int fid_net_create(fid_ctx *ctx, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors, const void *blob,
                   size_t blob_bytes, int in_h, int in_w, int max_batch, fid_net **out) {
    using namespace fid;
    FID_REQUIRE(ctx && ops && tensors && blob && out, "NULL argument");
    FID_REQUIRE(n_ops > 0 && n_tensors > 0 && max_batch > 0 && in_h > 0 && in_w > 0, "bad sizes");
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));
    fid_net *net = new fid_net();
    net->n_ops = n_ops; net->n_tensors = n_tensors; net->in_h = in_h; net->in_w = in_w; net->max_batch = max_batch;
    net->ops.assign(ops, ops + (size_t)n_ops * FID_OP_WORDS);
    net->tensors.assign(tensors, tensors + (size_t)n_tensors * FID_TENSOR_WORDS);
    int n_slots = 0;
    for (int t = 0; t < n_tensors; t++) {
        const int32_t *tt = &net->tensors[(size_t)t * FID_TENSOR_WORDS];
        if (tt[T_SLOT] < 0 || tt[T_CP] % 8 != 0 || tt[T_H] <= 0 || tt[T_W] <= 0) {
            delete net;
            set_error("tensor %d: bad record", t);
            return FID_E_INVALID;
        }
        n_slots = std::max(n_slots, tt[T_SLOT] + 1);
    }
    net->slot_bytes_per_image.assign(n_slots, 0);
    for (int t = 0; t < n_tensors; t++) {
        const int32_t *tt = &net->tensors[(size_t)t * FID_TENSOR_WORDS];
        const size_t b = (size_t)tt[T_H] * tt[T_W] * tt[T_CP] * (tt[T_DTYPE] == 1 ? 4 : 2);
        net->slot_bytes_per_image[tt[T_SLOT]] = std::max(net->slot_bytes_per_image[tt[T_SLOT]], b);
    }
    for (int oi = 0; oi < n_ops; oi++) {
        const int32_t *op = &net->ops[(size_t)oi * FID_OP_WORDS];
        // every blob region a kernel will build a buffer resource from must lie inside the blob: the sizes follow from the record and the
        // tensors' padded channel counts exactly as lower.py lays the regions out (a malformed table through the public C-ABI must be refused
        // here, not fault the GPU later)
        auto in_blob = [&](long long off, size_t bytes) { return off >= 0 && (size_t)off + bytes <= blob_bytes; };
        auto opt_in_blob = [&](long long off, size_t bytes) { return off < 0 || (size_t)off + bytes <= blob_bytes; };
        bool ok = op[W_DST] >= 0 && op[W_DST] < n_tensors && op[W_SRC] >= -1 && op[W_SRC] < n_tensors && op[W_RES] >= -1 && op[W_RES] < n_tensors &&
                  op[W_WOFF] >= -1 && opt_in_blob(op[W_WOFF], (size_t)std::max(0, op[W_WBYTES])) &&
                  (op[W_TYPE] == OP_STEM || op[W_TYPE] == OP_STEMFUSED || op[W_TYPE] == OP_STEMBLOCK) == (op[W_SRC] == -1) && op[W_WROWS] >= 0;
        if (ok) {
            const int cp_src = op[W_SRC] >= 0 ? net->tensors[(size_t)op[W_SRC] * FID_TENSOR_WORDS + T_CP] : 0;
            const int cp_dst = net->tensors[(size_t)op[W_DST] * FID_TENSOR_WORDS + T_CP];
            const size_t rows = (size_t)op[W_WROWS];
            switch (op[W_TYPE]) {
            case OP_CONV: {
                const int ncls = (op[W_FLAGS] & CF_BORDER) ? 9 : 1;
                ok = opt_in_blob(op[W_BOFF], (size_t)ncls * rows * 4) && opt_in_blob(op[W_SOFF], rows * 4);
                if (ok && op[W_X_SRC2] != 0) {
                    ok = op[W_X_SRC2] > 0 && op[W_X_SRC2] <= n_tensors && op[W_X_T2] >= 1 && op[W_X_KW2] >= 1 && op[W_X_T2] % op[W_X_KW2] == 0 &&
                         op[W_X_S2] >= 1 && op[W_X_DST2] == 0 && op[W_X_W2OFF] > 0 && op[W_X_B2OFF] > 0 && op[W_X_SCOP] >= 1 && op[W_X_SCOP] <= oi;
                    if (ok) {
                        const int cp2 = net->tensors[(size_t)(op[W_X_SRC2] - 1) * FID_TENSOR_WORDS + T_CP];
                        const size_t row_halfs = (size_t)op[W_KH] * op[W_KW] * cp_src + (size_t)op[W_X_T2] * cp2;   // rows [kh*kw * Cin_p | taps * Cin2_p]
                        ok = in_blob(op[W_X_W2OFF], rows * row_halfs * 2) && in_blob(op[W_X_B2OFF], rows * 4);
                    }
                } else if (ok && op[W_X_W2OFF] != 0) {                 // a shortcut op: word 29 = index + 1 of the conv that may absorb it
                    ok = op[W_X_W2OFF] > oi + 1 && op[W_X_W2OFF] <= n_ops;
                }
                break;
            }
            case OP_STEM:
            case OP_DWCONV:
                ok = opt_in_blob(op[W_BOFF], rows * 4) && opt_in_blob(op[W_SOFF], rows * 4);
                break;
            case OP_STEMBLOCK:
                ok = cp_dst == 64 && op[W_WOFF] >= 0 && (size_t)op[W_WBYTES] >= 64 * 27 * 4 && in_blob(op[W_BOFF], 64 * 4) && opt_in_blob(op[W_SOFF], 64 * 4) &&
                     in_blob(op[W_S_W1], (size_t)2 * 73728) && in_blob(op[W_S_B1], (size_t)((op[W_FLAGS] & CF_BORDER) ? 9 : 1) * 64 * 4) &&
                     opt_in_blob(op[W_S_S1], 64 * 4) && (op[W_S_ACT0] == ACT_RELU || (op[W_S_ACT0] == ACT_PRELU && op[W_SOFF] >= 0)) &&
                     (op[W_ACT] == ACT_RELU || (op[W_ACT] == ACT_PRELU && op[W_S_S1] >= 0)) && op[W_S_DST2] >= 0 && op[W_S_DST2] <= n_tensors &&
                     (op[W_S_DST2] == 0 || net->tensors[(size_t)(op[W_S_DST2] - 1) * FID_TENSOR_WORDS + T_CP] == 64);
                break;
            case OP_LATFPN:
                ok = cp_dst == 64 && (cp_src == 64 || cp_src == 96) && op[W_WOFF] >= 0 && (size_t)op[W_WBYTES] >= (size_t)2 * 73728 && in_blob(op[W_BOFF], 64 * 4) &&
                     in_blob(op[W_L_W0], (size_t)64 * cp_src * 2) && in_blob(op[W_L_B0], 64 * 4) && op[W_L_LAT] >= 0 && op[W_L_LAT] <= n_tensors &&
                     (op[W_L_LAT] == 0 || net->tensors[(size_t)(op[W_L_LAT] - 1) * FID_TENSOR_WORDS + T_CP] == 64) &&
                     (op[W_RES] < 0 || net->tensors[(size_t)op[W_RES] * FID_TENSOR_WORDS + T_CP] == 64);
                break;
            case OP_STEMFUSED:
                ok = in_blob(op[W_F_W0], 32 * 32 * 2) && in_blob(op[W_F_B0], 32 * 4) && in_blob(op[W_F_W1], (size_t)32 * 9 * 32 * 2) && in_blob(op[W_F_B1], 32 * 4) &&
                     in_blob(op[W_F_W2], (size_t)cp_dst * 9 * 32 * 2) && in_blob(op[W_F_B2], (size_t)cp_dst * 4);
                break;
            case OP_BBLOCK: {
                const size_t image = (size_t)(cp_src / 32) * 73728;    // repack kind 2: 73 728 B per 32-channel chunk (two chunks for the 64-channel block)
                ok = (cp_src == 32 || cp_src == 64) && in_blob(op[W_B_W1], image) && in_blob(op[W_B_W2], image) &&
                     in_blob(op[W_B_B1], (size_t)((op[W_FLAGS] & CF_BORDER) ? 9 : 1) * cp_src * 4) && in_blob(op[W_B_B2], (size_t)cp_src * 4) &&
                     (op[W_B_ACT1] == ACT_RELU || (op[W_B_ACT1] == ACT_PRELU && in_blob(op[W_B_S1], (size_t)cp_src * 4)));
                break;
            }
            case OP_DWPW:
                ok = op[W_WOFF] >= 0 && in_blob(op[W_D_WOFF], (size_t)9 * cp_src * 4) && in_blob(op[W_D_BOFF], (size_t)cp_src * 4) &&
                     (op[W_D_ACT] != ACT_PRELU || in_blob(op[W_D_SOFF], (size_t)cp_src * 4)) && opt_in_blob(op[W_BOFF], rows * 4) && opt_in_blob(op[W_SOFF], rows * 4);
                break;
            case OP_MBBLOCK: {
                const size_t gp = (size_t)std::max(0, op[W_M_GP]);
                ok = op[W_WOFF] >= 0 && gp > 0 && gp % 32 == 0 && (op[W_STRIDE] == 1 || op[W_STRIDE] == 2) && in_blob(op[W_M_W1], gp * cp_src * 2) &&
                     in_blob(op[W_M_B1], gp * 4) && (op[W_M_ACT1] != ACT_PRELU || in_blob(op[W_M_S1], gp * 4)) && in_blob(op[W_M_DW], 9 * gp * 4) &&
                     in_blob(op[W_M_DWB], gp * 4) && (op[W_M_DWACT] != ACT_PRELU || in_blob(op[W_M_DWS], gp * 4)) && (size_t)op[W_WBYTES] >= rows * gp * 2 &&
                     opt_in_blob(op[W_BOFF], rows * 4) && opt_in_blob(op[W_SOFF], rows * 4);
                break;
            }
            default: break;
            }
        }
        if (!ok) {
            delete net;
            set_error("op %d: bad record", oi);
            return FID_E_INVALID;
        }
        const int32_t *dt = &net->tensors[(size_t)op[W_DST] * FID_TENSOR_WORDS];
        if (op[W_TYPE] == OP_STEMFUSED || op[W_TYPE] == OP_BBLOCK || op[W_TYPE] == OP_DWPW || op[W_TYPE] == OP_MBBLOCK || op[W_TYPE] == OP_STEMBLOCK || op[W_TYPE] == OP_LATFPN)
            net->macs_per_image += (double)(((unsigned long long)(unsigned)op[W_F_MACS_HI] << 32) | (unsigned)op[W_F_MACS_LO]);
        if (op[W_TYPE] == OP_CONV && op[W_X_DST2] > 0) net->macs_per_image += (double)(unsigned)op[W_F_MACS_LO];   // the fused shortcut's share
        if (op[W_TYPE] == OP_CONV || op[W_TYPE] == OP_STEM || op[W_TYPE] == OP_DWCONV)
            net->macs_per_image += (double)dt[T_H] * dt[T_W] * op[W_COUT] * (op[W_CIN] / std::max(1, op[W_GROUPS])) * op[W_KH] * op[W_KW];
    }
    for (int s = 0; s < n_slots; s++) {
        const size_t bytes = net->slot_bytes_per_image[s] * (size_t)max_batch;
        if (bytes > 0x7FFFFFF0ull) {
            delete net;
            set_error("activation slot %d needs %zu bytes at max_batch=%d (> 2 GiB addressable per tensor): lower max_batch", s, bytes, max_batch);
            return FID_E_INVALID;
        }
    }
    net->slots.assign(n_slots, nullptr);
    for (int s = 0; s < n_slots; s++) FID_HIP(hipMalloc(&net->slots[s], net->slot_bytes_per_image[s] * (size_t)max_batch + 256));
    FID_HIP(hipMalloc(&net->blob, blob_bytes + 256));
    net->blob_bytes = blob_bytes;
    FID_HIP(hipMemcpy(net->blob, blob, blob_bytes, hipMemcpyHostToDevice));
    if (const char *e = getenv("FID_SUB_BATCH")) net->sub_batch = atoi(e);
    if (const char *e = getenv("FID_AUTOTUNE")) net->autotune = atoi(e);
    if (const char *e = getenv("FID_GRAPH")) net->graphs = atoi(e);
    net->tuned.resize(n_ops);
    net->plan_ok.resize(n_ops);
    net->tdir.assign(n_tensors, 0);
    if (getenv("FID_NO_REV")) net->alternate = 0;
    {
        hipDeviceProp_t prop;
        FID_HIP(hipGetDeviceProperties(&prop, ctx->device));
        // ISA name without its feature suffixes + CU count: the marketing name and the ":sramecc+:xnack-" tail vary with the driver stack
        // and with tools that wrap the process (under rocprofv3 the r3 plan did not match and every net re-tuned inside the profile)
        char arch[128];
        snprintf(arch, sizeof(arch), "%s", prop.gcnArchName);
        if (char *c = strchr(arch, ':')) *c = 0;
        char key[256];
        snprintf(key, sizeof(key), "%s/cus%d", arch, prop.multiProcessorCount);
        for (char *c = key; *c; c++)
            if (*c == '|' || *c == '\n') *c = '_';
        net->device_key = key;
        unsigned long long h = fnv1a(net->ops.data(), net->ops.size() * sizeof(int32_t));
        h = fnv1a(net->tensors.data(), net->tensors.size() * sizeof(int32_t), h);
        h = fnv1a(&blob_bytes, sizeof(blob_bytes), h);
        const int rev = FID_PLAN_REV;                         // candidate-set revision: plans of another library revision do not match
        h = fnv1a(&rev, sizeof(rev), h);
        net->table_hash = h;
    }
    int n_plan = -1;
    const char *plan_file = nullptr;
    if (const char *e = getenv("FID_PLAN")) {                 // load + append every new pick (plan generation: tools/make_plan.sh)
        net->plan_path = e;
        plan_file = e;
        (void)plan_load(net, e, &n_plan);
    } else if (const char *e = getenv("FID_PLAN_RO")) {        // load only: a tracked plan file is never written by a run (bench.py)
        plan_file = e;
        (void)plan_load(net, e, &n_plan);
    }
    if (getenv("FID_TUNE_LOG"))
        fprintf(stderr, "[plan] device key %s, table %016llx: %d picks from %s\n", net->device_key.c_str(), net->table_hash, n_plan, plan_file ? plan_file : "(no plan file)");
    *out = net;
    return FID_OK;
}

// Kernel plans of this net (per conv op and batch size: kernel family, tile, split-K) as text lines keyed by the device name
// and a hash of the layer table.  fid_net_plan_save writes every pick made so far; fid_net_plan_load installs matching
// lines (others are ignored) and returns their number -- a loaded pick is used instead of timing candidates, so outputs no
// longer depend on which box tuned.  FID_PLAN=<file> does both automatically (load at create,
// append every new pick).
int fid_net_plan_save(fid_net *net, const char *path) {
    FID_REQUIRE(net && path, "NULL argument");
    FILE *f = fopen(path, "a");
    FID_REQUIRE(f, "cannot open %s for appending", path);
    char line[512];
    for (int oi = 0; oi < net->n_ops; oi++)
        for (const auto &kv : net->tuned[oi]) {
            fid::plan_line(net, oi, kv.first, kv.second, line, sizeof(line));
            fputs(line, f);
        }
    fclose(f);
    return FID_OK;
}

int fid_net_plan_load(fid_net *net, const char *path, int *n_loaded) {
    FID_REQUIRE(net && path, "NULL argument");
    for (auto &kv : net->replays)      // recorded launch sequences no longer match
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    net->replays.clear();
    return fid::plan_load(net, path, n_loaded);
}

int fid_net_set_sub_batch(fid_net *net, int sub_batch) {
    FID_REQUIRE(net && sub_batch >= 0, "bad args");
    net->sub_batch = sub_batch;
    for (auto &kv : net->replays)      // recorded launch sequences no longer match
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    net->replays.clear();
    return FID_OK;
}

int fid_net_destroy(fid_ctx *ctx, fid_net *net) {
    if (!net) return FID_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : net->slots)
        if (p) (void)hipFree(p);
    if (net->blob) (void)hipFree(net->blob);
    for (auto &kv : net->alt_w)
        if (kv.second) (void)hipFree(kv.second);
    for (auto &kv : net->replays)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (net->prof_events) {
        for (int i = 0; i < net->n_prof_events; i++) (void)hipEventDestroy(net->prof_events[i]);
        delete[] net->prof_events;
    }
    delete net;
    return FID_OK;
}

int fid_net_run(fid_ctx *ctx, fid_net *net, const uint8_t *images_dev, int batch) {
    return fid::run_all(ctx, net, images_dev, batch, nullptr);
}

int fid_net_run_profiled(fid_ctx *ctx, fid_net *net, const uint8_t *images_dev, int batch, float *op_ms) {
    FID_REQUIRE(op_ms, "op_ms is NULL");
    return fid::run_all(ctx, net, images_dev, batch, op_ms);
}

int fid_net_tensor(fid_net *net, int tensor_id, void **dptr, int dims[4], int *dtype) {
    FID_REQUIRE(net && tensor_id >= 0 && tensor_id < net->n_tensors, "bad tensor id %d", tensor_id);
    const fid::TensorView v = fid::view(net, tensor_id);
    if (dptr) *dptr = v.ptr;
    if (dims) { dims[0] = v.H; dims[1] = v.W; dims[2] = v.C; dims[3] = v.Cp; }
    if (dtype) *dtype = v.dtype;
    return FID_OK;
}

int fid_net_macs(fid_net *net, double *macs_per_image) {
    FID_REQUIRE(net && macs_per_image, "NULL argument");
    *macs_per_image = net->macs_per_image;
    return FID_OK;
}

}  // extern "C"
