// Depthwise 3x3 (+BN +activation) followed by the pointwise 1x1 conv that consumes it, in one launch: the depthwise result of a tile of
// output pixels goes to LDS (fp16, exactly the values the unfused layer would store) and is the pixel operand of the 1x1 conv's MFMAs --
// it never travels to HBM and one launch replaces two.  MobileFaceNet (w600k_mbf, the recogniser the reference's main.py:19-30 defaults
// to) is a chain of such pairs: at 32 faces its 17 depthwise launches (9-41 us each, launch-bound: 0.01-0.23 GFLOP) were 39 % of the
// net's 0.64 ms and every one was followed by a 9-15 us pointwise launch (BASELINE configs[4] names this path).
//
//   item   = 32 consecutive output pixels (flattened over batch, rows, columns) x all output channels
//   stage1 = depthwise on the VALU: thread = (4-channel group cg, a pixel of the item); its 9 x 4 weights, bias and slopes live in registers
//            (the group is fixed per thread); the 9 taps are 8-byte loads straight from global memory (neighbouring pixels share them through
//            L1 / L2: the maps these layers see are a few hundred KB per image); fp32 fmaf chain in the order of dwconv_nhwc (net.hip), so the
//            fp16 values are bit-identical to the unfused layer's output.  Result -> LDS [pixel][Gp] with a 16-byte row pad (bank spread).
//   stage2 = pointwise on MFMA 16x16x32: wave w owns cout fragments w, w+4, ...; per fragment 2 pixel fragments x Gp/32 K-steps, the A
//            (weight) fragments come straight from global memory (L2-hot: 32 KB - 256 KB per layer), B from LDS.
//   epilogue = bias, residual (MobileFaceNet's conv_3.x / 4.x / 5.x blocks), activation, 8-byte stores from the accumulator layout.
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int PXI = 32;                        // output pixels per item (two MFMA pixel fragments)
constexpr int NT_ = 256;
constexpr int NPF = PXI / 16;                  // pixel fragments per item                       // threads: several small workgroups per CU -- these layers are a few hundred items of ~2 us of work

struct DWPWArgs {
    const _Float16 *in;        // [B, H, W, Gp]
    const float *dw_w;         // [9][Gp]
    const float *dw_b;         // [Gp]
    const float *dw_s;         // [Gp] PReLU slopes of the depthwise layer (dw_act == ACT_PRELU)
    const _Float16 *pw_w;      // [Cout_p][Gp]
    const float *pw_b;         // [Cout_p] or NULL
    const float *pw_s;         // [Cout_p] (pw_act == ACT_PRELU)
    const _Float16 *res;       // [B, Ho, Wo, Cout_p] or NULL
    _Float16 *out;             // [B, Ho, Wo, Cout_p]
    int H, W, Ho, Wo, Gp, Cout_p, stride, dw_act, pw_act;
    int M, n_items;
};

// KSM: K-steps of the pointwise conv (Gp / 32) rounded up to 2 | 4 | 8 | 16 -- ALL weight fragments of a cout fragment are requested at once
// (one trip to L2 instead of one per K-step: a K-step is two MFMAs, 32 cycles, against ~700 ns per dependent load)
template <int KSM>
__global__ void __launch_bounds__(NT_, KSM <= 2 ? 4 : (KSM <= 8 ? 3 : 2)) dwpw_kernel(const DWPWArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int ncg = a.Gp >> 2;                                  // 4-channel groups (a power of two <= 128)
    const int pitch = a.Gp * 2 + 16;                            // bytes per pixel row of the LDS tile
    const int cg = tid & (ncg - 1), px0 = tid / ncg, pxs = NT_ / ncg;

    // ---- depthwise parameters of my channel group (4 channels per thread: ~100 registers in all, four workgroups per CU -- every item of
    // these few-hundred-item layers is resident at once; with 8 channels per thread half of them waited for a free slot) ----
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; t++) wv[t] = *(const f32x4 *)(a.dw_w + t * a.Gp + cg * 4);
    const f32x4 bv = *(const f32x4 *)(a.dw_b + cg * 4);
    const f32x4 sv = a.dw_act == ACT_PRELU ? *(const f32x4 *)(a.dw_s + cg * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    const int hw = a.Ho * a.Wo, frags = a.Cout_p >> 4, ksteps = a.Gp >> 5;

    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int m0 = item * PXI;
        // the pointwise stage's first operands do not depend on stage 1: request them now (all K-steps of the wave's first cout fragment)
        half8 af[KSM];
        {
            const _Float16 *wr0 = a.pw_w + (size_t)((wave < frags ? wave : 0) * 16 + frow) * a.Gp + fq * 8;
#pragma unroll
            for (int ks = 0; ks < KSM; ks++) af[ks] = *(const half8 *)(wr0 + (ks < ksteps ? ks : 0) * 32);
        }
        const f32x4 b0 = a.pw_b ? *(const f32x4 *)(a.pw_b + (wave < frags ? wave : 0) * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        // ================= stage 1: depthwise -> LDS =================
        for (int p = px0; p < PXI; p += pxs) {
            const int m = m0 + p;
            half4 o = half4{0, 0, 0, 0};
            if (m < a.M) {
                const int n = m / hw, r = m - n * hw;
                const int oy = r / a.Wo, ox = r - oy * a.Wo;
                half4 v[9];
                bool ok[9];
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const int iy = oy * a.stride - 1 + dy, ix = ox * a.stride - 1 + dx;
                        ok[dy * 3 + dx] = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        const int cy = ok[dy * 3 + dx] ? iy : 0, cx = ok[dy * 3 + dx] ? ix : 0;
                        v[dy * 3 + dx] = *(const half4 *)(a.in + ((size_t)(n * a.H + cy) * a.W + cx) * a.Gp + cg * 4);
                    }
                f32x4 acc = bv;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    if (!ok[t]) continue;                        // (a skipped tap, not a zero product: the order of the fmaf chain is dwconv_nhwc's)
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = fmaf((float)v[t][e], wv[t][e], acc[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float x = acc[e];
                    if (a.dw_act == ACT_RELU) x = fmaxf(x, 0.f);
                    else if (a.dw_act == ACT_PRELU) x = x > 0.f ? x : x * sv[e];
                    o[e] = (_Float16)x;
                }
            }
            *(half4 *)(smem + p * pitch + cg * 8) = o;
        }
        __syncthreads();

        // ================= stage 2: pointwise on the matrix cores =================
        for (int cf = wave; cf < frags; cf += NT_ / 64) {
            f32x4 acc[NPF];
#pragma unroll
            for (int p = 0; p < NPF; p++) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
            const _Float16 *wrow = a.pw_w + (size_t)(cf * 16 + frow) * a.Gp + fq * 8;
            if (cf != wave) {
#pragma unroll
                for (int ks = 0; ks < KSM; ks++) af[ks] = *(const half8 *)(wrow + (ks < ksteps ? ks : 0) * 32);
            }
            const int co = cf * 16 + fq * 4;
            // the epilogue's operands are requested before the K loop (for the wave's first fragment: before stage 1): their latency hides behind it
            const f32x4 b = cf == wave ? b0 : (a.pw_b ? *(const f32x4 *)(a.pw_b + co) : f32x4{0.f, 0.f, 0.f, 0.f});
            const f32x4 sl = a.pw_act == ACT_PRELU ? *(const f32x4 *)(a.pw_s + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            half4 rs[NPF];
#pragma unroll
            for (int p = 0; p < NPF; p++) {
                const int m = m0 + p * 16 + frow;
                rs[p] = (a.res && m < a.M) ? *(const half4 *)(a.res + (size_t)m * a.Cout_p + co) : half4{0, 0, 0, 0};
            }
#pragma unroll
            for (int ks = 0; ks < KSM; ks++) {
                if (ks >= ksteps) break;
#pragma unroll
                for (int p = 0; p < NPF; p++) {
                    const half8 bf = *(const half8 *)(smem + (p * 16 + frow) * pitch + (ks * 32 + fq * 8) * 2);
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks], bf, acc[p], 0, 0, 0);
                }
            }
#pragma unroll
            for (int p = 0; p < NPF; p++) {
                const int m = m0 + p * 16 + frow;
                if (m >= a.M) continue;
                f32x4 v = acc[p] + b + __builtin_convertvector(rs[p], f32x4);
                if (a.pw_act == ACT_PRELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + sl * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
                half4 h = __builtin_convertvector(v, half4);
                if (a.pw_act == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                *(half4 *)(a.out + (size_t)m * a.Cout_p + co) = h;
            }
        }
        __syncthreads();                                        // everyone has read the tile before the next item overwrites it
    }
}

}  // namespace

bool dwpw_applicable(int Gp, int Cout_p) {
    return (Gp == 32 || Gp == 64 || Gp == 128 || Gp == 256 || Gp == 512) && Cout_p % 16 == 0 && Cout_p >= 16 && Cout_p <= 1024;
}

int dwpw_launch(fid_ctx *ctx, const void *in, const float *dw_w, const float *dw_b, const float *dw_s, int dw_act, const void *pw_w, const float *pw_b,
                const float *pw_s, int pw_act, const void *res, void *out, int B, int H, int W, int Ho, int Wo, int Gp, int Cout_p, int stride) {
    FID_REQUIRE(in && dw_w && dw_b && pw_w && out && B > 0 && dwpw_applicable(Gp, Cout_p) && (stride == 1 || stride == 2), "dwpw: bad arguments");
    FID_REQUIRE(Ho == (H + 2 - 3) / stride + 1 && Wo == (W + 2 - 3) / stride + 1, "dwpw: output shape %dx%d for input %dx%d stride %d", Ho, Wo, H, W, stride);
    DWPWArgs a{};
    a.in = (const _Float16 *)in; a.dw_w = dw_w; a.dw_b = dw_b; a.dw_s = dw_s; a.pw_w = (const _Float16 *)pw_w; a.pw_b = pw_b; a.pw_s = pw_s;
    a.res = (const _Float16 *)res; a.out = (_Float16 *)out;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Gp = Gp; a.Cout_p = Cout_p; a.stride = stride; a.dw_act = dw_act; a.pw_act = pw_act;
    a.M = B * Ho * Wo;
    a.n_items = cdiv(a.M, PXI);
    const int lds = PXI * (Gp * 2 + 16);
    const int grid = std::min(a.n_items, ctx->num_cus * 4);
#define DWPW_GO(K) do { FID_TRY(ensure_dyn_lds(ctx, (const void *)dwpw_kernel<K>, lds)); \
        hipLaunchKernelGGL(dwpw_kernel<K>, dim3(grid), dim3(NT_), lds, ctx->stream, a); } while (0)
    if (Gp <= 64) DWPW_GO(2);
    else if (Gp == 128) DWPW_GO(4);
    else if (Gp == 256) DWPW_GO(8);
    else DWPW_GO(16);
#undef DWPW_GO
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
