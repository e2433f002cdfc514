// MobileFaceNet's bottleneck in ONE launch:  pointwise 1x1 (Cin -> G, act) -> depthwise 3x3 / stride 1 | 2 (G, act) -> pointwise 1x1 (G -> Cout)
// [+ block input] (reference models/arcface.py:51 runs w600k_mbf inside session.run; main.py:19-30 defaults to this recogniser; BASELINE
// configs[4]).  At 32 faces the net was 50 launches of 6-12 us (profiles/r03: each a load -> compute -> store chain of a few dependent memory
// round trips, ~6 us even when empty of work): 45 of them are the three layers of 15 such blocks.  Here the two expanded maps never leave LDS
// and EVERY global load of a block is requested up front or a slice ahead; between them there are only LDS, MFMA and VALU stages.
//
//   item    = a tile of To x To output pixels of one image (To = 7 at stride 1, 4 at stride 2) x a block of 128 couts; its input region -- the
//             9 x 9 pixels the depthwise stage reads -- is fetched once into LDS (fp16, all Cin channels).  Small tiles on purpose: 4 x 32 faces x
//             cout blocks = a few hundred workgroups on the 14 x 14 maps (whole-map items left 3/4 of the CUs idle and were 2.5x SLOWER than the
//             three launches: the depthwise stage is VALU work that must be spread), and both expanded maps fit LDS WHOLE (no channel slicing:
//             every weight of the item is requested once, up front, behind the region fetch -- one trip to memory, not one per slice)
//   A) pw1 on the matrix cores: 81 region pixels x G channels -> E (LDS, fp16, bias + act: exactly the values the unfused layer stores); wave w
//      owns cout fragments w, w + 8, ..., two at a time (one pixel-fragment read feeds both)
//   B) depthwise on the VALU, fp32 fmaf chain in dwconv_nhwc's order, taps outside the image skipped -> D (LDS, fp16); thread = (4-channel
//      group, pixel lane), its 9 x 4 weights in registers
//   C) pw2: wave w owns cout fragment 8 cb + w for the tile's <= 4 pixel fragments, K = G
//   epilogue: bias, block input (the centre of the region in LDS), activation, 8-byte stores.  Three barriers per item.
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWV = 8, NTH = NWV * 64;
// the input region of an item: RW x RW pixels, RW = (To - 1) stride + 3 = 9 (7 x 7 outputs at stride 1, 4 x 4 at stride 2) or 15 (7 x 7 outputs at
// stride 2 where both expanded maps of 225 pixels fit LDS: the 56 -> 28 block, a quarter of the workgroups)
constexpr int MAXOF = 4;                                         // pixel fragments of an output tile (7 x 7 = 49 pixels at stride 1, 4 x 4 at stride 2)

struct MBArgs {
    const _Float16 *x;         // [B, H, W, Cin_p]
    const _Float16 *w1;        // [Gp][Cin_p]
    const float *b1, *s1;      // [Gp]
    const float *dww;          // [9][Gp]
    const float *dwb, *dws;    // [Gp]
    const _Float16 *w2;        // [Cout_p][Gp]
    const float *b2, *s2;      // [Cout_p] (b2 may be NULL)
    _Float16 *out;             // [B, Ho, Wo, Cout_p]
    int H, W, Ho, Wo, Cin_p, Gp, Cout_p, stride;
    int act1, dw_act, act2, res;
    int To;                    // output tile edge: 7 (stride 1) | 4 (stride 2); the region is ((To - 1) stride + 3)^2 = 9 x 9 input pixels
    int tiles_x, tiles_per_img, n_items, ncb;      // ncb = cout blocks of 128 per tile
    int cb_in_wg;              // 1: a workgroup walks the cout blocks of its tile itself (items = tiles); 0: one item per (tile, cout block)
    int xs_pitch, e_pitch, off_e, off_d;
    int ablate;                // FID_MB_ABLATE timing experiments (wrong results): 1 no stage A, 2 no stage B, 4 no stage C, 16 no region fetch
};

// KS1 = Cin_p / 32 (K-steps of pw1); RW = the region's edge (9 | 15).
template <int KS1, int RW>
__global__ void __launch_bounds__(NTH) mbf_block(const MBArgs a) {
    constexpr int RPX = RW * RW, NPF = (RPX + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int frags1 = a.Gp >> 4, frags2 = a.Cout_p >> 4, ks2 = a.Gp >> 5;
    const int cgs = a.Gp >> 2;                                  // 4-channel groups of the expanded map (32 | 64 | 128 ...)
    const int cg = tid % cgs, pl = tid / cgs, npl = NTH / cgs;  // stage B: my channel group, my pixel lane, pixel lanes (0 when cgs > NTH: see launch)
    char *sX = smem, *sE = smem + a.off_e, *sD = smem + a.off_d;
    const int chunks = a.Cin_p >> 3;                            // 16-byte chunks per pixel of x

    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
        const int cb0 = a.cb_in_wg ? 0 : item % a.ncb, it = a.cb_in_wg ? item : item / a.ncb;      // cout block(s) of this item
        const int n = it / a.tiles_per_img, t = it - n * a.tiles_per_img;
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int oy0 = ty * a.To, ox0 = tx * a.To;
        const int ry0 = oy0 * a.stride - 1, rx0 = ox0 * a.stride - 1;
        const int toh = min(a.To, a.Ho - oy0), tow = min(a.To, a.Wo - ox0), opx = toh * tow, nof = (opx + 15) >> 4;
        const int inv_tow = 65536 / tow + 1;                    // p / tow for p < 64 as (p * inv_tow) >> 16

        // ---- every global request of the item goes out before anything is waited for: the first pair of pw1 cout fragments, the depthwise
        // tables of my channel group, the first eight K-steps of pw2's fragment, the biases, and the 9 x 9 input region ----
        half8 w1f[2][KS1];
        f32x4 b1v[2], s1v[2];
        auto load_w1 = [&](int j0) {                            // cout fragments wave + 8 j0, wave + 8 (j0 + 1) of pw1
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int fr = wave + NWV * (j0 + u), frc = fr < frags1 ? fr : 0;
                const _Float16 *p = a.w1 + (size_t)(frc * 16 + frow) * a.Cin_p + fq * 8;
#pragma unroll
                for (int k = 0; k < KS1; k++) w1f[u][k] = *(const half8 *)(p + k * 32);
                b1v[u] = *(const f32x4 *)(a.b1 + frc * 16 + fq * 4);
                s1v[u] = a.act1 == ACT_PRELU ? *(const f32x4 *)(a.s1 + frc * 16 + fq * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
        };
        half8 w2f[8];
        int fr2 = cb0 * NWV + wave, fr2c = fr2 < frags2 ? fr2 : 0;
        auto load_w2 = [&](int k0) {                            // K-steps k0 .. k0 + 7 of my pw2 fragment
            const _Float16 *p = a.w2 + (size_t)(fr2c * 16 + frow) * a.Gp + fq * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) w2f[k] = *(const half8 *)(p + (k0 + k < ks2 ? k0 + k : 0) * 32);
        };
        load_w1(0);
        f32x4 dwv[9], dwbv, dwsv;
        {
            const int c4 = (cg < cgs ? cg : 0) * 4;
#pragma unroll
            for (int tp = 0; tp < 9; tp++) dwv[tp] = *(const f32x4 *)(a.dww + (size_t)tp * a.Gp + c4);
            dwbv = *(const f32x4 *)(a.dwb + c4);
            dwsv = a.dw_act == ACT_PRELU ? *(const f32x4 *)(a.dws + c4) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
        {
            const int total = RPX * chunks;
            for (int i0 = 0; i0 < total; i0 += NTH * 6) {
                u32x4 v[6];
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int i = i0 + u * NTH + tid;
                    const int p = i / chunks, c = i - p * chunks;
                    const int ry = p / RW, rx = p - ry * RW;
                    const int iy = ry0 + ry, ix = rx0 + rx;
                    const bool ok = i < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a.ablate & 16);
                    v[u] = ok ? *(const u32x4 *)(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin_p + c * 8) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int i = i0 + u * NTH + tid;
                    const int p = i / chunks, c = i - p * chunks;
                    if (i < total) *(u32x4 *)(sX + p * a.xs_pitch + c * 16) = v[u];
                }
            }
        }
        __syncthreads();                                        // the region is in LDS (and the previous item is done with E / D)

        // ======== A: pw1 on the region -> E (fp16, bias + activation: the values the unfused layer stores) ========
        for (int j0 = 0; j0 * NWV + wave < frags1 && !(a.ablate & 1); j0 += 2) {
            if (j0 > 0) load_w1(j0);                            // (layers with more than 256 expanded channels: the next pair, one more trip)
#pragma unroll 1
            for (int pf = 0; pf < NPF; pf += 2) {               // (two pixel fragments per iteration; a fragment past the region clamps its reads and skips its stores = four independent accumulator chains; not unrolled
                                                                //  further: six iterations' fragment reads in flight at once spilled the resident weights)
                f32x4 acc[2][2];
#pragma unroll
                for (int h = 0; h < 2; h++) acc[h][0] = acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const char *xr[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int px = (pf + h) * 16 + frow;
                    xr[h] = sX + (px < RPX ? px : 0) * a.xs_pitch + fq * 16;
                }
#pragma unroll
                for (int k = 0; k < KS1; k++)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const half8 bf = *(const half8 *)(xr[h] + k * 64);
                        acc[h][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[0][k], bf, acc[h][0], 0, 0, 0);
                        acc[h][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[1][k], bf, acc[h][1], 0, 0, 0);
                    }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int px = (pf + h) * 16 + frow;
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int fr = wave + NWV * (j0 + u);
                        f32x4 v = acc[h][u] + b1v[u];
                        if (a.act1 == ACT_PRELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + s1v[u] * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
                        else if (a.act1 == ACT_RELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
                        if (px < RPX && fr < frags1) *(half4 *)(sE + px * a.e_pitch + (fr * 16 + fq * 4) * 2) = __builtin_convertvector(v, half4);
                    }
                }
            }
        }
        // pw2's operands are requested now (not at the top: 40 more live registers through stage A spilled): they travel during stage B
        load_w2(0);
        f32x4 b2v = a.b2 ? *(const f32x4 *)(a.b2 + fr2c * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 s2v = a.act2 == ACT_PRELU ? *(const f32x4 *)(a.s2 + fr2c * 16 + fq * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        __syncthreads();                                        // E is complete
        // ======== B: depthwise 3x3 on E -> D (fp32 fmaf chain in dwconv_nhwc's order, taps outside the image skipped) ========
        if (cg < cgs && pl < npl && !(a.ablate & 2)) {
            for (int p = pl; p < opx; p += npl) {
                const int oyl = (p * inv_tow) >> 16, oxl = p - oyl * tow;
                const int iy0 = (oy0 + oyl) * a.stride - 1, ix0 = (ox0 + oxl) * a.stride - 1;      // image coordinates of tap (0, 0)
                const char *e0 = sE + ((iy0 - ry0) * RW + (ix0 - rx0)) * a.e_pitch + cg * 8;
                f32x4 acc = dwbv;
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        if ((unsigned)(iy0 + dy) >= (unsigned)a.H || (unsigned)(ix0 + dx) >= (unsigned)a.W) continue;       // (a skipped tap, as in dwconv_nhwc)
                        const half4 e = *(const half4 *)(e0 + (dy * RW + dx) * a.e_pitch);
                        acc = __builtin_elementwise_fma(__builtin_convertvector(e, f32x4), dwv[dy * 3 + dx], acc);     // (four fused multiply-adds, packed two by two: the same bits as fmaf per channel)
                    }
                half4 o;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    float xv = acc[jj];
                    if (a.dw_act == ACT_RELU) xv = fmaxf(xv, 0.f);
                    else if (a.dw_act == ACT_PRELU) xv = xv > 0.f ? xv : xv * dwsv[jj];
                    o[jj] = (_Float16)xv;
                }
                *(half4 *)(sD + p * a.e_pitch + cg * 8) = o;
            }
        }
        __syncthreads();                                        // D is complete
        // ======== C: pw2 for my cout fragment, K = the expanded channels (cb_in_wg: for my fragment of every cout block in turn) ========
        for (int cb = cb0; cb < (a.cb_in_wg ? a.ncb : cb0 + 1); cb++) {
        if (cb > cb0) {                                         // the next block's operands: one more trip (half the workgroups instead)
            fr2 = cb * NWV + wave; fr2c = fr2 < frags2 ? fr2 : 0;
            load_w2(0);
            b2v = a.b2 ? *(const f32x4 *)(a.b2 + fr2c * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            s2v = a.act2 == ACT_PRELU ? *(const f32x4 *)(a.s2 + fr2c * 16 + fq * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
        f32x4 acc2[MAXOF];
#pragma unroll
        for (int p = 0; p < MAXOF; p++) acc2[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < ks2 && !(a.ablate & 4); k0 += 8) {
            if (k0 > 0) load_w2(k0);
#pragma unroll
            for (int p = 0; p < MAXOF; p++) {
                if (p >= nof) continue;                         // (wave-uniform; `continue`, not `break`: the accumulator indices stay compile-time)
                const int px = p * 16 + frow;
                const char *dr = sD + (px < opx ? px : 0) * a.e_pitch + fq * 16;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (k0 + k < ks2) acc2[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2f[k], *(const half8 *)(dr + (k0 + k) * 64), acc2[p], 0, 0, 0);
            }
        }
        // ======== epilogue: bias, block input (from the region in LDS), activation, 8-byte stores ========
        if (fr2 < frags2) {
            const int co = fr2 * 16 + fq * 4;
#pragma unroll
            for (int p = 0; p < MAXOF; p++) {
                const int px = p * 16 + frow;
                if (p >= nof || px >= opx) continue;
                const int oyl = (px * inv_tow) >> 16, oxl = px - oyl * tow;
                f32x4 v = acc2[p] + b2v;
                if (a.res) {                                    // stride 1, Cin_p == Cout_p: the block input at the output pixel
                    const int rp = (oy0 + oyl - ry0) * RW + (ox0 + oxl - rx0);
                    v += __builtin_convertvector(*(const half4 *)(sX + rp * a.xs_pitch + co * 2), f32x4);
                }
                if (a.act2 == ACT_PRELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + s2v * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
                else if (a.act2 == ACT_RELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
                *(half4 *)(a.out + ((size_t)(n * a.Ho + oy0 + oyl) * a.Wo + ox0 + oxl) * a.Cout_p + co) = __builtin_convertvector(v, half4);
            }
        }
        }
        __syncthreads();                                        // the residual reads of sX are done before the next item's region overwrites it
    }
}

}  // namespace

// region + both expanded maps (lower.py's _mbf_block mirrors this)
static int mbf_lds_bytes(int Cin_p, int Gp, int To, int stride) {
    const int rw = (To - 1) * stride + 3, rpx = rw * rw, drows = ((To * To + 15) / 16) * 16;
    const int xs = ((rpx * (Cin_p * 2 + 16) + 1023) / 1024) * 1024, e = ((rpx * (Gp * 2 + 16) + 1023) / 1024) * 1024;
    return xs + e + drows * (Gp * 2 + 16);
}
// output tile edge: 7 at stride 1; at stride 2 7 when the 15 x 15 region fits LDS, else 4
static int mbf_tile(int Cin_p, int Gp, int stride) {
    if (stride == 1) return 7;
    return mbf_lds_bytes(Cin_p, Gp, 7, 2) <= 160 * 1024 ? 7 : 4;
}

bool mbf_block_applicable(int H, int W, int Cin_p, int Gp, int Cout_p, int stride, bool res) {
    if (getenv("FID_NO_MBF_FUSE")) return false;
    if (stride != 1 && stride != 2) return false;
    if (Cin_p % 32 || Gp % 32 || Cout_p % 16 || Cin_p > 256 || Cin_p < 32 || Cout_p > 256 || Gp > 512 || Gp < 32) return false;
    if (res && (stride != 1 || Cin_p != Cout_p)) return false;
    if ((Gp / 4) > NTH) return false;                            // stage B: at least one pixel lane per 4-channel group
    return H >= 1 && W >= 1 && mbf_lds_bytes(Cin_p, Gp, mbf_tile(Cin_p, Gp, stride), stride) <= 160 * 1024;
}

int mbf_block_launch(fid_ctx *ctx, const void *x, const void *w1, const float *b1, const float *s1, int act1, const float *dww, const float *dwb,
                     const float *dws, int dw_act, const void *w2, const float *b2, const float *s2, int act2, bool res, void *out, int B, int H, int W,
                     int Cin_p, int Gp, int Cout_p, int stride) {
    FID_REQUIRE(mbf_block_applicable(H, W, Cin_p, Gp, Cout_p, stride, res), "mbf_block: shape not applicable");
    MBArgs a{};
    a.x = (const _Float16 *)x; a.w1 = (const _Float16 *)w1; a.b1 = b1; a.s1 = s1; a.dww = dww; a.dwb = dwb; a.dws = dws; a.w2 = (const _Float16 *)w2;
    a.b2 = b2; a.s2 = s2; a.out = (_Float16 *)out;
    a.H = H; a.W = W; a.stride = stride; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.Cin_p = Cin_p; a.Gp = Gp; a.Cout_p = Cout_p; a.act1 = act1; a.dw_act = dw_act; a.act2 = act2; a.res = res;
    a.To = mbf_tile(Cin_p, Gp, stride);
    const int rw = (a.To - 1) * stride + 3, rpx = rw * rw;
    a.tiles_x = cdiv(a.Wo, a.To);
    a.tiles_per_img = a.tiles_x * cdiv(a.Ho, a.To);
    a.ncb = cdiv(Cout_p / 16, NWV);
    // cout blocks as separate items recompute the expanded maps per block: worth it only while the items still fit one round of workgroups
    a.cb_in_wg = a.ncb > 1 && B * a.tiles_per_img * a.ncb > ctx->num_cus;
    a.n_items = B * a.tiles_per_img * (a.cb_in_wg ? 1 : a.ncb);
    a.xs_pitch = Cin_p * 2 + 16;
    a.e_pitch = Gp * 2 + 16;
    a.off_e = ((rpx * a.xs_pitch + 1023) / 1024) * 1024;
    a.off_d = a.off_e + ((rpx * a.e_pitch + 1023) / 1024) * 1024;
    const int lds = mbf_lds_bytes(Cin_p, Gp, a.To, stride);
    const int ks1 = Cin_p / 32;
    static const int ablate = getenv("FID_MB_ABLATE") ? atoi(getenv("FID_MB_ABLATE")) : 0;
    a.ablate = ablate;
    const int grid = std::min(a.n_items, ctx->num_cus * std::max(1, (160 * 1024) / lds));
#define MB_GO(K) do { if (rw == 9) { FID_TRY(ensure_dyn_lds(ctx, (const void *)mbf_block<K, 9>, lds)); \
                                         hipLaunchKernelGGL((mbf_block<K, 9>), dim3(grid), dim3(NTH), lds, ctx->stream, a); } \
                      else { FID_TRY(ensure_dyn_lds(ctx, (const void *)mbf_block<K, 15>, lds)); \
                             hipLaunchKernelGGL((mbf_block<K, 15>), dim3(grid), dim3(NTH), lds, ctx->stream, a); } } while (0)
    switch (ks1) { case 1: MB_GO(1); break; case 2: MB_GO(2); break; case 3: MB_GO(3); break; case 4: MB_GO(4); break;
                   case 5: MB_GO(5); break; case 6: MB_GO(6); break; case 7: MB_GO(7); break; default: MB_GO(8); break; }
#undef MB_GO
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
