// Fused SCRFD "deep stem": uint8 frame -> conv3x3/s2+ReLU -> conv3x3+ReLU -> conv3x3+ReLU -> maxpool3x3/s2,
// one kernel, intermediates never leave the CU.
//
// Why: at 640x640 the three stem activations are 419 / 419 / 839 MB per 64-frame batch; as separate
// launches the stem moved ~4.4 GB through HBM and took 1.45 ms of the detector's 4.8 ms although it holds
// only 17 % of its MACs.  Fused, the only HBM traffic is the uint8 frames in (78.6 MB) and the pooled
// tensor out (210 MB).
//
// One workgroup (8 waves) produces an 8x8 tile of POOLED pixels per iteration of a persistent loop:
//   input patch 43x43x3 (uint8 -> exact integers 2p-255 as fp16; outside the frame = 0, the blob's padding)
//   S1 conv0 (K = 27 -> 32, stride 2)  21x21 x C0p   MFMA, pixel operand gathered element-wise from the patch
//   S2 conv1                            19x19 x C1p   MFMA 9 taps, patch + weights in LDS
//   S3 conv2                            17x17 x C2p   MFMA 9 taps
//   S4 maxpool 3x3/s2                   8x8  x C2p    -> global, 16 B per lane, whole pixel rows
// Positions of an intermediate map that fall outside the real feature map are written as 0 -- they are the
// next conv's zero padding (and harmless for the max-pool because every value is post-ReLU >= 0).
// The halo recompute costs 1.23x the MACs of the unfused stem.  LDS images are XOR-swizzled like
// conv_direct.hip; all weights (57 KB) stay resident for the workgroup's lifetime; the next tile's input
// pixels are prefetched into registers while the current tile is computed.
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 8;                          // pooled tile edge
constexpr int R2 = 2 * TP + 1, R1 = R2 + 2, R0 = R1 + 2, RI = 2 * R0 + 1;   // 17, 19, 21, 43
constexpr int N0 = R0 * R0, N1 = R1 * R1, N2 = R2 * R2;                      // 441, 361, 289
constexpr int C0P = 32, C1P = 32;

struct StemArgs {
    const uint8_t *img;       // [B, H, W, 3] BGR
    const _Float16 *w0;       // [32][32]  (k = tap*3 + c, zero padded), scale/2 + BN folded
    const float *b0;
    const _Float16 *w1;       // [C1P][9][C0P]
    const float *b1;
    const _Float16 *w2;       // [C2P][9][C1P]
    const float *b2;
    _Float16 *out;            // [B, Hp, Wp, C2P]
    int H, W, H1, W1, Hp, Wp; // frame, stride-2 maps, pooled map
    int tiles_x, tiles_y, n_tiles;
    int ablate;   // timing experiments only (FID_STEM_ABLATE bit mask: 1 S1, 2 S2, 4 S3, 8 S4, 16 input)
};

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ int swz128(int lin) { return lin & 7; }

template <int C2P, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 4) scrfd_stem_fused(const StemArgs a) {
    constexpr int NT = NW * 64;   // threads per workgroup
    constexpr int RS = 132;                             // halfs per patch row in LDS: the 33-dword window as fetched (3 bytes of lead-in)
    constexpr int INP_HALFS = RI * RS;                  // 5676, + 1 zero element
    constexpr int INP_BYTES = (INP_HALFS * 2 + 2 + 255) / 256 * 256;
    constexpr int O0_BYTES = ((N0 + 15) / 16 * 16) * 64;  // 448 rows
    constexpr int O1_BYTES = ((N1 + 15) / 16 * 16) * 64;  // 368 rows
    constexpr int ROW2 = C2P * 2;
    constexpr int O2_BYTES = ((N2 + 15) / 16 * 16) * ROW2;  // 304 rows
    constexpr int W0_BYTES = 32 * 64, W1_BYTES = 9 * C1P * 64, W2_BYTES = 9 * C2P * 64;
    constexpr int NI2 = C2P / 16;
    static_assert(INP_BYTES + O0_BYTES + O1_BYTES + O2_BYTES + W0_BYTES + W1_BYTES + W2_BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *sIn = (_Float16 *)smem;
    char *sO0 = smem + INP_BYTES;
    char *sO1 = sO0 + O0_BYTES;
    char *sO2 = sO1 + O1_BYTES;
    char *sW0 = sO2 + O2_BYTES;
    char *sW1 = sW0 + W0_BYTES;
    char *sW2 = sW1 + W1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // ---- weights -> LDS once (tap-major rows of 64 B, 16-byte chunks swizzled by row) ----
    for (int i = tid; i < W0_BYTES / 16; i += NT) ((u32x4 *)sW0)[i] = ((const u32x4 *)a.w0)[i];
    for (int i = tid; i < 9 * C1P * 4; i += NT) {        // 16-byte chunks: row = t*C1P + co, 4 per row
        const int row = i >> 2, slot = i & 3, t = row / C1P, co = row - t * C1P;
        const int chunk = slot ^ swz64(row);
        *(u32x4 *)(sW1 + row * 64 + slot * 16) = *(const u32x4 *)(a.w1 + ((co * 9 + t) * C0P + chunk * 8));
    }
    for (int i = tid; i < 9 * C2P * 4; i += NT) {
        const int row = i >> 2, slot = i & 3, t = row / C2P, co = row - t * C2P;
        const int chunk = slot ^ swz64(row);
        *(u32x4 *)(sW2 + row * 64 + slot * 16) = *(const u32x4 *)(a.w2 + ((co * 9 + t) * C1P + chunk * 8));
    }
    if (tid == 0) sIn[INP_HALFS] = (_Float16)0.f;       // the "k >= 27" / padding element

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // ---- input patch prefetch: aligned dwords of the frame rows (3 per thread) instead of single bytes.
    // A patch row starts at frame byte (32*tx - 7)*3 = 96*tx - 21; the dword-aligned window [96*tx - 24, +132)
    // covers it (the frame width is a multiple of 4, so rows are dword aligned).  When the registers are
    // committed each byte becomes the exact integer 2p-255 as fp16, or 0 where the pixel lies outside the
    // frame (the blob's zero padding; a real pixel 0 maps to -255, so validity comes from coordinates).
    constexpr int DROW = 33;                              // dwords per patch row
    constexpr int NDW = RI * DROW;                        // 1419 dwords per patch
    constexpr int DPT = (NDW + NT - 1) / NT;
    unsigned pre[DPT];
    auto prefetch = [&](int tile) {
        const int n = tile / tiles_per_img, r = tile - n * tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int iy0 = 2 * (2 * ty * TP - 3) - 1, bx0 = 96 * tx - 24;
        const uint8_t *base = a.img + (size_t)n * a.H * a.W * 3;
        const int rowbytes = a.W * 3;
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            const int d = tid + NT * i;
            const int pr = d / DROW, dc = d - pr * DROW;
            const int iy = iy0 + pr, bx = bx0 + dc * 4;
            const bool in = d < NDW && (unsigned)iy < (unsigned)a.H && bx >= 0 && bx + 4 <= rowbytes;
            pre[i] = in ? *(const unsigned *)(base + (size_t)iy * rowbytes + bx) : 0u;
        }
    };
    auto commit = [&](int tile) {
        const int n = tile / tiles_per_img, r = tile - n * tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int iy0 = 2 * (2 * ty * TP - 3) - 1, ix0 = 2 * (2 * tx * TP - 3) - 1;
        (void)n;
        // window element w (= byte of the dword window) holds pixel (w - 3) / 3 of the patch row; it lies inside the frame
        // for w in [wlo, whi).  Each dword becomes four consecutive halfs: ONE 8-byte LDS store instead of four 2-byte ones.
        const int wlo = 3 + 3 * max(0, -ix0), whi = 3 + 3 * min(RI, a.W - ix0);
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            const int d = tid + NT * i;
            if (d < NDW) {
                const int pr = d / DROW, dc = d - pr * DROW;
                const bool rin = (unsigned)(iy0 + pr) < (unsigned)a.H;
                half4 h;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int w = dc * 4 + j;
                    const int v = (pre[i] >> (8 * j)) & 0xFF;
                    h[j] = (rin && w >= wlo && w < whi) ? (_Float16)(float)(2 * v - 255) : (_Float16)0.f;
                }
                *(half4 *)(sIn + pr * RS + dc * 4) = h;
            }
        }
    };

    // ---- per-lane gather table of conv0's pixel operand: k = 8*fq + j -> element offset from the pixel base ----
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = fq * 8 + j;
        const int tap = k / 3, c = k - tap * 3, dy = tap / 3, dx = tap - dy * 3;
        koff[j] = k < 27 ? dy * RS + dx * 3 + c : -1;
    }

    // biases in registers for the workgroup's lifetime: read inside the tile loop they are a global load + full wait per
    // (subtile, cout group) -- up to 12 serial round trips per stage (measured: 477 -> 412 us at 32 frames).
    // (Software-pipelining the S2/S3 tap loops like conv_chunked.hip was measured 8 % SLOWER here, same box A/B.)
    f32x4 bias0[2], bias1[2], bias2[NI2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
        bias0[ni] = *(const f32x4 *)(a.b0 + ni * 16 + fq * 4);
        bias1[ni] = *(const f32x4 *)(a.b1 + ni * 16 + fq * 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI2; ni++) bias2[ni] = *(const f32x4 *)(a.b2 + ni * 16 + fq * 4);

    int tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(tile);
    __syncthreads();

    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int n = tile / tiles_per_img, r = tile - n * tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int oy2 = 2 * ty * TP - 1, ox2 = 2 * tx * TP - 1;     // conv2-region origin in the stride-2 map
        const int oy1 = oy2 - 1, ox1 = ox2 - 1, oy0 = oy2 - 2, ox0 = ox2 - 2;

        if (!(a.ablate & 16)) commit(tile);
        const int next = tile + gridDim.x;
        if (next < a.n_tiles && !(a.ablate & 16)) prefetch(next);
        __syncthreads();

        // ---------------- S1: conv0, K = 27 (padded 32), stride 2, C0P outputs ----------------
        if (!(a.ablate & 1))
        if (!(a.ablate & 1))
        for (int sub = wave; sub < (N0 + 15) / 16; sub += NW) {
            const int q = min(sub * 16 + frow, N0 - 1);
            const int y = q / R0, x = q - y * R0;
            const int base = 2 * y * RS + 2 * x * 3 + 3;
            half8 pf;
#pragma unroll
            for (int j = 0; j < 8; j++) pf[j] = sIn[koff[j] >= 0 ? base + koff[j] : INP_HALFS];
            const int qd = sub * 16 + frow;
            const bool inside = qd < N0 && (unsigned)(oy0 + y) < (unsigned)a.H1 && (unsigned)(ox0 + x) < (unsigned)a.W1;
#pragma unroll
            for (int ni = 0; ni < 2; ni++) {
                const half8 wf = *(const half8 *)(sW0 + (ni * 16 + frow) * 64 + fq * 16);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, pf, acc, 0, 0, 0);
                const f32x4 b = bias0[ni];
                half4 h = __builtin_elementwise_max(__builtin_convertvector(acc + b, half4), half4{0, 0, 0, 0});   // ReLU after the rounding: same result
                if (!inside) h = half4{0, 0, 0, 0};
                if (qd < N0) *(half4 *)(sO0 + qd * 64 + (((ni * 2 + (fq >> 1)) ^ swz64(qd)) << 4) + (fq & 1) * 8) = h;
            }
        }
        __syncthreads();

        // ---------------- S2: conv1 3x3, C0P -> C1P on the 21x21 patch -> 19x19 ----------------
        if (!(a.ablate & 2)) {
            constexpr int NSUB = (N1 + 15) / 16;          // 23
            constexpr int MI = (NSUB + NW - 1) / NW;      // subtiles per wave (contiguous)
            int lin0[MI], qd[MI];
            bool have[MI], inside[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int sub = wave * MI + mi;
                have[mi] = sub < NSUB;
                qd[mi] = sub * 16 + frow;
                const int q = min(qd[mi], N1 - 1);
                const int y = q / R1, x = q - y * R1;
                lin0[mi] = y * R0 + x;
                inside[mi] = have[mi] && qd[mi] < N1 && (unsigned)(oy1 + y) < (unsigned)a.H1 && (unsigned)(ox1 + x) < (unsigned)a.W1;
            }
            f32x4 acc[2][MI];
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (have[0])
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int dy = t / 3, dx = t % 3;
                half8 wf[2], pf[MI];
#pragma unroll
                for (int ni = 0; ni < 2; ni++) {
                    const int row = t * C1P + ni * 16 + frow;
                    wf[ni] = *(const half8 *)(sW1 + row * 64 + ((fq ^ swz64(row)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    const int lin = lin0[mi] + dy * R0 + dx;
                    pf[mi] = *(const half8 *)(sO0 + lin * 64 + ((fq ^ swz64(lin)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[mi], acc[ni][mi], 0, 0, 0);
            }
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                if (!have[mi] || qd[mi] >= N1) continue;
#pragma unroll
                for (int ni = 0; ni < 2; ni++) {
                    const f32x4 b = bias1[ni];
                    half4 h = __builtin_elementwise_max(__builtin_convertvector(acc[ni][mi] + b, half4), half4{0, 0, 0, 0});
                    if (!inside[mi]) h = half4{0, 0, 0, 0};
                    *(half4 *)(sO1 + qd[mi] * 64 + (((ni * 2 + (fq >> 1)) ^ swz64(qd[mi])) << 4) + (fq & 1) * 8) = h;
                }
            }
        }
        __syncthreads();

        // ---------------- S3: conv2 3x3, C1P -> C2P on the 19x19 patch -> 17x17 ----------------
        if (!(a.ablate & 4)) {
            constexpr int NSUB = (N2 + 15) / 16;          // 19
            constexpr int MI = (NSUB + NW - 1) / NW;
            int lin0[MI], qd[MI];
            bool have[MI], inside[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                // subtiles dealt round-robin so that every wave gets 2 and three waves get a third
                const int sub = wave + NW * mi;
                have[mi] = sub < NSUB;
                qd[mi] = sub * 16 + frow;
                const int q = min(qd[mi], N2 - 1);
                const int y = q / R2, x = q - y * R2;
                lin0[mi] = y * R1 + x;
                inside[mi] = have[mi] && qd[mi] < N2 && (unsigned)(oy2 + y) < (unsigned)a.H1 && (unsigned)(ox2 + x) < (unsigned)a.W1;
            }
            f32x4 acc[NI2][MI];
#pragma unroll
            for (int ni = 0; ni < NI2; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int dy = t / 3, dx = t % 3;
                half8 wf[NI2], pf[MI];
#pragma unroll
                for (int ni = 0; ni < NI2; ni++) {
                    const int row = t * C2P + ni * 16 + frow;
                    wf[ni] = *(const half8 *)(sW2 + row * 64 + ((fq ^ swz64(row)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    const int lin = lin0[mi] + dy * R1 + dx;
                    pf[mi] = *(const half8 *)(sO1 + lin * 64 + ((fq ^ swz64(lin)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    if (mi == MI - 1 && !have[mi]) continue;       // wave-uniform: only waves 0..2 own a third subtile
#pragma unroll
                    for (int ni = 0; ni < NI2; ni++)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[mi], acc[ni][mi], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                if (!have[mi] || qd[mi] >= N2) continue;
#pragma unroll
                for (int ni = 0; ni < NI2; ni++) {
                    const f32x4 b = bias2[ni];
                    half4 h = __builtin_elementwise_max(__builtin_convertvector(acc[ni][mi] + b, half4), half4{0, 0, 0, 0});
                    if (!inside[mi]) h = half4{0, 0, 0, 0};
                    const int chunk = ni * 2 + (fq >> 1);
                    const int sw = ROW2 == 128 ? swz128(qd[mi]) : swz64(qd[mi]);
                    *(half4 *)(sO2 + qd[mi] * ROW2 + ((chunk ^ sw) << 4) + (fq & 1) * 8) = h;
                }
            }
        }
        __syncthreads();

        // ---------------- S4: maxpool 3x3 / stride 2 (pad 1) -> 8x8 x C2P, 16 B per lane ----------------
        if (!(a.ablate & 8)) {
            constexpr int CPP = ROW2 / 16;                 // chunks per pixel (8 or 4)
            for (int i = tid; i < TP * TP * CPP; i += NT) {
                const int pp = i / CPP, c = i - pp * CPP;
                const int py = pp / TP, px = pp - py * TP;
                const int gy = ty * TP + py, gx = tx * TP + px;
                half8 m;
#pragma unroll
                for (int e = 0; e < 8; e++) m[e] = (_Float16)0.f;       // all candidates are >= 0 (post-ReLU; outside = 0)
#pragma unroll
                for (int dy = 0; dy < 3; dy++)
#pragma unroll
                    for (int dx = 0; dx < 3; dx++) {
                        const int q = (2 * py + dy) * R2 + (2 * px + dx);
                        const int sw = ROW2 == 128 ? swz128(q) : swz64(q);
                        const half8 v = *(const half8 *)(sO2 + q * ROW2 + ((c ^ sw) << 4));
                        m = __builtin_elementwise_max(m, v);       // v_pk_max_f16 x 4 (all values are finite and >= 0)
                    }
                if (gy < a.Hp && gx < a.Wp) *(half8 *)(a.out + (((size_t)n * a.Hp + gy) * a.Wp + gx) * C2P + c * 8) = m;
            }
        }
        __syncthreads();
    }
}

template <int C2P, int NW>
int launch_stem(fid_ctx *ctx, const StemArgs &a) {
    constexpr int INP_BYTES = ((RI * 132) * 2 + 2 + 255) / 256 * 256;
    constexpr size_t lds = INP_BYTES + ((N0 + 15) / 16 * 16) * 64 + ((N1 + 15) / 16 * 16) * 64 + ((N2 + 15) / 16 * 16) * (C2P * 2) +
                           32 * 64 + 9 * C1P * 64 + 9 * C2P * 64;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)scrfd_stem_fused<C2P, NW>, (int)((int)lds)));
    const int grid = std::min(a.n_tiles, ctx->num_cus);
    hipLaunchKernelGGL((scrfd_stem_fused<C2P, NW>), dim3(grid), dim3(NW * 64), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace

// img [B,H,W,3] u8 -> out [B, H/4, W/4, C2p]; blob offsets resolved by the caller
int stem_fused_launch(fid_ctx *ctx, const uint8_t *img, int B, int H, int W, const void *w0, const float *b0, const void *w1,
                      const float *b1, const void *w2, const float *b2, void *out, int C2p) {
    FID_REQUIRE(H % 4 == 0 && W % 4 == 0, "fused stem: frame %dx%d not a multiple of 4", W, H);
    StemArgs a{};
    a.img = img; a.w0 = (const _Float16 *)w0; a.b0 = b0; a.w1 = (const _Float16 *)w1; a.b1 = b1;
    a.w2 = (const _Float16 *)w2; a.b2 = b2; a.out = (_Float16 *)out;
    a.H = H; a.W = W; a.H1 = H / 2; a.W1 = W / 2; a.Hp = H / 4; a.Wp = W / 4;
    a.tiles_x = cdiv(a.Wp, TP); a.tiles_y = cdiv(a.Hp, TP);
    a.n_tiles = B * a.tiles_x * a.tiles_y;
    if (const char *e = getenv("FID_STEM_ABLATE")) a.ablate = atoi(e);
    const bool w16 = getenv("FID_STEM_W16") != nullptr;  // 16 waves were measured: 1.6x SLOWER (128-VGPR cap spills, costlier barriers)
    if (C2p == 64) return w16 ? launch_stem<64, 16>(ctx, a) : launch_stem<64, 8>(ctx, a);
    if (C2p == 32) return w16 ? launch_stem<32, 16>(ctx, a) : launch_stem<32, 8>(ctx, a);
    set_error("fused stem: C2p=%d unsupported", C2p);
    return FID_E_INVALID;
}

}  // namespace fid
