// Direct 3x3 / stride-1 convolution with the WHOLE filter bank resident in LDS and a haloed
// activation patch per tile -- the kernel for the high-resolution, <= 64-channel layers of SCRFD
// (stem, layer1, PAFPN) and of IResNet-50's first stage, i.e. the layers where the implicit-GEMM
// kernel (conv.hip) is limited by L2->LDS traffic: there every activation byte is re-fetched once per
// tap (9x) and a 64-wide output gives only 64 flop per byte.  Here each CU
//
//   * loads the layer's weights [Cout_p][9][Cin_p] fp16 ONCE (<= 72 KB) and keeps them for all its tiles,
//   * walks 16x16-pixel output tiles (persistent grid-stride loop), fetching for each an 18x18 halo
//     patch [(TH+2)(TW+2)][Cin_p] by LDS-DMA (buffer_load ... lds: no VGPR staging; image borders and
//     the zero padding come from the buffer descriptor's bounds check) -- 1.27 input bytes per output
//     pixel-channel instead of 9 -- double-buffered: the next tile's patch lands while this one is
//     multiplied,
//   * runs the 9 taps x Cin_p/32 MFMA steps back to back out of LDS with no barrier in between
//     (one barrier per tile), v_mfma_f32_16x16x32_f16, A = weights, B = pixels, wave tile
//     64 pixels (4 tile rows) x Cout_p.
//
// LDS images are XOR-swizzled in 16-byte chunks (f = pixel&7 for 128-byte pixels, (pixel>>1)&3 for
// 64-byte pixels): conflict-free ds_read_b128 for ANY patch offset (checked against the gfx950 bank
// map for all alignments); the LDS-DMA destination stays lane-linear, the swizzle is applied to each
// lane's SOURCE address.
//
// Epilogue identical to conv.hip (bias with optional 9 border classes, residual, ReLU/PReLU, sigmoid,
// fp16/fp32 NHWC store).
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TH = 16, TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;  // 324 patch pixels

template <int ROWB>
__device__ __forceinline__ int swz(int lin) {
    return ROWB == 128 ? (lin & 7) : ((lin >> 1) & 3);
}

struct DirectArgs {
    const void *in;
    const void *w;
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, B;
    int act, flags, nsig;
    int res_Cp;
    int tiles_x, tiles_y, n_tiles;
    unsigned in_bytes, w_bytes;
};

template <int CIN_P, int COUT_P>
__global__ void __launch_bounds__(256, 1) conv3x3_direct(const DirectArgs a) {
    constexpr int ROWB = CIN_P * 2;            // bytes per pixel / per (tap, cout) weight row
    constexpr int CPP = ROWB / 16;             // 16-byte chunks per row
    constexpr int PXI = 64 / CPP;              // rows written by one wave-wide LDS-DMA instruction (1 KB)
    constexpr int N_PINSTR = (NPIX + PXI - 1) / PXI;
    constexpr int PATCH_ROWS = N_PINSTR * PXI;
    constexpr int PATCH_BYTES = PATCH_ROWS * ROWB;
    constexpr int W_ROWS = 9 * COUT_P;
    constexpr int W_BYTES = W_ROWS * ROWB;
    constexpr int N_WINSTR = W_BYTES / 1024;
    constexpr int MAX_PI = (N_PINSTR + 3) / 4; // patch instructions per wave
    constexpr int NI = COUT_P / 16, MI = 4, KK = CIN_P / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sW = smem;                            // [9][COUT_P][CIN_P] fp16, tap-major
    char *sP = smem + W_BYTES;                  // [2][PATCH_ROWS][CIN_P]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);

    // ---- filter bank -> LDS (once per workgroup) ----
    for (int j = wave; j < N_WINSTR; j += 4) {
        const int p = j * 64 + lane;            // 16-byte position in the LDS image
        const int row = p / CPP, slot = p % CPP;
        const int t = row / COUT_P, co = row - t * COUT_P;
        const int chunk = slot ^ swz<ROWB>(row);
        const unsigned vo = (unsigned)(((co * 9 + t) * CIN_P + chunk * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(sW + j * 1024), 16, vo, 0, 0, 0);
    }

    // ---- per-lane constants of the patch fetch: which patch pixel / chunk each of my DMA lanes fills ----
    int p_py[MAX_PI], p_px[MAX_PI], p_ch[MAX_PI];
#pragma unroll
    for (int k = 0; k < MAX_PI; k++) {
        const int j = wave + 4 * k;
        const int lin = j * PXI + lane / CPP;
        p_py[k] = lin / PW;
        p_px[k] = lin - p_py[k] * PW;
        p_ch[k] = ((lane % CPP) ^ swz<ROWB>(lin)) * 8;
        if (j >= N_PINSTR || lin >= NPIX) p_py[k] = -100000;  // never inside an image
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    auto fetch_patch = [&](int tile, int buf) {
        const int n = tile / tiles_per_img;
        const int r = tile - n * tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
        for (int k = 0; k < MAX_PI; k++) {
            const int j = wave + 4 * k;
            if (j < N_PINSTR) {
                const int iy = y0 + p_py[k], ix = x0 + p_px[k];
                const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * CIN_P + p_ch[k]) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(sP + buf * PATCH_BYTES + j * 1024),
                                                         16, vo, 0, 0, 0);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < a.n_tiles) fetch_patch(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    EpiArgs ep{a.bias, a.slope, a.res, a.out, COUT_P, a.H, a.W, a.act, a.flags, a.nsig, a.H, a.W, a.res_Cp};
    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wave * MI) * PW + frow;   // patch pixel of (tile row wave*4, tile col frow), tap (0,0)
    int it = 0;
    for (; tile < a.n_tiles; tile += gridDim.x, it++) {
        const int cur = it & 1;
        const int next = tile + gridDim.x;
        if (next < a.n_tiles) fetch_patch(next, cur ^ 1);
        const char *P = sP + cur * PATCH_BYTES;

        // epilogue operands (bias rows, residual values) are requested now and arrive during the K loop
        EpiPix px[MI];
        int co0[NI];
        {
            const int n = tile / tiles_per_img;
            const int r = tile - n * tiles_per_img;
            const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
            const int ox = tx * TW + frow;
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int oy = ty * TH + wave * MI + mi;
                px[mi].valid = oy < a.H && ox < a.W;
                px[mi].n = n; px[mi].oy = oy; px[mi].ox = ox;
                px[mi].m = px[mi].valid ? ((long long)n * a.H + oy) * a.W + ox : 0;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++) co0[ni] = ni * 16 + fq * 4;
        }
        EpiRegs<NI, MI> R;
        epilogue_prefetch<NI, MI>(ep, px, co0, R);

        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (!(a.flags & (1 << 28)))
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int dy = t / 3, dx = t % 3;
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                half8 wf[NI], pf[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    const int row = t * COUT_P + ni * 16 + frow;
                    wf[ni] = *(const half8 *)(sW + row * ROWB + (((kk * 4 + fq) ^ swz<ROWB>(row)) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    const int lin = lin0 + (mi + dy) * PW + dx;
                    pf[mi] = *(const half8 *)(P + lin * ROWB + (((kk * 4 + fq) ^ swz<ROWB>(lin)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[mi], acc[ni][mi], 0, 0, 0);
            }
        }

        // the next patch (requested before the K loop) has long landed: this wait is free, and it is placed
        // BEFORE the output stores so that those stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(a.flags & (1 << 29))) epilogue_finish<NI, MI>(ep, acc, px, co0, R);
        else if (acc[0][0][0] == 12345.678f) epilogue_finish<NI, MI>(ep, acc, px, co0, R);
        __syncthreads();
    }
}

template <int CIN_P, int COUT_P>
int launch_direct(fid_ctx *ctx, const DirectArgs &a) {
    constexpr int ROWB = CIN_P * 2, PXI = 64 / (ROWB / 16);
    constexpr int PATCH_BYTES = ((NPIX + PXI - 1) / PXI) * PXI * ROWB;
    constexpr size_t lds = (size_t)9 * COUT_P * ROWB + 2 * PATCH_BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        FID_HIP(hipFuncSetAttribute((const void *)conv3x3_direct<CIN_P, COUT_P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int grid = std::min(a.n_tiles, ctx->num_cus);
    hipLaunchKernelGGL((conv3x3_direct<CIN_P, COUT_P>), dim3(grid), dim3(256), lds, ctx->stream, a);
    return FID_OK;
}

}  // namespace

bool conv_direct_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_DIRECT")) return false;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (a.Cin_p == 32 || a.Cin_p == 64) &&
           (a.Cout_p == 32 || a.Cout_p == 64) && a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 16 && a.W >= 16 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX)) && (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo));
}

int conv_direct_launch(fid_ctx *ctx, const ConvArgs &c) {
    DirectArgs a{};
    a.in = c.in; a.w = c.w; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.B = c.M / (c.Ho * c.Wo);
    a.act = c.act; a.flags = c.flags; a.nsig = c.nsig; a.res_Cp = c.res_Cp;
    if (const char *dbg = getenv("FID_DIRECT_ABLATE")) a.flags |= atoi(dbg) << 28;   // timing experiments only
    a.tiles_x = cdiv(c.W, TW); a.tiles_y = cdiv(c.H, TH);
    a.n_tiles = a.B * a.tiles_x * a.tiles_y;
    a.in_bytes = c.in_bytes; a.w_bytes = c.w_bytes;
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB");
    int rc;
    if (c.Cin_p == 64 && c.Cout_p == 64) rc = launch_direct<64, 64>(ctx, a);
    else if (c.Cin_p == 32 && c.Cout_p == 64) rc = launch_direct<32, 64>(ctx, a);
    else if (c.Cin_p == 64 && c.Cout_p == 32) rc = launch_direct<64, 32>(ctx, a);
    else rc = launch_direct<32, 32>(ctx, a);
    FID_TRY(rc);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
