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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
__global__ void __launch_bounds__(512, 2) conv3x3_direct(const DirectArgs a) {
    constexpr int ROWB = CIN_P * 2;            // bytes per pixel / per (tap, cout) weight row
    constexpr int CPP = ROWB / 16;             // 16-byte chunks per row
    constexpr int PXI = 64 / CPP;              // rows written by one wave-wide LDS-DMA instruction (1 KB)
    constexpr int N_PINSTR = (NPIX + PXI - 1) / PXI;
    constexpr int PATCH_ROWS = N_PINSTR * PXI;
    constexpr int PATCH_BYTES = PATCH_ROWS * ROWB;
    constexpr int W_ROWS = 9 * COUT_P;
    constexpr int W_BYTES = W_ROWS * ROWB;
    constexpr int N_WINSTR = W_BYTES / 1024;
    constexpr int MAX_PI = (N_PINSTR + 3) / 4 + 1; // patch instructions per wave of a group (upper bound)
    constexpr int NI = COUT_P / 16, MI = 4, KK = CIN_P / 32;
    // output staging: each wave transposes its 64-pixel x COUT_P tile through LDS so that the global stores
    // are 16 bytes per lane and cover whole pixel rows (1 KB contiguous per instruction) instead of 8-byte
    // pieces scattered over 16 cache lines -- the stores were the bottleneck of the HBM-bound layers.
    constexpr int OROWB = COUT_P * 2;           // bytes per output pixel
    constexpr int OCPP = OROWB / 16;            // 16-byte chunks per output pixel
    constexpr int SCR_BYTES = 64 * OROWB;       // per wave
    constexpr bool SCR_DEDICATED = W_BYTES + 2 * PATCH_BYTES + 8 * SCR_BYTES <= 160 * 1024;
    static_assert(SCR_DEDICATED || 4 * SCR_BYTES <= (N_PINSTR - 4) * 1024, "patch buffer too small to double as staging");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sW = smem;                            // [9][COUT_P][CIN_P] fp16, tap-major, shared by both groups

    // 8 waves = two groups of 4 (one wave of each group per SIMD).  The groups work on different tiles
    // and run half a period apart: while one group's waves feed the matrix cores from LDS, the other
    // group stores its finished tile and fetches its next patch (ping-pong; one workgroup barrier per
    // half period keeps them interleaved).
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;
    char *sP = smem + W_BYTES + grp * PATCH_BYTES;   // this group's patch [PATCH_ROWS][CIN_P]
    // staging area of this wave: a dedicated region when LDS allows, else the first 4*SCR_BYTES of the group's
    // (then dead) patch buffer; in that case each wave only DMAs into the 1 KB blocks it owns (its own staging
    // blocks + a share of the tail), so no wave's next patch can land on another wave's unread staging data
    char *sS = SCR_DEDICATED ? smem + W_BYTES + 2 * PATCH_BYTES + (grp * 4 + wg) * SCR_BYTES : sP + wg * SCR_BYTES;
    constexpr int SCR_BLKS = SCR_BYTES / 1024;       // 1 KB DMA blocks per wave staging area
    auto dma_block = [&](int k) {                    // k-th patch block this wave fills
        if (SCR_DEDICATED) return wg + 4 * k;
        return k < SCR_BLKS ? wg * SCR_BLKS + k : 4 * SCR_BLKS + wg + 4 * (k - SCR_BLKS);
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);

    // ---- filter bank -> LDS (once per workgroup) ----
    for (int j = wave; j < N_WINSTR; j += 8) {
        const int p = j * 64 + lane;            // 16-byte position in the LDS image
        const int row = p / CPP, slot = p % CPP;
        const int t = row / COUT_P, co = row - t * COUT_P;
        const int chunk = slot ^ swz<ROWB>(row);
        const unsigned vo = (unsigned)(((co * 9 + t) * CIN_P + chunk * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(sW + j * 1024), 16, vo, 0, 0, 0);
    }

    // ---- per-lane constants of the patch fetch: which patch pixel / chunk each of my DMA lanes fills ----
    // packed py | px << 8 | channel offset << 16 (one register per block; py = 255 marks a padding row)
    int p_pk[MAX_PI];
#pragma unroll
    for (int k = 0; k < MAX_PI; k++) {
        const int j = dma_block(k);
        const int lin = j * PXI + lane / CPP;
        int py = lin / PW;
        const int px = lin - py * PW;
        const int ch = ((lane % CPP) ^ swz<ROWB>(lin)) * 8;
        if (j >= N_PINSTR || lin >= NPIX) py = 255;   // never inside an image
        p_pk[k] = py | (px << 8) | (ch << 16);
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    auto fetch_patch = [&](int tile) {
        const int n = tile / tiles_per_img;
        const int r = tile - n * tiles_per_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
        for (int k = 0; k < MAX_PI; k++) {
            const int j = dma_block(k);
            if (j < N_PINSTR) {
                int pk = p_pk[k];
                asm volatile("" : "+v"(pk));                     // opaque: unpack here, do not hoist three registers per block
                const int py = pk & 255, iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
                const bool in = py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * CIN_P + (pk >> 16)) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(sP + j * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    // pair q of this workgroup = tiles (2q, 2q+1): group 0 takes the even one, group 1 the odd one
    const int n_pairs = (a.n_tiles + 1) >> 1;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);   // XCD-major order: consecutive tile pairs (overlapping halos) in one L2
    const int my_pairs = bid < n_pairs ? (n_pairs - 1 - bid) / gridDim.x + 1 : 0;
    auto tile_of = [&](int i) { return (bid + i * gridDim.x) * 2 + grp; };

    if (my_pairs > 0 && tile_of(0) < a.n_tiles) fetch_patch(tile_of(0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    EpiArgs ep{a.bias, a.slope, a.res, a.out, COUT_P, a.H, a.W, a.act, a.flags, a.nsig, a.H, a.W, a.res_Cp};
    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wg * MI) * PW + frow;   // patch pixel of (tile row wg*4, tile col frow), tap (0,0)
    if (grp == 1) __syncthreads();            // half-period phase shift of group 1
    for (int i = 0; i < my_pairs; i++) {
        const int tile = tile_of(i);
        const bool have = tile < a.n_tiles;
        const int next = tile_of(i + 1);
        const bool have_next = (i + 1 < my_pairs) && next < a.n_tiles;

        f32x4 acc[NI][MI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- compute half-period: 9 taps x Cin_p/32 MFMA steps out of LDS, no barrier inside ----
        // Order: K chunk, column shift dx, then the 6 patch rows of this wave.  A pixel fragment (row r, shift dx) feeds
        // every output row mi = r - dy, so a chunk needs 18 pixel + 9*NI weight fragment reads instead of 36 + 9*NI
        // (the naive per-tap order made the LDS reads as long as the MFMAs).  The column's three taps keep their
        // weights in registers; set dy is refetched for the next column right after its last use (row 3 + dy).
        if (have && !(a.flags & (1 << 28))) {
            int plin = lin0, wrow = frow;
            asm volatile("" : "+v"(plin), "+v"(wrow));    // opaque: fragment addresses are recomputed per tile, not kept live
            half8 wq[3][NI], pq[3];
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                auto load_w = [&](int dy, int dx) {        // rows t*COUT_P + ni*16 + frow: the swizzle term only depends on frow
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        wq[dy][ni] = *(const half8 *)(sW + (wrow + (dy * 3 + dx) * COUT_P + ni * 16) * ROWB + (((kk * 4 + fq) ^ swz<ROWB>(wrow)) << 4));
                };
                auto load_p = [&](int q, int set) {        // q = dx*6 + r
                    const int lin = plin + (q % 6) * PW + q / 6;
                    pq[set] = *(const half8 *)(sP + lin * ROWB + (((kk * 4 + fq) ^ swz<ROWB>(lin)) << 4));
                };
                load_w(0, 0); load_p(0, 0); load_w(1, 0); load_p(1, 1); load_w(2, 0);
#pragma unroll
                for (int q = 0; q < 18; q++) {
                    const int dx = q / 6, r = q % 6;
                    if (q + 2 < 18) load_p(q + 2, (q + 2) % 3);
                    __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of this row's MFMAs
#pragma unroll
                    for (int dy = 0; dy < 3; dy++) {
                        const int mi = r - dy;
                        if (mi < 0 || mi >= MI) continue;
#pragma unroll
                        for (int ni = 0; ni < NI; ni++)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[dy][ni], pq[q % 3], acc[ni][mi], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (dx < 2 && r >= 3) load_w(r - 3, dx + 1);
                }
            }
        }
        __syncthreads();   // my group is done reading its patch; the other group starts computing

        // ---- memory half-period: store my finished tile, fetch my next patch (latency hides behind the other group) ----
        if (SCR_DEDICATED && have_next) fetch_patch(next);
        if (have && !(a.flags & (1 << 29))) {
            int lo = lane;                                       // opaque lane id: the pixel / staging / store address arithmetic of
            asm volatile("" : "+v"(lo));                         // the epilogue is done here, per tile, instead of living across the
            const int lane = lo, frow = lo & 15, fq = lo >> 4;   // MFMA section (where it spilled)
            EpiPix px[MI];
            int co0[NI];
            {
                const int tl = have ? tile : 0;
                const int n = tl / tiles_per_img;
                const int r = tl - n * tiles_per_img;
                const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
                const int ox = tx * TW + frow;
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    const int oy = ty * TH + wg * MI + mi;
                    px[mi].valid = have && oy < a.H && ox < a.W;
                    px[mi].n = n; px[mi].oy = oy; px[mi].ox = ox;
                    px[mi].m = px[mi].valid ? ((long long)n * a.H + oy) * a.W + ox : 0;
                }
#pragma unroll
                for (int ni = 0; ni < NI; ni++) co0[ni] = ni * 16 + fq * 4;
            }
            if (COUT_P == 32 && (a.flags & CF_OUT_F32)) {     // fp32 outputs (detector head maps) only exist 32 wide
                epilogue_tile<NI, MI>(ep, acc, px, co0);
            } else {
                ep_half4 hv[NI][MI];
                epilogue_values<NI, MI>(ep, acc, px, co0, hv);
                // wave-local transpose: lane (pixel frow of row mi, cout group fq) -> [pixel][chunk ^ (pixel & (OCPP-1))]
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) {
                        const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                        *(ep_half4 *)(sS + p * OROWB + ((c ^ (p & (OCPP - 1))) << 4) + (fq & 1) * 8) = hv[ni][mi];
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int n = tile / tiles_per_img;
                const int r = tile - n * tiles_per_img;
                const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
                constexpr int PPI = 64 / OCPP;         // pixels per store instruction
#pragma unroll
                for (int s2 = 0; s2 < 64 / PPI; s2++) {
                    const int p = s2 * PPI + lane / OCPP, c = lane % OCPP;
                    const u32x4 v = *(const u32x4 *)(sS + p * OROWB + ((c ^ (p & (OCPP - 1))) << 4));
                    const int oy = ty * TH + wg * MI + (p >> 4), ox = tx * TW + (p & 15);
                    if (oy < a.H && ox < a.W)
                        *(u32x4 *)((char *)a.out + ((((size_t)n * a.H + oy) * a.W + ox) * OROWB) + c * 16) = v;
                }
            }
        }
        if (!SCR_DEDICATED) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads returned before the DMA may overwrite them
            if (have_next) fetch_patch(next);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // patches landed; roles swap
    }
    if (grp == 0) __syncthreads();
}

template <int CIN_P, int COUT_P>
int launch_direct(fid_ctx *ctx, const DirectArgs &a) {
    constexpr int ROWB = CIN_P * 2, PXI = 64 / (ROWB / 16);
    constexpr int PATCH_BYTES = ((NPIX + PXI - 1) / PXI) * PXI * ROWB;
    constexpr size_t base = (size_t)9 * COUT_P * ROWB + 2 * PATCH_BYTES;
    constexpr size_t scr = (size_t)8 * 64 * COUT_P * 2;
    constexpr size_t lds = base + scr <= 160 * 1024 ? base + scr : base;
    static_assert(lds <= 160 * 1024, "LDS budget");
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_direct<CIN_P, COUT_P>, (int)((int)lds)));
    const int grid = std::min((a.n_tiles + 1) / 2, ctx->num_cus);
    hipLaunchKernelGGL((conv3x3_direct<CIN_P, COUT_P>), dim3(grid), dim3(512), lds, ctx->stream, a);
    return FID_OK;
}

}  // namespace

bool conv_direct_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_DIRECT")) return false;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (a.Cin_p == 32 || a.Cin_p == 64) &&
           (a.Cout_p == 32 || a.Cout_p == 64) && a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 16 && a.W >= 16 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX)) && (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo)) &&
           (a.Cout_p == 32 || (!(a.flags & CF_OUT_F32) && a.nsig == 0));
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
