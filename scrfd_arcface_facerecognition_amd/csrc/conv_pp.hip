// Ping-pong channel-chunked 3x3 / stride-1 convolution (autotuner generation 4).
//
// What the stamped profile of conv_chunked.hip showed (s_memtime per step, SCRFD layer2, 96 -> 96 channels):
//   * right after the per-chunk barrier all 8 waves are aligned, so everything that is not an MFMA -- decoding the
//     next prefetch, issuing its ~10 LDS-DMA instructions per wave (a wave BLOCKS in a DMA issue while the CU's
//     address path serves the other waves: 16 cycles per 1 KB instruction), the epilogue's loads, the output
//     transpose and stores -- runs on all four SIMDs at once while their matrix pipes idle: 3456 MFMA cycles in a
//     7000-cycle step, 12000 in steps with an epilogue;
//   * moving that work INTO the MFMA section does not help: both waves of a SIMD reach it at the same time.
//
// Here the two wave groups of a workgroup therefore play different ROLES per phase, like conv_direct.hip, while
// sharing the streamed weights, unlike a plain two-tile ping-pong:
//
//   work item   = (PAIR of 16x16-pixel tiles, block of CB = NI*16 output channels); group g owns tile 2*pair + g
//   step c      = one chunk of 32 input channels of the item: weight chunk W(c) [9][CB][32] (shared by both
//                 groups) + one haloed 18x18x32 patch chunk per group, P0(c) and P1(c)
//   phase 2c    : group 0 multiplies  W(c) x P0(c)  (one wave per SIMD, back-to-back MFMAs, 64 pixels x CB per wave)
//                 group 1 does memory: issues W(c+1) and P0(c+1), and the epilogue of its tile if c-1 finished it
//   phase 2c+1  : group 1 multiplies  W(c) x P1(c);  group 0 issues P1(c+1) and runs its epilogue if c finished its tile
//   one workgroup barrier per phase.
//
// Every prefetch is issued two phases before its first use and is waited for (vmcnt(0)) by the issuing wave at the
// end of its own compute phase in between, i.e. after a whole phase of matrix work.  A weight chunk crosses
// L2 -> LDS once per 512 pixels (conv_chunked: per 256).
// LDS: weights 2 x 9*CB*64 B + patches 2 groups x 2 buffers x 21 KB = 157.5 KB for CB = 64.
// The finished tile is transposed through the group's just-consumed patch buffer (32 couts per pass) into 16-byte
// stores.  Epilogue loads (bias / residual) are issued one memory phase early and land during the last multiply.
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TH = 16, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;   // 324 patch pixels
constexpr int CK = 32;                                               // input channels per chunk
constexpr int P_BLKS = 21;                                           // 1 KB DMA blocks of a patch chunk (16 pixels x 64 B each)
constexpr int P_BYTES = P_BLKS * 1024;

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }


struct PPArgs {
    const void *in;
    const void *w;
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, Cin_p, Cout_p;
    int act, flags, res_Cp;
    int tiles_x, n_tiles, n_cblk, n_items, n_chunks;
    FastDiv d_cblk, d_tpi, d_tx, d_chunks;   // divisions by n_cblk, tiles per image, tiles_x, n_chunks
    int tiles_per_img;
    unsigned in_bytes, w_bytes;
    int ablate;   // timing experiments only (FID_PP_ABLATE: 1 = no MFMA, 2 = no DMA, 4 = no epilogue)
};

template <int NI>   // CB = NI*16 output channels per item; wave tile 64 pixels x CB
__global__ void __launch_bounds__(512, 2) conv3x3_pp(const PPArgs a) {
    constexpr int CB = NI * 16, MI = 4;
    constexpr int W_BLKS = 9 * CB * 64 / 1024, W_BYTES = W_BLKS * 1024;
    constexpr int MAX_W = (W_BLKS + 3) / 4, MAX_P = (P_BLKS + 3) / 4;   // DMA instructions per wave of the issuing group
    static_assert(2 * W_BYTES + 4 * P_BYTES <= 160 * 1024, "LDS budget");
    static_assert(4 * 4096 <= P_BYTES, "staging pass must fit the patch buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sWr = smem;                                   // [2][W_BYTES]
    char *sPr = smem + 2 * W_BYTES;                     // [group][buffer][P_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);

    // my items: blockIdx.x, + gridDim.x, ...; steps = (local item, chunk) linearised
    const int my_items = blockIdx.x < a.n_items ? (a.n_items - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int S = my_items * a.n_chunks;
    if (S == 0) return;

    // step -> (item, chunk); (item, group) -> (image, tile row, tile column, cout block); false if the tile does not exist
    auto step_item = [&](int c, int &item, int &ck) {
        const int li = fastdiv(c, a.d_chunks);
        ck = c - li * a.n_chunks;
        item = blockIdx.x + li * gridDim.x;
    };
    auto decode = [&](int item, int g, int &n, int &ty, int &tx, int &cb) {
        const int pair = fastdiv(item, a.d_cblk);
        cb = item - pair * a.n_cblk;
        const int tile = pair * 2 + g;
        n = fastdiv(tile, a.d_tpi);
        const int r = tile - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx);
        tx = r - ty * a.tiles_x;
        return tile < a.n_tiles;
    };

    // ---- prefetch DMAs, issued by the 4 waves of ONE group: wave wg fills blocks wg, wg+4, ... ----
    auto issue_W = [&](int c) {                         // weight chunk of step c -> slot c & 1
        if (a.ablate & 2) return;
        int item, ck, n, ty, tx, cb;
        step_item(c, item, ck);
        decode(item, 0, n, ty, tx, cb);
        const int ubase = (cb * CB * 9 * a.Cin_p + ck * CK) * 2;
        char *dst = sWr + (c & 1) * W_BYTES;
#pragma unroll
        for (int k = 0; k < MAX_W; k++) {
            const int j = wg + 4 * k;
            if (j < W_BLKS) {
                // LDS row = t*CB + co (64 B each); rows past the filter bank fall outside the descriptor and read as 0
                const int row = j * 16 + (lane >> 2);
                const int t = row / CB, co = row - t * CB;
                const unsigned vo = (unsigned)(((co * 9 + t) * a.Cin_p + ((lane & 3) ^ swz64(row)) * 8) * 2 + ubase);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        }
    };
    auto issue_P = [&](int g, int c) {                  // patch chunk of group g's tile for step c -> buffer c & 1 of group g
        if (a.ablate & 2) return;
        int item, ck, n, ty, tx, cb;
        step_item(c, item, ck);
        if (!decode(item, g, n, ty, tx, cb)) return;    // (odd tile count: the buffer stays unread)
        const int y0 = ty * TH - 1, x0 = tx * TW - 1, c0 = ck * CK;
        char *dst = sPr + (g * 2 + (c & 1)) * P_BYTES;
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wg + 4 * k;
            if (j < P_BLKS) {
                const int row = j * 16 + (lane >> 2);
                const int py = row / PW, px = row - py * PW;
                const int iy = y0 + py, ix = x0 + px;
                const bool in = row < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + ((lane & 3) ^ swz64(row)) * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    // ---- epilogue state of my group's tile ----
    EpiArgs ep{a.bias, a.slope, a.res, a.out, a.Cout_p, a.H, a.W, a.act, a.flags, 0, a.H, a.W, a.res_Cp};
    EpiPix px[MI];
    int co0[NI];
    EpiRegs<NI, MI> R;
    bool epi_have = false;                              // px / co0 / R describe a tile of mine that exists
    auto epi_prefetch = [&](int item) {                 // one memory phase BEFORE the tile's last multiply
        int n, ty, tx, cb;
        epi_have = decode(item, grp, n, ty, tx, cb);
        if (!epi_have) return;
        int lo = lane;
        asm volatile("" : "+v"(lo));                    // opaque: per-lane address arithmetic stays here, is not hoisted
        const int frow = lo & 15, fq = lo >> 4;
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int oy = ty * TH + wg * MI + mi, ox = tx * TW + frow;
            px[mi].valid = oy < a.H && ox < a.W;
            px[mi].n = n; px[mi].oy = oy; px[mi].ox = ox;
            px[mi].m = px[mi].valid ? ((long long)n * a.H + oy) * a.W + ox : 0;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) co0[ni] = cb * CB + ni * 16 + fq * 4;
        epilogue_prefetch<NI, MI>(ep, px, co0, R);
    };

    f32x4 acc[NI][MI];
    auto epi_store = [&](char *stage) {                 // values -> fp16 -> transposed through `stage` (my 4 KB of it) -> 16-byte stores
        ep_half4 hv[NI][MI];
        epilogue_values_fast<NI, MI>(ep, acc, px, co0, R, hv);
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int frow = lo & 15, fq = lo >> 4;
        char *sS = stage + wg * 4096;
        const int co_item = co0[0] - fq * 4;            // first cout of the item
#pragma unroll
        for (int n0 = 0; n0 < NI; n0 += 2) {            // passes of 32 couts: staging rows of 64 B, chunk c ^ ((p >> 2) & 3)
            const int nn = NI - n0 < 2 ? NI - n0 : 2;   // couts of this pass / 16
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++) {
                    if (ni >= nn) continue;
                    const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                    *(ep_half4 *)(sS + p * 64 + ((c ^ ((p >> 2) & 3)) << 4) + (fq & 1) * 8) = hv[n0 + ni][mi];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) {
                const int p = s2 * 16 + (lo >> 2), c = lo & 3;
                const u32x4 v = *(const u32x4 *)(sS + p * 64 + ((c ^ ((p >> 2) & 3)) << 4));
                const int oy = px[0].oy + (p >> 4), ox = px[0].ox - frow + (p & 15);
                const int co = co_item + n0 * 16 + c * 8;
                if (c < nn * 2 && oy < a.H && ox < a.W && co < a.Cout_p)
                    *(u32x4 *)((char *)a.out + ((((size_t)px[0].n * a.H + oy) * a.W + ox) * a.Cout_p + co) * 2) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads returned before the next pass overwrites the rows
        }
    };

    // ---- prologue: W(0), P0(0), P1(0) ----
    if (grp == 0) { issue_P(0, 0); issue_P(1, 0); }
    else issue_W(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wg * MI) * PW + frow;
    bool epi_pending = false;                           // my tile's last multiply ran in my previous compute phase
    const int n_phases = 2 * S;
    for (int ph = 0; ph < n_phases; ph++) {
        const int c = ph >> 1;
        if (grp == (ph & 1)) {
            // =================== compute phase: W(c) x P_grp(c) ===================
            int item, ck;
            step_item(c, item, ck);
            if (ck == 0) {
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (!(a.ablate & 1)) {
                const char *sW = sWr + (c & 1) * W_BYTES, *sP = sPr + (grp * 2 + (c & 1)) * P_BYTES;
                // column dx, then the 6 patch rows of this wave: a pixel fragment (row r, shift dx) feeds every output row
                // mi = r - dy; the column's three taps keep their weights in registers, set dy is refetched for the next
                // column right after its last use (row 3 + dy)
                int plin = lin0, wlane = frow * 64 + ((fq ^ swz64(frow)) << 4);
                asm volatile("" : "+v"(plin), "+v"(wlane));
                half8 wq[3][NI], pq[3];
                auto load_w = [&](int dy, int dx) {     // rows t*CB + ni*16 + frow: the swizzle term only depends on frow
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) wq[dy][ni] = *(const half8 *)(sW + wlane + ((dy * 3 + dx) * CB + ni * 16) * 64);
                };
                auto load_p = [&](int q, int set) {     // q = dx*6 + r
                    const int lin = plin + (q % 6) * PW + q / 6;
                    pq[set] = *(const half8 *)(sP + lin * 64 + ((fq ^ swz64(lin)) << 4));
                };
                load_w(0, 0); load_p(0, 0); load_w(1, 0); load_p(1, 1); load_w(2, 0);
#pragma unroll
                for (int q = 0; q < 18; q++) {
                    const int dx = q / 6, r = q % 6;
                    if (q + 2 < 18) load_p(q + 2, (q + 2) % 3);
                    __builtin_amdgcn_sched_barrier(0);
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
            epi_pending = (ck == a.n_chunks - 1) && epi_have && !(a.ablate & 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetches I issued last phase (first use: next phase)
        } else {
            // =================== memory phase ===================
            // prefetches, two phases ahead of their first use
            if (c + 1 < S) {
                if (!(ph & 1)) { issue_W(c + 1); issue_P(0, c + 1); }   // I am group 1
                else issue_P(1, c + 1);                                  // I am group 0
            }
            // my previous compute phase was step cp: its chunk index tells whether my tile is complete
            const int cp = grp == 0 ? c : c - 1;
            if (epi_pending) {
                epi_store(sPr + (grp * 2 + (cp & 1)) * P_BYTES);
                epi_pending = false;
            }
            // my next compute phase is step cp + 1: if it is the last chunk of its item, fetch the epilogue operands now
            if (cp + 1 < S && cp + 1 >= 0) {
                int item, ck;
                step_item(cp + 1, item, ck);
                if (ck == a.n_chunks - 1) epi_prefetch(item);
            }
        }
        __syncthreads();
    }
    // group 1's last tile: its multiply was the final phase
    if (epi_pending) epi_store(sPr + (grp * 2 + ((S - 1) & 1)) * P_BYTES);
}

template <int NI>
int launch_pp(fid_ctx *ctx, const PPArgs &a) {
    constexpr size_t lds = 2 * (size_t)9 * NI * 16 * 64 + 4 * P_BYTES;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_pp<NI>, (int)((int)lds)));
    const int grid = std::min(a.n_items, ctx->num_cus);
    hipLaunchKernelGGL((conv3x3_pp<NI>), dim3(grid), dim3(512), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace

bool conv_pp_applicable(const ConvArgs &a) {
    // Measured (MI355X, round 1): on par with conv_chunked (SCRFD layer2 54.6 vs 53.6 us, IResNet layer2 28 vs 29 us) --
    // a single MFMA wave per SIMD reaches ~62 % of the pipe rate on LDS-fed fragments, which cancels what the hidden
    // epilogue buys.  Kept as an opt-in autotuner candidate (FID_PP=1, also forced by FID_FORCE_GEN=4 in the tests).
    const char *fg = getenv("FID_FORCE_GEN");
    if (!getenv("FID_PP") && !(fg && atoi(fg) == 4)) return false;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin_p % 32 == 0 && a.Cin_p >= 64 && a.Cout_p >= 32 &&
           a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 12 && a.W >= 12 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo));
}

// cb: output channels per work item (32, 48 or 64)
int conv_pp_launch(fid_ctx *ctx, const ConvArgs &c, int cb) {
    PPArgs a{};
    a.in = c.in; a.w = c.w; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.act = c.act; a.flags = c.flags; a.res_Cp = c.res_Cp;
    const int B = c.M / (c.Ho * c.Wo);
    a.tiles_x = cdiv(c.W, TW);
    a.tiles_per_img = a.tiles_x * cdiv(c.H, TH);
    a.n_tiles = B * a.tiles_per_img;
    a.n_cblk = cdiv(c.Cout_p, cb);
    a.n_items = cdiv(a.n_tiles, 2) * a.n_cblk;
    a.n_chunks = c.Cin_p / CK;
    a.d_cblk = fastdiv_make(a.n_cblk); a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    a.d_chunks = fastdiv_make(a.n_chunks);
    a.in_bytes = c.in_bytes;
    a.w_bytes = (unsigned)std::min<size_t>(c.w_bytes, (size_t)c.w_rows * 9 * c.Cin_p * 2);   // rows past the bank read as 0
    if (const char *e = getenv("FID_PP_ABLATE")) a.ablate = atoi(e);
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB");
    FID_REQUIRE(a.n_chunks >= 2, "ping-pong conv needs at least 64 input channels");
    if (cb == 32) return launch_pp<2>(ctx, a);
    if (cb == 48) return launch_pp<3>(ctx, a);
    if (cb == 64) return launch_pp<4>(ctx, a);
    set_error("ping-pong conv: cb=%d unsupported", cb);
    return FID_E_INVALID;
}

}  // namespace fid
