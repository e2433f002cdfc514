// Producer / consumer 3x3 / stride-1 convolution with RESIDENT weights for 64 -> 64 channels (autotuner generation 7):
// conv_direct.hip's data flow (whole filter bank in LDS for the workgroup's lifetime, one haloed 18x18x64 patch per
// 16x16 tile, two patch slots) with conv_pc.hip's division of labour.
//
// conv_direct.hip splits its 8 waves into two ping-pong groups; a group multiplies with ONE wave per SIMD (~62 % of the
// pipe rate on LDS-fed fragments) and its memory half-period (epilogue + stores + patch fetch, serial) is 2.5x the
// matrix time: 39 % MFMA utilisation on SCRFD layer1.  Here
//   waves 0..7   consumers: all eight multiply the SAME tile (4 pixel groups x 2 cout groups, two waves per SIMD),
//                K = 64 as two 32-channel passes over the resident weights, no barrier inside a tile
//   waves 8..11  producers: fetch the patch of tile t+1 while tile t is multiplied, carry the residual tile in
//                (coalesced) and the finished tile out (conv_pc.hip's staging protocol; the staging area is the patch
//                slot the finished tile was read from, which the next patch prefetch refills afterwards)
// One barrier set per TILE (conv_pc.hip: per 32-channel chunk).
// LDS: weights 2 x 36 KB + 2 tile slots x (2 x 21 KB) = 156 KB.
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TH = 16, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;   // 324 patch pixels
constexpr int CB = 64, NI = 2, MI = 4;                               // couts per item; per consumer wave 64 pixels x 32 couts
constexpr int NCH = 2, CK = 32;                                      // K = 64 as two chunks of 32 input channels
constexpr int P_BLKS = 21, P_BYTES = P_BLKS * 1024;                  // one chunk of a patch: 1 KB DMA blocks (16 pixels x 64 B)
constexpr int T_BLKS = NCH * P_BLKS, T_BYTES = T_BLKS * 1024;        // a tile slot: both chunks
constexpr int W_BLKS = 9 * CB * 64 / 1024, W_BYTES = W_BLKS * 1024;  // one chunk of the filter bank
constexpr int N_CONS = 8, N_PROD = 4;
constexpr int OROWB = NI * 32, OCPP = NI * 2, OMASK = OCPP - 1;      // staging rows of a consumer wave: 64 B, 4 chunks
constexpr int S_BLKS = N_CONS * OCPP;                                // staging blocks (1 KB) of a tile: 32
constexpr int MAX_W = NCH * W_BLKS / N_PROD, MAX_P = (T_BLKS + N_PROD - 1) / N_PROD, MAX_S = S_BLKS / N_PROD;
static_assert(2 * W_BYTES + 2 * T_BYTES <= 160 * 1024 && S_BLKS * 1024 <= T_BYTES, "LDS budget");

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct PCRArgs {
    const void *in;
    const void *w;
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W;
    int act, flags;
    int tiles_x, tiles_per_img, n_tiles;
    FastDiv d_tpi, d_tx;
    unsigned in_bytes, w_bytes;
    int plain_order;   // experiment (FID_PCR_PLAIN_ORDER): tiles in workgroup-id order instead of XCD-major order
};

__global__ void __launch_bounds__((N_CONS + N_PROD) * 64, 3) conv3x3_pcr(const PCRArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sWr = smem, *sTr = smem + NCH * W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // bid = my place in XCD-major order: the workgroups of one XCD work on consecutive tiles (their halos overlap in that L2)
    const int bid = a.plain_order ? (int)blockIdx.x : xcd_major_id(blockIdx.x, gridDim.x);
    const int my_tiles = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    auto decode_tile = [&](int tile, int &n, int &ty, int &tx) {
        n = fastdiv(tile, a.d_tpi);
        const int r = tile - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const bool has_res = a.res != nullptr;

    if (wave >= N_CONS) {
        // ======================================= PRODUCERS =======================================
        const int pw = wave - N_CONS;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        // ---- the filter bank, once: LDS row = t*CB + co of chunk ck, 64 B each, 16-byte chunks swizzled by row ----
#pragma unroll
        for (int k = 0; k < MAX_W; k++) {
            const int j = pw + N_PROD * k, ck = j / W_BLKS, jj = j - ck * W_BLKS;
            const int row = jj * 16 + (lane >> 2);
            const int t = row / CB, co = row - t * CB;
            const unsigned vo = (unsigned)(((co * 9 + t) * 64 + ck * CK + ((lane & 3) ^ swz64(row)) * 8) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(sWr + j * 1024), 16, vo, 0, 0, 0);
        }
        // ---- per-lane constants of my patch blocks: block j = chunk j / 21, 16 patch pixels x 64 B ----
        auto patch_pk = [&](int k) {                            // py | px << 8 | channel offset << 16 (py = 255: padding row)
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int j = pw + N_PROD * k, ck = j / P_BLKS, jj = j - ck * P_BLKS;
            const int row = jj * 16 + (lo >> 2);
            int py = row / PW;
            const int px = row - py * PW;
            if (row >= NPIX) py = 255;
            return py | (px << 8) | ((ck * CK + ((lo & 3) ^ swz64(row)) * 8) << 16);
        };
        int p_off[MAX_P];                                       // interior tiles: byte offset relative to the patch's top-left pixel
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int pk = patch_pk(k), py = pk & 255, px = (pk >> 8) & 255;
            p_off[k] = py == 255 ? -1 : ((py * a.W + px) * 64 + (pk >> 16)) * 2;
        }
        auto issue_patch = [&](int tile, int slot) {
            int n, ty, tx;
            decode_tile(tile, n, ty, tx);
            const int y0 = ty * TH - 1, x0 = tx * TW - 1;
            char *dst = sTr + slot * T_BYTES;
            if (y0 >= 0 && x0 >= 0 && y0 + TH + 2 <= a.H && x0 + TW + 2 <= a.W) {   // interior tile
                const int base = ((n * a.H + y0) * a.W + x0) * 64 * 2;
#pragma unroll
                for (int k = 0; k < MAX_P; k++) {
                    const int j = pw + N_PROD * k;
                    if (j >= T_BLKS) continue;
                    const unsigned vo = p_off[k] < 0 ? OOB : (unsigned)(p_off[k] + base);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < MAX_P; k++) {
                const int j = pw + N_PROD * k;
                if (j >= T_BLKS) continue;
                const int pk = patch_pk(k);
                const int py = pk & 255, iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
                const bool in = py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * 64 + (pk >> 16)) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        };
        // ---- staged tile <-> global: block b = consumer wave b / OCPP, chunks (b % OCPP)*64 ... +63 of its [64 px][64 B] rows ----
        auto stage_pk = [&](int k) {                            // pixel | cout chunk << 8 | consumer wave << 16
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int b = pw + N_PROD * k, wv = b / OCPP;
            const int gl = (b - wv * OCPP) * 64 + lo;
            const int p = gl / OCPP, c = (gl - p * OCPP) ^ (p & OMASK);
            return p | (c << 8) | (wv << 16);
        };
        int s_off[MAX_S];                                       // byte offset of my segment relative to the tile's first output
#pragma unroll
        for (int k = 0; k < MAX_S; k++) {
            const int pk = stage_pk(k), p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
            s_off[k] = ((((wv & 3) * MI + (p >> 4)) * a.W + (p & 15)) * 64 + (wv >> 2) * NI * 16 + c * 8) * 2;
        }
        u32x4 sv[MAX_S], rv[MAX_S];
        auto read_tile = [&](char *slot) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++) sv[k] = *(const u32x4 *)(slot + (pw + N_PROD * k) * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read back before my patch DMAs refill the slot
        };
        auto move_tile = [&](int tile, const void *src, bool load) {   // load: rv <- src (residual), else: out <- sv; true: whole tile inside
            int n, ty, tx;
            decode_tile(tile, n, ty, tx);
            if (ty * TH + TH <= a.H && tx * TW + TW <= a.W) {
                char *base = (char *)(load ? const_cast<void *>(src) : a.out) + (((size_t)n * a.H + ty * TH) * a.W + tx * TW) * 64 * 2;
#pragma unroll
                for (int k = 0; k < MAX_S; k++) {
                    if (load) rv[k] = *(const u32x4 *)(base + (unsigned)s_off[k]);
                    else *(u32x4 *)(base + (unsigned)s_off[k]) = sv[k];
                }
                return true;
            }
#pragma unroll
            for (int k = 0; k < MAX_S; k++) {
                const int pk = stage_pk(k);
                const int p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
                const int oy = ty * TH + (wv & 3) * MI + (p >> 4), ox = tx * TW + (p & 15);
                const size_t off = ((((size_t)n * a.H + oy) * a.W + ox) * 64 + (wv >> 2) * NI * 16 + c * 8) * 2;
                const bool in = oy < a.H && ox < a.W;
                if (load) {
                    rv[k] = u32x4{0u, 0u, 0u, 0u};
                    if (in) rv[k] = *(const u32x4 *)((const char *)src + off);
                } else if (in) {
                    *(u32x4 *)((char *)a.out + off) = sv[k];
                }
            }
            return false;
        };
        auto write_residual = [&](char *slot) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++) *(u32x4 *)(slot + (pw + N_PROD * k) * 1024 + lane * 16) = rv[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };

        // prologue: (weights issued above) patches of my first two tiles
        issue_patch(bid, 0);
        if (my_tiles > 1) issue_patch(bid + gridDim.x, 1);
        if (has_res) move_tile(bid, a.res, true);
        bool stores_young = false;                              // the youngest MAX_S memory operations are output stores
        int tile = bid;                                         // the tile the consumers multiply in iteration t
        for (int t = 0; t < my_tiles; t++) {
            // the patch of tile t (and at t = 0 the weights) and my residual segments must have landed; stores may fly on
            if (stores_young) wait_vmcnt_n<MAX_S>();
            else wait_vmcnt_n<0>();
            stores_young = false;
            raw_barrier();                                      // T(t)
            if (t > 0) {
                char *slot = sTr + ((t - 1) & 1) * T_BYTES;     // the finished tile's patch slot is the staging area
                if (has_res) {
                    write_residual(slot);                       // residual of tile t-1 (loaded during iteration t-1)
                    raw_barrier();                              // R(t)
                }
                raw_barrier();                                  // F(t): the consumers have staged tile t-1
                read_tile(slot);
                if (t + 1 < my_tiles) issue_patch(tile + gridDim.x, (t + 1) & 1);   // prefetch first, into the slot just drained
                if (has_res) move_tile(tile, a.res, true);      // residual of tile t: needed right after T(t+1)
                stores_young = move_tile(tile - gridDim.x, nullptr, false) && !has_res ? true : false;
                if (has_res) stores_young = false;              // (loads are older than the stores: wait for everything)
            }
            tile += gridDim.x;
        }
        wait_vmcnt_n<0>();
        raw_barrier();                                          // tail T
        {
            char *slot = sTr + ((my_tiles - 1) & 1) * T_BYTES;
            if (has_res) {
                write_residual(slot);
                raw_barrier();                                  // tail R
            }
            raw_barrier();                                      // tail F
            read_tile(slot);
            move_tile(tile - gridDim.x, nullptr, false);
        }
        return;
    }

    // ========================================= CONSUMERS =========================================
    const int grp = wave >> 2, wg = wave & 3;               // cout group, pixel group
    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wg * MI) * PW + frow;
    EpiArgs ep{a.bias, a.slope, a.res, a.out, 64, a.H, a.W, a.act, a.flags, 0, a.H, a.W, 64};
    EpiPix px[MI];
    int co0[NI];
    EpiRegs<NI, MI> R;
    ep_half4 hv[NI][MI];
    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        co0[ni] = grp * NI * 16 + ni * 16 + fq * 4;
        R.bb[ni] = (a.bias != nullptr && !(a.flags & CF_BORDER)) ? *(const ep_f32x4 *)(a.bias + co0[ni]) : ep_f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.act == ACT_PRELU) R.sl[ni] = *(const ep_f32x4 *)(a.slope + co0[ni]);
    }
    auto set_pixels = [&](int tile) {                        // only the border-class bias needs the pixel coordinates
        if (!(a.flags & CF_BORDER)) return;
        int n, ty, tx;
        decode_tile(tile, n, ty, tx);
        int lo = lane;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            px[mi].n = n; px[mi].oy = ty * TH + wg * MI + mi; px[mi].ox = tx * TW + (lo & 15);
            px[mi].valid = true; px[mi].m = 0;
        }
    };
    auto stage_addr = [&](char *slot, int mi, int ni, int lo) {
        const int p = mi * 16 + (lo & 15), c = ni * 2 + (lo >> 5);
        return slot + wave * (64 * OROWB) + p * OROWB + ((c ^ (p & OMASK)) << 4) + ((lo >> 4) & 1) * 8;
    };
    auto epi_values = [&](char *slot, int tile) {            // slot: where the producers dropped the residual tile (res layers)
        set_pixels(tile);
        if (has_res) {
            int lo = lane;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) R.rr[ni][mi] = *(const ep_half4 *)stage_addr(slot, mi, ni, lo);
        }
        epilogue_values_fast<NI, MI>(ep, acc, px, co0, R, hv);
    };
    auto epi_stage = [&](char *slot) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++) *(ep_half4 *)stage_addr(slot, mi, ni, lo) = hv[ni][mi];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staged before the barrier that hands the slot to the producers
    };

    int tile = bid;
    for (int t = 0; t < my_tiles; t++) {
        raw_barrier();                                      // T(t): patch of tile t landed; everyone is done with tile t-1's slot
        if (t > 0) {
            char *slot = sTr + ((t - 1) & 1) * T_BYTES;
            if (has_res) {
                raw_barrier();                              // R(t): the residual tile is in the slot
                epi_values(slot, tile - gridDim.x);         // (tile t-1's sums are still in acc)
            }
            epi_stage(slot);
            raw_barrier();                                  // F(t)
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
#pragma unroll
            for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        // two 32-channel passes, row-sharing tap order (conv_chunked.hip): column dx, then the 6 patch rows of this wave
#pragma unroll
        for (int ck = 0; ck < NCH; ck++) {
            const char *sW = sWr + ck * W_BYTES, *sP = sTr + (t & 1) * T_BYTES + ck * P_BYTES;
            int plin = lin0, wlane = ((grp * NI) * 16 + frow) * 64 + ((fq ^ swz64(frow)) << 4);
            asm volatile("" : "+v"(plin), "+v"(wlane));   // opaque: recompute the fragment addresses per pass
            half8 wq[3][NI], pq[3];
            auto load_w = [&](int dy, int dx) {
#pragma unroll
                for (int ni = 0; ni < NI; ni++) wq[dy][ni] = *(const half8 *)(sW + wlane + ((dy * 3 + dx) * CB + ni * 16) * 64);
            };
            auto load_p = [&](int q, int set) {            // q = dx*6 + r
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
        if (!has_res) epi_values(nullptr, tile);            // kept in hv until the next iteration stages them
        tile += gridDim.x;
    }
    raw_barrier();                                          // tail T
    {
        char *slot = sTr + ((my_tiles - 1) & 1) * T_BYTES;
        if (has_res) {
            raw_barrier();                                  // tail R
            epi_values(slot, tile - gridDim.x);
        }
        epi_stage(slot);
        raw_barrier();                                      // tail F
    }
}

}  // namespace

bool conv_pcr_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_PCR")) return false;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin_p == 64 && a.Cout_p == 64 && a.w_rows == 64 &&
           a.H == a.Ho && a.W == a.Wo && a.H >= 16 && a.W >= 16 && !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo && a.res_Cp == 64));
}

int conv_pcr_launch(fid_ctx *ctx, const ConvArgs &c) {
    PCRArgs a{};
    a.in = c.in; a.w = c.w; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.act = c.act; a.flags = c.flags;
    const int B = c.M / (c.Ho * c.Wo);
    a.tiles_x = cdiv(c.W, TW);
    a.tiles_per_img = a.tiles_x * cdiv(c.H, TH);
    a.n_tiles = B * a.tiles_per_img;
    a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    a.in_bytes = c.in_bytes;
    a.plain_order = getenv("FID_PCR_PLAIN_ORDER") != nullptr;
    a.w_bytes = (unsigned)std::min<size_t>(c.w_bytes, (size_t)64 * 9 * 64 * 2);
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB");
    constexpr size_t lds = NCH * (size_t)W_BYTES + 2 * (size_t)T_BYTES;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_pcr, (int)((int)lds)));
    const int grid = std::min(a.n_tiles, ctx->num_cus);
    hipLaunchKernelGGL(conv3x3_pcr, dim3(grid), dim3((N_CONS + N_PROD) * 64), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
