// Producer / consumer channel-chunked 3x3 / stride-1 convolution, TWO tiles per fetched weight chunk (autotuner generation 8).
//
// conv_pc.hip's stamps say what bounds its steps: the global->LDS fill.  A CU takes one 1-KB piece per ~62-76 cycles whatever
// the mix of weight and patch pieces and however many waves issue them (13-16 B/clk), and a CB = 64 step needs 57 KB of fill
// (36 KB weights + 21 KB patch) for 2304 MFMA cycles per SIMD: 4500 cycles of fill, the consumers wait at the barrier.
// Here an item is a PAIR of 16x16 tiles (consecutive in (image, tile row, tile column) order -- on 14x14 maps two faces)
// times 64 couts: per 32-channel step one weight chunk (36 KB) and two patches (2 x 21 KB) feed 4608 MFMA cycles per SIMD,
// 78 KB of fill for twice the matrix work, and every weight fragment a consumer reads from LDS is used for both tiles.
//
//   waves 0..7   CONSUMERS: wave (wg, grp) owns output rows 4wg..4wg+3 of BOTH tiles x couts 32grp..32grp+31
//                           (acc: 2 tiles x 2 x 4 fragments); row-sharing tap order as in conv_pc.hip
//   waves 8..11  PRODUCERS: all LDS-DMA (a quarter of the blocks each), output stores and residual loads
//
// LDS: 2 weight slots x 36 KB + 2 patch slots x 2 x 21 KB = 156 KB; both streams one step ahead.  A step that starts a new
// item first writes out the previous pair: tile 0 is staged in the weight slot of the step before, tile 1 in its patch slot
// (both free by then), one barrier (F) hands them to the producers, which read them back as 16-byte cout segments, refill the
// slots with the next chunks and store.  Residual tiles travel the other way through the same two areas (barrier R).
// The bias / activation arithmetic runs in that flush step, one tile at a time (the packed results of both tiles would not fit
// the register budget beside 64 accumulator registers).
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TH = 16, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;   // 324 patch pixels
constexpr int CK = 32;                                               // input channels per chunk
constexpr int P_BLKS = 21, P_BYTES = P_BLKS * 1024;                  // 1 KB DMA blocks of one tile's patch chunk
constexpr int P2_BYTES = 2 * P_BYTES;                                // a patch slot: both tiles
constexpr int N_CONS = 8, N_PROD = 4;
constexpr int NI = 2, MI = 4, CB = 2 * NI * 16;                      // 64 couts per workgroup
constexpr int W_BLKS = 9 * CB * 64 / 1024, W_BYTES = W_BLKS * 1024;  // 36 blocks
constexpr int LDS_BYTES = 2 * W_BYTES + 2 * P2_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct PC2Args {
    const void *in;
    const void *w;
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, Cin_p, Cout_p;
    int act, flags, res_Cp;
    int tiles_x, tiles_per_img, n_tiles, n_cblk, n_items, n_chunks;
    FastDiv d_cblk, d_tpi, d_tx;
    unsigned in_bytes, w_bytes;
    int w_packed;   // a.w is repack.hip's kind-1 image: every 1-KB weight piece contiguous
};

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ void __launch_bounds__((N_CONS + N_PROD) * 64, 3) conv3x3_pc2(const PC2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sWr = smem, *sPr = smem + 2 * W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);     // consecutive items (the cout blocks of a pair) in one L2
    const int my_items = bid < a.n_items ? (a.n_items - 1 - bid) / gridDim.x + 1 : 0;
    const int n_steps = my_items * a.n_chunks;
    if (n_steps == 0) return;

    // item -> (tile pair, cout block); tile t -> (image, tile row, tile column); t >= n_tiles: the odd pair's missing half
    auto decode_item = [&](int item, int &pair, int &cb) {
        pair = fastdiv(item, a.d_cblk);
        cb = item - pair * a.n_cblk;
    };
    auto decode_tile = [&](int t, int &n, int &ty, int &tx) {
        n = fastdiv(t, a.d_tpi);
        const int r = t - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const bool has_res = a.res != nullptr;
    constexpr int OROWB = NI * 32, OCPP = NI * 2, OMASK = OCPP - 1;      // staged tile: [wave][64 pixels][64 B], chunks swizzled by pixel
    static_assert(N_CONS * 64 * OROWB <= W_BYTES && N_CONS * 64 * OROWB <= P2_BYTES, "staging areas");

    if (wave >= N_CONS) {
        // ======================================= PRODUCERS =======================================
        const int pw = wave - N_CONS;
        __builtin_assume(pw >= 0 && pw < N_PROD);
        constexpr int MAX_W = W_BLKS / N_PROD, MAX_P = (P_BLKS + N_PROD - 1) / N_PROD;     // 9, 6
        static_assert(W_BLKS % N_PROD == 0, "weight blocks");
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        int w_off[MAX_W];                                      // my 16 B inside the cout block's [CB][9][Cin_p] rows (LDS row = t*CB + co)
#pragma unroll
        for (int k = 0; k < MAX_W; k++) {
            const int row = (pw + N_PROD * k) * 16 + (lane >> 2);
            const int t = row / CB, co = row - t * CB;
            w_off[k] = ((co * 9 + t) * a.Cin_p + ((lane & 3) ^ swz64(row)) * 8) * 2;
            if (a.w_packed) w_off[k] = (pw + N_PROD * k) * 1024 + lane * 16;
        }
        int p_pk[MAX_P];                                       // py | px << 8 | channel offset << 16 (py = 255: padding row) of my k-th patch block
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int row = (pw + N_PROD * k) * 16 + (lane >> 2);
            int py = row / PW;
            const int px = row - py * PW;
            if (row >= NPIX) py = 255;
            p_pk[k] = py | (px << 8) | ((((lane & 3) ^ swz64(row)) * 8) << 16);
        }
        auto patch_pk = [&](int k) {
            int pk = p_pk[k];
            asm volatile("" : "+v"(pk));                       // opaque: unpack at the use, do not hoist three registers per block
            return pk;
        };
        // staged tile: block b = consumer wave b / SK, 16-byte segments (b % SK)*64 ... +63 of that wave's 64 pixels x 4 segments
        constexpr int SK = OCPP, S_BLKS = N_CONS * SK, MAX_S = S_BLKS / N_PROD;            // 4, 32, 8
        auto stage_pk = [&](int k) {                            // pixel | cout chunk << 8 | consumer wave << 16 of my k-th staging block
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int b = pw + N_PROD * k, wv = b / SK;
            const int gl = (b - wv * SK) * 64 + lo;
            const int p = gl / OCPP, c = (gl - p * OCPP) ^ (p & OMASK);
            return p | (c << 8) | (wv << 16);
        };
        int s_off[MAX_S];                                       // byte offset of my segment relative to the tile's first output
#pragma unroll
        for (int k = 0; k < MAX_S; k++) {
            const int pk = stage_pk(k), p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
            s_off[k] = ((((wv & 3) * MI + (p >> 4)) * a.W + (p & 15)) * a.Cout_p + (wv >> 2) * NI * 16 + c * 8) * 2;
        }
        u32x4 tv[2][MAX_S];                                     // my share of the two staged tiles (outputs on their way out, residuals on their way in)
        auto read_tiles = [&](const char *s0, const char *s1) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++) {
                tv[0][k] = *(const u32x4 *)(s0 + (pw + N_PROD * k) * 1024 + lane * 16);
                tv[1][k] = *(const u32x4 *)(s1 + (pw + N_PROD * k) * 1024 + lane * 16);
            }
            lds_done();                                         // read back before my DMAs refill the two areas
        };
        auto write_tiles = [&](char *s0, char *s1) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++) {
                *(u32x4 *)(s0 + (pw + N_PROD * k) * 1024 + lane * 16) = tv[0][k];
                *(u32x4 *)(s1 + (pw + N_PROD * k) * 1024 + lane * 16) = tv[1][k];
            }
            lds_done();
        };
        // global <-> tv for the pair of `item`; load: the residual tiles, else: store the outputs
        auto move_tiles = [&](int item, bool load) {            // true: exactly 2*MAX_S memory operations were issued
            int pair, cb;
            decode_item(item, pair, cb);
            bool exact = true;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int t = pair * 2 + h;
                if (t >= a.n_tiles) {                           // (wave-uniform) the odd pair's missing half
                    exact = false;
                    if (load) {
#pragma unroll
                        for (int k = 0; k < MAX_S; k++) tv[h][k] = u32x4{0u, 0u, 0u, 0u};
                    }
                    continue;
                }
                int n, ty, tx;
                decode_tile(t, n, ty, tx);
                if (ty * TH + TH <= a.H && tx * TW + TW <= a.W && cb * CB + CB <= a.Cout_p) {   // whole tile inside the tensor
                    const size_t base = ((((size_t)n * a.H + ty * TH) * a.W + tx * TW) * a.Cout_p + cb * CB) * 2;
#pragma unroll
                    for (int k = 0; k < MAX_S; k++) {
                        if (load) tv[h][k] = *(const u32x4 *)((const char *)a.res + base + (unsigned)s_off[k]);
                        else *(u32x4 *)((char *)a.out + base + (unsigned)s_off[k]) = tv[h][k];
                    }
                    continue;
                }
                exact = false;
#pragma unroll
                for (int k = 0; k < MAX_S; k++) {
                    const int pk = stage_pk(k);
                    const int p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
                    const int oy = ty * TH + (wv & 3) * MI + (p >> 4), ox = tx * TW + (p & 15);
                    const int co = cb * CB + (wv >> 2) * NI * 16 + c * 8;
                    const bool in = oy < a.H && ox < a.W && co < a.Cout_p;
                    const size_t off = ((((size_t)n * a.H + oy) * a.W + ox) * a.Cout_p + co) * 2;
                    if (load) {
                        tv[h][k] = u32x4{0u, 0u, 0u, 0u};
                        if (in) tv[h][k] = *(const u32x4 *)((const char *)a.res + off);
                    } else if (in) {
                        *(u32x4 *)((char *)a.out + off) = tv[h][k];
                    }
                }
            }
            return exact;
        };
        struct Cursor {
            int item, ck;      // work item / chunk the NEXT issue fetches
            int w_base;        // weights: byte offset of the item's cout block (chunk 0)
            int n[2], y0[2], x0[2];   // patches: image and top-left input pixel of each tile's haloed patch (n < 0: no tile)
        };
        auto cursor_decode = [&](Cursor &c) {
            int pair, cb;
            decode_item(c.item, pair, cb);
            c.w_base = cb * CB * 9 * a.Cin_p * 2;       // also the packed image's block offset: n_chunks * W_BYTES per cout block
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int t = pair * 2 + h;
                int n, ty, tx;
                decode_tile(t < a.n_tiles ? t : 0, n, ty, tx);
                c.n[h] = t < a.n_tiles ? n : -1; c.y0[h] = ty * TH - 1; c.x0[h] = tx * TW - 1;
            }
        };
        auto cursor_next = [&](Cursor &c) {
            if (++c.ck == a.n_chunks) {
                c.ck = 0;
                c.item += gridDim.x;
                cursor_decode(c);
            }
        };
        auto issue_weights = [&](const Cursor &c, int slot) {  // exactly MAX_W instructions
            const int ubase = c.w_base + c.ck * (a.w_packed ? W_BYTES : CK * 2);
            char *dst = sWr + slot * W_BYTES;
#pragma unroll
            for (int k = 0; k < MAX_W; k++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(dst + (pw + N_PROD * k) * 1024), 16,
                                                         (unsigned)(w_off[k] + ubase), 0, 0, 0);
        };
        auto issue_patches = [&](const Cursor &c, int slot) {  // exactly 2 * my_p instructions
            const int c0 = c.ck * CK;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                char *dst = sPr + slot * P2_BYTES + h * P_BYTES;
                const int n = c.n[h], y0 = c.y0[h], x0 = c.x0[h];
#pragma unroll
                for (int k = 0; k < MAX_P; k++) {
                    const int j = pw + N_PROD * k;
                    if (j >= P_BLKS) continue;
                    const int pk = patch_pk(k);
                    const int py = pk & 255, iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
                    const bool in = n >= 0 && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + (pk >> 16)) * 2) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
                }
            }
        };

        Cursor cf;                                             // both streams fetch the same (item, chunk) sequence, one step ahead
        cf.item = bid; cf.ck = 0;
        cursor_decode(cf);
        issue_weights(cf, 0);
        issue_patches(cf, 0);
        cursor_next(cf);                                       // -> step 1
        int ck = 0, item = bid;                                // chunk / item of the step the consumers are in
        bool stores_young = false;                             // my youngest operations are the 2*MAX_S output stores (they may fly on)
        for (int s = 0; s < n_steps; s++) {
            if (stores_young) wait_vmcnt_n<2 * MAX_S>(); else wait_vmcnt_n<0>();    // W(s), P(s) have landed
            raw_barrier();                                     // T(s)
            stores_young = false;
            const bool more = s + 1 < n_steps;
            char *s0 = sWr + ((s + 1) & 1) * W_BYTES, *s1 = sPr + ((s + 1) & 1) * P2_BYTES;   // step s-1's slots = where chunk s+1 goes
            if (ck == 0 && s > 0) {                            // the consumers write out the previous pair first
                if (has_res) {                                 // (its residual tiles were loaded during the step before)
                    write_tiles(s0, s1);
                    raw_barrier();                             // R(s): the consumers pick their residual values up
                }
                raw_barrier();                                 // F(s): both tiles are staged
                read_tiles(s0, s1);
                if (more) { issue_weights(cf, (s + 1) & 1); issue_patches(cf, (s + 1) & 1); }
                stores_young = move_tiles(item - gridDim.x, false);
            } else if (more) {
                issue_weights(cf, (s + 1) & 1);
                issue_patches(cf, (s + 1) & 1);
            }
            if (has_res && ck == a.n_chunks - 1) {             // the item's last chunk: fetch its residual tiles (after the prefetches)
                move_tiles(item, true);
                stores_young = false;                          // the loads are the youngest entries and are needed right after T(s+1)
            }
            if (more) cursor_next(cf);
            if (++ck == a.n_chunks) { ck = 0; item += gridDim.x; }
        }
        raw_barrier();                                         // tail A: every consumer is done with the last slots
        char *s0 = sWr + ((n_steps - 1) & 1) * W_BYTES, *s1 = sPr + ((n_steps - 1) & 1) * P2_BYTES;
        if (has_res) {
            wait_vmcnt_n<0>();
            write_tiles(s0, s1);
            raw_barrier();                                     // tail R
        }
        raw_barrier();                                         // tail B: the last pair is staged
        read_tiles(s0, s1);
        move_tiles(item - gridDim.x, false);
        return;
    }

    // ========================================= CONSUMERS =========================================
    const int grp = wave >> 2, wg = wave & 3;               // cout group, pixel group
    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wg * MI) * PW + frow;
    EpiArgs ep{a.bias, a.slope, a.res, a.out, a.Cout_p, a.H, a.W, a.act, a.flags, 0, a.H, a.W, a.res_Cp};

    f32x4 acc[2][NI][MI];
    EpiRegs<NI, MI> R;
    int bias_cb = -1;                                        // cout block whose bias / slopes sit in R
    // bias + residual + activation of the pair `item` (sums in acc), tile by tile, and the fp16 results into the two staging areas
    auto flush_pair = [&](int item, char *s0, char *s1) {
        int pair, cb;
        decode_item(item, pair, cb);
        int lo = lane;                                       // opaque lane id: keeps this block's per-lane arithmetic out of the step loop
        asm volatile("" : "+v"(lo));
        const int frow = lo & 15, fq = lo >> 4;
        int co0[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++) co0[ni] = cb * CB + (grp * NI + ni) * 16 + fq * 4;
        if (cb != bias_cb) {                                 // bias row / PReLU slopes only change with the cout block
            bias_cb = cb;
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int c = co0[ni] < a.Cout_p ? co0[ni] : 0;
                R.bb[ni] = (a.bias != nullptr && !(a.flags & CF_BORDER)) ? *(const ep_f32x4 *)(a.bias + c) : ep_f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.act == ACT_PRELU) R.sl[ni] = *(const ep_f32x4 *)(a.slope + c);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            char *sS = (h == 0 ? s0 : s1) + wave * (64 * OROWB);
            EpiPix px[MI];
            if (a.flags & CF_BORDER) {                       // pixel coordinates: only the border-class bias needs them
                int n, ty, tx;
                decode_tile(min(pair * 2 + h, a.n_tiles - 1), n, ty, tx);
#pragma unroll
                for (int mi = 0; mi < MI; mi++) { px[mi].oy = ty * TH + wg * MI + mi; px[mi].ox = tx * TW + frow; }
            }
            if (has_res) {                                   // my residual elements sit where my outputs will go
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int ni = 0; ni < NI; ni++) {
                        const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                        R.rr[ni][mi] = *(const ep_half4 *)(sS + p * OROWB + ((c ^ (p & OMASK)) << 4) + (fq & 1) * 8);
                    }
            }
            ep_half4 hv[NI][MI];
            epilogue_values_fast<NI, MI>(ep, acc[h], px, co0, R, hv);
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                    *(ep_half4 *)(sS + p * OROWB + ((c ^ (p & OMASK)) << 4) + (fq & 1) * 8) = hv[ni][mi];
                }
        }
        lds_done();                                          // staged before the barrier that hands the areas to the producers
    };

    int li = 0, ck = 0;                                      // local item index / chunk of the current step
    for (int s = 0; s < n_steps; s++) {
        raw_barrier();                                       // T(s): the producers saw W(s), P(s) land; everyone is done with step s-1
        if (ck == 0 && s > 0) {
            if (has_res) raw_barrier();                      // R(s): the producers have put the residual tiles there
            flush_pair(bid + (li - 1) * gridDim.x, sWr + ((s + 1) & 1) * W_BYTES, sPr + ((s + 1) & 1) * P2_BYTES);
            raw_barrier();                                   // F(s): the producers write the pair out, then refill the slots
        }
        if (ck == 0) {
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) acc[h][ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char *sW = sWr + (s & 1) * W_BYTES, *sP = sPr + (s & 1) * P2_BYTES;
        // row-sharing tap order (conv_chunked.hip): column dx, then the 6 patch rows of this wave; a pixel fragment (row r, shift dx)
        // feeds every output row mi = r - dy of its tile; the column's three taps keep their weights in registers for BOTH tiles
        {
            int plin = lin0, wlane = ((grp * NI) * 16 + frow) * 64 + ((fq ^ swz64(frow)) << 4);
            asm volatile("" : "+v"(plin), "+v"(wlane));     // opaque: recompute the fragment addresses per step
            half8 wq[3][NI], pq[3][2];
            auto load_w = [&](int dy, int dx) {
#pragma unroll
                for (int ni = 0; ni < NI; ni++) wq[dy][ni] = *(const half8 *)(sW + wlane + ((dy * 3 + dx) * CB + ni * 16) * 64);
            };
            auto load_p = [&](int q, int set) {              // q = dx*6 + r
                const int lin = plin + (q % 6) * PW + q / 6;
                const int off = lin * 64 + ((fq ^ swz64(lin)) << 4);
                pq[set][0] = *(const half8 *)(sP + off);
                pq[set][1] = *(const half8 *)(sP + P_BYTES + off);
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
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int ni = 0; ni < NI; ni++)
                            acc[h][ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[dy][ni], pq[q % 3][h], acc[h][ni][mi], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dx < 2 && r >= 3) load_w(r - 3, dx + 1);
            }
        }
        if (++ck == a.n_chunks) { ck = 0; li++; }
    }
    raw_barrier();                                           // tail A: all consumers are done reading the last slots
    if (has_res) raw_barrier();                              // tail R
    flush_pair(bid + (li - 1) * gridDim.x, sWr + ((n_steps - 1) & 1) * W_BYTES, sPr + ((n_steps - 1) & 1) * P2_BYTES);
    raw_barrier();                                           // tail B
}

}  // namespace

bool conv_pc2_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_PC2")) return false;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin_p % 32 == 0 && a.Cin_p >= 64 && a.Cout_p >= 64 && a.Cout_p % 64 == 0 &&
           a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 12 && a.W >= 12 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo));
}

int conv_pc2_launch(fid_ctx *ctx, const ConvArgs &c) {
    PC2Args a{};
    a.in = c.in; a.w = c.w_alt ? c.w_alt : c.w; a.w_packed = c.w_alt != nullptr; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.act = c.act; a.flags = c.flags; a.res_Cp = c.res_Cp;
    const int B = c.M / (c.Ho * c.Wo);
    a.tiles_x = cdiv(c.W, TW);
    a.tiles_per_img = a.tiles_x * cdiv(c.H, TH);
    a.n_tiles = B * a.tiles_per_img;
    a.n_cblk = cdiv(c.Cout_p, CB);
    a.n_items = cdiv(a.n_tiles, 2) * a.n_cblk;
    a.n_chunks = c.Cin_p / CK;
    a.d_cblk = fastdiv_make(a.n_cblk); a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    a.in_bytes = c.in_bytes; a.w_bytes = c.w_bytes;
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB");
    FID_REQUIRE(c.res == nullptr || c.res_Cp == c.Cout_p, "producer/consumer conv: residual with %d channels for %d outputs", c.res_Cp, c.Cout_p);
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_pc2, (int)(LDS_BYTES)));
    const int grid = std::min(a.n_items, ctx->num_cus);
    hipLaunchKernelGGL(conv3x3_pc2, dim3(grid), dim3((N_CONS + N_PROD) * 64), LDS_BYTES, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
