// 3x3 / stride-1 convolution with the weights in REGISTERS and the waves split by output channel (autotuner generation 9).
//
// What bounds the producer/consumer halo-patch kernels (conv_pc*.hip) is what a CU can take in: activation patches arrive
// from beyond L2 at ~12 B/clk/CU and the weight chunks from L2 at ~29 B/clk/CU through the same path, and a conv3x3_pc2 step
// (2 tiles x 64 couts x 32 channels) needs 42 KB + 37 KB for 4608 matrix cycles -- the fill is as long as the arithmetic.
// Twice the couts per item halves the patch bytes per flop, but 2 x 74 KB of weight slots do not fit LDS beside the patches.
// Here the weights never touch LDS:
//
//   item   = a PAIR of tiles (TH x 16 output pixels each, consecutive in (image, tile row, tile column) order) x 128 couts
//   wave w = couts 16w .. 16w+15 of the block for ALL pixels of both tiles: its A fragments (9 taps x 16 couts x 32 channels
//            = 9 KB per step) are private, so each lane loads them straight from global memory into VGPRs (repack.hip kind 2:
//            fragment order, 1 KB contiguous per tap).  ONE register set: as soon as a tap column (dx) of the step is done its
//            three fragments are re-loaded with the NEXT step's column (inline-asm buffer loads the compiler does not track,
//            counted s_waitcnt by hand: every wait names only OLDER operations, so nothing in flight is drained early)
//   LDS    = only the two tiles' haloed patches (2 slots x 2 x ~21 KB), filled by LDS-DMA one step ahead, every wave issuing an
//            eighth of the pieces; every wave reads every pixel fragment (B operand) with the row-sharing tap order of
//            conv_chunked.hip: the fragment of patch row r shifted by dx feeds output rows r - dy for the three taps (dy, dx)
//   step   = 32 input channels: 42 KB of patches + 74 KB of weights for 9216 matrix cycles per SIMD (2 waves each)
//
// TH = 14 serves the 14 / 28 / 56 / 112-pixel maps of IResNet exactly (a 16-row tile wastes two of sixteen rows there and
// fetches a halo that is all padding); lanes 14, 15 of a fragment then compute pixels that are not stored.
// Epilogue: straight from the accumulator layout (a lane holds 4 consecutive couts of a pixel): bias (or the 9-class border
// bias), residual, activation, 8-byte buffer stores -- out-of-range rows / lanes go to an out-of-bounds offset, so the number
// of stores per item is exact and the next step's wait can leave them in flight.
#include <type_traits>

#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
#ifndef FID_STRIP_GW
#define FID_STRIP_GW 1
#endif
constexpr int WR_STRIP_GW = FID_STRIP_GW == 8 ? 8 : 1;           // patch positions between a STRIP tile's two images
constexpr int CK = 32;                                            // channels per step (patch width: 18 positions, 19 with STRIP tiles)

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WRArgs {
    const void *in;
    const void *w;        // repack.hip kind 2
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, Cin_p, Cout_p;
    int act, flags;
    int tiles_x, tiles_per_img, n_tiles, n_cblk, n_items, n_chunks;
    FastDiv d_cblk, d_tpi, d_tx;
    unsigned in_bytes, out_bytes, w_bytes;
    int tab_off, ncls;   // LDS offset of the bias [ncls][Cout_p] + slope [Cout_p] fp32 tables; ncls = 9 with CF_BORDER else 1 (0: no bias)
    int n_img, tiles_y;  // STRIP: images of the strip, tile rows; tile t = (strip tile t / tiles_y, tile row t % tiles_y), d_tx divides by tiles_y, d_tpi by W
    int rev;      // walk the items from the last to the first (see ConvArgs::rev)
    int stagger;  // experiment: workgroups in the second half of the grid (the co-resident ones) start this many x 64 cycles late
    int ablate;   // FID_WR_ABLATE timing experiments (wrong results): 1 no step barrier, 2 no patch pieces, 4 no weight reloads, 8 no epilogue, 32 stores dropped, 64 no residual loads
};

// TH: tile rows (14 | 16); NT: tiles per item; NW: waves = 16-cout fragments per item;
// NCH = 0: the weights stream (one register set, re-loaded column by column one step ahead);
// NCH = 2 | 3: layers with NCH*32 input channels and at most NW*16 couts -- ALL weights of the layer (NCH x 9 fragments per wave) stay
//              in registers for the kernel's lifetime: only patches are fetched (SCRFD's 64 / 96-channel stacks at 160x160 / 80x80)
// NS: patch slots -- pieces are requested NS - 1 steps ahead (4: the one-tile variant on small batches, where a step is ~1 us, shorter
//     than a trip to memory, and there is no second workgroup on the CU to hide it)
// STRIP (round 5): x-packed pixel fragments -- a tile is TH rows x 16 consecutive columns of the strip formed by the rows of all images side by
//     side (conv_ks.hip describes the addressing): no idle lanes on 14 / 28 / 56 / 112 / 20 / 40-wide maps
// XF: bit 0 = STRIP, bit 1 = ONE_ROW (the map is one tile high, H = TH: patch rows 0 and TH + 1 lie outside the image -- zeros, neither read nor multiplied:
//     120 of 126 products per tile and step on IResNet's 14x14 stage, as conv3x3_ks<14, 2> does)
template <int TH, int NT, int NW, int NCH, int NS = 2, int XF = 0>
__global__ void __launch_bounds__(NW * 64, 2) conv3x3_wr(const WRArgs a) {
    constexpr bool STRIP = (XF & 1) != 0, ONE_ROW = (XF & 2) != 0;
    static_assert(!ONE_ROW || NCH == 0, "ONE_ROW: streaming variants only (the deferred write-out is scheduled on the full row list)");
    constexpr int CBW = NW * 16;                                // couts per item
    // STRIP: GW = 8 patch positions between the tile's two images: the lanes behind the boundary read 512 bytes further right -- the same bank row
    // quarter and XOR swizzle as without the gap, so the fragment reads stay conflict-free (one shared zero column, GW = 1, put five of a fragment's
    // sixteen pixels on one bank row: 33-50 % of the LDS cycles were conflicts, profiles/r05); maps one tile high store only their TH real rows
    // MEASURED equal-to-slower (conv_ks.hip, profiles/r05/ab_gutter8.txt): the default is the one shared zero column, GW = 1; -DFID_STRIP_GW=8 builds the gap
    constexpr int GW = STRIP ? WR_STRIP_GW : 0;
    constexpr int PW = 18 + GW;                                 // patch positions per row
    constexpr bool COMPACT = STRIP && ONE_ROW && GW == 8;
    constexpr int R0 = COMPACT ? 1 : 0;
    constexpr int TW = STRIP ? 16 : TH;                         // tile stride in x (16 lanes per fragment; lanes >= TW are not stored)
    constexpr int PH = TH + 2, NPIX = (COMPACT ? TH : PH) * PW;
    constexpr int P_BLKS = (NPIX * 64 + 1023) / 1024, P_BYTES = P_BLKS * 1024, SLOT = NT * P_BYTES;
    constexpr int MAX_P = (NT * P_BLKS + NW - 1) / NW;          // patch pieces per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);        // consecutive items (the cout blocks of a pair) in one L2
    const int my_items = bid < a.n_items ? (a.n_items - 1 - bid) / gridDim.x + 1 : 0;
    const int n_steps = my_items * a.n_chunks;
    if (n_steps == 0) return;
    const int frow = lane & 15, fq = lane >> 4;

    auto decode_item = [&](int item, int &pair, int &cb) {
        if (a.rev) item = a.n_items - 1 - item;
        pair = fastdiv(item, a.d_cblk);
        cb = item - pair * a.n_cblk;
    };
    auto decode_tile = [&](int t, int &n, int &ty, int &tx) {
        if (STRIP) {                                            // n = image of lane 0, tx = its column there (lane l: strip column 16 ft + l)
            const int ft = fastdiv(t, a.d_tx);
            ty = t - ft * a.tiles_y;
            n = fastdiv(ft * 16, a.d_tpi); tx = ft * 16 - n * a.W;
            return;
        }
        n = fastdiv(t, a.d_tpi);
        const int r = t - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    // STRIP: lane / tile column l of a tile whose lane 0 is column c0 of image n -> its image and column
    auto strip_px = [&](int n, int c0, int l, int &img, int &x) {
        const bool sec = l >= a.W - c0;
        img = n + (sec ? 1 : 0); x = c0 + l - (sec ? a.W : 0);
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.out), 0, a.out_bytes, 0x00020000);

    // ---- my patch pieces: piece j = wave + 8k covers patch pixels 16*(j % P_BLKS) .. +15 of tile j / P_BLKS, 4 lanes per pixel
    int p_pk[MAX_P];                                            // py | px << 8 | channel offset << 16 | tile << 24; py = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P; k++) {
        const int j = wave + NW * k;
        const int h = j / P_BLKS, blk = j - h * P_BLKS;
        const int row = blk * 16 + (lane >> 2);
        int py = row / PW;
        const int px = row - py * PW;
        if (row >= NPIX || px >= (STRIP ? PW : TW + 2) || j >= NT * P_BLKS) py = 255;
        p_pk[k] = py | (px << 8) | ((((lane & 3) ^ swz64(row)) * 8) << 16) | (h << 24);
    }
    struct Cursor {
        int item, ck, cb;
        int n[NT], y0[NT], x0[NT];       // image and top-left input pixel of each tile's haloed patch (n < 0: no tile)
    };
    auto cursor_decode = [&](Cursor &c) {
        int pair;
        decode_item(c.item, pair, c.cb);
#pragma unroll
        for (int h = 0; h < NT; h++) {
            const int t = pair * NT + h;
            int n, ty, tx;
            decode_tile(t < a.n_tiles ? t : 0, n, ty, tx);
            c.n[h] = t < a.n_tiles ? n : -1; c.y0[h] = ty * TH - 1; c.x0[h] = (STRIP ? tx : tx * TW) - 1;
        }
    };
    auto cursor_next = [&](Cursor &c) {
        if (++c.ck == a.n_chunks) {
            c.ck = 0;
            c.item += gridDim.x;
            cursor_decode(c);
        }
    };
    auto issue_patches = [&](const Cursor &c, int slot) {      // exactly MAX_P instructions
        const int c0 = c.ck * CK;
        char *dst = smem + slot * SLOT;
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + NW * k < NT * P_BLKS ? wave + NW * k : (NS - slot) * NT * P_BLKS;   // surplus piece: zeros into the spare KB behind the slots
            int pk = p_pk[k];
            asm volatile("" : "+v"(pk));                        // opaque: unpack at the use
            const int h = pk >> 24, py = pk & 255;
            const int n = h ? c.n[NT - 1] : c.n[0], y0 = h ? c.y0[NT - 1] : c.y0[0], x0 = h ? c.x0[NT - 1] : c.x0[0];
            const int iy = y0 + py + R0;
            int ix = x0 + ((pk >> 8) & 255), img = n;
            bool in = n >= 0 && py != 255 && (unsigned)iy < (unsigned)a.H;
            if (STRIP) {                                        // position p holds image A's column x0 + p up to its right padding (column W), behind it image B from column 0
                const bool second = ix > a.W;
                img += second ? 1 : 0; ix = second ? ix - a.W - GW : ix;     // (positions W + 1 .. W + GW - 2 are never read; W + GW - 1 is image B's left padding)
                in = in && img < a.n_img;
            }
            in = in && (unsigned)ix < (unsigned)a.W;
            const unsigned vo = in ? (unsigned)((((img * a.H + iy) * a.W + ix) * a.Cin_p + c0 + ((pk >> 16) & 255)) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
        }
    };
    // ---- weights: buffer loads with a per-lane offset (wave's fragment block + lane*16), the (cout block, chunk, column) base in
    // an SGPR and the tap inside the column as the immediate; repack kind 2 stores the taps of a column contiguously ([dx][dy])
    const unsigned long long wp = (unsigned long long)a.w;      // buffer descriptor by hand (an inline-asm "s" operand): base, stride 0, bytes, flags
    const i32x4 rs_w = i32x4{(int)(unsigned)wp, (int)((unsigned)(wp >> 32) & 0xFFFFu), (int)a.w_bytes, 0x00020000};
    const int w_voff = wave * 9216 + lane * 16;
    half8 w[NCH > 0 ? NCH * 9 : 9];                             // w[chunk*9 + dy*3 + dx] (streaming: chunk = 0)
    auto load_col = [&](int cb, int ck, int dx, half8 &t0, half8 &t1, half8 &t2) {   // exactly 3 instructions
        const int gf = cb * NW;                                 // first 16-cout fragment of the block; repack kind 2 groups 8 fragments per chunk
        const int soff = __builtin_amdgcn_readfirstlane(((gf >> 3) * a.n_chunks + ck) * (8 * 9216) + (gf & 7) * 9216 + dx * 3072);
        asm volatile("buffer_load_dwordx4 %0, %3, %4, %5 offen\n\t"
                     "buffer_load_dwordx4 %1, %3, %4, %5 offen offset:1024\n\t"
                     "buffer_load_dwordx4 %2, %3, %4, %5 offen offset:2048"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2)
                     : "v"(w_voff), "s"(rs_w), "s"(soff)
                     : "memory");
    };

    // ---- pixel fragment addresses: lin = q' + frow with q' a compile-time constant; the swizzled 16-byte group is one of four
    // per lane, selected by the parity of q' and (q' >> 1) & 3 -- precomputed, so a fragment read is one ds_read with an immediate
    // STRIP: lanes behind the tile's image boundary read one position further right (the shared zero column lies between the images): per tile of the item
    // (the 16-row pair variant has no registers for a second set: the host gives it pairs of tiles with one boundary position -- an even number of
    // tile rows, so that a pair is two rows of one strip tile, or a width that is a multiple of 16)
    constexpr int NPB = (STRIP && !(TH == 16 && NT == 2)) ? NT : 1;
    int pbase[NPB][2][4];
    auto set_pbase = [&](int t, int fs) {
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) pbase[t][par][c] = fs * 64 + ((fq ^ ((((fs + par) >> 1) + c) & 3)) << 4);
    };
#pragma unroll
    for (int t = 0; t < NPB; t++) set_pbase(t, frow);
    auto strip_item = [&](int item_) {                          // STRIP: the lane shifts of the item about to be multiplied
        int pair, cb;
        decode_item(item_, pair, cb);
        int fr = frow;
        asm volatile("" : "+v"(fr));
#pragma unroll
        for (int t = 0; t < NPB; t++) {
            const int tile = pair * NT + t;
            int n, ty, c0;
            decode_tile(tile < a.n_tiles ? tile : 0, n, ty, c0);
            set_pbase(t, fr + (fr >= a.W - c0 ? GW : 0));
        }
    };

    f32x4 acc[NT][TH];

    // one tap column of a step: PH pixel-fragment rows x (up to) 3 taps x NT tiles
#ifndef WR_PD_NT2
#define WR_PD_NT2 2
#endif
    constexpr int PD = NT == 1 ? (NCH == 3 ? 2 : 4) : WR_PD_NT2;                         // pixel fragments read ahead (a fragment feeds <= 3*NT MFMAs = 48*NT cycles; an LDS read takes > 100)
    auto compute_col = [&](const char *sP, int dx_, auto wb_tag, auto &&row_hook) {
        constexpr int WB = decltype(wb_tag)::value * 9;
        // the eight per-lane fragment bases of this slot, made opaque: every read is then "base register + immediate" -- left to
        // itself the compiler hoists one address register PER READ out of the step loop (54 VGPRs of loop-invariant addresses)
        int pb[NPB][2][4];
        const int slot_off = (int)(sP - smem);
#pragma unroll
        for (int t = 0; t < NPB; t++)
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[t][par][c] = pbase[t][par][c] + slot_off;
#ifndef WR_NO_PB
                asm volatile("" : "+v"(pb[t][par][c]));
#endif
            }
        half8 pq[PD + 1][NT];
        auto load_p = [&](int q, int set) {                     // q = dx * PH + patch row
            const int K = (q % PH - R0) * PW + q / PH;          // lin = K + frow
#pragma unroll
            for (int t = 0; t < NT; t++)
                pq[set][t] = *(const half8 *)(smem + (pb[NPB > 1 ? t : 0][K & 1][(K >> 1) & 3] + (K * 64 + t * P_BYTES)));
        };
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            if (dx != dx_) continue;
            constexpr int NR = ONE_ROW ? TH : PH, RF = ONE_ROW ? 1 : 0;     // patch rows read: RF .. RF + NR - 1
#pragma unroll
            for (int j = 0; j < PD; j++) load_p(dx * PH + RF + j, j % (PD + 1));
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int r = RF + j;
                const int q = dx * PH + r;
                if (j + PD < NR) load_p(q + PD, (j + PD) % (PD + 1));
                row_hook(dx, r);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int mi = r - dy;
                    if (mi < 0 || mi >= TH) continue;
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[WB + dy * 3 + dx], pq[j % (PD + 1)][t], acc[t][mi], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // bias + residual + activation of the pair `item` (sums in acc) in the accumulator layout, then the fp16 results through LDS so
    // that they leave as 16 bytes per lane / whole 256-byte cout rows per pixel (8-byte stores from the accumulator layout touch 16
    // lines per instruction: measured 12k cycles per item, every matrix pipe idle meanwhile).  Staging area = the patch slot the
    // item's last step has just finished with; four passes of half a tile (HR rows x 16 pixels x 128 couts).
    // Exactly EPI_ST stores (+ NT*TH residual loads before them) per wave.
    constexpr int ROWB = CBW * 2, CPX = NW * 2;                 // bytes / 16-byte chunks of a staged pixel row
    // rows per pass: what the free patch slot holds -- or, resident variant, the whole tile at once in the residual / staging region
    constexpr int HR = NCH > 0 ? TH : ((TH / 2) < (SLOT / (16 * ROWB)) ? (TH / 2) : (SLOT / (16 * ROWB)));
    constexpr int NPASS = (TH + HR - 1) / HR;
#ifndef WR_DIRECT
#define WR_DIRECT 0
#endif
#ifndef WR_DEFER
#define WR_DEFER 1
#endif
    // resident variants: an item's finished tile is only STAGED (fp16, pixel rows) at the end of its last step; the 16-byte row stores
    // are issued one by one between the matrix rows of the next step, so the write burst of all CUs no longer sits between two steps
    // with every matrix pipe idle (measured on SCRFD layer1: 127 us with the epilogue, 99 with its stores dropped, 85 without it)
    constexpr bool DEFER = WR_DEFER && NCH == 2 && TH >= 14;    // (the store schedule below needs 15 matrix rows per column)
    constexpr bool DIRECT = WR_DIRECT && NCH == 0;               // streaming variants: 8-byte stores straight from the accumulator layout, no staging, no barriers
    constexpr int ST_I = (HR * 16 * CPX + NW * 64 - 1) / (NW * 64), EPI_ST = DIRECT ? NT * TH : NPASS * NT * ST_I;   // write-out instructions per wave and pass / item
    constexpr int RG = NCH > 0 ? 2 : (((TH == 16 && NT == 2) && HR > 4) ? 4 : HR);   // residual rows in registers at a time (resident variant: they come from LDS, just in time)
    constexpr int EPI_RL = NT * NPASS * ((HR + RG - 1) / RG) * RG;                   // residual loads per item (rows past a pass / the tile: out of bounds)
    static_assert(HR >= 1 && (NCH > 0 || HR * 16 * ROWB <= SLOT), "staging area");
    // resident variant: one cout block, so the bias row / PReLU slopes of a lane never change -- fetched once, not per tile; and the
    // residual tile comes in by LDS-DMA with the item's LAST patch prefetch (a whole step ahead) in the staged layout, so the epilogue
    // reads it from LDS: a tile's epilogue then has no global round trip on its critical path (measured before: 64-100 us of a 146-192 us
    // layer were the per-tile bias / residual load latencies, with only two workgroups per CU to hide them)
    constexpr int RS_BLKS = (TH * 16 * ROWB + 1023) / 1024, RP = (RS_BLKS + NW - 1) / NW;
    char *sR = smem + NS * SLOT + 1024;
    f32x4 k_bias = f32x4{0.f, 0.f, 0.f, 0.f}, k_sl = f32x4{1.f, 1.f, 1.f, 1.f};
    if constexpr (NCH > 0) {
        const int c0 = wave * 16 + fq * 4, cc = c0 < a.Cout_p ? c0 : 0;
        if (a.bias && !(a.flags & CF_BORDER)) k_bias = *(const f32x4 *)(a.bias + cc);
        if (a.act == ACT_PRELU) k_sl = *(const f32x4 *)(a.slope + cc);
    }
    // bias rows / PReLU slopes in LDS for the kernel's lifetime: an item's epilogue reads them with ds_read instead of global loads whose
    // latency (one to three L2 round trips per item) sat on its critical path
    // Called AFTER the first patch pieces and weight fragments have been requested: the three cold fetches of a launch (tables, weights,
    // first patch: ~1-2 us each from HBM / the Infinity Cache) then share one wait instead of running back to back.
    const float *sTab = (const float *)(smem + a.tab_off);
    auto fill_tables = [&]() {
        float *tb = (float *)(smem + a.tab_off);
        const int nb = a.ncls * a.Cout_p;
        for (int i = tid; i < nb; i += NW * 64) tb[i] = a.bias[i];
        for (int i = tid; i < a.Cout_p; i += NW * 64) tb[nb + i] = a.act == ACT_PRELU ? a.slope[i] : 1.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        raw_barrier();
    };
    auto issue_residual = [&](int item_) {                      // exactly RP instructions (resident variant, layers with a residual)
        int pair, cb;
        decode_item(item_, pair, cb);
        int n, ty, tx;
        decode_tile(pair < a.n_tiles ? pair : 0, n, ty, tx);
        int lo = lane;
        asm volatile("" : "+v"(lo));                            // opaque: the per-piece constants are recomputed per call, not kept in registers
#pragma unroll
        for (int k = 0; k < RP; k++) {
            const int j = wave + NW * k;
            const int off = j * 1024 + lo * 16;
            const int pl = off / ROWB, pos = (off - pl * ROWB) >> 4;
            const int pr = pl >> 4, pc = pl & 15;
            const int c = (pos + CPX - pc % CPX) % CPX;         // the staged layout rotates a pixel's chunks by its column
            int nn = n, ox = tx * TW + pc;
            if (STRIP) strip_px(n, tx, pc, nn, ox);
            const int oy = ty * TH + pr;
            const bool ok = j < RS_BLKS && pl < TH * 16 && pair < a.n_tiles && pc < TW && oy < a.H && ox < a.W && c * 8 < a.Cout_p && (!STRIP || nn < a.n_img);
            const unsigned vo = ok ? (unsigned)((((nn * a.H + oy) * a.W + ox) * a.Cout_p + c * 8) * 2) : OOB;
            char *dst = j < RS_BLKS ? sR + j * 1024 : smem + NS * SLOT;     // surplus piece: the spare KB
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (__attribute__((address_space(3))) void *)dst, 16, vo, 0, 0, 0);
        }
    };
    auto epilogue_body = [&](int item, char *stage, auto act_tag, auto res_tag, auto border_tag) {
        constexpr bool STAGE_ONLY = DEFER;                       // resident variants: the write-out happens inside the NEXT step's matrix columns
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool RES = decltype(res_tag)::value, BORDER = decltype(border_tag)::value;
        int pair, cb;
        decode_item(item, pair, cb);
        int lo = lane;
        asm volatile("" : "+v"(lo));                            // opaque lane id: keeps this block's per-lane arithmetic out of the step loop
        const int fr = lo & 15, q4 = lo >> 4;
        const int co0 = cb * CBW + wave * 16 + q4 * 4;
        const bool co_ok = co0 < a.Cout_p;
        const int cc = co_ok ? co0 : 0;
        f32x4 bmid = f32x4{0.f, 0.f, 0.f, 0.f}, sl = f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (NCH > 0) {
            bmid = k_bias; sl = k_sl;
        } else {
            if (a.ncls && !BORDER) bmid = *(const f32x4 *)(sTab + cc);
            if (ACT == ACT_PRELU) sl = *(const f32x4 *)(sTab + a.ncls * a.Cout_p + cc);
        }
        const unsigned rstride = (unsigned)(a.W * a.Cout_p * 2);
        // my 8 bytes of a staged pixel row (256 B = 16 chunks of 8 couts): chunk wave*2 + q4/2, XOR-swizzled by the pixel column
        const int st_w = fr * ROWB + (((wave * 2 + (q4 >> 1) + fr) % CPX) << 4) + (q4 & 1) * 8;    // chunk rotated by the pixel column: no bank pile-up
        if (NCH > 0 && RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of the residual tile (issued a step ago) are in LDS
        if (!DIRECT && (NCH == 0 || RES)) raw_barrier();        // every wave is done reading the slot / everybody's residual pieces landed
        // residual groups (streaming variant: 8-byte loads in the accumulator layout), software-pipelined: group g + 1 is requested before
        // group g is consumed, so only the item's first group waits for memory
        constexpr int GPP = (HR + RG - 1) / RG, NGRP = NT * NPASS * GPP;
        unsigned t_base[NT];
        bool t_ok[NT];
        int t_oy0[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int tile = pair * NT + t;
            int n, ty, tx;
            decode_tile(tile < a.n_tiles ? tile : 0, n, ty, tx);
            t_oy0[t] = ty * TH;
            int nn = n, oxl = tx * TW + fr;
            if (STRIP) strip_px(n, tx, fr, nn, oxl);
            t_ok[t] = tile < a.n_tiles && co_ok && fr < TW && oxl < a.W && (!STRIP || nn < a.n_img);
            t_base[t] = (unsigned)((((nn * a.H + ty * TH) * a.W + oxl) * a.Cout_p + co0) * 2);
        }
        u32x2 rrA[RG], rrB[RG];
        auto load_group = [&](int gi, u32x2 (&rr)[RG]) {        // gi -> (tile, pass, group in pass); exactly RG loads
            const int t = gi / (NPASS * GPP), rem = gi - t * (NPASS * GPP), r0 = (rem / GPP) * HR, r1 = r0 + (rem % GPP) * RG;
#pragma unroll
            for (int r = 0; r < RG; r++)
                rr[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, (t_ok[t] && r1 + r < r0 + HR && r1 + r < TH && t_oy0[t] + r1 + r < a.H && !(a.ablate & 64)) ? t_base[t] + (r1 + r) * rstride : OOB, 0, 0);
        };
        if (RES && NCH == 0) load_group(0, rrA);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int tile = pair * NT + t;
            int n, ty, tx;
            decode_tile(tile < a.n_tiles ? tile : 0, n, ty, tx);
            const int oy0 = ty * TH, ox0 = tx * TW;
            int ox = ox0 + fr, n_l = n;
            if (STRIP) strip_px(n, tx, fr, n_l, ox);
            f32x4 btop = bmid, bbot = bmid;
            if (BORDER) {       // exact fold of a BatchNorm in front of the zero-padded conv: the bias row depends on the pixel's border class
                const int xc = ox == 0 ? 0 : (ox == a.W - 1 ? 2 : 1);
                btop = *(const f32x4 *)(sTab + (0 + xc) * a.Cout_p + cc);
                bmid = *(const f32x4 *)(sTab + (3 + xc) * a.Cout_p + cc);
                bbot = *(const f32x4 *)(sTab + (6 + xc) * a.Cout_p + cc);
            }
#pragma unroll
            for (int r0 = 0; r0 < TH; r0 += HR) {
#pragma unroll
                for (int r1 = r0; r1 < r0 + HR && r1 < TH; r1 += RG) {
                    constexpr int dummy = 0; (void)dummy;
                    const int gi = (t * NPASS + r0 / HR) * GPP + (r1 - r0) / RG;     // (compile-time: the nest is fully unrolled)
                    u32x2 rr[RG];
                    if (RES && NCH > 0) {                        // resident variant: the residual tile is in LDS, staged layout
#pragma unroll
                        for (int r = 0; r < RG; r++)
                            if (r1 + r < r0 + HR && r1 + r < TH) rr[r] = *(const u32x2 *)(sR + (r1 + r) * (16 * ROWB) + st_w);
                    } else if (RES) {
                        if (gi + 1 < NGRP) { if (gi & 1) load_group(gi + 1, rrA); else load_group(gi + 1, rrB); }
#pragma unroll
                        for (int r = 0; r < RG; r++) rr[r] = (gi & 1) ? rrB[r] : rrA[r];
                    }
                    if (!DIRECT && r1 == r0 && t + r0 > 0) raw_barrier();  // the pass before has been read back
#pragma unroll
                    for (int r = r1; r < r1 + RG && r < r0 + HR && r < TH; r++) {
                        const int oy = oy0 + r;
                        f32x4 v;
                        if (!BORDER) v = acc[t][r] + bmid;
                        else v = acc[t][r] + (oy == 0 ? btop : (oy == a.H - 1 ? bbot : bmid));   // (a scalar compare + 4 selects per row)
                        if (RES) {
                            const half4 h = __builtin_bit_cast(half4, rr[r - r1]);
                            v += __builtin_convertvector(h, f32x4);
                        }
                        if (ACT == ACT_PRELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + sl * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
                        half4 h = __builtin_convertvector(v, half4);
                        if (ACT == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                        if (DIRECT) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rs_out, (t_ok[t] && oy < a.H && !(a.ablate & 32)) ? t_base[t] + r * rstride : OOB, 0, 0);
                        else *(half4 *)(stage + (r - r0) * (16 * ROWB) + st_w) = h;
                    }
                }
                if (DIRECT) continue;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (STAGE_ONLY) continue;                       // (the next step's barrier publishes the staged tile)
                raw_barrier();                                  // the half tile is staged
                // write-out: 16-byte slot g = (i*NW + wave)*64 + lane of the pass = pixel g / CPX, chunk g % CPX.  NW*64 / CPX = 32 for every
                // NW, so pixel = 32 i + q0 and chunk = c with q0, c per-lane constants: both addresses are affine in i
                int lo2 = lo;
                asm volatile("" : "+v"(lo2));                   // opaque again: the constants are recomputed per pass, not kept in registers
                const int wl = wave * 64 + lo2, q0 = wl / CPX, c = wl - q0 * CPX;
                const int pr0 = q0 >> 4, pc = q0 & 15;          // row / column of the lane's first pixel inside the pass
                int oxx = ox0 + pc, n2 = n;
                if (STRIP) strip_px(n, tx, pc, n2, oxx);
                const int co = cb * CBW + c * 8;
                const bool okc = tile < a.n_tiles && pc < TW && oxx < a.W && co < a.Cout_p && (!STRIP || n2 < a.n_img);
                const char *lsrc = stage + q0 * ROWB + (((c + pc) % CPX) << 4);
                const unsigned g0 = (unsigned)((((n2 * a.H + oy0 + r0 + pr0) * a.W + oxx) * a.Cout_p + co) * 2);
                const int rows_left = (a.H - oy0 < TH ? a.H - oy0 : TH) - r0 - pr0;     // rows of the pass this lane may store: 2 i < rows_left
#pragma unroll
                for (int i = 0; i < ST_I; i++) {
                    const bool in_pass = 2 * i + pr0 < HR;      // pixel 32 i + q0 < HR*16
                    const u32x4 v = *(const u32x4 *)(lsrc + (in_pass ? i * 32 * ROWB : 0));
                    const bool ok = okc && in_pass && 2 * i < rows_left;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (ok && !(a.ablate & 32)) ? g0 + (unsigned)(2 * i) * rstride : OOB, 0, 0);
                }
            }
        }
    };
    // one straight-line body per (activation, residual, border bias) -- chosen once per item (deciding per value on runtime flags
    // costs scalar branches and selects between every few vector instructions: the epilogue was 15k cycles per item that way)
    auto epilogue = [&](int item, char *stage) {
        using std::integral_constant;
        const bool res = a.res != nullptr, border = (a.flags & CF_BORDER) != 0;
#define WR_EPI(A) \
        do { \
            if (res) { if (border) epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, true>{}, integral_constant<bool, true>{}); \
                       else epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, true>{}, integral_constant<bool, false>{}); } \
            else { if (border) epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{}); \
                   else epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{}); } \
        } while (0)
        if (a.act == ACT_PRELU) WR_EPI(ACT_PRELU);
        else if (a.act == ACT_RELU) WR_EPI(ACT_RELU);
        else WR_EPI(ACT_NONE);
#undef WR_EPI
    };

    if (a.stagger > 0 && (int)blockIdx.x >= (int)gridDim.x / 2) {
        for (int i = 0; i < a.stagger; i += 100) __builtin_amdgcn_s_sleep(100);
    }
    if constexpr (NCH > 0) {
        // ================= resident weights: a.n_chunks == NCH, one cout block =================
        static_assert(NT == 1 && NS == 2, "resident variant: one tile per item, two patch slots");
        Cursor cf;
        cf.item = bid; cf.ck = 0;
        cursor_decode(cf);
        issue_patches(cf, 0);                                   // first patch, all weights, tables: one trip to memory
        cursor_next(cf);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            load_col(0, c, 0, w[c * 9 + 0], w[c * 9 + 3], w[c * 9 + 6]);
            load_col(0, c, 1, w[c * 9 + 1], w[c * 9 + 4], w[c * 9 + 7]);
            load_col(0, c, 2, w[c * 9 + 2], w[c * 9 + 5], w[c * 9 + 8]);
        }
        fill_tables();                                          // (ends with vmcnt(0) + barrier)
#pragma unroll
        for (int i = 0; i < NCH * 9; i++) asm volatile("" : "+v"(w[i]));
        constexpr int E1 = EPI_ST;
        int e_prev = 0, item = bid, s = 0;
        const bool has_res = a.res != nullptr;
        auto no_hook = [](int, int) {};
        // deferred write-out of the tile staged in sR (item `pend`): per-lane constants as in epilogue_body's write-out loop
        int pend = -1;
        const char *wo_src = sR;
        unsigned wo_g0 = OOB;
        int wo_rows = 0, wo_pr0 = 0;
        const unsigned wo_rstride = (unsigned)(a.W * a.Cout_p * 2);
        auto wo_setup = [&]() {
            int lo2 = lane;
            asm volatile("" : "+v"(lo2));
            int n, ty, tx;
            const int ptile = (pend >= 0 && a.rev) ? a.n_items - 1 - pend : pend;     // (resident variant: item = tile)
            decode_tile(ptile >= 0 && ptile < a.n_tiles ? ptile : 0, n, ty, tx);
            const int wl = wave * 64 + lo2, q0 = wl / CPX, c = wl - q0 * CPX;
            const int pr0 = q0 >> 4, pc = q0 & 15;
            int oxx = tx * TW + pc, n2 = n;
            if (STRIP) strip_px(n, tx, pc, n2, oxx);
            const int oy0 = ty * TH, co = c * 8;
            const bool okc = ptile >= 0 && ptile < a.n_tiles && pc < TW && oxx < a.W && co < a.Cout_p && !(a.ablate & 32) && (!STRIP || n2 < a.n_img);
            wo_src = sR + q0 * ROWB + (((c + pc) % CPX) << 4);
            wo_g0 = (unsigned)((((n2 * a.H + oy0 + pr0) * a.W + oxx) * a.Cout_p + co) * 2);
            wo_rows = okc ? (a.H - oy0 < TH ? a.H - oy0 : TH) - pr0 : 0;      // 2 i < wo_rows: the lane may store row 2 i + pr0 of the tile
            wo_pr0 = pr0;
        };
        u32x4 wo_v = u32x4{0u, 0u, 0u, 0u};
        auto wo_one_read = [&](int i) {
            const bool in_pass = 2 * i + wo_pr0 < HR;
            wo_v = *(const u32x4 *)(wo_src + (in_pass ? i * 32 * ROWB : 0));
        };
        auto wo_one_store = [&](int i) {
            const bool ok = 2 * i + wo_pr0 < HR && 2 * i < wo_rows;
            __builtin_amdgcn_raw_buffer_store_b128(wo_v, rs_out, ok ? wo_g0 + (unsigned)(2 * i) * wo_rstride : OOB, 0, 0);
        };
        // write-out i = 3 dx + j of a step: LDS read in front of matrix row 1 + 5 j of column dx, store in front of row 4 + 5 j
        // (exactly ST_I stores per wave in every first step of an item, whether or not a tile is pending: the counts stay exact)
        auto wo_hook = [&](int dx, int r) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int i = 3 * dx + j;
                if (i >= ST_I) continue;
                if (r == 1 + 5 * j) wo_one_read(i);
                if (r == 4 + 5 * j) wo_one_store(i);
            }
        };
        static_assert(!DEFER || (ST_I <= 9 && PH >= 15), "deferred write-out schedule");
        auto step = [&](auto c_tag) {
            constexpr int C = decltype(c_tag)::value;
            if constexpr (DEFER) {
                // younger than my pieces of step s (requested at the top of step s - 1): the ST_I stores of a first step -- nothing else
                if (C == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST_I) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (e_prev == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // my pieces of step s have landed
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E1 > 63 ? 63 : E1) : "memory");            // (an item's stores may fly on)
            }
            raw_barrier();
            if (s + 1 < n_steps) { issue_patches(cf, (s + 1) & 1); cursor_next(cf); }
            if (C == NCH - 1 && has_res) issue_residual(item);  // lands during this step; read in the epilogue behind a vmcnt(0)
            if (C == 0) {
#pragma unroll
                for (int r = 0; r < TH; r++) acc[0][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (STRIP) strip_item(item);
            }
            const char *sP = smem + (s & 1) * SLOT;
            if constexpr (DEFER && C == 0) {
                wo_setup();
                compute_col(sP, 0, c_tag, wo_hook);
                compute_col(sP, 1, c_tag, wo_hook);
                compute_col(sP, 2, c_tag, wo_hook);
            } else {
                compute_col(sP, 0, c_tag, no_hook);
                compute_col(sP, 1, c_tag, no_hook);
                compute_col(sP, 2, c_tag, no_hook);
            }
            e_prev = 0;
            if (C == NCH - 1) {
                if (!(a.ablate & 8)) epilogue(item, sR);    // outputs are staged in place of the residual tile
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                e_prev = 1;
                pend = (a.ablate & 8) ? -1 : item;
                item += gridDim.x;
            }
            s++;
        };
        for (int it = 0; it < my_items; it++) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            if constexpr (NCH > 2) step(std::integral_constant<int, 2>{});
        }
        if constexpr (DEFER) {                                  // the last item's tile
            raw_barrier();
            wo_setup();
#pragma unroll
            for (int i = 0; i < ST_I; i++) { wo_one_read(i); wo_one_store(i); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // ---- operation order per wave and step s (vmcnt retires in order, so every wait below names only OLDER operations):
    //   top:  MAX_P patch pieces of step s+D (D = NS - 1 slots ahead)
    //   after column 0 / 1: 3 weight loads = that column of step s+1
    //   after column 2: [the item's last chunk: E = EPI_ST stores (+ NT*TH residual loads) of the epilogue], then column 2 of step s+1
    // wait at the top for the pieces of step s:            younger = D x 9 weight loads + (D - 1) x MAX_P pieces (+ E of an epilogue in between)
    // wait before column dx for its 3 fragments (step s):  younger = 6 weight loads + MAX_P pieces (+ E for columns 0 and 1)
    constexpr int D = NS - 1;                                   // steps a patch is requested ahead
    Cursor cf;                                                  // what the next patch issue fetches
    cf.item = bid; cf.ck = 0;
    cursor_decode(cf);
    int w_cb = cf.cb, w_ck = 0, w_item = bid;                   // what the next weight column loads belong to
    Cursor none = cf;                                           // "no tile": keeps the operation count exact past the last step
#pragma unroll
    for (int h = 0; h < NT; h++) none.n[h] = -1;
#pragma unroll
    for (int d = 0; d < D; d++) {                               // pieces of steps 0 .. D-1
        if (d < n_steps) { issue_patches(cf, d); cursor_next(cf); }
        else issue_patches(none, d);
    }
    load_col(w_cb, w_ck, 0, w[0], w[3], w[6]);
    load_col(w_cb, w_ck, 1, w[1], w[4], w[7]);
    load_col(w_cb, w_ck, 2, w[2], w[5], w[8]);
    auto w_next = [&]() {
        if (++w_ck == a.n_chunks) {
            w_ck = 0; w_item += gridDim.x;
            int pair;
            decode_item(w_item, pair, w_cb);
        }
    };
    w_next();
    fill_tables();                                              // drains the pieces and the first weight set too: the counted waits of step 0 then find less in flight than they allow
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]));
    int ck = 0, item = bid;
    constexpr int CAP = 63;                                      // vmcnt is a 6-bit counter
    // younger than the pieces of step s (requested at the top of step s - D): D x 9 weight loads, the pieces of the D - 1 steps between
    constexpr int N_TOP = D * 9 + (D - 1) * MAX_P, N_COL = 6 + MAX_P;
    constexpr int E1 = EPI_ST, E2 = EPI_ST + EPI_RL;
    int e_type = 0, e_age = 99;                                 // the last epilogue: 1 / 2 (without / with residual loads), and how many steps ago
#define WR_WAIT_E(N0, ET, REGS)                                                                                      \
    do {                                                                                                             \
        if ((ET) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N0) > CAP ? CAP : (N0)) : "memory");                \
        else if ((ET) == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N0) + E1 > CAP ? CAP : (N0) + E1) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N0) + E2 > CAP ? CAP : (N0) + E2) : "memory");                \
        REGS;                                                                                                        \
    } while (0)
    int slot = 0, slot_in = D % NS;                             // slot of step s / slot the pieces of step s + D go to
    for (int s = 0; s < n_steps; s++) {
        const int e_top = e_age <= D ? e_type : 0, e_col = e_age == 1 ? e_type : 0;   // (two epilogues inside the window: only one is counted -- a longer wait, never a shorter one)
        WR_WAIT_E(N_TOP, e_top, (void)0);                       // my pieces of step s have landed
        if (!(a.ablate & 1)) raw_barrier();                     // ... everybody's; and everyone is done with step s-1's slot
        if (a.ablate & 2) { if (s + D < n_steps) cursor_next(cf); }
        else if (s + D < n_steps) { issue_patches(cf, slot_in); cursor_next(cf); }
        else issue_patches(none, slot_in);
        if (ck == 0) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < TH; r++) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (STRIP) strip_item(item);
        }
        const char *sP = smem + slot * SLOT;
        WR_WAIT_E(N_COL, e_col, asm volatile("" : "+v"(w[0]), "+v"(w[3]), "+v"(w[6])));
        __builtin_amdgcn_sched_barrier(0);
        compute_col(sP, 0, std::integral_constant<int, 0>{}, [](int, int) {});
        if (!(a.ablate & 4)) load_col(w_cb, w_ck, 0, w[0], w[3], w[6]);   // (past the last step these fetch a valid, unused block: the count stays exact)
        WR_WAIT_E(N_COL, e_col, asm volatile("" : "+v"(w[1]), "+v"(w[4]), "+v"(w[7])));
        __builtin_amdgcn_sched_barrier(0);
        compute_col(sP, 1, std::integral_constant<int, 0>{}, [](int, int) {});
        if (!(a.ablate & 4)) load_col(w_cb, w_ck, 1, w[1], w[4], w[7]);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_COL) : "memory");   // column 2 was loaded AFTER the epilogue of the step before: no E term
        asm volatile("" : "+v"(w[2]), "+v"(w[5]), "+v"(w[8]));
        __builtin_amdgcn_sched_barrier(0);
        compute_col(sP, 2, std::integral_constant<int, 0>{}, [](int, int) {});
        e_age++;
        if (++ck == a.n_chunks) {                               // the epilogue's own loads (bias rows, residual) make the compiler wait for
            if (!(a.ablate & 8)) epilogue(item, smem + slot * SLOT);      // everything older: the pieces and columns 0 / 1, issued >= 1/3 step ago --
            e_type = a.res ? 2 : 1; e_age = 1;                  // which is why column 2 is re-loaded only after it
            ck = 0; item += gridDim.x;
        }
        if (!(a.ablate & 4)) load_col(w_cb, w_ck, 2, w[2], w[5], w[8]);
        if (s + 2 < n_steps) w_next();
        slot = slot + 1 == NS ? 0 : slot + 1;
        slot_in = slot_in + 1 == NS ? 0 : slot_in + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the surplus loads target registers / LDS of this wave: drain before exit
    }
#undef WR_WAIT_E
}

}  // namespace

bool conv_wr_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_WR")) return false;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin_p % 32 == 0 && a.Cin_p >= 64 && a.Cout_p >= 64 && a.Cout_p % 16 == 0 &&
           a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 12 && a.W >= 12 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo && a.res_Cp == a.Cout_p));
}

template <int TH, int NT, int NW, int NCH, int NS = 2, int XF = 0>
static int wr_launch_t(fid_ctx *ctx, WRArgs &a, int n_tiles) {
    constexpr int PW = (XF & 1) ? 18 + WR_STRIP_GW : 18;
    constexpr int P_BYTES = ((((XF & 3) == 3 && WR_STRIP_GW == 8 ? TH : TH + 2) * PW * 64 + 1023) / 1024) * 1024;
    constexpr int RS_BYTES = NCH > 0 ? ((TH * 16 * NW * 32 + 1023) / 1024) * 1024 : 0;      // resident variant: the residual tile
    a.ncls = a.bias ? ((a.flags & CF_BORDER) ? 9 : 1) : 0;
    a.tab_off = NS * NT * P_BYTES + 1024 + RS_BYTES;
    const int LDS = a.tab_off + (a.ncls + 1) * a.Cout_p * 4;
    FID_REQUIRE(LDS <= 160 * 1024, "conv3x3_wr: %d bytes of LDS", LDS);
    a.n_cblk = cdiv(a.Cout_p, NW * 16);
    a.n_items = cdiv(n_tiles, NT) * a.n_cblk;
    a.d_cblk = fastdiv_make(a.n_cblk);
    FID_REQUIRE(NCH == 0 || (a.n_chunks == NCH && a.n_cblk == 1), "conv3x3_wr: resident variant %d x %d on %d chunks / %d cout blocks", NW * 16, NCH, a.n_chunks, a.n_cblk);
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_wr<TH, NT, NW, NCH, NS, XF>, (int)(LDS)));
    // waves per CU by registers: > 168 VGPRs -> two per SIMD (8 per CU); the one-tile streaming variant stays below 168 -> three per SIMD
    const int wg_per_cu = std::max(1, std::min((NT == 1 && NCH == 0 && TH <= 14 ? 12 : 8) / NW, (160 * 1024) / LDS));
    static const int wgpc_env = getenv("FID_WR_WGPC") ? atoi(getenv("FID_WR_WGPC")) : 0;
    const int grid = std::min(a.n_items, ctx->num_cus * (wgpc_env > 0 ? wgpc_env : wg_per_cu));
    hipLaunchKernelGGL((conv3x3_wr<TH, NT, NW, NCH, NS, XF>), dim3(grid), dim3(NW * 64), LDS, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

// resident-weight variant for this layer: 64 / 96 / 128 couts (one block of 4 / 6 / 8 waves) x 2 | 3 channel chunks
bool conv_wr_resident_ok(const ConvArgs &a) {
    const int nch = a.Cin_p / CK;
    if (!conv_wr_applicable(a) || (nch != 2 && nch != 3)) return false;
    return (a.Cout_p == 64) || (a.Cout_p == 96) || (a.Cout_p == 128 && nch == 2);
}

// nt x cb: 2 x 128 (large batches: a pair of tiles x 128 couts on 8 waves) | 1 x 64 (four waves, two workgroups per CU: enough items
// for every CU when there are only a few hundred tiles); resident: one tile x all couts with the layer's weights kept in registers
int conv_wr_launch(fid_ctx *ctx, const ConvArgs &c, int nt, int cb, int resident, int ring, bool strip) {
    FID_REQUIRE(c.w_alt, "conv3x3_wr needs the fragment-order weights (repack kind 2)");
    FID_REQUIRE(!strip || (conv_strip_ok(c) && ring != 4), "conv3x3_wr: no STRIP tiling for a %d-wide map (ring %d)", c.W, ring);
    FID_REQUIRE(resident ? conv_wr_resident_ok(c) : ((nt == 2 && cb == 128) || (nt == 1 && cb == 64) || (nt == 1 && cb == 128 && strip)), "conv3x3_wr: no %d x %d variant (resident %d)", nt, cb, resident);
    // tile edge 14 or 16: whichever computes fewer pixels for this map (14 fits IResNet's 112 / 56 / 28 / 14 maps exactly and a 40x40
    // map in 3x3 tiles of 196 = 91 % useful pixels, where 16x16 tiles compute 48x48 for 69 %)
    auto padded = [&](int t) { return (long long)cdiv(c.H, t) * t * cdiv(c.W, t) * t; };
    const bool t14 = padded(14) * 16 <= padded(16) * 14;      // 14-wide tiles leave 2 of 16 lanes idle: require the pixel saving to cover that
    // 10-row tiles for the maps neither edge fits (20x20: two tiles of 10 per side = 62 % useful lanes x rows, where 14 / 16-row tiles reach 45 / 39 %);
    // instantiated for the one-tile variants those small maps take (resident, and streaming x 64 couts)
    auto work = [&](int t) { return (double)padded(t) * 16.0 / t; };      // matrix rows x 16 lanes the tiling computes
    const bool t10 = (resident || (nt == 1 && ring != 4)) && work(10) < 0.9 * std::min(work(14), work(16));
    int TH = t10 ? 10 : (t14 ? 14 : 16);
    if (strip) {                                              // rows only: the strip has no padded columns (10-row tiles exist for the one-tile variants)
        TH = conv_strip_rows(c);
        if (TH == 10 && !(resident || nt == 1)) TH = cdiv(c.H, 14) * 14 <= cdiv(c.H, 16) * 16 ? 14 : 16;
        if (TH == 16 && nt == 2 && c.W % 16 != 0 && cdiv(c.H, 16) % 2 != 0) TH = 14;      // (the 16-row pair kernel keeps ONE set of lane shifts per item)
    }
    WRArgs a{};
    a.in = c.in; a.w = c.w_alt; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.act = c.act; a.flags = c.flags; a.rev = c.rev;
    static const int ablate = getenv("FID_WR_ABLATE") ? atoi(getenv("FID_WR_ABLATE")) : 0;
    a.ablate = ablate;
    static const int stagger = getenv("FID_WR_STAGGER") ? atoi(getenv("FID_WR_STAGGER")) : 0;
    a.stagger = stagger;
    const int B = c.M / (c.Ho * c.Wo);
    a.tiles_x = cdiv(c.W, TH);
    a.tiles_per_img = a.tiles_x * cdiv(c.H, TH);
    a.n_tiles = B * a.tiles_per_img;
    a.n_chunks = c.Cin_p / CK;
    a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    a.n_img = B;
    if (strip) {                                              // tile t = strip tile t / tiles_y, tile row t % tiles_y
        a.tiles_y = cdiv(c.H, TH);
        a.n_tiles = cdiv(B * c.W, 16) * a.tiles_y;
        a.d_tpi = fastdiv_make(c.W); a.d_tx = fastdiv_make(a.tiles_y);
        FID_REQUIRE((long long)B * c.W + 16 < (1ll << 27), "conv3x3_wr: strip of %d x %d columns", B, c.W);
    }
    a.in_bytes = c.in_bytes;
    const size_t ob = (size_t)c.M * c.Cout_p * 2;
    FID_REQUIRE(a.in_bytes <= OOB && ob <= OOB, "conv: tensor larger than 2 GiB");
    a.out_bytes = (unsigned)ob;
    a.w_bytes = (unsigned)repack_bytes(2, c.Cout_p, c.Cin_p);
    if (resident) {
        const int key = (c.Cout_p / 16) * 10 + a.n_chunks;
#define WR_RES(NWV, NCHV) (strip ? (TH == 10 ? wr_launch_t<10, 1, NWV, NCHV, 2, 1>(ctx, a, a.n_tiles) : TH == 14 ? wr_launch_t<14, 1, NWV, NCHV, 2, 1>(ctx, a, a.n_tiles) : wr_launch_t<16, 1, NWV, NCHV, 2, 1>(ctx, a, a.n_tiles)) \
                                : (t10 ? wr_launch_t<10, 1, NWV, NCHV>(ctx, a, a.n_tiles) : t14 ? wr_launch_t<14, 1, NWV, NCHV>(ctx, a, a.n_tiles) : wr_launch_t<16, 1, NWV, NCHV>(ctx, a, a.n_tiles)))
        switch (key) {
            case 42: return WR_RES(4, 2);
            case 43: return WR_RES(4, 3);
            case 62: return WR_RES(6, 2);
            case 63: return WR_RES(6, 3);
            case 82: return WR_RES(8, 2);
        }
#undef WR_RES
        set_error("conv3x3_wr: no resident variant for %d couts x %d chunks", c.Cout_p, a.n_chunks);
        return FID_E_INVALID;
    }
    if (strip) {
        const bool one = c.H == 14 && TH == 14;                // the map is one tile high: the two halo rows are skipped
        if (nt == 2) return TH == 14 ? (one ? wr_launch_t<14, 2, 8, 0, 2, 3>(ctx, a, a.n_tiles) : wr_launch_t<14, 2, 8, 0, 2, 1>(ctx, a, a.n_tiles)) : wr_launch_t<16, 2, 8, 0, 2, 1>(ctx, a, a.n_tiles);
        if (cb == 128) {                                       // one tile x 128 couts on eight waves: a quarter of the pair item's work, the epilogue and prologue of ONE item where conv3x3_ks pays two
            if (TH == 10) return wr_launch_t<10, 1, 8, 0, 2, 1>(ctx, a, a.n_tiles);
            return TH == 14 ? (one ? wr_launch_t<14, 1, 8, 0, 2, 3>(ctx, a, a.n_tiles) : wr_launch_t<14, 1, 8, 0, 2, 1>(ctx, a, a.n_tiles)) : wr_launch_t<16, 1, 8, 0, 2, 1>(ctx, a, a.n_tiles);
        }
        if (TH == 10) return wr_launch_t<10, 1, 4, 0, 2, 1>(ctx, a, a.n_tiles);
        return TH == 14 ? (one ? wr_launch_t<14, 1, 4, 0, 2, 3>(ctx, a, a.n_tiles) : wr_launch_t<14, 1, 4, 0, 2, 1>(ctx, a, a.n_tiles)) : wr_launch_t<16, 1, 4, 0, 2, 1>(ctx, a, a.n_tiles);
    }
    if (nt == 2) return t14 ? wr_launch_t<14, 2, 8, 0>(ctx, a, a.n_tiles) : wr_launch_t<16, 2, 8, 0>(ctx, a, a.n_tiles);
    if (ring == 4) return t14 ? wr_launch_t<14, 1, 4, 0, 4>(ctx, a, a.n_tiles) : wr_launch_t<16, 1, 4, 0, 4>(ctx, a, a.n_tiles);
    if (t10) return wr_launch_t<10, 1, 4, 0>(ctx, a, a.n_tiles);
    return t14 ? wr_launch_t<14, 1, 4, 0>(ctx, a, a.n_tiles) : wr_launch_t<16, 1, 4, 0>(ctx, a, a.n_tiles);
}

}  // namespace fid
