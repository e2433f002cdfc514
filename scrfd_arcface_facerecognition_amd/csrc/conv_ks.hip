// 3x3 / stride-1 convolution for FEW tiles (autotuner generation 9, variant ns = 6): conv3x3_wr's one-tile x 64-cout item with the K axis
// split over two wave groups of the workgroup.
//
// IResNet-50 at 64 faces has 64 tiles of 14x14 on its 26 stage-3 layers: 256 items (x 4 cout blocks) for 256 CUs, ONE item per CU, so a launch
// is one item long: prologue + 8 channel steps of 126 MFMAs per wave + epilogue, with one wave per SIMD (conv3x3_wr<14,1,4,0>: 21 us steady of
// which 8 us are matrix work).  More items per CU do not exist; more waves per item do: here 8 waves = (cout fragment cw = 0..3) x (K half kg
// = 0..1).  A step covers TWO 32-channel chunks -- the patch slot holds both, wave group kg multiplies chunk 2s + kg with its own weight
// fragments (weights in registers, loaded straight from global memory in repack.hip kind 2 order, one register set re-loaded column by column
// one step ahead behind hand-counted waits exactly like conv3x3_wr) -- so an item takes half the steps and half the barriers, every SIMD
// runs two waves, and no weight or patch byte is fetched twice.  At the item's end the two partial sums meet in LDS: each wave hands the rows it
// does not finish to its partner (rows [0, TH/2) are finished by kg = 0, the rest by kg = 1), adds what it receives, applies bias / residual /
// activation to ITS rows and stages them; all 8 waves write the tile out as 16-byte rows.  The fp32 sum order differs from conv3x3_wr's
// (chunk pairs are summed inside a group, the two groups last): a different kernel pick, like every other family.
//
// STRIP (round 5; MODE bit 2): x-packed pixel fragments.  The 16 lanes of an MFMA pixel operand are 16 per-lane LDS addresses -- nothing forces
// them into one image.  The rows of ALL images of the batch laid side by side form a strip of B * W columns, and a tile is TH rows x 16 CONSECUTIVE
// STRIP COLUMNS: on a 14-wide map eight images are seven exact fragments (a 14-column tile leaves lanes 14, 15 idle: 1.14 x the matrix work), on
// 28 / 56 / 112 / 20 / 40-wide maps likewise.  A tile crosses at most one image boundary (host-checked): lanes behind it belong to the next image
// and read GW patch positions further right: the right padding of image A and the left padding of image B sit between the two images in the
// patch.  The default is GW = 1: ONE shared zero column (PW = 19).  It breaks the bank pattern of the fragment reads -- the 16 pixels of a fragment
// are no longer 16 CONSECUTIVE 64-byte positions, so one 256-byte bank row gets five of them: SQ_LDS_BANK_CONFLICT 47 % of the LDS cycles against
// 5 % (profiles/r05) -- and is still the faster form: GW = 8 (PW = 26; -DFID_STRIP_GW=8) moves the lanes behind the boundary by 512 bytes (the same
// bank row quarter and XOR swizzle as without the gap: conflict-free again, maps one tile high then store only their TH real rows so that the
// slots fit) and measured equal-to-slower (below: seven more cells per row in every LDS-DMA piece list).
// Row-sharing, the tap order and every counted wait are unchanged; what changes is the
// piece -> pixel mapping of the patch fetch, a per-item lane shift in the fragment addresses, and the lane -> pixel mapping of the epilogue.
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
constexpr int CK = 32;                                            // channels per chunk (patch width: 18 positions, 19 with STRIP)
constexpr int NW = 4, KS = 2, NWT = NW * KS, CBW = NW * 16;       // cout fragments, K halves, waves, couts per item
constexpr int ROWB = CBW * 2, CPX = ROWB / 16;                    // bytes / 16-byte chunks of a staged pixel row

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct KSArgs {
    const void *in;
    const void *w;        // repack.hip kind 2
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, Cin_p, Cout_p;
    int act, flags;
    int tiles_x, tiles_per_img, n_tiles, n_cblk, n_items, n_chunks, n_steps_item;
    FastDiv d_cblk, d_tpi, d_tx;
    unsigned in_bytes, out_bytes, w_bytes;
    int ncls;             // bias classes: 9 with CF_BORDER, 1 with a plain bias, 0 without
    int rev;              // ConvArgs::rev
    int n_img;            // images (MOSAIC: the faces four of which share a tile; STRIP: the strip's images)
    int tiles_y;          // STRIP: tile rows; tile t = (strip tile t / tiles_y, tile row t % tiles_y), d_tx divides by tiles_y, d_tpi by W
    int ablate;           // FID_KS_ABLATE timing experiments (wrong results): 1 no step barrier, 2 no patch pieces, 4 no weight reloads, 8 no epilogue, 16 no matrix work
};

// MOSAIC (TH = 16, 7x7 maps -- IResNet's last stage): a tile is a 2 x 2 mosaic of four images, image (r, c) of the mosaic at tile pixels
// [8r, 8r + 7) x [8c, 8c + 7); tile row / column 7 and 15 are zero gutters (the zero padding between neighbours: never fetched, computed and
// dropped), so one 16-pixel matrix column carries two images and an item's 64 x 9 Cin weights serve four images instead of one.
// MODE 2 (the map is one tile high, H = TH -- IResNet's 14x14 stage): patch rows 0 and TH + 1 lie outside the image: zeros, not multiplied.
// STRIP (MODE & 4): see the head of the file; MODE 6 = STRIP on a map one tile high.
// gap between a STRIP tile's two images: 8 positions (conflict-free fragment reads) where two patch slots of two chunks each + the K-half exchange
// area fit the 160 KB (10-row tiles; 14-row tiles of maps one tile high, which store 14 patch rows), else the one shared zero column
// MEASURED (profiles/r05/ab_gutter8.txt, same box, two alternations): the conflict-free 8-position gap is NOT faster -- IResNet-50 at batch 500
// 6.40 -> 6.46 ms, at 128 and SCRFD-10G +-0: the seven extra positions per row are seven more 64-byte cells every LDS-DMA piece list carries (19 -> 23 KB
// per tile and chunk, one more piece per wave and step), and the fragment reads were not what the step waits for.  Default: the one shared column;
// `make EXTRA=-DFID_STRIP_GW=8` builds the wide gap.
#ifndef FID_STRIP_GW
#define FID_STRIP_GW 1
#endif
constexpr int ks_gutter(int th, int mode) { return (FID_STRIP_GW == 8 && (th == 10 || (mode & 3) == 2)) ? 8 : 1; }
template <int TH, int MODE>
__global__ void __launch_bounds__(NWT * 64, 2) conv3x3_ks(const KSArgs a) {
    constexpr bool MOSAIC = (MODE & 3) == 1, ONE_ROW = (MODE & 3) == 2, STRIP = (MODE & 4) != 0;
    static_assert(!(STRIP && MOSAIC), "one packing at a time");
    constexpr int GW = STRIP ? ks_gutter(TH, MODE) : 0;        // patch positions between the tile's two images
    constexpr int PW = 18 + GW;
    constexpr bool COMPACT = STRIP && ONE_ROW && GW == 8;       // the two halo rows are never read: the slot holds patch rows 1 .. TH only (what lets the wide gap fit)
    constexpr int R0 = COMPACT ? 1 : 0;
    constexpr int TW = STRIP ? 16 : TH, PH = TH + 2, NPIX = (COMPACT ? TH : PH) * PW, RH = TH / 2;
    constexpr int P_BLKS = (NPIX * 64 + 1023) / 1024, P_BYTES = P_BLKS * 1024, SLOT = KS * P_BYTES, NS = 2;
    constexpr int N_PIECES = KS * P_BLKS, MAX_P = (N_PIECES + NWT - 1) / NWT;
    constexpr int EX_BYTES = NWT * RH * 1024;                    // one KB per (wave, handed-over row)
    constexpr int ST_I = (TH * 16 * CPX + NWT * 64 - 1) / (NWT * 64);
    constexpr int OFF_SPARE = NS * SLOT, OFF_EX = OFF_SPARE + 1024, OFF_TAB = OFF_EX + EX_BYTES;
    static_assert(TH % 2 == 0 && TH * 16 * ROWB <= SLOT, "the finished tile is staged in the step's patch slot");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & (NW - 1), kg = wave >> 2;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_items ? (a.n_items - 1 - bid) / gridDim.x + 1 : 0;
    const int n_steps = my_items * a.n_steps_item;
    if (n_steps == 0) return;
    const int frow = lane & 15, fq = lane >> 4;

    auto decode_item = [&](int item, int &tile, int &cb) {
        if (a.rev) item = a.n_items - 1 - item;
        tile = fastdiv(item, a.d_cblk);
        cb = item - tile * a.n_cblk;
    };
    auto decode_tile = [&](int t, int &n, int &ty, int &tx) {
        if (MOSAIC) { n = t; ty = 0; tx = 0; return; }           // n = the mosaic; its images are 4n .. 4n + 3
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
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.out), 0, a.out_bytes, 0x00020000);

    // ---- my patch pieces of a step: piece j = wave + 8k covers chunk half j / P_BLKS, patch pixels 16 (j % P_BLKS) .. +15, 4 lanes per pixel
    int p_pk[MAX_P];                                            // py | px << 8 | channel offset (halfs) << 16; py = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P; k++) {
        const int j = wave + NWT * k;
        const int hf = j / P_BLKS, blk = j - hf * P_BLKS;
        const int row = blk * 16 + (lane >> 2);
        int py = row / PW;
        const int px = row - py * PW;
        if (row >= NPIX || px >= (STRIP ? PW : TW + 2) || j >= N_PIECES) py = 255;
        p_pk[k] = py | (px << 8) | (((((lane & 3) ^ swz64(row)) * 8) + hf * CK) << 16);
    }
    struct Cursor {
        int item, ck, cb;                // ck = step inside the item
        int n, y0, x0;                   // image and top-left input pixel of the tile's haloed patch (n < 0: no tile)
    };
    auto cursor_decode = [&](Cursor &c) {
        int tile, n, ty, tx;
        decode_item(c.item, tile, c.cb);
        decode_tile(tile < a.n_tiles ? tile : 0, n, ty, tx);
        c.n = tile < a.n_tiles ? n : -1; c.y0 = ty * TH - 1; c.x0 = (STRIP ? tx : tx * TW) - 1;
    };
    auto cursor_next = [&](Cursor &c) {
        if (++c.ck == a.n_steps_item) {
            c.ck = 0;
            c.item += gridDim.x;
            cursor_decode(c);
        }
    };
    auto issue_patches = [&](const Cursor &c, int slot) {      // exactly MAX_P instructions
        const int c0 = c.ck * (KS * CK);
        char *dst = smem + slot * SLOT;
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + NWT * k;
            int pk = p_pk[k];
            asm volatile("" : "+v"(pk));                        // opaque: unpack at the use
            const int py = pk & 255, iy = c.y0 + py + R0, ix = c.x0 + ((pk >> 8) & 255);
            bool in;
            unsigned vo;
            if (MOSAIC) {
                const int img = c.n * 4 + (iy >> 3) * 2 + (ix >> 3), ly = iy & 7, lx = ix & 7;
                in = c.n >= 0 && py != 255 && (unsigned)iy < 16u && (unsigned)ix < 16u && ly < 7 && lx < 7 && img < a.n_img && !(a.ablate & 2);
                vo = in ? (unsigned)((((img * 7 + ly) * 7 + lx) * a.Cin_p + c0 + (pk >> 16)) * 2) : OOB;
            } else if (STRIP) {                                 // position p holds image A's column x0 + p up to its right padding (column W), behind it image B from column 0
                const bool second = ix > a.W;                   // (positions W + 1 .. W + GW - 2 of a wide gap are never read; W + GW - 1 is image B's left padding)
                const int img = c.n + (second ? 1 : 0), lx = second ? ix - a.W - GW : ix;
                in = c.n >= 0 && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)lx < (unsigned)a.W && img < a.n_img && !(a.ablate & 2);
                vo = in ? (unsigned)((((img * a.H + iy) * a.W + lx) * a.Cin_p + c0 + (pk >> 16)) * 2) : OOB;
            } else {
                in = c.n >= 0 && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a.ablate & 2);
                vo = in ? (unsigned)((((c.n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + (pk >> 16)) * 2) : OOB;
            }
            char *d = j < N_PIECES ? dst + j * 1024 : smem + OFF_SPARE;      // surplus piece: zeros into the spare KB
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)d, 16, vo, 0, 0, 0);
        }
    };
    // ---- weights: one register set, the three taps of a column per load group (repack kind 2 keeps them contiguous), SGPR base per (cout block, chunk, column)
    const unsigned long long wp = (unsigned long long)a.w;
    const i32x4 rs_w = i32x4{(int)(unsigned)wp, (int)((unsigned)(wp >> 32) & 0xFFFFu), (int)a.w_bytes, 0x00020000};
    const int w_voff = cw * 9216 + lane * 16;
    half8 w[9];                                                 // w[dy*3 + dx]
    auto load_col = [&](int cb, int step, int dx, half8 &t0, half8 &t1, half8 &t2) {   // exactly 3 instructions
        const int gf = cb * NW, ck = step * KS + kg;
        const int soff = __builtin_amdgcn_readfirstlane(((gf >> 3) * a.n_chunks + ck) * (8 * 9216) + (gf & 7) * 9216 + dx * 3072);
        asm volatile("buffer_load_dwordx4 %0, %3, %4, %5 offen\n\t"
                     "buffer_load_dwordx4 %1, %3, %4, %5 offen offset:1024\n\t"
                     "buffer_load_dwordx4 %2, %3, %4, %5 offen offset:2048"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2)
                     : "v"(w_voff), "s"(rs_w), "s"(soff)
                     : "memory");
    };

    // fragment addresses: lin = q' + frow (+ 1 for the STRIP lanes behind the tile's image boundary: the shared zero column lies between the images)
    int pbase[2][4];
    auto set_pbase = [&](int fs) {
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) pbase[par][c] = fs * 64 + ((fq ^ ((((fs + par) >> 1) + c) & 3)) << 4) + kg * P_BYTES;
    };
    set_pbase(frow);
    auto strip_item = [&](int item_) {                          // STRIP: the lane shift of the item about to be multiplied
        int tile, cb, n, ty, c0;
        decode_item(item_, tile, cb);
        decode_tile(tile < a.n_tiles ? tile : 0, n, ty, c0);
        int fr = frow;
        asm volatile("" : "+v"(fr));
        set_pbase(fr + (fr >= a.W - c0 ? GW : 0));
    };

    f32x4 acc[TH];
    constexpr int PD = 4;                                       // pixel fragments read ahead
    auto compute_col = [&](int slot_off, int dx_) {
        int pb[2][4];
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = pbase[par][c] + slot_off;
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int q, int set) {                     // q = dx * PH + patch row
            const int K = (q % PH - R0) * PW + q / PH;          // lin = K + frow
            pq[set] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + K * 64));
        };
        // MOSAIC: patch rows 0, 8, 16, 17 (the halo above / below and the two gutters) are zeros and output rows 7, 15 are dropped:
        // neither read nor multiplied -- 14 of 18 fragments, 38 of 48 products per column
        constexpr int NR = MOSAIC ? 14 : (ONE_ROW ? TH : PH);
        auto row_of = [](int j) { return MOSAIC ? (j < 7 ? j + 1 : j + 2) : (ONE_ROW ? j + 1 : j); };
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            if (dx != dx_) continue;
#pragma unroll
            for (int j = 0; j < PD; j++) load_p(dx * PH + row_of(j), j % (PD + 1));
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int r = row_of(j);
                if (j + PD < NR) load_p(dx * PH + row_of(j + PD), (j + PD) % (PD + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int mi = r - dy;
                    if (mi < 0 || mi >= TH || (MOSAIC && (mi & 7) == 7)) continue;
                    acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy * 3 + dx], pq[j % (PD + 1)], acc[mi], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // bias rows [ncls][64] + PReLU slopes [64] of cout block `cb` in LDS (the epilogue reads them with ds_read)
    const float *sTab = (const float *)(smem + OFF_TAB);
    auto fill_tables = [&](int cb) {
        float *tb = (float *)(smem + OFF_TAB);
        const int c0 = cb * CBW;
        for (int i = tid; i < a.ncls * CBW; i += NWT * 64) {
            const int cls = i / CBW, c = i - cls * CBW;
            tb[i] = c0 + c < a.Cout_p ? a.bias[cls * a.Cout_p + c0 + c] : 0.f;
        }
        for (int i = tid; i < CBW; i += NWT * 64) tb[a.ncls * CBW + i] = (a.act == ACT_PRELU && c0 + i < a.Cout_p) ? a.slope[c0 + i] : 1.f;
    };

    // ---- the item's end: the two K halves meet, each wave finishes RH rows.  Exactly RH residual loads (if any) + ST_I stores per wave.
    auto epilogue_body = [&](int item, char *stage, auto act_tag, auto res_tag, auto border_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool RES = decltype(res_tag)::value, BORDER = decltype(border_tag)::value;
        int tile, cb, n, ty, tx;
        decode_item(item, tile, cb);
        decode_tile(tile < a.n_tiles ? tile : 0, n, ty, tx);
        int lo = lane;
        asm volatile("" : "+v"(lo));                            // opaque lane id: keeps this block's per-lane arithmetic out of the step loop
        const int fr = lo & 15, q4 = lo >> 4;
        // hand the rows my partner finishes over: kg = 0 gives rows RH .. TH-1, kg = 1 rows 0 .. RH-1 (one KB per row and wave, lane-major)
        char *ex_w = smem + OFF_EX + ((cw * KS + (kg ^ 1)) * RH) * 1024 + lo * 16;
        const char *ex_r = smem + OFF_EX + ((cw * KS + kg) * RH) * 1024 + lo * 16;
#pragma unroll
        for (int i = 0; i < RH; i++) *(f32x4 *)(ex_w + i * 1024) = kg ? acc[i] : acc[RH + i];
        const int co0 = cb * CBW + cw * 16 + q4 * 4, cl = cw * 16 + q4 * 4;
        const bool co_ok = co0 < a.Cout_p;
        // MOSAIC: wave group kg finishes mosaic row kg (RH = 8 tile rows = one image's 7 rows + the gutter); the lane's column picks the image
        if (MOSAIC) n = n * 4 + kg * 2 + (fr >> 3);
        const int c0 = tx;                                      // STRIP: lane 0's column in image n; lanes fr >= W - c0 are columns fr - (W - c0) of image n + 1
        const bool sec = STRIP && fr >= a.W - c0;
        const int n_tile = n;
        if (STRIP) n += sec ? 1 : 0;
        const int oy0 = MOSAIC ? 0 : ty * TH + kg * RH, ox = MOSAIC ? (fr & 7) : (STRIP ? c0 + fr - (sec ? a.W : 0) : tx * TW + fr);
        const bool t_ok = tile < a.n_tiles && co_ok && fr < TW && ox < a.W && (!(MOSAIC || STRIP) || n < a.n_img);
        const unsigned rstride = (unsigned)(a.W * a.Cout_p * 2);
        const unsigned t_base = (unsigned)((((n * a.H + oy0) * a.W + ox) * a.Cout_p + co0) * 2);
        u32x2 rr[RH];
        if (RES) {
#pragma unroll
            for (int i = 0; i < RH; i++)
                rr[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, (t_ok && oy0 + i < a.H) ? t_base + i * rstride : OOB, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // the partial sums are in LDS; everyone is done reading the step's patch slot
        f32x4 bmid = f32x4{0.f, 0.f, 0.f, 0.f}, btop = bmid, bbot = bmid, sl = f32x4{1.f, 1.f, 1.f, 1.f};
        if (BORDER) {       // exact fold of a BatchNorm in front of the zero-padded conv: the bias row depends on the pixel's border class
            const int xc = ox == 0 ? 0 : (ox == a.W - 1 ? 2 : 1);
            btop = *(const f32x4 *)(sTab + (0 + xc) * CBW + cl);
            bmid = *(const f32x4 *)(sTab + (3 + xc) * CBW + cl);
            bbot = *(const f32x4 *)(sTab + (6 + xc) * CBW + cl);
        } else if (a.ncls) {
            bmid = *(const f32x4 *)(sTab + cl);
        }
        if (ACT == ACT_PRELU) sl = *(const f32x4 *)(sTab + a.ncls * CBW + cl);
        // my 8 bytes of a staged pixel row (128 B = 8 chunks of 8 couts), the chunk rotated by the pixel column: no bank pile-up
        char *sp = stage + (kg * RH) * (16 * ROWB) + fr * ROWB + (((cw * 2 + (q4 >> 1) + fr) % CPX) << 4) + (q4 & 1) * 8;
#pragma unroll
        for (int i = 0; i < RH; i++) {
            const f32x4 other = *(const f32x4 *)(ex_r + i * 1024);
            f32x4 v = (kg ? acc[RH + i] : acc[i]) + other;
            const int oy = oy0 + i;
            if (!BORDER) v += bmid;
            else v += (oy == 0 ? btop : (oy == a.H - 1 ? bbot : bmid));
            if (RES) v += __builtin_convertvector(__builtin_bit_cast(half4, rr[i]), f32x4);
            if (ACT == ACT_PRELU) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + sl * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
            half4 h = __builtin_convertvector(v, half4);
            if (ACT == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
            *(half4 *)(sp + i * (16 * ROWB)) = h;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // the tile is staged
        // write-out: 16-byte slot g = i * 512 + thread = (pixel g / 8, chunk g % 8): 64 pixels = 4 tile rows per round
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int q0 = t2 >> 3, c = t2 & 7;
        const int pr0 = q0 >> 4, pc = q0 & 15;
        const bool sec2 = STRIP && pc >= a.W - c0;
        const int n2 = STRIP ? n_tile + (sec2 ? 1 : 0) : n;
        const int oxx = MOSAIC ? (pc & 7) : (STRIP ? c0 + pc - (sec2 ? a.W : 0) : tx * TW + pc), co = cb * CBW + c * 8;
        const bool okc = tile < a.n_tiles && pc < TW && oxx < a.W && co < a.Cout_p && (!STRIP || n2 < a.n_img);
        const char *lsrc = stage + q0 * ROWB + (((c + pc) % CPX) << 4);
        if (MOSAIC) {       // rounds 0, 1: mosaic row 0 (image rows pr0, 4 + pr0), rounds 2, 3: mosaic row 1
            const int img0 = tile * 4 + (pc >> 3);
#pragma unroll
            for (int i = 0; i < ST_I; i++) {
                const int img = img0 + (i >> 1) * 2, ly = 4 * (i & 1) + pr0;
                const bool ok = okc && ly < 7 && img < a.n_img;
                const u32x4 v = *(const u32x4 *)(lsrc + i * (64 * ROWB));
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, ok ? (unsigned)((((img * 7 + ly) * 7 + oxx) * a.Cout_p + co) * 2) : OOB, 0, 0);
            }
            return;
        }
        const unsigned g0 = (unsigned)((((n2 * a.H + ty * TH + pr0) * a.W + oxx) * a.Cout_p + co) * 2);
#pragma unroll
        for (int i = 0; i < ST_I; i++) {
            const int row = 4 * i + pr0;
            const bool ok = okc && row < TH && ty * TH + row < a.H;
            const u32x4 v = *(const u32x4 *)(lsrc + (row < TH ? i * (64 * ROWB) : 0));
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, ok ? g0 + (unsigned)(4 * i) * rstride : OOB, 0, 0);
        }
    };
    auto epilogue = [&](int item, char *stage) {
        using std::integral_constant;
        const bool res = a.res != nullptr, border = (a.flags & CF_BORDER) != 0;
#define KS_EPI(A) \
        do { \
            if (res) { if (border) epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, true>{}, integral_constant<bool, true>{}); \
                       else epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, true>{}, integral_constant<bool, false>{}); } \
            else { if (border) epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{}); \
                   else epilogue_body(item, stage, integral_constant<int, A>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{}); } \
        } while (0)
        if (a.act == ACT_PRELU) KS_EPI(ACT_PRELU);
        else if (a.act == ACT_RELU) KS_EPI(ACT_RELU);
        else KS_EPI(ACT_NONE);
#undef KS_EPI
    };

    // ---- operation order per wave and step s (vmcnt retires in order; every wait names only OLDER operations; an item's epilogue adds
    // operations between column 1's and column 2's re-load, which makes the waits after it stricter than needed, never looser):
    //   top: MAX_P pieces of step s + 1;  after column dx: its 3 weight loads for step s + 1 (column 2: after the epilogue, if any)
    //   wait at the top for my pieces of step s: younger = 9 weight loads;  wait before column dx for its fragments: younger = 6 loads + MAX_P pieces
    Cursor cf;
    cf.item = bid; cf.ck = 0;
    cursor_decode(cf);
    int w_cb = cf.cb, w_ck = 0, w_item = bid;                   // what the next weight column loads belong to
    Cursor none = cf;
    none.n = -1;
    issue_patches(cf, 0);
    cursor_next(cf);
    load_col(w_cb, w_ck, 0, w[0], w[3], w[6]);
    load_col(w_cb, w_ck, 1, w[1], w[4], w[7]);
    load_col(w_cb, w_ck, 2, w[2], w[5], w[8]);
    auto w_next = [&]() {
        if (++w_ck == a.n_steps_item) {
            w_ck = 0; w_item += gridDim.x;
            int tile;
            decode_item(w_item < a.n_items ? w_item : bid, tile, w_cb);     // (past the last item: a valid, unused block)
        }
    };
    w_next();
    int tab_cb = cf.cb;
    {
        int tile0, cb0;
        decode_item(bid, tile0, cb0);
        tab_cb = cb0;
        fill_tables(cb0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]));
    constexpr int N_TOP = 9, N_COL = 6 + MAX_P;
    int ck = 0, item = bid, slot = 0;
    for (int s = 0; s < n_steps; s++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP) : "memory");      // my pieces of step s have landed
        if (!(a.ablate & 1)) raw_barrier();                     // ... everybody's; everyone is done with the other slot (and with the tables of the item before)
        if (s + 1 < n_steps) { issue_patches(cf, slot ^ 1); cursor_next(cf); }
        else issue_patches(none, slot ^ 1);
        if (ck == 0) {
#pragma unroll
            for (int r = 0; r < TH; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (STRIP) strip_item(item);
        }
        const int so = slot * SLOT;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_COL) : "memory");
        asm volatile("" : "+v"(w[0]), "+v"(w[3]), "+v"(w[6]));
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.ablate & 16)) compute_col(so, 0);
        if (!(a.ablate & 4)) load_col(w_cb, w_ck, 0, w[0], w[3], w[6]);              // (past the last step these fetch a valid, unused block: the count stays exact)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_COL) : "memory");
        asm volatile("" : "+v"(w[1]), "+v"(w[4]), "+v"(w[7]));
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.ablate & 16)) compute_col(so, 1);
        if (!(a.ablate & 4)) load_col(w_cb, w_ck, 1, w[1], w[4], w[7]);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_COL) : "memory");
        asm volatile("" : "+v"(w[2]), "+v"(w[5]), "+v"(w[8]));
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.ablate & 16)) compute_col(so, 2);
        if (++ck == a.n_steps_item) {
            if (!(a.ablate & 8)) epilogue(item, smem + so);
            ck = 0; item += gridDim.x;
            if (s + 1 < n_steps) {                              // the next item's cout block (the same one whenever the grid is a multiple of the block count)
                int tile, cb;
                decode_item(item, tile, cb);
                if (cb != tab_cb) { tab_cb = cb; fill_tables(cb); }      // (published by the next step's barrier; read an item later)
            }
        }
        if (!(a.ablate & 4)) load_col(w_cb, w_ck, 2, w[2], w[5], w[8]);
        w_next();
        slot ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the surplus loads target registers / LDS of this wave: drain before exit
}

template <int TH, int MODE = 0>
static int ks_launch_t(fid_ctx *ctx, KSArgs &a, int per_wg) {
    constexpr int PW = 18 + ((MODE & 4) ? ks_gutter(TH, MODE) : 0);
    constexpr int P_BYTES = ((((MODE & 7) == 6 && PW == 26 ? TH : TH + 2) * PW * 64 + 1023) / 1024) * 1024;
    const int LDS = 2 * KS * P_BYTES + 1024 + NWT * (TH / 2) * 1024 + (a.ncls + 1) * CBW * 4;
    FID_REQUIRE(LDS <= 160 * 1024, "conv3x3_ks: %d bytes of LDS", LDS);
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_ks<TH, MODE>, LDS));
    const int grid = per_wg >= 2 ? std::min(cdiv(a.n_items, per_wg), std::max(1, ctx->num_cus / per_wg)) : std::min(a.n_items, ctx->num_cus);   // per_wg = 2: at most half the CUs
    hipLaunchKernelGGL((conv3x3_ks<TH, MODE>), dim3(grid), dim3(NWT * 64), LDS, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace

// 7x7 maps (below conv3x3_wr's 12-pixel minimum): four images per 16x16 tile
bool conv_ks_mosaic(const ConvArgs &a) {
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.H == 7 && a.W == 7 && a.Ho == 7 && a.Wo == 7 && a.Cin_p % 64 == 0 &&
           a.Cout_p % 16 == 0 && a.Cout_p >= 64 && a.w_rows == a.Cout_p && !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == 7 && a.res_W == 7 && a.res_Cp == a.Cout_p));
}

bool conv_ks_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_KS")) return false;
    if ((a.Cin_p / CK) % KS != 0) return false;
    if (conv_ks_mosaic(a)) return true;
    return conv_wr_applicable(a);
}

// STRIP tiles (16 consecutive columns of the strip of all images' rows): every tile may cross at most ONE image boundary
bool conv_strip_ok(const ConvArgs &a) {
    if (getenv("FID_NO_STRIP") || a.W < 12) return false;
    for (int k = 0, c0 = 0; k < a.W; k++, c0 = (c0 + 16) % a.W)
        if (c0 + 15 >= 2 * a.W) return false;
    return true;
}
// tile rows of a STRIP launch: the edge that pads the map's height least (ties: the taller tile re-fetches fewer halo rows)
int conv_strip_rows(const ConvArgs &a) {
    int best = 16;
    for (int t : {14, 10})
        if (cdiv(a.H, t) * t < cdiv(a.H, best) * best) best = t;
    return best;
}
bool conv_ks_strip_applicable(const ConvArgs &a) { return conv_ks_applicable(a) && !conv_ks_mosaic(a) && conv_strip_ok(a); }

static int ks_tile_rows(const ConvArgs &c) {
    if (conv_ks_mosaic(c)) return 16;
    auto padded = [&](int t) { return (long long)cdiv(c.H, t) * t * cdiv(c.W, t) * t; };
    const bool t14 = padded(14) * 16 <= padded(16) * 14;
    auto work = [&](int t) { return (double)padded(t) * 16.0 / t; };
    const bool t10 = work(10) < 0.9 * std::min(work(14), work(16));
    return t10 ? 10 : (t14 ? 14 : 16);
}

int conv_ks_items(const ConvArgs &c, bool strip) {
    const int TH = strip ? conv_strip_rows(c) : ks_tile_rows(c), B = c.M / (c.Ho * c.Wo);
    const int tiles = strip ? cdiv(B * c.W, 16) * cdiv(c.H, TH) : (conv_ks_mosaic(c) ? cdiv(B, 4) : B * cdiv(c.W, TH) * cdiv(c.H, TH));
    return tiles * cdiv(c.Cout_p, CBW);
}

// per_wg = 2: half the workgroups (at most half the CUs), two or more items each -- longer alone (IResNet-50 stage 3 at 64 faces: 27 vs 20 us per
// layer) but on half the chip; with two batches in flight the other lane's kernels run on the free half.  The plan decides (tile 512 vs 256).
int conv_ks_launch(fid_ctx *ctx, const ConvArgs &c, int per_wg, bool strip) {
    FID_REQUIRE(c.w_alt, "conv3x3_ks needs the fragment-order weights (repack kind 2)");
    FID_REQUIRE(strip ? conv_ks_strip_applicable(c) : conv_ks_applicable(c), "conv3x3_ks: layer not applicable (strip %d)", (int)strip);
    const bool mosaic = conv_ks_mosaic(c);
    const int TH = strip ? conv_strip_rows(c) : ks_tile_rows(c);
    KSArgs a{};
    a.in = c.in; a.w = c.w_alt; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.act = c.act; a.flags = c.flags; a.rev = c.rev;
    const int B = c.M / (c.Ho * c.Wo);
    a.n_img = B;
    a.tiles_x = cdiv(c.W, TH);
    a.tiles_per_img = a.tiles_x * cdiv(c.H, TH);
    a.n_tiles = mosaic ? cdiv(B, 4) : B * a.tiles_per_img;
    if (strip) {                                                // tile t = strip tile t / tiles_y, tile row t % tiles_y
        a.tiles_y = cdiv(c.H, TH);
        a.n_tiles = cdiv(B * c.W, 16) * a.tiles_y;
        FID_REQUIRE((long long)B * c.W + 16 < (1ll << 27), "conv3x3_ks: strip of %d x %d columns", B, c.W);
    }
    a.n_chunks = c.Cin_p / CK;
    a.n_steps_item = a.n_chunks / KS;
    a.n_cblk = cdiv(c.Cout_p, CBW);
    a.n_items = a.n_tiles * a.n_cblk;
    a.d_cblk = fastdiv_make(a.n_cblk); a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    if (strip) { a.d_tpi = fastdiv_make(c.W); a.d_tx = fastdiv_make(a.tiles_y); }
    a.in_bytes = c.in_bytes;
    const size_t ob = (size_t)c.M * c.Cout_p * 2;
    FID_REQUIRE(a.in_bytes <= OOB && ob <= OOB, "conv: tensor larger than 2 GiB");
    a.out_bytes = (unsigned)ob;
    a.w_bytes = (unsigned)repack_bytes(2, c.Cout_p, c.Cin_p);
    a.ncls = c.bias ? ((c.flags & CF_BORDER) ? 9 : 1) : 0;
    static const int ablate = getenv("FID_KS_ABLATE") ? atoi(getenv("FID_KS_ABLATE")) : 0;
    a.ablate = ablate;
    if (strip) {
        if (TH == 10) return ks_launch_t<10, 4>(ctx, a, per_wg);
        if (TH == 14) return c.H == 14 ? ks_launch_t<14, 6>(ctx, a, per_wg) : ks_launch_t<14, 4>(ctx, a, per_wg);
        return ks_launch_t<16, 4>(ctx, a, per_wg);
    }
    if (mosaic) return ks_launch_t<16, 1>(ctx, a, per_wg);
    if (TH == 10) return ks_launch_t<10>(ctx, a, per_wg);
    if (TH == 14) return c.H == 14 ? ks_launch_t<14, 2>(ctx, a, per_wg) : ks_launch_t<14>(ctx, a, per_wg);
    return ks_launch_t<16>(ctx, a, per_wg);
}

}  // namespace fid
