// 3x3 / stride-2 / pad-1 convolution for layers with 64, 96 or 128 input channels and at most 128 couts (autotuner generation 10).
//
// The stride-2 convs ran on the implicit GEMM: every output pixel fetches its 9 taps through L2 (2.25 fetches of every input
// pixel) and 128x128 tiles of a 96-cout layer waste a quarter of the matrix work -- SCRFD's layer2.0.conv1 (64 -> 96, 409 600
// output pixels) took 120 us at 12 % of the MFMA peak while moving only 1.75 TB/s, bound by neither.  These layers are
// bandwidth-shaped (4 input pixels per output pixel, few channels): the floor is one pass over the input.
//
//   tile    = 8 x 16 OUTPUT pixels of one image; its input patch = 17 rows x 33 columns, fetched once per 32-channel chunk by
//             LDS-DMA one step ahead into two slots.  The patch is stored in LDS with its even and its odd columns in separate
//             planes ([row][E: 17 pixels][O: 16 pixels]): output column ox reads input column 2 ox + dx - 1, i.e. E[ox],
//             O[ox], E[ox + 1] for dx = 0, 1, 2 -- 16 CONSECUTIVE pixels of a plane per fragment, so the B-operand reads are the
//             conflict-free ds_read_b128 pattern of the stride-1 kernels (a stride-2 gather in an interleaved image is 2-way
//             conflicted and the reads, not the matrix pipes, would set the pace)
//   wave w  = couts 16w .. 16w+15 for all 128 pixels (as in conv_wr.hip): ALL its weights (NCH chunks x 9 taps x 16 couts x 32
//             channels) stay in registers for the kernel's lifetime -- nothing but patches is fetched in the loop
//   row sharing: patch row pr feeds output row o with tap row dy where 2 o + dy = pr (one or two output rows per patch row)
//   epilogue: bias / PReLU slopes in registers (one cout block), optional residual by 8-byte loads, the tile is staged through
//             the patch slot the step has just finished with and leaves as 16 bytes per lane / whole cout rows per pixel
//
// Operation counts per wave are exact (surplus DMA pieces go to a spare KB, out-of-range stores to an out-of-bounds offset), so
// the step's top wait can leave an item's stores in flight.
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
constexpr int CK = 32;                       // channels per step
constexpr int TOH = 8, TOW = 16;             // output tile
constexpr int PR = 2 * TOH + 1;              // 17 patch rows
constexpr int NE = TOW + 1, NO = TOW;        // even / odd plane pixels per patch row
constexpr int RW = NE + NO;                  // 33 pixels per patch row
constexpr int NPIX = PR * RW;                // 561
constexpr int P_BLKS = (NPIX * 64 + 1023) / 1024, P_BYTES = P_BLKS * 1024;   // 36 pieces

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct S2Args {
    const void *in;
    const void *w;        // repack.hip kind 2 (3x3 taps, fragment order)
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    void *out2;           // DUAL: the second output (the block's shortcut), same shape as out
    int H, W, Ho, Wo, Cin_p, Cout_p;
    int act, flags;
    int tiles_x, tiles_per_img, n_tiles;
    FastDiv d_tpi, d_tx;
    unsigned in_bytes, out_bytes, w_bytes;
    // NX > 0: the block's 1x1 / stride-2 shortcut conv rides along (IResNet's downsampling block; lower.py, the generation-12 form of this
    // kernel): x = the block input [B, H2, W2, NX*32] read at (xs oy, xs ox), its weights = columns [9 Cin_p, 9 Cin_p + NX*32) of the plain
    // second weight image w2 (rows of `krow` halfs), the bias table is the two convs' summed bias -- no residual read
    const void *in2, *w2;
    int H2, W2, krow, xs;     // xs: x is sampled at (xs oy, xs ox): 2 = the block input itself, 1 = its even-pixel copy (stem_block.hip)
    unsigned in2_bytes, w2_bytes;
};

// NW: waves = 16-cout fragments (Cout_p <= NW*16); NCH: 32-channel chunks (Cin_p = NCH*32)
// DUAL: TWO convs over the same patch in one launch -- waves 0 .. NW/2-1 are the 3x3 / stride-2 conv (output `out`, activation a.act),
// waves NW/2 .. NW-1 the block's shortcut, a 2x2 / stride-2 conv stored as a 3x3 one whose taps (0, *) and (*, 0) are zero (output
// `out2`, no activation): they skip those taps' fragment reads and MFMAs.  The weight / bias tables hold both convs' rows back to back;
// the tile is written out in two passes (each half staged through the finished patch slot), all NW waves storing in both.
// NSLOT: patch slots (2: the next step's patch is requested one step ahead; 3: TWO steps ahead -- the eight-wave variants run one workgroup
// per CU, so nothing but a deeper ring hides the ~2 us a cold patch takes to arrive behind a 1.1 us step: round 4)
// NX: 32-channel chunks of the block input whose 1x1 / stride-2 shortcut conv is absorbed (0: none).  Every item then has one more step: its
// "patch" is the 8 x 16 sampled pixels of x (NX chunks of 8 KB, pixel-linear like a plane row), its matrix work 8 x NX MFMAs per wave.
template <int NW, int NCH, bool DUAL = false, int NSLOT = 2, int NX = 0>
__global__ void __launch_bounds__(NW * 64, DUAL ? 3 : ((NCH * NW >= 32 || NSLOT > 2) ? 1 : 2)) conv3x3_s2(const S2Args a) {   // (four chunks x eight fragments: 144 weight VGPRs per wave, one workgroup per CU)
    constexpr int NCHT = NCH + (NX > 0 ? 1 : 0);                                 // steps per item
    static_assert(!(DUAL && NX > 0) && NX * 8 <= P_BLKS, "variants");
    constexpr int NH = DUAL ? NW / 2 : NW;                                      // waves (= cout fragments) of one output
    constexpr int CBW = NH * 16, ROWB = CBW * 2, CPX = NH * 2;
    constexpr int MAX_P = (P_BLKS + NW - 1) / NW;
    constexpr int PPI = NW * 64 / CPX, RPI = PPI / 16;                          // pixels / tile rows one write-out instruction of the workgroup covers
    static_assert(PPI % 16 == 0, "a write-out instruction covers whole tile rows");
    constexpr int ST_I = (TOH * TOW * CPX + NW * 64 - 1) / (NW * 64);          // write-out instructions per wave and pass
    static_assert(TOH * TOW * ROWB <= P_BYTES, "staging area");
    extern __shared__ __attribute__((aligned(16))) char smem[];                 // NSLOT patch slots + 1 spare KB

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_tiles = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    const int n_steps = my_tiles * NCHT;
    if (n_steps == 0) return;
    const int frow = lane & 15, fq = lane >> 4;

    auto decode_tile = [&](int t, int &n, int &ty, int &tx) {
        n = fastdiv(t, a.d_tpi);
        const int r = t - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.out), 0, a.out_bytes, 0x00020000);
    const auto rs_out2 = __builtin_amdgcn_make_buffer_rsrc((void *)(DUAL ? a.out2 : a.out), 0, a.out_bytes, 0x00020000);
    const auto rs_in2 = __builtin_amdgcn_make_buffer_rsrc((void *)(NX > 0 ? a.in2 : a.in), 0, NX > 0 ? a.in2_bytes : a.in_bytes, 0x00020000);
    const int part = DUAL ? (wave >= NH ? 1 : 0) : 0, wave_l = wave - part * NH;   // which output this wave computes; its fragment inside it

    // ---- my patch pieces: piece j = wave + NW k covers LDS pixels 16 j .. 16 j + 15 (4 lanes per pixel); LDS pixel lin = pr*33 + q,
    // q < 17: even plane, input column 2 q; q >= 17: odd plane, input column 2 (q - 17) + 1 (columns relative to the patch's first)
    int p_pk[MAX_P];                                            // pr | pc << 8 | channel offset << 16; pr = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P; k++) {
        const int j = wave + NW * k;
        const int lin = j * 16 + (lane >> 2);
        int pr = lin / RW;
        const int q = lin - pr * RW;
        const int pc = q < NE ? 2 * q : 2 * (q - NE) + 1;
        if (lin >= NPIX || j >= P_BLKS) pr = 255;
        p_pk[k] = pr | (pc << 8) | ((((lane & 3) ^ swz64(lin)) * 8) << 16);
    }
    struct Cursor { int tile, ck, n, y0, x0; };
    auto cursor_decode = [&](Cursor &c) __attribute__((always_inline)) {
        int n, ty, tx;
        decode_tile(c.tile < a.n_tiles ? c.tile : 0, n, ty, tx);
        c.n = c.tile < a.n_tiles ? n : -1; c.y0 = 2 * ty * TOH - 1; c.x0 = 2 * tx * TOW - 1;
    };
    auto cursor_next = [&](Cursor &c) __attribute__((always_inline)) {
        if (++c.ck == NCHT) { c.ck = 0; c.tile += gridDim.x; cursor_decode(c); }
    };
    auto issue_patches = [&](const Cursor &c, int slot) __attribute__((always_inline)) {      // exactly MAX_P instructions (always inlined: an outlined copy takes the argument block through scratch)
        const int c0 = c.ck * CK;
        char *dst = smem + slot * P_BYTES;
        if (NX > 0 && c.ck == NCH) {                            // the shortcut's step: x at (2 oy, 2 ox); piece j = chunk j / 8, tile row j % 8
            const int oy0 = (c.y0 + 1) >> 1, ox0 = (c.x0 + 1) >> 1;
#pragma unroll
            for (int k = 0; k < MAX_P; k++) {
                const int jr = wave + NW * k;
                const bool real = jr < NX * 8;
                const int j = real ? jr : NSLOT * P_BLKS - slot * P_BLKS;
                const int r = jr & 7, cx = lane >> 2, lin = r * 16 + cx;
                const int oy = oy0 + r, ox = ox0 + cx;
                const bool in = real && c.n >= 0 && oy < a.Ho && ox < a.Wo;
                // (every surplus piece gets an offset of its own: the compiler folds two loads with identical operands into one -- found by
                //  tools/check_waitcnt.py, the step then issued MAX_P - 1 operations and the counted waits were one short)
                const unsigned vo = in ? (unsigned)((((c.n * a.H2 + a.xs * oy) * a.W2 + a.xs * ox) * (NX * CK) + (jr >> 3) * CK + (((lane & 3) ^ swz64(lin)) * 8)) * 2) : OOB - (unsigned)k * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in2, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + NW * k < P_BLKS ? wave + NW * k : NSLOT * P_BLKS - slot * P_BLKS;   // surplus piece: the spare KB behind the slots
            int pk = p_pk[k];
            asm volatile("" : "+v"(pk));
            const int pr = pk & 255;
            const int iy = c.y0 + pr, ix = c.x0 + ((pk >> 8) & 255);
            const bool in = c.n >= 0 && pr != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned vo = in ? (unsigned)((((c.n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + ((pk >> 16) & 255)) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
        }
    };

    // first patch and all weights in one trip to memory (both cold at launch: ~1-2 us each)
    constexpr int D = NSLOT - 1;                                // steps a patch is requested ahead
    Cursor cf;
    cf.tile = bid; cf.ck = 0;
    cursor_decode(cf);
    Cursor none = cf;                                           // "no tile": every piece out of bounds -- keeps the operation count exact past the last step
    none.n = -1;
#pragma unroll
    for (int d = 0; d < D; d++) {                               // pieces of steps 0 .. D-1
        if (d < n_steps) { issue_patches(cf, d); cursor_next(cf); }
        else issue_patches(none, d);
    }
    // ---- weights: all NCH x 9 fragments of this wave's 16 couts, resident (repack kind 2: [cb128][chunk][cf 0..7][dx][dy][lane])
    const unsigned long long wp = (unsigned long long)a.w;
    const i32x4 rs_w = i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)wp), __builtin_amdgcn_readfirstlane((int)((unsigned)(wp >> 32) & 0xFFFFu)),
                             __builtin_amdgcn_readfirstlane((int)a.w_bytes), 0x00020000};   // (scalar registers whatever the pressure: the asm below names them as such)
    const int w_voff = ((wave >> 3) * NCH * 8 + (wave & 7)) * 9216 + lane * 16;    // (more than 8 fragments: the next 128-cout block)
    half8 w[NCH * 9];                                           // w[chunk*9 + dy*3 + dx]
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int soff = __builtin_amdgcn_readfirstlane(c * (8 * 9216) + dx * 3072);
            asm volatile("buffer_load_dwordx4 %0, %3, %4, %5 offen\n\t"
                         "buffer_load_dwordx4 %1, %3, %4, %5 offen offset:1024\n\t"
                         "buffer_load_dwordx4 %2, %3, %4, %5 offen offset:2048"
                         : "=&v"(w[c * 9 + 0 + dx]), "=&v"(w[c * 9 + 3 + dx]), "=&v"(w[c * 9 + 6 + dx])
                         : "v"(w_voff), "s"(rs_w), "s"(soff)
                         : "memory");
        }
    half8 wx[NX > 0 ? NX : 1];                                 // the shortcut's A fragments: row = cout, 8 consecutive channels per lane, from the plain image
    if (NX > 0) {
        const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, a.w2_bytes, 0x00020000);
        const int row = wave * 16 + (lane & 15);
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const unsigned off = row < a.Cout_p ? (unsigned)((row * a.krow + 9 * a.Cin_p + k * CK + (lane >> 4) * 8) * 2) : OOB;
            wx[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, off, 0, 0));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NCH * 9; i++) asm volatile("" : "+v"(w[i]));
#pragma unroll
    for (int i = 0; i < (NX > 0 ? NX : 0); i++) asm volatile("" : "+v"(wx[i]));

    // bias / slopes of this lane's 4 couts: one cout block, so they never change
    f32x4 k_bias = f32x4{0.f, 0.f, 0.f, 0.f}, k_sl = f32x4{1.f, 1.f, 1.f, 1.f};
    {
        const int c0 = wave * 16 + fq * 4, cc = (DUAL || c0 < a.Cout_p) ? c0 : 0;     // (DUAL: the table has both outputs' rows)
        if (a.bias) k_bias = *(const f32x4 *)(a.bias + cc);
        if (!DUAL && a.act == ACT_PRELU) k_sl = *(const f32x4 *)(a.slope + cc);
    }

    // ---- pixel fragment addresses: lin = K + frow with K a compile-time constant (conv_wr.hip): base register + immediate
    int pbase[2][4];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int c = 0; c < 4; c++) pbase[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c) & 3)) << 4);

    f32x4 acc[TOH];
    constexpr int PD = DUAL ? 2 : 3;                            // fragments read ahead (DUAL: twelve waves at <= 168 VGPRs)
    constexpr int NF = PR * 3;                                  // fragments per step: (patch row, dx), dx fastest
    auto compute = [&](const char *sP, auto c_tag, auto short_tag) __attribute__((always_inline)) {
        constexpr int WB = decltype(c_tag)::value * 9;
        constexpr bool SHORT = decltype(short_tag)::value;       // the shortcut's waves: taps dy, dx in {1, 2} only
        auto used = [](int f) {                                  // does fragment (patch row, dx) feed any MFMA of this wave
            const int pr = f / 3, dx = f % 3;
            if (!SHORT) return true;
            if (dx == 0) return false;
            for (int dy = 1; dy < 3; dy++)
                if (pr - dy >= 0 && !((pr - dy) & 1) && (pr - dy) / 2 < TOH) return true;
            return false;
        };
        int pb[2][4];
        const int slot_off = (int)(sP - smem);
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (DUAL) {                                     // (recomputed per step: eight fewer registers across the loop)
                    int lo_ = lane;
                    asm volatile("" : "+v"(lo_));
                    const int fr_ = lo_ & 15, fq_ = lo_ >> 4;
                    pb[par][c] = fr_ * 64 + ((fq_ ^ ((((fr_ + par) >> 1) + c) & 3)) << 4) + slot_off;
                } else {
                    pb[par][c] = pbase[par][c] + slot_off;
                }
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int f, int set) {                     // f = pr*3 + dx: E[ox] / O[ox] / E[ox + 1]
            if (!used(f)) return;
            const int pr = f / 3, dx = f % 3;
            const int K = pr * RW + (dx == 0 ? 0 : (dx == 1 ? NE : 1));
            pq[set] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + K * 64));
        };
#pragma unroll
        for (int f = 0; f < PD; f++) load_p(f, f % (PD + 1));
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const int pr = f / 3, dx = f % 3;
            if (f + PD < NF) load_p(f + PD, (f + PD) % (PD + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                if ((pr - dy) < 0 || ((pr - dy) & 1)) continue;
                if (SHORT && (dy == 0 || dx == 0)) continue;
                const int o = (pr - dy) / 2;
                if (o >= TOH) continue;
                acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[WB + dy * 3 + dx], pq[f % (PD + 1)], acc[o], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the shortcut's step: fragment (row r, chunk k) = the 16 sampled pixels of tile row r, pixel-linear in chunk k's 8 KB
    auto compute_x = [&](const char *sP) __attribute__((always_inline)) {
        const int slot_off = (int)(sP - smem);
        int pb0 = pbase[0][0] + slot_off;
        asm volatile("" : "+v"(pb0));
        half8 px[TOH];
#pragma unroll
        for (int k = 0; k < (NX > 0 ? NX : 0); k++) {
#pragma unroll
            for (int r = 0; r < TOH; r++) px[r] = *(const half8 *)(smem + (pb0 + k * 8192 + r * 1024));
#pragma unroll
            for (int r = 0; r < TOH; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[k], px[r], acc[r], 0, 0, 0);
        }
    };

    // ---- epilogue of `tile` (sums in acc): bias (+ residual) + activation in the accumulator layout, staged, written out as 16-byte slots
    constexpr int EPI_ST = (DUAL ? 2 : 1) * ST_I, EPI_RL = TOH;
    auto epilogue_body = [&](int tile, char *stage, auto act_tag, auto res_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool RES = decltype(res_tag)::value && !DUAL;
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int fr = lo & 15, q4 = lo >> 4;
        const int co0 = wave_l * 16 + q4 * 4;
        const bool co_ok = co0 < a.Cout_p;
        int n, ty, tx;
        decode_tile(tile < a.n_tiles ? tile : 0, n, ty, tx);
        const int oy0 = ty * TOH, ox0 = tx * TOW, ox = ox0 + fr;
        const unsigned rstride = (unsigned)(a.Wo * a.Cout_p * 2);
        const int st_w = fr * ROWB + (((wave_l * 2 + (q4 >> 1) + fr) % CPX) << 4) + (q4 & 1) * 8;
        u32x2 rr[TOH];
        if (RES) {
            const bool lane_ok = tile < a.n_tiles && co_ok && ox < a.Wo;
            const unsigned base = (unsigned)((((n * a.Ho + oy0) * a.Wo + ox) * a.Cout_p + co0) * 2);
#pragma unroll
            for (int r = 0; r < TOH; r++) rr[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, (lane_ok && oy0 + r < a.Ho) ? base + r * rstride : OOB, 0, 0);
        }
#pragma unroll
        for (int pass = 0; pass < (DUAL ? 2 : 1); pass++) {
            raw_barrier();                                      // every wave is done reading the slot / the pass before has been read back
            if (part == pass) {
#pragma unroll
                for (int r = 0; r < TOH; r++) {
                    f32x4 v = acc[r] + k_bias;
                    if (RES) v += __builtin_convertvector(__builtin_bit_cast(half4, rr[r]), f32x4);
                    if (ACT == ACT_PRELU && pass == 0) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + k_sl * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
                    half4 h = __builtin_convertvector(v, half4);
                    if (ACT == ACT_RELU && pass == 0) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                    *(half4 *)(stage + r * (16 * ROWB) + st_w) = h;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            raw_barrier();                                      // the (half) tile is staged
            // write-out: slot (i*NW + wave)*64 + lane = pixel PPI i + q0, chunk c (constants recomputed per pass from an opaque lane id:
            // they must not stay in registers across the step loop)
            int lo2 = lo;
            asm volatile("" : "+v"(lo2));
            const int wl = wave * 64 + lo2, q0 = wl / CPX, c = wl - q0 * CPX;
            const int pr0 = q0 >> 4, pc = q0 & 15;
            const int oxx = ox0 + pc, co = c * 8;
            const bool okc = tile < a.n_tiles && oxx < a.Wo && co < a.Cout_p;
            const char *lsrc = stage + q0 * ROWB + (((c + pc) % CPX) << 4);
            const unsigned g0 = (unsigned)((((n * a.Ho + oy0 + pr0) * a.Wo + oxx) * a.Cout_p + co) * 2);
            const int rows_left = (a.Ho - oy0 < TOH ? a.Ho - oy0 : TOH) - pr0;
#pragma unroll
            for (int i = 0; i < ST_I; i++) {
                const bool in_tile = RPI * i + pr0 < TOH;
                const u32x4 v = *(const u32x4 *)(lsrc + (in_tile ? i * PPI * ROWB : 0));
                const unsigned off = (okc && in_tile && RPI * i < rows_left) ? g0 + (unsigned)(RPI * i) * rstride : OOB;
                if (pass == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rs_out2, off, 0, 0);
            }
        }
    };
    auto epilogue = [&](int tile, char *stage) __attribute__((always_inline)) {
        using std::integral_constant;
        const bool res = a.res != nullptr;
#define S2_EPI(A) \
        do { \
            if (res) epilogue_body(tile, stage, integral_constant<int, A>{}, integral_constant<bool, true>{}); \
            else epilogue_body(tile, stage, integral_constant<int, A>{}, integral_constant<bool, false>{}); \
        } while (0)
        if (a.act == ACT_PRELU) S2_EPI(ACT_PRELU);
        else if (a.act == ACT_RELU) S2_EPI(ACT_RELU);
        else S2_EPI(ACT_NONE);
#undef S2_EPI
    };

    // ---- steps: (tile, chunk); patches D = NSLOT - 1 steps ahead.  Operation order per wave and step t: [top: MAX_P pieces of step t + D
    // (real, or out-of-bounds ones past the last step: the count stays exact)] [last chunk: EPI_RL residual loads (if any) + EPI_ST stores].
    // Top wait for the pieces of step s (requested at the top of step s - D): younger = the pieces of the D - 1 steps between and the
    // epilogue of at most ONE of the steps s - D .. s - 1 (NCH >= 2 >= D: two item ends are never less than two steps apart).
    constexpr int E1 = EPI_ST, E2 = EPI_ST + EPI_RL, N_TOP = (D - 1) * MAX_P;
    static_assert(NCHT >= D && N_TOP + E2 <= 63, "counted waits");
    int e_type = 0, e_age = 99, tile = bid, s = 0;             // the last epilogue: 1 / 2 (without / with residual loads), and how many steps ago
    int slot = 0, slot_in = D % NSLOT;                          // slot of step s / slot the pieces of step s + D go to
    auto step = [&](auto c_tag) __attribute__((always_inline)) {
        constexpr int C = decltype(c_tag)::value;
        const int e_top = e_age <= D ? e_type : 0;
        if (e_top == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP) : "memory");
        else if (e_top == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP + E1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP + E2) : "memory");
        raw_barrier();
        if (s + D < n_steps) { issue_patches(cf, slot_in); cursor_next(cf); }
        else if (D > 1) issue_patches(none, slot_in);           // (D = 1: no later wait counts them)
        if (C == 0) {
#pragma unroll
            for (int r = 0; r < TOH; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (NX > 0 && C == NCH) compute_x(smem + slot * P_BYTES);
        else if (DUAL && part) compute(smem + slot * P_BYTES, std::integral_constant<int, (C < NCH ? C : 0)>{}, std::integral_constant<bool, true>{});
        else compute(smem + slot * P_BYTES, std::integral_constant<int, (C < NCH ? C : 0)>{}, std::integral_constant<bool, false>{});
        e_age++;
        if (C == NCHT - 1) {
            epilogue(tile, smem + slot * P_BYTES);
            e_type = a.res ? 2 : 1; e_age = 1;
            tile += gridDim.x;
        }
        s++;
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        slot_in = slot_in + 1 == NSLOT ? 0 : slot_in + 1;
    };
    for (int it = 0; it < my_tiles; it++) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        if constexpr (NCHT > 2) step(std::integral_constant<int, 2>{});
        if constexpr (NCHT > 3) step(std::integral_constant<int, 3>{});
        if constexpr (NCHT > 4) step(std::integral_constant<int, 4>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// the fused form: 64 input channels, 96 + 96 couts, no residual / activation on the second output
static bool conv_s2_dual_ok(const ConvArgs &a) {
    return a.out2 != nullptr && a.kh == 3 && a.kw == 3 && a.stride == 2 && a.pad == 1 && a.Cin_p == 64 && a.Cout_p == 96 && a.w_rows == 192 &&
           a.Ho == (a.H - 1) / 2 + 1 && a.Wo == (a.W - 1) / 2 + 1 && a.Ho >= 8 && a.Wo >= 8 && a.flags == 0 && a.nsig == 0 && a.res == nullptr &&
           a.act != ACT_PRELU;
}

// the shortcut-absorbing form (lower.py's second weight image; a generation-12 pick with ns = 10): IResNet's downsampling block on 64 -> 64 and
// 128 -> 128 channels whose block input has 64 channels (layer1.0 / layer2.0 of IResNet-50)
static bool conv_s2_sc_ok(const ConvArgs &a) {
    return a.in2 != nullptr && a.T2 == 1 && (a.s2 == 2 || a.s2 == 1) && a.Cin2_p == 64 && a.res == nullptr && a.H2 > (a.Ho - 1) * a.s2 && a.W2 > (a.Wo - 1) * a.s2 &&
           ((a.Cin_p == 64 && a.Cout_p == 64) || (a.Cin_p == 128 && a.Cout_p == 128));
}

bool conv_s2_applicable(const ConvArgs &a) {
    if (a.out2) return conv_s2_dual_ok(a);
    if (getenv("FID_NO_S2")) return false;
    if (a.in2 && !conv_s2_sc_ok(a)) return false;
    const int nch = a.Cin_p / CK;
    // (128 -> 128 channels, IResNet's layer2.0.conv2: all 4 x 9 fragments of a wave's 16 couts = 144 VGPRs, eight waves, one workgroup per CU)
    return a.kh == 3 && a.kw == 3 && a.stride == 2 && a.pad == 1 && a.Cin_p % 32 == 0 && (nch == 2 || nch == 3 || (nch == 4 && a.Cout_p == 128)) &&
           (a.Cout_p == 64 || a.Cout_p == 96 || (a.Cout_p == 128 && nch != 3)) && a.w_rows == a.Cout_p &&
           a.Ho == (a.H - 1) / 2 + 1 && a.Wo == (a.W - 1) / 2 + 1 && a.Ho >= 8 && a.Wo >= 8 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32 | CF_BORDER)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo && a.res_Cp == a.Cout_p));
}

template <int NW, int NCH, bool DUAL = false, int NSLOT = 2, int NX = 0>
static int s2_launch_t(fid_ctx *ctx, const S2Args &a) {
    constexpr int LDS = NSLOT * P_BYTES + 1024;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_s2<NW, NCH, DUAL, NSLOT, NX>, (int)(LDS)));
    // workgroups per CU: two patch-slot pairs fit LDS; registers allow 12 waves per CU (<= 168 VGPRs) for two chunks, 8 for three
    const int wg_per_cu = (NCH * NW >= 32 || NSLOT > 2) ? 1 : std::max(1, std::min((NCH == 2 ? 12 : 8) / NW, 2));
    const int grid = std::min(a.n_tiles, ctx->num_cus * wg_per_cu);
    hipLaunchKernelGGL((conv3x3_s2<NW, NCH, DUAL, NSLOT, NX>), dim3(grid), dim3(NW * 64), LDS, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

int conv_s2_launch(fid_ctx *ctx, const ConvArgs &c) {
    FID_REQUIRE(c.w_alt, "conv3x3_s2 needs the fragment-order weights (repack kind 2)");
    FID_REQUIRE(conv_s2_applicable(c), "conv3x3_s2: layer not supported");
    S2Args a{};
    a.in = c.in; a.w = c.w_alt; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out; a.out2 = c.out2;
    a.H = c.H; a.W = c.W; a.Ho = c.Ho; a.Wo = c.Wo; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.act = c.act; a.flags = c.flags;
    const int B = c.M / (c.Ho * c.Wo);
    a.tiles_x = cdiv(c.Wo, TOW);
    a.tiles_per_img = a.tiles_x * cdiv(c.Ho, TOH);
    a.n_tiles = B * a.tiles_per_img;
    a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    a.in_bytes = c.in_bytes;
    const size_t ob = (size_t)c.M * c.Cout_p * 2;
    FID_REQUIRE(a.in_bytes <= OOB && ob <= OOB, "conv: tensor larger than 2 GiB");
    a.out_bytes = (unsigned)ob;
    a.w_bytes = (unsigned)repack_bytes(2, c.w_rows, c.Cin_p);
    if (c.out2) return s2_launch_t<12, 2, true>(ctx, a);
    if (c.in2) {                                                // the shortcut rides along: plain second image for its fragments, summed bias
        a.in2 = c.in2; a.w2 = c.w; a.H2 = c.H2; a.W2 = c.W2; a.krow = 9 * c.Cin_p + c.T2 * c.Cin2_p; a.xs = c.s2;
        a.in2_bytes = c.in2_bytes; a.w2_bytes = c.w_bytes;
        FID_REQUIRE(c.w_bytes >= (unsigned)((size_t)c.w_rows * a.krow * 2), "conv3x3_s2: second weight image of %u bytes for %d rows of %d halfs", c.w_bytes, c.w_rows, a.krow);
        if (c.Cin_p == 64) return s2_launch_t<4, 2, false, 2, 2>(ctx, a);
        return s2_launch_t<8, 4, false, 3, 2>(ctx, a);
    }
    const int key = (c.Cout_p / 16) * 10 + c.Cin_p / CK;
    switch (key) {
        case 42: return s2_launch_t<4, 2>(ctx, a);
        case 43: return s2_launch_t<4, 3>(ctx, a);
        case 62: return s2_launch_t<6, 2>(ctx, a);
        case 63: return s2_launch_t<6, 3>(ctx, a);
        case 82: return s2_launch_t<8, 2, false, 3>(ctx, a);     // (eight waves = one workgroup per CU: three patch slots)
        case 84: return s2_launch_t<8, 4, false, 3>(ctx, a);
    }
    set_error("conv3x3_s2: no variant for %d couts x %d chunks", c.Cout_p, c.Cin_p / CK);
    return FID_E_INVALID;
}

}  // namespace fid
