// conv_gw.hip (generation 11): implicit GEMM with the WEIGHTS IN REGISTERS -- for the GEMM-shaped layers the halo-patch kernels
// cannot take: stride-2 convs with >= 128 input channels, 1x1 / 2x2 shortcut convs, 3x3 convs on 7x7 / 20x20 maps.
//
// Why: the LDS-staged implicit GEMM (conv.hip, generations 1 / 2) pushes BOTH operands of a 128 x 128 x 64 step (32 KB) through
// the global -> LDS path, which a CU serves at ~29 B/clk L2-hot: 1100 cycles of fill for 512 cycles of matrix work (measured
// 14-30 % MFMA-busy on IResNet-50's stride-2 and 7x7 layers).  Here only the PIXEL operand goes through LDS; every wave loads the
// MFMA A fragments of ITS OWN 64 couts straight from global memory into VGPRs (repack kind 3: one contiguous KB per fragment), one
// K-step ahead.  An item = BM flattened output pixels x NW*64 couts; per 32-channel K-step: BM*64 B of LDS fill (8 KB at BM = 128),
// 4 KB of register loads per wave, PF = BM/16 pixel-fragment reads from LDS each feeding 4 MFMAs.  With two workgroups per CU
// (NW = 4) a step pair costs 1024 matrix cycles against 16 KB of fill (16 B/clk), 32 KB of register loads (32 B/clk of the 64 B/clk
// L1 path) and 64 KB of LDS reads (64 of 128 B/clk): every stream at about half of what the CU can take.
//
// K order: tap-major, 32-channel chunks inside a tap (consecutive steps read the two halves of a pixel's 128-byte lines).
// Zero padding, stride and ragged tiles come from the buffer descriptor: a lane whose input pixel lies outside the image (or whose
// output pixel is past M) requests an out-of-range offset and the LDS-DMA writes zeros.
// vmcnt is counted by hand (it retires in order; every wait names only OLDER operations).  Per wave and step s:
//   top: wait for MY pieces of step s (requested at step s - D), barrier, request the MAX_P pieces of step s + D,
//        wait for the 4 A fragments of step s (requested during step s - 1), request those of step s + 1, multiply;
//   last step of an item: epilogue = [4*PF residual loads, consumed in place] + E = 2*PF 16-byte row stores.
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

__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct GWArgs {
    const void *in;
    const void *w;        // repack.hip kind 3: [cout block of 128][tap][32-channel chunk][cout fragment 0..7][lane][8 halfs]
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, Ho, Wo, Cin_p, Cout_p;
    int kh, kw, stride, pad, act;
    int M, n_nblk, n_items, n_chunks, taps, n_steps;     // n_steps = taps * n_chunks K-steps per item, rounded up to the kernel's step group
    FastDiv d_nblk, d_hw, d_w;                           // item -> (pixel tile, cout block); pixel -> (image, oy, ox)
    unsigned in_bytes, out_bytes, w_bytes;
    int tab_off, ncls;                                   // LDS offset of the fp32 bias [ncls][Cout_p] + slope [Cout_p] tables; ncls = 9 with CF_BORDER, 1, or 0 (no bias)
    int ablate;                                          // FID_GW_ABLATE != 0: no epilogue (timing experiment, wrong results)
};

// BM: output pixels per item (64 | 128); NW: waves, 64 couts each; NS: pixel slots (pieces are requested D = NS - 1 steps ahead);
// WD: steps the weight fragments are requested ahead (WD + 1 register sets).  Large batches run 2-4 workgroups per CU, which hide a
// trip to memory behind each other (NS = 3, WD = 1); with a few hundred items in all (64 faces) a CU holds one workgroup whose step
// is ~120 ns of matrix work, so the requests must be 5 / 3 steps ahead to cover the ~0.6-1 us trip themselves.
// KC: 32-channel chunks per K-step.  A single wave issues one instruction every ~4-5 cycles, and a step's bookkeeping (cursors, addresses,
// waits: ~180 scalar / vector instructions) costs the same whatever the step multiplies: with KC = 1 a BM = 64 step is 16 MFMAs (256 matrix
// cycles) behind ~850 cycles of instruction issue -- measured 430 ns per step with every memory operation switched off.  KC chunks per step
// put 16*KC MFMAs behind the same bookkeeping.
template <int BM, int NW, int NS, int WD, int KC>
__global__ void __launch_bounds__(NW * 64, 2) conv_gw(const GWArgs a) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only needs the launch stub; the body is gfx950 builtins and inline assembly)
    constexpr int PF = BM / 16;                                 // pixel fragments = 1-KB pieces per step
    constexpr int MAX_P = PF / NW;                              // pieces per wave, step and chunk
    static_assert(PF % NW == 0 && MAX_P >= 1, "every wave requests the same number of pieces");
    constexpr int SUB = BM * 64, SLOT = SUB * KC;               // a slot holds the step's KC chunks, chunk-major
    constexpr int D = NS - 1;
    constexpr int E = 2 * PF;                                   // stores per wave and item
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sStage = smem + NS * SLOT;                            // NW x 2 KB: per-wave transpose of one pixel fragment's 16 x 64 results

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);        // the cout blocks of a pixel tile are consecutive items: one L2
    const int my_items = bid < a.n_items ? (a.n_items - 1 - bid) / gridDim.x + 1 : 0;
    if (my_items == 0) return;
    const int frow = lane & 15, fq = lane >> 4;

    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.out), 0, a.out_bytes, 0x00020000);
    const unsigned long long wp = (unsigned long long)a.w;
    const i32x4 rs_w = i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)wp), __builtin_amdgcn_readfirstlane((int)((unsigned)(wp >> 32) & 0xFFFFu)),
                             __builtin_amdgcn_readfirstlane((int)a.w_bytes), 0x00020000};

    // ---- pixel pieces: piece j = wave + NW*k holds tile pixels 16 j .. 16 j + 15, 4 lanes x 16 B per pixel; the 16-byte channel
    // group a lane fetches is XOR-swizzled by the pixel row, so that a fragment read (16 pixels x 4 groups) is bank-conflict-free
    // Fetch cursor (D steps ahead of the multiply): item, K-step of the item, tap and the byte offset of the step's first chunk.  A lane's
    // request offsets pv[] = (image, tap-shifted pixel) are recomputed only when the tap changes (every n_chunks / KC steps); inside a tap
    // a step's pieces differ by the scalar chunk offset alone.  Steps past taps * n_chunks / KC are padding: no pixels (zeros).
    int f_item = bid, f_k = 0, f_t = 0, f_dy = 0, f_dx = 0, f_coff = 0;
    unsigned pc_base[MAX_P];                                    // byte offset of the pixel's image (+ the lane's swizzled 16-byte group)
    int pc_y0[MAX_P], pc_x0[MAX_P];                             // top-left input coordinate of the pixel's window (y0 = -30000: no pixel)
    unsigned pv[MAX_P];                                         // the current tap's request offset per piece (OOB: zeros)
    const int sw_g = ((lane & 3) ^ (((lane >> 2) >> 1) & 3)) * 16;   // (row = 16 j + (lane >> 2): the swizzle only sees lane bits)
    const int cin_b = a.Cin_p * 2;
    auto tap_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int iy = pc_y0[k] + f_dy, ix = pc_x0[k] + f_dx;
            const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && f_t < a.taps;
            pv[k] = in ? pc_base[k] + (unsigned)((iy * a.W + ix) * cin_b) : OOB;
        }
    };
    auto item_pixels = [&]() __attribute__((always_inline)) {   // per item: where this lane's pixels of the tile live
        const int mt = fastdiv(f_item, a.d_nblk);
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + NW * k;
            const int p = mt * BM + j * 16 + (lane >> 2);
            const int n = fastdiv(p, a.d_hw), r = p - n * (a.Ho * a.Wo);
            const int oy = fastdiv(r, a.d_w), ox = r - oy * a.Wo;
            const bool ok = p < a.M && f_item < a.n_items;
            pc_base[k] = (unsigned)(n * a.H * a.W) * (unsigned)cin_b + (unsigned)sw_g;
            pc_y0[k] = ok ? oy * a.stride - a.pad : -30000;
            pc_x0[k] = ox * a.stride - a.pad;
        }
        tap_offsets();
    };
    auto fetch_next = [&]() __attribute__((always_inline)) {
        f_coff += 64 * KC;
        if (++f_k == a.n_steps) {
            f_k = 0; f_coff = 0; f_t = 0; f_dy = 0; f_dx = 0;
            f_item += gridDim.x;
            item_pixels();
        } else if (f_coff == cin_b) {
            f_coff = 0;
            ++f_t;
            if (++f_dx == a.kw) { f_dx = 0; ++f_dy; }
            tap_offsets();
        }
    };
    auto issue_pixels = [&](int slot_b) __attribute__((always_inline)) {        // exactly MAX_P * KC instructions; slot_b = byte offset of the slot
        char *dst = smem + slot_b;
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + NW * k;
#pragma unroll
            for (int q = 0; q < KC; q++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + q * SUB + j * 1024), 16, pv[k], f_coff + q * 64, 0, 0);
        }
    };

    // ---- weights: the wave's 4 cout fragments of one K-step = 4 buffer loads of 1 KB (fragment i at +i KB)
    const int w_voff = lane * 16;
    half8 wS[WD + 1][4 * KC];                                   // register sets: step s multiplies with set s % (WD + 1); [chunk q][fragment i]
    // weight cursor (WD steps ahead): repack kind 3 stores a cout block's K-steps back to back, so the scalar offset just advances
    int w_item = bid, w_k = 0, w_soff = 0;
    const int w_blk = a.taps * a.n_chunks * 8192;               // bytes of one 128-cout block
    auto w_base = [&]() __attribute__((always_inline)) {
        const int it = w_item < a.n_items ? w_item : 0;
        const int mt = fastdiv(it, a.d_nblk), nb = it - mt * a.n_nblk;
        const int gf = nb * (NW * 4) + wave * 4;
        w_soff = __builtin_amdgcn_readfirstlane((gf >> 3) * w_blk + (gf & 7) * 1024);
    };
    auto load_w = [&](half8 (&w)[4 * KC]) __attribute__((always_inline)) {  // exactly 4 * KC instructions, then the cursor advances one step
#pragma unroll
        for (int q = 0; q < KC; q++) {
            asm volatile("buffer_load_dwordx4 %0, %4, %5, %6 offen\n\t"
                         "buffer_load_dwordx4 %1, %4, %5, %6 offen offset:1024\n\t"
                         "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:2048\n\t"
                         "buffer_load_dwordx4 %3, %4, %5, %6 offen offset:3072"
                         : "=&v"(w[q * 4 + 0]), "=&v"(w[q * 4 + 1]), "=&v"(w[q * 4 + 2]), "=&v"(w[q * 4 + 3])
                         : "v"(w_voff), "s"(rs_w), "s"(w_soff)
                         : "memory");
            w_soff += 8192;                                     // (padding steps run on into the next block / past the end: zeros or finite values, times zero pixels)
        }
        if (++w_k == a.n_steps) { w_k = 0; w_item += gridDim.x; w_base(); }
    };

    f32x4 acc[PF][4];
    const int lds_lane = frow * 64 + ((fq ^ ((frow >> 1) & 3)) << 4);   // fragment read: pixel frow of a piece, channel group fq
    constexpr int PD = 3;                                       // pixel fragments read ahead
    auto compute = [&](int slot_b, const half8 (&w)[4 * KC], auto &&mid) __attribute__((always_inline)) {
        int pb = slot_b + lds_lane;
        asm volatile("" : "+v"(pb));                            // one base register; every read is base + immediate
        constexpr int NR = KC * PF;                             // fragment reads of the step: r = q * PF + pf at q * SUB + pf * 1024
        half8 pq[PD + 1];
#pragma unroll
        for (int r = 0; r < PD && r < NR; r++) pq[r] = *(const half8 *)(smem + pb + (r / PF) * SUB + (r % PF) * 1024);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int q = r / PF, pf = r % PF;
            if (r + PD < NR) pq[(r + PD) % (PD + 1)] = *(const half8 *)(smem + pb + ((r + PD) / PF) * SUB + ((r + PD) % PF) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[pf][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[q * 4 + i], pq[r % (PD + 1)], acc[pf][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mid(r);                                             // address arithmetic + requests of later steps, issued under the matrix pipe's shadow
        }
    };

    // ---- epilogue: bias + residual + activation in the accumulator layout (a lane holds 4 consecutive couts of a pixel), fp16,
    // then one pixel fragment at a time through the wave's 2-KB staging block so that the results leave as 16 bytes per lane /
    // whole 128-byte rows per pixel.  No workgroup barrier: the block is private to the wave, LDS serves a wave's requests in order.
    const float *sTab = (const float *)(smem + a.tab_off);
    auto epilogue_body = [&](int item, auto act_tag, auto res_tag, auto border_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool RES = decltype(res_tag)::value, BORDER = decltype(border_tag)::value;
        int lo = lane;
        asm volatile("" : "+v"(lo));                            // opaque lane id: this block's per-lane arithmetic stays out of the step loop
        const int fr = lo & 15, q4 = lo >> 4;
        const int mt = fastdiv(item, a.d_nblk), nb = item - mt * a.n_nblk;
        const int co_w = nb * (NW * 64) + wave * 64;            // first cout of this wave
        // bias / slope rows are re-read from the LDS tables per pixel fragment (an opaque offset keeps the compiler from hoisting 32
        // registers of them across the fragment loop: the accumulators already take 128)
        const int cc0 = co_w + q4 * 4;
        char *st = sStage + wave * 2048;
        // staged row = pixel (128 B = 8 chunks of 8 couts), chunks rotated by the pixel so that neither side piles up on a bank
        const int st_w = fr * 128 + (q4 & 1) * 8;               // + ((2 i + (q4 >> 1) + fr) & 7) * 16
        const int rq = lo >> 3, rc = lo & 7;                    // read-back: pixel rq (+ 8), chunk rc
        const int p_r = mt * BM + fr;                           // residual: my pixel of fragment 0 (+ 16 pf)
        const int co_r = co_w + q4 * 4;                         //           my couts of fragment 0 (+ 16 i)
        u32x2 rr[2][4];
        auto load_res = [&](int pf, u32x2 (&r)[4]) __attribute__((always_inline)) {            // exactly 4 loads
            const int p = p_r + pf * 16;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int co = co_r + i * 16;
                r[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, (p < a.M && co < a.Cout_p) ? (unsigned)((p * a.Cout_p + co) * 2) : OOB, 0, 0);
            }
        };
        if (RES) load_res(0, rr[0]);
#pragma unroll
        for (int pf = 0; pf < PF; pf++) {
            if (RES && pf + 1 < PF) load_res(pf + 1, rr[(pf + 1) & 1]);
            int trow = 0;                                       // bias row of this fragment's pixel
            if (BORDER) {       // exact fold of a BatchNorm in front of the zero-padded conv: the bias row depends on the pixel's border class
                const int p = p_r + pf * 16;
                const int n = fastdiv(p, a.d_hw), r = p - n * (a.Ho * a.Wo);
                const int oy = fastdiv(r, a.d_w), ox = r - oy * a.Wo;
                trow = ((oy == 0 ? 0 : (oy == a.Ho - 1 ? 6 : 3)) + (ox == 0 ? 0 : (ox == a.Wo - 1 ? 2 : 1))) * a.Cout_p;
            }
            asm volatile("" : "+v"(trow));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c0 = cc0 + i * 16, cc = c0 < a.Cout_p ? c0 : 0;
                f32x4 v = acc[pf][i];
                if (a.ncls) v += *(const f32x4 *)(sTab + trow + cc);
                if (RES) {
                    const half4 h = __builtin_bit_cast(half4, rr[pf & 1][i]);
                    v += __builtin_convertvector(h, f32x4);
                }
                if (ACT == ACT_PRELU) {
                    const f32x4 sl = *(const f32x4 *)(sTab + a.ncls * a.Cout_p + cc + (trow & 0));
                    v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}) + sl * __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f});
                }
                half4 h = __builtin_convertvector(v, half4);
                if (ACT == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                *(half4 *)(st + st_w + (((2 * i + (q4 >> 1) + fr) & 7) << 4)) = h;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const int q = rq + h2 * 8;
                const u32x4 v = *(const u32x4 *)(st + q * 128 + (((rc + q) & 7) << 4));
                const int p = mt * BM + pf * 16 + q, co = co_w + rc * 8;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (p < a.M && co < a.Cout_p) ? (unsigned)((p * a.Cout_p + co) * 2) : OOB, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next fragment overwrites the block)
        }
    };
    auto epilogue = [&](int item) __attribute__((always_inline)) {
        using std::integral_constant;
        const bool res = a.res != nullptr, border = a.ncls == 9;
#define GW_EPI(A) \
        do { \
            if (res) { if (border) epilogue_body(item, integral_constant<int, A>{}, integral_constant<bool, true>{}, integral_constant<bool, true>{}); \
                       else epilogue_body(item, integral_constant<int, A>{}, integral_constant<bool, true>{}, integral_constant<bool, false>{}); } \
            else { if (border) epilogue_body(item, integral_constant<int, A>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{}); \
                   else epilogue_body(item, integral_constant<int, A>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{}); } \
        } while (0)
        if (a.act == ACT_PRELU) GW_EPI(ACT_PRELU);
        else if (a.act == ACT_RELU) GW_EPI(ACT_RELU);
        else GW_EPI(ACT_NONE);
#undef GW_EPI
    };

    // ---- prologue: the first D steps' pieces, the first step's weights, the tables -- one trip to memory
    item_pixels();
#pragma unroll
    for (int d = 0; d < D; d++) {
        issue_pixels(d * SLOT);                                 // (past the last step: out-of-range pixels, zeros into a slot nobody reads)
        fetch_next();
    }
    w_base();
#pragma unroll
    for (int d = 0; d < WD; d++) load_w(wS[d]);                 // weights of steps 0 .. WD - 1
    {
        float *tb = (float *)(smem + a.tab_off);
        const int nb = a.ncls * a.Cout_p;
        for (int i = tid; i < nb; i += NW * 64) tb[i] = a.bias[i];
        for (int i = tid; i < a.Cout_p; i += NW * 64) tb[nb + i] = a.act == ACT_PRELU ? a.slope[i] : 1.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        raw_barrier();
    }
#pragma unroll
    for (int d = 0; d < WD; d++)
#pragma unroll
        for (int i = 0; i < 4 * KC; i++) asm volatile("" : "+v"(wS[d][i]));

    // An item runs n_steps = a multiple of WD + 1 K-steps (the host pads with steps that fetch no pixels), so that step k of an item always
    // multiplies with register set k % (WD + 1) and ONE copy of the epilogue follows the step loop.
    // Operation order per wave and step s: [pieces of step s + D] [weights of step s + WD] -- both requested under the first matrix rows.
    // Younger than the pieces of step s (requested in step s - D): that step's weights and, for each of the D - 1 steps between, pieces +
    // weights.  Younger than the weights of step s (requested in step s - WD): pieces + weights of the WD - 1 steps between.  An epilogue's
    // E stores count on top while it is inside the window.
    constexpr int OPS_P = MAX_P * KC, OPS_W = 4 * KC;             // requests per wave and step
    constexpr int N_TOP = OPS_W + (D - 1) * (OPS_P + OPS_W), N_W = (WD - 1) * (OPS_P + OPS_W);
    static_assert(N_TOP + E <= 63 && N_W + E <= 63, "vmcnt is a 6-bit counter");
    int item = bid, e_age = 99, slot_b = 0, slot_in_b = (D % NS) * SLOT;
    auto step = [&](auto k_tag) __attribute__((always_inline)) {
        constexpr int K = decltype(k_tag)::value;               // step index modulo the register sets
        half8 (&w_cur)[4 * KC] = wS[K];
        half8 (&w_nxt)[4 * KC] = wS[(K + WD) % (WD + 1)];
        if (e_age <= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP + E) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_TOP) : "memory");
        raw_barrier();                                          // everybody's pieces of this step have landed; everybody is done with the slot of the step before
        if (e_age <= WD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_W + E) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_W) : "memory");
#pragma unroll
        for (int i = 0; i < 4 * KC; i++) asm volatile("" : "+v"(w_cur[i]));
        compute(slot_b, w_cur, [&](int r) __attribute__((always_inline)) {
            if (r == 0) { issue_pixels(slot_in_b); fetch_next(); }
            if (r == 2) load_w(w_nxt);
        });
        e_age++;
        slot_b = slot_b + SLOT == NS * SLOT ? 0 : slot_b + SLOT;
        slot_in_b = slot_in_b + SLOT == NS * SLOT ? 0 : slot_in_b + SLOT;
    };
    const int n_groups = a.n_steps / (WD + 1);
    for (int it = 0; it < my_items; it++) {
#pragma unroll
        for (int pf = 0; pf < PF; pf++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[pf][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < n_groups; g++) {
            step(std::integral_constant<int, 0>{});
            if constexpr (WD >= 1) step(std::integral_constant<int, 1>{});
            if constexpr (WD >= 2) step(std::integral_constant<int, 2>{});
            if constexpr (WD >= 3) step(std::integral_constant<int, 3>{});
            if constexpr (WD >= 4) step(std::integral_constant<int, 4>{});
            if constexpr (WD >= 5) step(std::integral_constant<int, 5>{});
        }
        if (!a.ablate) epilogue(item);
        else {
#pragma unroll
            for (int i = 0; i < E; i++) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rs_out, OOB, 0, 0);
        }
        e_age = 1;
        item += gridDim.x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the surplus requests target this wave's registers / LDS: drain before exit
#endif
}

template <int BM, int NW, int NS, int WD, int KC>
int gw_launch_t(fid_ctx *ctx, GWArgs &a) {
    FID_REQUIRE(a.n_chunks % KC == 0, "conv_gw: %d chunks in steps of %d", a.n_chunks, KC);
    a.n_steps = cdiv(a.taps * (a.n_chunks / KC), WD + 1) * (WD + 1);   // padded to whole groups of WD + 1 steps (padding steps fetch no pixels)
    a.n_nblk = cdiv(a.Cout_p, NW * 64);
    a.n_items = cdiv(a.M, BM) * a.n_nblk;
    a.d_nblk = fastdiv_make(a.n_nblk);
    a.tab_off = NS * BM * 64 * KC + NW * 2048;
    const int LDS = a.tab_off + (a.ncls + 1) * a.Cout_p * 4;
    FID_REQUIRE(LDS <= 160 * 1024, "conv_gw: %d bytes of LDS", LDS);
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_gw<BM, NW, NS, WD, KC>, (int)(LDS)));
    const int wg_per_cu = std::max(1, std::min(8 / NW, (160 * 1024) / LDS));      // 8 waves per CU by registers
    static const int wgpc_env = getenv("FID_GW_WGPC") ? atoi(getenv("FID_GW_WGPC")) : 0;
    const int grid = std::min(a.n_items, ctx->num_cus * (wgpc_env > 0 ? wgpc_env : wg_per_cu));
    hipLaunchKernelGGL((conv_gw<BM, NW, NS, WD, KC>), dim3(grid), dim3(NW * 64), LDS, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace

bool conv_gw_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_GW")) return false;
    return a.kh == a.kw && (a.kh == 1 || a.kh == 2 || a.kh == 3) && a.Cin_p % 32 == 0 && a.Cin_p >= 64 && a.Cout_p >= 96 && a.Cout_p % 32 == 0 &&
           a.w_rows == a.Cout_p && !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 && a.M >= 512 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo && a.res_Cp == a.Cout_p));
}

// bm = 64 | 128 output pixels per item, bn = 128 | 256 couts per item (2 | 4 waves)
int conv_gw_launch(fid_ctx *ctx, const ConvArgs &c, int bm, int bn) {
    FID_REQUIRE(c.w_alt, "conv_gw needs the fragment-order weights (repack kind 3)");
    FID_REQUIRE(conv_gw_applicable(c) && (bm == 64 || bm == 128) && (bn == 128 || bn == 256), "conv_gw: no %d x %d variant for this layer", bm, bn);
    GWArgs a{};
    a.in = c.in; a.w = c.w_alt; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.Ho = c.Ho; a.Wo = c.Wo; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.kh = c.kh; a.kw = c.kw; a.stride = c.stride; a.pad = c.pad; a.act = c.act;
    a.ncls = c.bias ? ((c.flags & CF_BORDER) ? 9 : 1) : 0;
    a.M = c.M;
    a.n_chunks = c.Cin_p / 32; a.taps = c.kh * c.kw; a.n_steps = a.taps * a.n_chunks;
    a.d_hw = fastdiv_make(c.Ho * c.Wo); a.d_w = fastdiv_make(c.Wo);
    a.in_bytes = c.in_bytes;
    const size_t ob = (size_t)c.M * c.Cout_p * 2;
    FID_REQUIRE(a.in_bytes <= OOB && ob <= OOB, "conv: tensor larger than 2 GiB");
    a.out_bytes = (unsigned)ob;
    a.w_bytes = (unsigned)repack_bytes(3, c.Cout_p, c.Cin_p, a.taps);
    static const int ablate = getenv("FID_GW_ABLATE") ? atoi(getenv("FID_GW_ABLATE")) : 0;
    a.ablate = ablate;
    // chunks per step: two with 64-pixel items when the channel count allows (the weight sets live in registers: 128-pixel items have
    // room for one chunk only -- with two, the compiler spills around the hand-counted loads)
    const int kc = (bm == 64 && a.n_chunks % 2 == 0) ? 2 : 1;
#define GW_GO(BMV, NWV) (kc == 2 ? gw_launch_t<BMV, NWV, 3, 1, (BMV == 64 ? 2 : 1)>(ctx, a) : gw_launch_t<BMV, NWV, 3, 1, 1>(ctx, a))
    if (bm == 128) return bn == 256 ? GW_GO(128, 4) : GW_GO(128, 2);
    return bn == 256 ? GW_GO(64, 4) : GW_GO(64, 2);
#undef GW_GO
}

}  // namespace fid
