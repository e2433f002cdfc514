// IResNet's first two convs as ONE launch (round 4):
//   uint8 crop --blob + conv3x3 (3 -> 64) + BN + PReLU--> x --[BN] conv3x3 (64 -> 64) + BN + PReLU--> out      (arcface_r50: stem, layer1.0.conv1)
// reference models/arcface.py:44-51 (blobFromImage + the first graph nodes inside session.run).
//
// Why: unfused, the stem writes x (112 x 112 x 64 fp16 = 1.6 MB per face: 803 MB per 500 faces) and layer1.0.conv1 reads it back -- 188 us at
// 4.3 TB/s for the write alone at batch 500, the read keeps conv1 (1 039 TFLOP/s) at the CU's ingest limit.  The stem is 5 % of conv1's matrix
// work, so recomputing it on the 1-pixel halo costs nothing: here x only ever exists in LDS.  The block's shortcut conv (1x1 / stride 2 on x)
// needs x at the even pixels only: those leave as a COMPACT second output x_even [B, H/2, W/2, 64] (a quarter of x), read with stride 1 by
// the shortcut op or by the conv that absorbs it (lower.py).
//
//   item   = a 16 x 16 tile of `out` x all 64 couts; x on the 18 x 18 region around it (outside the image: 0 = conv1's zero padding), the
//            uint8 patch on 20 x 20 pixels (dword window of a row: 2 bytes of lead-in, so that it starts on a dword of the frame row)
//   waves  = 4 (two workgroups per CU run out of phase: one's gather / epilogue phases under the other's matrix phase, as conv_bb32 found):
//            stage S: wave w takes pixel fragments w, w + 4, .. of the 21 flattened ones, ALL four cout fragments.  K = 12 tap slots x 4 halfs
//                     (B, G, R, 0) in three v_mfma_f32_16x16x16_f16 steps: a lane's operand of a step is ONE pixel of the patch = one aligned
//                     8-byte LDS read (the first form gathered K = 27 as eight 2-byte reads and spent 1 600 VALU / LDS instructions per wave and
//                     item on it and on per-item index arithmetic: 237 of 615 us); every per-fragment address is computed once per kernel;
//                     bias rides in as the accumulator's initial value; PReLU -> x in conv3x3_wr's patch layout
//            stage C: wave w = couts 16 w .. 16 w + 15 of conv1 for all 16 rows, its 2 x 9 weight fragments resident in registers (repack kind 2),
//                     row-sharing tap order; bias by border class (9 rows: the exact fold of the BatchNorm in front of the zero-padded conv)
//                     + PReLU -> staging -> 16-byte row stores issued at the start of the NEXT item
//   LDS    = input patch 2.8 KB + x 40.5 KB + staging 32 KB + tables 3 KB = 78.3 KB (two workgroups: 156.5 of 160 KB).  Three barriers per item.  Only full vmcnt(0) drains.
#include <type_traits>

#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TO = 16, PW = 18, NPIX = PW * PW;                  // output tile edge, x region edge
constexpr int NF = (NPIX + 15) / 16;                             // 21 flattened pixel fragments of the x region
constexpr int P_BYTES = NPIX * 64;                                // one 32-channel chunk of x: 20.25 KB (no LDS-DMA here: no 1-KB rounding)
constexpr int IPR = 20, IPA = IPR + 2;                           // input patch rows (= pixel columns); rows allocated (two zero rows behind: taps 9 .. 11 of the K axis read there, weight 0)
constexpr int NW = 4, NT = NW * 64;
constexpr int ROWB = 128, CPX = 8;                               // bytes / 16-byte chunks of a staged output pixel
constexpr int ST_I = TO * 16 * CPX / NT;                         // 8 stores per thread and item
constexpr int IN_BYTES = (IPA * IPR * 8 + 255) / 256 * 256;       // the patch as 4 halfs per pixel (B, G, R, 0): every tap of a pixel is ONE aligned 8-byte LDS read
constexpr int STG_BYTES = TO * 16 * ROWB;
constexpr int TAB_FLOATS = 64 + 64 + 9 * 64 + 64;                // stem bias, stem slopes, conv1 bias rows, conv1 slopes
constexpr int OFF_IN = 0, OFF_X = IN_BYTES, OFF_STG = OFF_X + 2 * P_BYTES, OFF_TAB = OFF_STG + STG_BYTES, LDS_BYTES = OFF_TAB + TAB_FLOATS * 4, OFF_DUMP = OFF_STG;   // (OFF_DUMP: where the lanes of a partial fragment store, 8 bytes each per chunk, so that no branch surrounds the stores -- the staging area, free during stage S)
static_assert(2 * LDS_BYTES + 2048 <= 160 * 1024, "two workgroups per CU, with room for the allocation granularity");
constexpr int NIP = IPR * IPR, PPT = (NIP + NT - 1) / NT;        // 400 pixels of a patch, 2 per thread

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
// PReLU on four sums in TWO instructions per value instead of three (ReLU = slope 0): m = max(v, 0); t = v - m is min(v, 0) exactly; fma(s, t, m)
// has one zero addend or a zero product, so it rounds like the separate multiply and add of epilogue.h (identical results)
// (max(v, 0) as v_med3_f32(v, 0, +inf): a plain max makes the compiler canonicalise the MFMA result first -- a second v_max per value.  No inline
// asm here: an asm statement that reads an MFMA result directly is outside the compiler's hazard tracking and returned garbage when tried)
__device__ __forceinline__ f32x4 prelu4(f32x4 v, f32x4 s) {
    const float inf = __builtin_inff();
    const f32x4 m = f32x4{__builtin_amdgcn_fmed3f(v[0], 0.f, inf), __builtin_amdgcn_fmed3f(v[1], 0.f, inf), __builtin_amdgcn_fmed3f(v[2], 0.f, inf),
                          __builtin_amdgcn_fmed3f(v[3], 0.f, inf)};
    return __builtin_elementwise_fma(s, v - m, m);
}

struct SBArgs {
    const uint8_t *img;       // [B, H, W, 3] BGR
    const float *w0;          // stem: fp32 [64][27] (k = (dy*3 + dx)*3 + c_bgr; blob scale / 2, channel swap and BN folded by lower.py)
    const float *b0, *s0;     // stem bias, PReLU slopes (act0 == ACT_PRELU) or NULL
    const void *w1;           // conv1: repack kind 2 image of the 64 x 9 x 64 filter bank
    const float *b1, *s1;     // conv1 bias rows fp32 [ncls1][64], PReLU slopes or NULL
    void *out;                // [B, H, W, 64] fp16
    void *xe;                 // x at the even pixels [B, He, We, 64] fp16, or NULL
    int H, W, He, We, act0, act1, ncls1;
    int tiles_x, tiles_per_img, n_tiles;
    FastDiv d_tpi, d_tx;
    unsigned out_bytes, xe_bytes;
    int stagger;              // start delay of the second half of the grid in units of 64 cycles (the co-resident workgroups under round-robin dispatch; speed only)
    int ablate;               // FID_SB_ABLATE timing experiments (wrong results): 1 no stage S, 2 no stage C, 4 no stores, 8 no even-pixel copy, 16 no input fetch
};

__global__ void __launch_bounds__(NT, 2) ir_stem_block(const SBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    if (my_items == 0) return;

    auto decode_tile = [&](int item, int &n, int &ty, int &tx) __attribute__((always_inline)) {
        n = fastdiv(item, a.d_tpi);
        const int r = item - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);
    const auto rs_xe = __builtin_amdgcn_make_buffer_rsrc((void *)(a.xe ? a.xe : a.out), 0, a.xe ? a.xe_bytes : a.out_bytes, 0x00020000);

    // ---- tables -> LDS; stem weights (this lane's K group of all four cout fragments) and conv1's 18 fragments -> registers
    float *tb = (float *)(smem + OFF_TAB);                       // [0, 64) b0, [64, 128) s0, [128, 704) b1 rows, [704, 768) s1
    for (int i = tid; i < 64; i += NT) {
        tb[i] = a.b0 ? a.b0[i] : 0.f;
        tb[64 + i] = (a.act0 == ACT_PRELU && a.s0) ? a.s0[i] : 0.f;      // (ReLU = PReLU with slope 0)
        tb[704 + i] = (a.act1 == ACT_PRELU && a.s1) ? a.s1[i] : 0.f;
    }
    for (int i = tid; i < a.ncls1 * 64; i += NT) tb[128 + i] = a.b1[i];
    for (int i = tid; i < 2 * IPR; i += NT) *(unsigned long long *)(smem + OFF_IN + (IPR * IPR + i) * 8) = 0ull;      // the two zero rows behind the patch
    half4 wf[4][3];                                             // stem weights: A fragment of step s = tap slot 4 s + fq, halfs = channels B, G, R, 0
#pragma unroll
    for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int st = 0; st < 3; st++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int g = st * 4 + fq;
                wf[ni][st][j] = (g < 9 && j < 3) ? (_Float16)a.w0[(ni * 16 + frow) * 27 + g * 3 + j] : (_Float16)0.f;
            }
    half8 w1[18];                                               // [chunk * 9 + dy * 3 + dx] of couts 16 wave .. 16 wave + 15
    {
        const char *p1 = (const char *)a.w1 + wave * 9216 + lane * 16;
#pragma unroll
        for (int ck = 0; ck < 2; ck++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int dy = 0; dy < 3; dy++) w1[ck * 9 + dy * 3 + dx] = *(const half8 *)(p1 + ck * (8 * 9216) + dx * 3072 + dy * 1024);
    }
    // ---- stage S constants of this lane (item-independent): per fragment the pixel's byte offset in the input patch, its 8-byte slot in an x
    // chunk with the swizzle term in the low bits, and its region coordinates; per step the tap's offset
    constexpr int MF = (NF + NW - 1) / NW;                      // fragments per wave (the last one of waves 1 .. 3 does not exist)
    int f_in[MF], f_x[MF], f_yx[MF];
#pragma unroll
    for (int i = 0; i < MF; i++) {
        const int fi = wave + NW * i, qd = fi * 16 + frow, q = (fi < NF && qd < NPIX) ? qd : 0;
        const int py = q / PW, px = q - py * PW;
        f_in[i] = (py * IPR + px) * 8;
        f_x[i] = (fi < NF && qd < NPIX) ? (q * 64 + (fq & 1) * 8) | swz64(q) : (OFF_DUMP - OFF_X + lane * 8);      // (a lane without a pixel: its own 8 bytes of the dump area, swizzle term 0)
        f_yx[i] = py | (px << 8);
    }
    int t_off[3];
#pragma unroll
    for (int st = 0; st < 3; st++) {
        const int g = st * 4 + fq;
        t_off[st] = ((g / 3) * IPR + g % 3) * 8;                 // (slots 9 .. 11: a row further down, weight 0; the rows behind the patch are zeros)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w1[i]));

    // ---- input patch of an item: patch pixel (r, c) = image pixel (16 ty - 2 + r, 16 tx - 2 + c); thread t owns pixels t and t + 256: their
    // three bytes are fetched an item ahead and committed as 4 halfs (2 p - 255: exact integers; outside the frame 0 = the blob's zero padding)
    int p_rc[PPT];
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int p = tid + NT * i, r = p / IPR;
        p_rc[i] = p < NIP ? (r | ((p - r * IPR) << 8)) : -1;
    }
    unsigned pre[PPT][3];
    unsigned pre_ok = 0;
    auto prefetch = [&](int item) __attribute__((always_inline)) {
        int n, ty, tx;
        decode_tile(item, n, ty, tx);
        const uint8_t *base = a.img + (size_t)n * a.H * a.W * 3;
        pre_ok = 0;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int iy = ty * TO - 2 + (p_rc[i] & 255), ix = tx * TO - 2 + (p_rc[i] >> 8);
            const bool in = p_rc[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a.ablate & 16);
            const uint8_t *pp = base + ((size_t)iy * a.W + ix) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) pre[i][c] = in ? pp[c] : 0u;
            pre_ok |= in ? (1u << i) : 0u;
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            if (p_rc[i] < 0) continue;
            half4 h = half4{0, 0, 0, 0};
            if ((pre_ok >> i) & 1u) {
#pragma unroll
                for (int c = 0; c < 3; c++) h[c] = (_Float16)(float)(2 * (int)pre[i][c] - 255);
            }
            *(half4 *)(smem + OFF_IN + (tid + NT * i) * 8) = h;
        }
    };

    int pbase[2][4];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int c = 0; c < 4; c++) pbase[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c) & 3)) << 4);

    f32x4 acc[TO];
#ifndef SB_PD
#define SB_PD 2
#endif
    constexpr int PD = SB_PD;                                   // fragment rows read ahead (conv_bb.hip)
    auto conv_phase = [&](int base_off) __attribute__((always_inline)) {
        constexpr int PH = TO + 2, NQ = 6 * PH;
        int pb[2][4];
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = pbase[par][c] + base_off;
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int q) __attribute__((always_inline)) {     // q = (chunk * 3 + dx) * PH + fragment row
            const int pass = q / PH, r = q - pass * PH, ck = pass / 3, dx = pass - ck * 3;
            const int K = r * PW + dx;
            pq[q % (PD + 1)] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + (K * 64 + ck * P_BYTES)));
        };
#pragma unroll
        for (int q = 0; q < PD; q++) load_p(q);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int pass = q / PH, r = q - pass * PH, ck = pass / 3, dx = pass - ck * 3;
            if (q + PD < NQ) load_p(q + PD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const int mi = r - dy;
                if (mi < 0 || mi >= TO) continue;
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[ck * 9 + dy * 3 + dx], pq[q % (PD + 1)], acc[mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // write-out of the tile staged by the item before (pn < 0: none): 16-byte slot g = i * 256 + thread = (pixel g / 8, chunk g % 8); 32 pixels = 2 tile rows per round
    auto write_out = [&](int pn, int pty, int ptx) __attribute__((always_inline)) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int q0 = t2 >> 3, c = t2 & 7;
        const int pr0 = q0 >> 4, pc = q0 & 15;
        const int oy0 = pty * TO, ox = ptx * TO + pc;
        const bool okc = pn >= 0 && ox < a.W && !(a.ablate & 4);
        const char *lsrc = smem + OFF_STG + q0 * ROWB + (((c + pc) % CPX) << 4);
        const unsigned g0 = (unsigned)((((pn * a.H + oy0 + pr0) * a.W + ox) * 64 + c * 8) * 2);
        const unsigned rstride = (unsigned)(a.W * 64 * 2);
#pragma unroll
        for (int i = 0; i < ST_I; i++) {
            const bool ok = okc && oy0 + 2 * i + pr0 < a.H;
            const u32x4 v = *(const u32x4 *)(lsrc + i * (32 * ROWB));
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, ok ? g0 + (unsigned)(2 * i) * rstride : OOB, 0, 0);
        }
    };

    int item = bid, pn = -1, pty = 0, ptx = 0;
    prefetch(item);
    // Two co-resident workgroups that start together stay in lock-step (same program, items of equal cost): both in the matrix phase C (sharing
    // the pipes), then both in the VALU / LDS phases (pipes idle).  The second half of the grid starts about half an item late.
    if (a.stagger > 0 && (int)blockIdx.x >= (int)gridDim.x / 2)
        for (int i = 0; i < a.stagger; i += 100) __builtin_amdgcn_s_sleep(100);
    for (int it = 0; it < my_items; it++, item += gridDim.x) {
        int n, ty, tx;
        decode_tile(item, n, ty, tx);
        raw_barrier();                                          // B0: everyone is done with the x region of the item before; its tile is staged
        write_out(pn, pty, ptx);
        commit();
        if (it + 1 < my_items) prefetch(item + gridDim.x);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // B1: the input patch is in LDS

        // ================= S: the stem conv on the 18 x 18 region, K = 27 in one MFMA step =================
        if (!(a.ablate & 1)) {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int q4 = lo >> 4;
            f32x4 bias0[4], sl0[4];
#pragma unroll
            for (int ni = 0; ni < 4; ni++) {
                bias0[ni] = *(const f32x4 *)(tb + ni * 16 + q4 * 4);
                sl0[ni] = *(const f32x4 *)(tb + 64 + ni * 16 + q4 * 4);
            }
            const bool interior = ty * TO >= 1 && tx * TO >= 1 && ty * TO + PW - 1 <= a.H && tx * TO + PW - 1 <= a.W;   // the whole region lies inside the image (wave-uniform)
            constexpr int FB = 3;                               // fragments whose operands are read together (one LDS round trip)
#pragma unroll
            for (int i0 = 0; i0 < MF; i0 += FB) {
                half4 pv[FB][3];
#pragma unroll
                for (int i = 0; i < FB; i++)
#pragma unroll
                    for (int st = 0; st < 3; st++) pv[i][st] = *(const half4 *)(smem + OFF_IN + f_in[i0 + i] + t_off[st]);
#pragma unroll
                for (int i = 0; i < FB; i++) {
                    if (wave + NW * (i0 + i) >= NF) continue;   // (wave-uniform)
                    const int fx = f_x[i0 + i];
                    bool inside = true;
                    if (!interior) {
                        const int py = f_yx[i0 + i] & 255, px = f_yx[i0 + i] >> 8;
                        inside = (unsigned)(ty * TO - 1 + py) < (unsigned)a.H && (unsigned)(tx * TO - 1 + px) < (unsigned)a.W;
                    }
                    const int sw = fx & 3;
                    char *xp = smem + OFF_X + (fx & ~7);
#pragma unroll
                    for (int ni = 0; ni < 4; ni++) {
                        f32x4 v = bias0[ni];
#pragma unroll
                        for (int st = 0; st < 3; st++) v = __builtin_amdgcn_mfma_f32_16x16x16f16(wf[ni][st], pv[i][st], v, 0, 0, 0);
                        half4 h = __builtin_convertvector(prelu4(v, sl0[ni]), half4);
                        if (!inside) h = half4{0, 0, 0, 0};     // outside the image: conv1's zero padding, NOT the stem evaluated there
                        *(half4 *)(xp + ((fx >= 2 * P_BYTES) ? (ni >> 1) * 512 : (ni >> 1) * P_BYTES + ((((ni & 1) * 2 + (q4 >> 1)) ^ sw) << 4))) = h;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // B2: x is complete

        // x at the even image pixels of the tile -> the compact second output: region pixel (2 ey + 1, 2 ex + 1) = image (16 ty + 2 ey, 16 tx + 2 ex)
        if (a.xe && !(a.ablate & 8)) {
            int t2 = tid;
            asm volatile("" : "+v"(t2));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int s = t2 + NT * j, e = s >> 3, c = s & 7, ey = e >> 3, ex = e & 7;
                const int lin = (2 * ey + 1) * PW + 2 * ex + 1;
                const u32x4 v = *(const u32x4 *)(smem + OFF_X + (c >> 2) * P_BYTES + lin * 64 + (((c & 3) ^ swz64(lin)) << 4));
                const int gy = ty * (TO / 2) + ey, gx = tx * (TO / 2) + ex;
                const bool ok = gy < a.He && gx < a.We;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_xe, ok ? (unsigned)((((n * a.He + gy) * a.We + gx) * 64 + c * 8) * 2) : OOB, 0, 0);
            }
        }

        // ================= C: conv1 on the 16 x 16 tile =================
        // the accumulators start as the rows' bias (by border class: 9 rows = the exact fold of the BatchNorm in front of the zero-padded conv)
        {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const int gx = tx * TO + fr, gy0 = ty * TO;
            const int xc = a.ncls1 == 9 ? (gx == 0 ? 0 : (gx == a.W - 1 ? 2 : 1)) : 0;
            const float *bt = tb + 128 + xc * 64 + wave * 16 + q4 * 4;
#pragma unroll
            for (int i = 0; i < TO; i++) {
                const int gy = gy0 + i, yc = a.ncls1 == 9 ? (gy == 0 ? 0 : (gy == a.H - 1 ? 2 : 1)) : 0;
                acc[i] = *(const f32x4 *)(bt + yc * 192);
            }
        }
        if (!(a.ablate & 2)) conv_phase(OFF_X);
        {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const f32x4 sl = *(const f32x4 *)(tb + 704 + wave * 16 + q4 * 4);
            char *sp = smem + OFF_STG + fr * ROWB + (((wave * 2 + (q4 >> 1) + fr) % CPX) << 4) + (q4 & 1) * 8;   // chunk rotated by the pixel column
#pragma unroll
            for (int i = 0; i < TO; i++) *(half4 *)(sp + i * (16 * ROWB)) = __builtin_convertvector(prelu4(acc[i], sl), half4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pn = n; pty = ty; ptx = tx;
    }
    raw_barrier();                                              // the last tile is staged
    write_out(pn, pty, ptx);
}

}  // namespace

// img uint8 [B, H, W, 3] -> out fp16 [B, H, W, 64] (+ xe fp16 [B, (H+1)/2, (W+1)/2, 64] or NULL); act0 / act1: ACT_RELU | ACT_PRELU; ncls1: 1 | 9
bool stem_block_applicable(int H, int W) { return H >= 3 && W >= 4 && W % 4 == 0; }
int stem_block_launch(fid_ctx *ctx, const uint8_t *img, int B, int H, int W, const float *w0, const float *b0, const float *s0, int act0, const void *w1,
                      const float *b1, int ncls1, const float *s1, int act1, void *out, void *xe) {
    FID_REQUIRE(img && w0 && w1 && b1 && out && B > 0 && stem_block_applicable(H, W), "stem_block: bad arguments (%d x %d)", H, W);
    FID_REQUIRE((act0 == ACT_RELU || (act0 == ACT_PRELU && s0)) && (act1 == ACT_RELU || (act1 == ACT_PRELU && s1)) && (ncls1 == 1 || ncls1 == 9),
                "stem_block: activations %d / %d, %d bias rows", act0, act1, ncls1);
    SBArgs a{};
    a.img = img; a.w0 = w0; a.b0 = b0; a.s0 = s0; a.w1 = w1; a.b1 = b1; a.s1 = s1; a.out = out; a.xe = xe;
    a.H = H; a.W = W; a.He = (H + 1) / 2; a.We = (W + 1) / 2; a.act0 = act0; a.act1 = act1; a.ncls1 = ncls1;
    a.tiles_x = cdiv(W, TO);
    a.tiles_per_img = a.tiles_x * cdiv(H, TO);
    a.n_tiles = B * a.tiles_per_img;
    a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    const size_t ob = (size_t)B * H * W * 64 * 2, xb = (size_t)B * a.He * a.We * 64 * 2;
    FID_REQUIRE(ob <= OOB, "stem_block: tensor larger than 2 GiB");
    a.out_bytes = (unsigned)ob; a.xe_bytes = (unsigned)xb;
    static const int ablate = getenv("FID_SB_ABLATE") ? atoi(getenv("FID_SB_ABLATE")) : 0;
    a.ablate = ablate;
    static const int stagger = getenv("FID_SB_STAGGER") ? atoi(getenv("FID_SB_STAGGER")) : 0;
    a.stagger = stagger;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)ir_stem_block, LDS_BYTES));
    static const bool log_occ = getenv("FID_TUNE_LOG") != nullptr;
    if (log_occ) {
        static bool once = false;
        if (!once) {
            once = true;
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)ir_stem_block, NT, LDS_BYTES);
            fprintf(stderr, "[stem_block] %d bytes of LDS, %d workgroups per CU resident\n", LDS_BYTES, nb);
        }
    }
    hipLaunchKernelGGL(ir_stem_block, dim3(std::min(a.n_tiles, 2 * ctx->num_cus)), dim3(NT), LDS_BYTES, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
