// Fused residual BasicBlock on 64 channels:  out = act2( conv2( relu( conv1(x) + b1 ) ) + b2 + x ),  both convs 3x3 / stride 1 / pad 1, 64 -> 64
// (SCRFD-10G layer1: three such blocks at 160x160 x 64 frames, reference models/scrfd.py:83 runs them inside session.run).
//
// Why: unfused, each of the two convs of such a block moves 420 / 630 MB per 64 frames (input + output, + the residual for the second)
// through a CU-side memory path that takes ~9-10 B/clk/CU (profiles/r02: 105 / 138 us = 4.0 / 4.6 TB/s with the matrix pipes 46 % busy):
// the layers are bound by bytes per CU, not by HBM and not by arithmetic.  Here the intermediate map never leaves the CU and the
// residual is the centre of the input patch that is in LDS anyway: per 14x14 output tile 41.5 KB come in (18x18 haloed patch) and
// 25 KB go out, against 2 x 41.5 + 32 (residual) + 2 x 32 KB per 16x16 tile for the two launches -- 2.1x fewer bytes per output pixel.
//
//   item  = one 14x14 OUTPUT tile x all 64 couts.  It needs conv1's result on the 16x16 region around it (one 16-pixel MFMA fragment
//           per row, no partial fragments) and x on the 18x18 region around that: exactly conv3x3_wr's haloed patch.  The price is
//           recompute: conv1 runs on 256 pixels and conv2 on 14 rows x 16 lanes per 196 stored pixels (1.22x the MACs of the pair).
//   waves = 8: wave = (cout fragment cw = 0..3: couts 16cw..16cw+15) x (row group rg = 0..1: rows 8rg..8rg+7 of the intermediate tile,
//           rows 7rg..7rg+6 of the output tile).  BOTH filter banks of a wave's 16 couts (2 convs x 2 chunks x 9 taps x 4 VGPRs = 144
//           registers) stay in registers for the kernel's lifetime (repack.hip kind 2 layout, packed by lower.py).
//   LDS   = two x-patch buffers (this item / the next one, fetched by LDS-DMA a whole item ahead) + the intermediate tile in patch
//           layout (16x16 pixels x 2 chunks of 32 channels, outside-the-image pixels written as 0: they are conv2's zero padding) + the
//           staging area of the finished tile (16-byte row stores, whole pixel rows).
//   order = A: conv1 (2 chunks x 3 tap columns, row-sharing tap order of conv_chunked.hip) -> bias + ReLU -> LDS;  B: conv2 from LDS ->
//           + bias + residual (read from the x patch) -> activation -> staging.  The staged tile's stores are issued at the start of the
//           NEXT item and the next patch's LDS-DMA pieces one per tap column of conv1, so neither sits between two matrix phases with
//           every pipe idle (measured with all of it at the item boundaries: 46 + 56 us of a 272 us block exposed).  Two barriers per item.
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TO = 14;                                           // output tile edge
constexpr int PW = 18, MW = 16;                                  // x patch / intermediate tile width in pixels
constexpr int NPIX = PW * PW;                                    // 324
constexpr int P_BLKS = (NPIX * 64 + 1023) / 1024, P_BYTES = P_BLKS * 1024;     // one 32-channel chunk of the patch: 21 KB
constexpr int X_ITEM = 2 * P_BYTES;                              // both chunks
constexpr int N_PIECES = 2 * P_BLKS;                             // 42 one-KB pieces per item
constexpr int NWT = 8;
constexpr int MAX_P = (N_PIECES + NWT - 1) / NWT;                // 6 per wave
constexpr int MID_CH = MW * MW * 64, MID_BYTES = 2 * MID_CH;     // 32 KB
constexpr int ROWB = 128, CPX = 8;                               // bytes / 16-byte chunks of a staged output pixel (64 couts)
constexpr int ST_SLOTS = TO * 16 * CPX;                          // 16-byte slots of a staged tile (14 rows x 16 pixel columns)
constexpr int ST_I = (ST_SLOTS + NWT * 64 - 1) / (NWT * 64);     // 4 stores per thread and item
constexpr int STG_BYTES = TO * 16 * ROWB;                        // staged output tile (its 16-byte row stores are issued during the NEXT item)
constexpr int TAB_BYTES = 9 * 64 * 4;                            // conv1's bias rows (IResNet form: 9 border classes)
constexpr int OFF_X = 0, OFF_SPARE = 2 * X_ITEM, OFF_MID = OFF_SPARE + 1024, OFF_STG = OFF_MID + MID_BYTES + 512, OFF_TAB = OFF_STG + STG_BYTES, LDS_BYTES = OFF_TAB + TAB_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct BBArgs {
    const void *in;
    const void *w1, *w2;      // repack kind 2 images of the two filter banks (64 couts padded to a block of 128)
    const float *b1, *b2;     // fp32 [ncls1][64] / [64]
    const float *s1;          // fp32 [64] PReLU slopes of conv1 (act1 == ACT_PRELU)
    int act1, ncls1;          // conv1: ACT_RELU | ACT_PRELU; 1 bias row, or 9 border-class rows (a BatchNorm folded in front of the zero-padded conv)
    void *out;
    int H, W;
    int act2;                 // activation after the residual add (ACT_RELU | ACT_NONE)
    int tiles_x, tiles_per_img, n_tiles;
    FastDiv d_tpi, d_tx;
    unsigned io_bytes;        // bytes of the input (= output) tensor
    int rev;                  // ConvArgs::rev
    int ablate;               // FID_BB_ABLATE timing experiments (wrong results): 1 no conv1, 2 no conv2, 4 no stores, 8 no patch fetch; conv_bb2 also 16 no conv1 epilogue, 32 no conv2 epilogue, 64 two barriers fewer
    int stagger;              // FID_BB_STAGGER (conv_bb2): the second half of the grid starts this many x 64 sleep ticks late
};

// IR = false: SCRFD's block (plain bias + ReLU after conv1; the bias lives in registers).  IR = true: any conv1 epilogue -- bias rows by border
// class from an LDS table, ReLU or PReLU (IResNet: BN - conv - BN - PReLU - conv - BN, + input).
template <bool IR>
__global__ void __launch_bounds__(NWT * 64, 2) conv_bb(const BBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, rg = wave >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    if (my_items == 0) return;

    auto decode_tile = [&](int item, int &n, int &ty, int &tx) {
        if (a.rev) item = a.n_tiles - 1 - item;
        n = fastdiv(item, a.d_tpi);
        const int r = item - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.io_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.io_bytes, 0x00020000);

    // ---- my pieces of an item's patch: piece j = wave + 8k -> chunk j / 21, pixels 16 (j % 21) .. +15, four lanes per pixel
    int p_pk[MAX_P];                                            // py | px << 8 | channel offset (halfs) << 16; py = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P; k++) {
        const int j = wave + NWT * k;
        const int ch = j / P_BLKS, blk = j - ch * P_BLKS;
        const int row = blk * 16 + (lane >> 2);
        int py = row / PW;
        const int px = row - py * PW;
        if (row >= NPIX || j >= N_PIECES) py = 255;
        p_pk[k] = py | (px << 8) | (((((lane & 3) ^ swz64(row)) * 8) + ch * 32) << 16);
    }
    // piece k of the patch of tile (n, ty, tx) into buffer `buf` (live = false: a piece of zeros: the operation count per item stays exact)
    auto issue_piece = [&](int k, int n, int ty, int tx, bool live, int buf) {
        const int y0 = ty * TO - 2, x0 = tx * TO - 2;
        const int j = wave + NWT * k;
        int pk = p_pk[k];
        asm volatile("" : "+v"(pk));                            // opaque: unpack at the use
        const int py = pk & 255, iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
        const bool in = live && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a.ablate & 8);
        const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * 64 + (pk >> 16)) * 2) : OOB;
        char *d = j < N_PIECES ? smem + OFF_X + buf * X_ITEM + j * 1024 : smem + OFF_SPARE;   // surplus piece: zeros into the spare KB
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)d, 16, vo, 0, 0, 0);
    };
    {
        int n, ty, tx;
        decode_tile(bid, n, ty, tx);
#pragma unroll
        for (int k = 0; k < MAX_P; k++) issue_piece(k, n, ty, tx, true, 0);
    }

    // ---- both filter banks of my 16 couts: kind 2 = [chunk][cout fragment 0..7][dx][dy][lane] x 16 B
    half8 w1[18], w2[18];                                       // [chunk * 9 + dy * 3 + dx]
    {
        const char *p1 = (const char *)a.w1 + cw * 9216 + lane * 16, *p2 = (const char *)a.w2 + cw * 9216 + lane * 16;
#pragma unroll
        for (int ck = 0; ck < 2; ck++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    w1[ck * 9 + dy * 3 + dx] = *(const half8 *)(p1 + ck * (8 * 9216) + dx * 3072 + dy * 1024);
                    w2[ck * 9 + dy * 3 + dx] = *(const half8 *)(p2 + ck * (8 * 9216) + dx * 3072 + dy * 1024);
                }
    }
    const f32x4 bias1 = *(const f32x4 *)(a.b1 + cw * 16 + fq * 4), bias2 = *(const f32x4 *)(a.b2 + cw * 16 + fq * 4);
    f32x4 slope1 = f32x4{0.f, 0.f, 0.f, 0.f};                   // (ReLU = PReLU with slope 0)
    if constexpr (IR) {
        if (a.act1 == ACT_PRELU) slope1 = *(const f32x4 *)(a.s1 + cw * 16 + fq * 4);
        float *tb = (float *)(smem + OFF_TAB);
        for (int i = tid; i < a.ncls1 * 64; i += NWT * 64) tb[i] = a.b1[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w1[i]), "+v"(w2[i]));

    // ---- pixel fragment addresses (conv3x3_wr's scheme): lin = K + frow with K a compile-time constant per read; the swizzled 16-byte
    // group of the lane is one of four, selected by K & 1 and (K >> 1) & 3.  The row groups start at patch rows 8 rg (x 18 pixels) and 7 rg
    // (x 16 pixels): both multiples of 8 pixels, so they only add to the base.
    int pbase[2][4];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int c = 0; c < 4; c++) pbase[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c) & 3)) << 4);

    f32x4 acc[8];
#ifndef BB_PD
#define BB_PD 2
#endif
    constexpr int PD = BB_PD;                                   // fragment rows read ahead
    // one conv: ROWS output rows of this wave from ROWS + 2 fragment rows x 3 tap columns x 2 chunks of 32 channels.  The six (chunk, column)
    // passes are ONE flattened sequence of fragment rows, read PD rows ahead ACROSS the pass boundaries: a pass that starts its own
    // read-ahead waits an LDS round trip (~150 cycles) with the matrix pipe idle -- six times per conv, twelve per item.
    auto conv_phase = [&](int base_off, auto rows_tag, auto pw_tag, auto cs_tag, const half8 *wv, auto &&col_hook) {
        constexpr int ROWS = decltype(rows_tag)::value, PWV = decltype(pw_tag)::value, CS = decltype(cs_tag)::value, PH = ROWS + 2, NQ = 6 * PH;
        int pb[2][4];
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = pbase[par][c] + base_off;
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int q) {                              // q = (chunk * 3 + dx) * PH + fragment row
            const int pass = q / PH, r = q - pass * PH, ck = pass / 3, dx = pass - ck * 3;
            const int K = r * PWV + dx;
            pq[q % (PD + 1)] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + (K * 64 + ck * CS)));
        };
#pragma unroll
        for (int q = 0; q < PD; q++) load_p(q);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int pass = q / PH, r = q - pass * PH, ck = pass / 3, dx = pass - ck * 3;
            if (r == 0) col_hook(pass);
            if (q + PD < NQ) load_p(q + PD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const int mi = r - dy;
                if (mi < 0 || mi >= ROWS) continue;
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ck * 9 + dy * 3 + dx], pq[q % (PD + 1)], acc[mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;

    // write-out of the tile staged by the item before (tile coordinates pn, pty, ptx; pn < 0: none -- the stores still issue, out of range,
    // so that every item has exactly ST_I of them): 16-byte slot g = i * 512 + thread = (pixel g / 8, chunk g % 8); 64 pixels = 4 tile rows per round
    auto write_out = [&](int pn, int pty, int ptx) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int q0 = t2 >> 3, c = t2 & 7;
        const int pr0 = q0 >> 4, pc = q0 & 15;
        const int oy0 = pty * TO, ox = ptx * TO + pc;
        const bool okc = pn >= 0 && pc < TO && ox < a.W && !(a.ablate & 4);
        const char *lsrc = smem + OFF_STG + q0 * ROWB + (((c + pc) % CPX) << 4);
        const unsigned g0 = (unsigned)((((pn * a.H + oy0 + pr0) * a.W + ox) * 64 + c * 8) * 2);
        const unsigned rstride = (unsigned)(a.W * 64 * 2);
#pragma unroll
        for (int i = 0; i < ST_I; i++) {
            const int row = 4 * i + pr0;
            const bool ok = okc && row < TO && oy0 + row < a.H;
            const u32x4 v = *(const u32x4 *)(lsrc + (row < TO ? i * (64 * ROWB) : 0));
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, ok ? g0 + (unsigned)(4 * i) * rstride : OOB, 0, 0);
        }
    };
    auto no_hook = [](int) {};

    int item = bid, pn = -1, pty = 0, ptx = 0;
    for (int it = 0; it < my_items; it++, item += gridDim.x) {
        const int buf = it & 1;
        // operation order per wave and item: [top: ST_I stores of the tile staged by the item before] [conv1's six tap columns: one piece each
        // of the NEXT item's patch].  So my pieces of this item's patch are the youngest operations in flight, requested most of an item ago.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();                                          // everybody's pieces; the tile of the item before is staged; its patch buffer is free
        write_out(pn, pty, ptx);
        int n, ty, tx, nn, nty, ntx;
        decode_tile(item, n, ty, tx);
        const bool nlive = it + 1 < my_items;
        decode_tile(nlive ? item + gridDim.x : 0, nn, nty, ntx);
        auto fetch_hook = [&](int pass) { issue_piece(pass, nn, nty, ntx, nlive, buf ^ 1); };      // one piece of the next patch per (chunk, column) pass

        // ================= A: conv1 on the 16x16 region (rows 8 rg .. 8 rg + 7 here) =================
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = IR ? f32x4{0.f, 0.f, 0.f, 0.f} : bias1;      // (plain form: the bias is the accumulators' start value, no add in the epilogue)
        if (!(a.ablate & 1)) {
            const int xo = OFF_X + buf * X_ITEM + rg * (8 * PW * 64);
            conv_phase(xo, integral_constant<int, 8>{}, integral_constant<int, PW>{}, integral_constant<int, P_BYTES>{}, w1, fetch_hook);
        } else {
#pragma unroll
            for (int k = 0; k < MAX_P; k++) issue_piece(k, nn, nty, ntx, nlive, buf ^ 1);
        }
        {
            // intermediate pixel (row 8 rg + i, column frow) = image pixel (ty*14 - 1 + row, tx*14 - 1 + frow); outside the image it is
            // conv2's zero padding, NOT conv1 evaluated there
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const int gx = tx * TO - 1 + fr, gy0 = ty * TO - 1 + rg * 8;
            const bool xin = (unsigned)gx < (unsigned)a.W;
            // (a tile whose 16x16 region lies inside the image has no padding pixel to clear: wave-uniform, most tiles)
            const bool edge = ty == 0 || tx == 0 || ty * TO + 15 > a.H || tx * TO + 15 > a.W;
            char *mp = smem + OFF_MID + (cw >> 1) * MID_CH + (rg * 8 * MW + fr) * 64 + ((((cw & 1) * 2 + (q4 >> 1)) ^ swz64(fr)) << 4) + (q4 & 1) * 8;
            const int xc = a.ncls1 == 9 ? (gx == 0 ? 0 : (gx == a.W - 1 ? 2 : 1)) : 0;
            const float *tb = (const float *)(smem + OFF_TAB) + xc * 64 + cw * 16 + q4 * 4;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                half4 h;
                if constexpr (IR) {
                    const int gy = gy0 + i, yc = a.ncls1 == 9 ? (gy == 0 ? 0 : (gy == a.H - 1 ? 2 : 1)) : 0;
                    f32x4 v = acc[i] + *(const f32x4 *)(tb + yc * 192);
                    v = __builtin_elementwise_fma(slope1, __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f}), __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}));   // (one rounding: max + slope * min)
                    h = __builtin_convertvector(v, half4);
                } else {
                    h = __builtin_elementwise_max(__builtin_convertvector(acc[i], half4), half4{0, 0, 0, 0});
                }
                if (edge && !(xin && (unsigned)(gy0 + i) < (unsigned)a.H)) h = half4{0, 0, 0, 0};
                *(half4 *)(mp + i * (MW * 64)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // the intermediate tile is complete (and everyone has read the staged tile of the item before)

        // ================= B: conv2 on the 14x14 tile (rows 7 rg .. 7 rg + 6 here) =================
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = bias2;
        if (!(a.ablate & 2)) {
            const int mo = OFF_MID + rg * (7 * MW * 64);
            conv_phase(mo, integral_constant<int, 7>{}, integral_constant<int, MW>{}, integral_constant<int, MID_CH>{}, w2, no_hook);
        }
        {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            // residual = x at the output pixel = patch pixel (row + 2, column + 2) of the chunk my couts live in
            const int g0 = (cw & 1) * 2 + (q4 >> 1);
            const char *xp = smem + OFF_X + buf * X_ITEM + (cw >> 1) * P_BYTES + (q4 & 1) * 8;
            char *sp = smem + OFF_STG + fr * ROWB + (((cw * 2 + (q4 >> 1) + fr) % CPX) << 4) + (q4 & 1) * 8;   // chunk rotated by the pixel column
            half4 rs[7];                                        // all residual reads first: one LDS round trip, not one per row (the staging writes below may alias for the compiler)
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const int r = rg * 7 + i, lin = (r + 2) * PW + fr + 2;
                rs[i] = *(const half4 *)(xp + lin * 64 + ((g0 ^ swz64(lin)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const int r = rg * 7 + i;
                f32x4 v = acc[i] + __builtin_convertvector(rs[i], f32x4);
                half4 h = __builtin_convertvector(v, half4);
                if (a.act2 == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                *(half4 *)(sp + r * (16 * ROWB)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pn = n; pty = ty; ptx = tx;
    }
    raw_barrier();                                              // the last tile is staged
    write_out(pn, pty, ptx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the surplus pieces target this workgroup's LDS: drain before exit
}


// ======================================================================================================================================
// The 64-channel block as TWO independent workgroups of FOUR waves per CU (round 4, second session; opt-in with `FID_BB_V=2`: measured equal
// alone and 1 % slower in the two-lane bench step, docs/FINDINGS.md).
// The eight-wave kernel above is one lock-step group per CU: both waves of a SIMD reach the conv1 / conv2 epilogues, the barriers and the
// patch wait together, and the matrix pipe idles through all of them (60 % MFMA-busy, conv phases 211 of 272 us).  conv_bb32 showed what two
// workgroups per CU buy (their epilogues and waits interleave); here they have to fit 80 KB of LDS each:
//   wave  = cout fragment cw = 0..3 for ALL rows: the two row groups of the eight-wave kernel (conv1 rows 8 rg .. 8 rg + 7, conv2 rows
//           7 rg .. 7 rg + 6) run one after the other on the same 8 accumulator rows -- same registers per wave (both filter banks: 144)
//   LDS   = ONE x-patch buffer (43 KB) + the intermediate tile (32 KB) + bias table: 79.6 KB.  No second patch buffer and no staging area:
//           the finished tile is written IN PLACE over the x patch (a lane's residual values sit exactly where its results go: patch pixel
//           (row + 2, column + 2), its couts' 8 bytes), the 16-byte row stores read it from there, and only then is the next item's patch
//           requested -- that wait (~2-3 us per item) is what the other workgroup's matrix phases cover.
//   order = [wait my pieces | barrier] conv1 rg 0, rg 1 -> mid [barrier] conv2 rg 0, rg 1 -> x in place [barrier] row stores [barrier] request
//           the next patch.  Every wait is a full drain: no hand-counted operation.
// ======================================================================================================================================
constexpr int NW2 = 4;
constexpr int MAX_P2 = (N_PIECES + NW2 - 1) / NW2;               // 11 pieces per wave (two surplus ones go to the spare KB)
constexpr int ST_I2 = (TO * 16 * CPX + NW2 * 64 - 1) / (NW2 * 64);   // 7 stores per thread and item: two tile rows per round
constexpr int OFF2_X = 0, OFF2_SPARE = X_ITEM, OFF2_MID = OFF2_SPARE + 1024, OFF2_TAB = OFF2_MID + MID_BYTES + 512, LDS2_BYTES = OFF2_TAB + TAB_BYTES;
static_assert(LDS2_BYTES <= 80 * 1024 && ST_I2 * 2 == TO, "two workgroups per CU; a store round covers two tile rows");

template <bool IR>
__global__ void __launch_bounds__(NW2 * 64, 2) conv_bb2(const BBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, cw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    if (my_items == 0) return;

    auto decode_tile = [&](int item, int &n, int &ty, int &tx) {
        if (a.rev) item = a.n_tiles - 1 - item;
        n = fastdiv(item, a.d_tpi);
        const int r = item - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.io_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.io_bytes, 0x00020000);

    int p_pk[MAX_P2];                                           // py | px << 8 | channel offset (halfs) << 16; py = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P2; k++) {
        const int j = cw + NW2 * k;
        const int ch = j / P_BLKS, blk = j - ch * P_BLKS;
        const int row = blk * 16 + (lane >> 2);
        int py = row / PW;
        const int px = row - py * PW;
        if (row >= NPIX || j >= N_PIECES) py = 255;
        p_pk[k] = py | (px << 8) | (((((lane & 3) ^ swz64(row)) * 8) + ch * 32) << 16);
    }
    auto issue_patch = [&](int n, int ty, int tx, bool live) {  // my MAX_P2 pieces of the patch of tile (n, ty, tx)
        const int y0 = ty * TO - 2, x0 = tx * TO - 2;
#pragma unroll
        for (int k = 0; k < MAX_P2; k++) {
            const int j = cw + NW2 * k;
            int pk = p_pk[k];
            asm volatile("" : "+v"(pk));                        // opaque: unpack at the use
            const int py = pk & 255, iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
            const bool in = live && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a.ablate & 8);
            const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * 64 + (pk >> 16)) * 2) : OOB;
            char *d = j < N_PIECES ? smem + OFF2_X + j * 1024 : smem + OFF2_SPARE;     // surplus piece: zeros into the spare KB
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)d, 16, vo, 0, 0, 0);
        }
    };
    if (a.stagger && blockIdx.x >= gridDim.x / 2)               // experiment (FID_BB_STAGGER): the co-resident half of the grid starts late
        for (int i = 0; i < a.stagger; i++) __builtin_amdgcn_s_sleep(64);
    {
        int n, ty, tx;
        decode_tile(bid, n, ty, tx);
        issue_patch(n, ty, tx, true);
    }

    // ---- both filter banks of my 16 couts: kind 2 = [chunk][cout fragment 0..7][dx][dy][lane] x 16 B
    half8 w1[18], w2[18];                                       // [chunk * 9 + dy * 3 + dx]
    {
        const char *p1 = (const char *)a.w1 + cw * 9216 + lane * 16, *p2 = (const char *)a.w2 + cw * 9216 + lane * 16;
#pragma unroll
        for (int ck = 0; ck < 2; ck++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    w1[ck * 9 + dy * 3 + dx] = *(const half8 *)(p1 + ck * (8 * 9216) + dx * 3072 + dy * 1024);
                    w2[ck * 9 + dy * 3 + dx] = *(const half8 *)(p2 + ck * (8 * 9216) + dx * 3072 + dy * 1024);
                }
    }
    const f32x4 bias1 = *(const f32x4 *)(a.b1 + cw * 16 + fq * 4), bias2 = *(const f32x4 *)(a.b2 + cw * 16 + fq * 4);
    f32x4 slope1 = f32x4{0.f, 0.f, 0.f, 0.f};                   // (ReLU = PReLU with slope 0)
    if constexpr (IR) {
        if (a.act1 == ACT_PRELU) slope1 = *(const f32x4 *)(a.s1 + cw * 16 + fq * 4);
        float *tb = (float *)(smem + OFF2_TAB);
        for (int i = tid; i < a.ncls1 * 64; i += NW2 * 64) tb[i] = a.b1[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w1[i]), "+v"(w2[i]));

    int pbase[2][4];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int c = 0; c < 4; c++) pbase[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c) & 3)) << 4);

    f32x4 acc[8];
    constexpr int PD = BB_PD;
    auto conv_phase = [&](int base_off, auto rows_tag, auto pw_tag, auto cs_tag, const half8 *wv) {
        constexpr int ROWS = decltype(rows_tag)::value, PWV = decltype(pw_tag)::value, CS = decltype(cs_tag)::value, PH = ROWS + 2, NQ = 6 * PH;
        int pb[2][4];
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = pbase[par][c] + base_off;
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int q) {                              // q = (chunk * 3 + dx) * PH + fragment row
            const int pass = q / PH, r = q - pass * PH, ck = pass / 3, dx = pass - ck * 3;
            const int K = r * PWV + dx;
            pq[q % (PD + 1)] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + (K * 64 + ck * CS)));
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
                if (mi < 0 || mi >= ROWS) continue;
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ck * 9 + dy * 3 + dx], pq[q % (PD + 1)], acc[mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;

    // row stores of the finished tile, read from where it was written in place: 16-byte slot = (pixel q0 = thread / 8 of a two-row round,
    // couts 8 c .. 8 c + 7 = group c & 3 of chunk c >> 2)
    auto write_out = [&](int n, int ty, int tx) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int q0 = t2 >> 3, c = t2 & 7;
        const int pr0 = q0 >> 4, pc = q0 & 15;
        const int oy0 = ty * TO, ox = tx * TO + pc;
        const bool okc = pc < TO && ox < a.W && !(a.ablate & 4);
        const char *lsrc = smem + OFF2_X + (c >> 2) * P_BYTES;
        const unsigned g0 = (unsigned)((((n * a.H + oy0 + pr0) * a.W + ox) * 64 + c * 8) * 2);
        const unsigned rstride = (unsigned)(a.W * 64 * 2);
#pragma unroll
        for (int i = 0; i < ST_I2; i++) {
            const int row = 2 * i + pr0, lin = (row + 2) * PW + pc + 2;
            const u32x4 v = *(const u32x4 *)(lsrc + lin * 64 + (((c & 3) ^ swz64(lin)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (okc && oy0 + row < a.H) ? g0 + (unsigned)(2 * i) * rstride : OOB, 0, 0);
        }
    };

    int item = bid;
    for (int it = 0; it < my_items; it++, item += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();                                          // everybody's pieces of this item's patch have landed
        int n, ty, tx;
        decode_tile(item, n, ty, tx);

        // ================= A: conv1 on the 16x16 region, rows 8 rg .. 8 rg + 7 per pass =================
#pragma nounroll
        for (int rg = 0; rg < 2; rg++) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc[r] = IR ? f32x4{0.f, 0.f, 0.f, 0.f} : bias1;
            if (!(a.ablate & 1))
                conv_phase(OFF2_X + rg * (8 * PW * 64), integral_constant<int, 8>{}, integral_constant<int, PW>{}, integral_constant<int, P_BYTES>{}, w1);
            if (a.ablate & 16) continue;
            // intermediate pixel (row 8 rg + i, column frow) = image pixel (ty*14 - 1 + row, tx*14 - 1 + frow); outside the image it is
            // conv2's zero padding, NOT conv1 evaluated there
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const int gx = tx * TO - 1 + fr, gy0 = ty * TO - 1 + rg * 8;
            const bool xin = (unsigned)gx < (unsigned)a.W;
            const bool edge = ty == 0 || tx == 0 || ty * TO + 15 > a.H || tx * TO + 15 > a.W;
            char *mp = smem + OFF2_MID + (cw >> 1) * MID_CH + (rg * 8 * MW + fr) * 64 + ((((cw & 1) * 2 + (q4 >> 1)) ^ swz64(fr)) << 4) + (q4 & 1) * 8;
            const int xc = a.ncls1 == 9 ? (gx == 0 ? 0 : (gx == a.W - 1 ? 2 : 1)) : 0;
            const float *tb = (const float *)(smem + OFF2_TAB) + xc * 64 + cw * 16 + q4 * 4;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                half4 h;
                if constexpr (IR) {
                    const int gy = gy0 + i, yc = a.ncls1 == 9 ? (gy == 0 ? 0 : (gy == a.H - 1 ? 2 : 1)) : 0;
                    f32x4 v = acc[i] + *(const f32x4 *)(tb + yc * 192);
                    v = __builtin_elementwise_fma(slope1, __builtin_elementwise_min(v, f32x4{0.f, 0.f, 0.f, 0.f}), __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f}));   // (one rounding: max + slope * min)
                    h = __builtin_convertvector(v, half4);
                } else {
                    h = __builtin_elementwise_max(__builtin_convertvector(acc[i], half4), half4{0, 0, 0, 0});
                }
                if (edge && !(xin && (unsigned)(gy0 + i) < (unsigned)a.H)) h = half4{0, 0, 0, 0};
                *(half4 *)(mp + i * (MW * 64)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(a.ablate & 64)) raw_barrier();                    // the intermediate tile is complete; nobody reads the x patch's halo any more

        // ================= B: conv2 on the 14x14 tile, rows 7 rg .. 7 rg + 6 per pass; results in place over the x patch =================
#pragma nounroll
        for (int rg = 0; rg < 2; rg++) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc[r] = bias2;
            if (!(a.ablate & 2))
                conv_phase(OFF2_MID + rg * (7 * MW * 64), integral_constant<int, 7>{}, integral_constant<int, MW>{}, integral_constant<int, MID_CH>{}, w2);
            if (a.ablate & 32) continue;
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            // residual = x at the output pixel = patch pixel (row + 2, column + 2) of the chunk my couts live in; the result replaces it
            const int g0 = (cw & 1) * 2 + (q4 >> 1);
            char *xp = smem + OFF2_X + (cw >> 1) * P_BYTES + (q4 & 1) * 8;
            half4 rs[7];                                        // all residual reads first: one LDS round trip, not one per row
            int ro[7];
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const int lin = (rg * 7 + i + 2) * PW + fr + 2;
                ro[i] = lin * 64 + ((g0 ^ swz64(lin)) << 4);
                rs[i] = *(const half4 *)(xp + ro[i]);
            }
#pragma unroll
            for (int i = 0; i < 7; i++) {
                f32x4 v = acc[i] + __builtin_convertvector(rs[i], f32x4);
                half4 h = __builtin_convertvector(v, half4);
                if (a.act2 == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                *(half4 *)(xp + ro[i]) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(a.ablate & 64)) raw_barrier();                    // the tile is complete (all four cout fragments)
        write_out(n, ty, tx);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of the tile are in registers (the stores themselves may fly on)
        raw_barrier();                                          // everybody's are: the buffer is free for the next patch
        {
            const bool nlive = it + 1 < my_items;
            int nn, nty, ntx;
            decode_tile(nlive ? item + gridDim.x : 0, nn, nty, ntx);
            if (nlive) issue_patch(nn, nty, ntx, true);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ======================================================================================================================================
// The same block on 32 stored channels (SCRFD-2.5G layer1: 24 -> 32 padded, three blocks at 160x160): ONE 32-channel chunk, two cout fragments.
// FOUR waves = (cout fragment cw = 0..1) x (row group rg = 0..1): the 64-channel kernel's row split (conv1 rows 8 rg .. 8 rg + 7, conv2 rows
// 7 rg .. 7 rg + 6) with half the cout fragments.  Both filter banks of a wave are 2 x 9 fragments = 72 VGPRs; LDS = two 21 KB patch buffers +
// 16 KB intermediate tile + 14 KB staging = 74 KB: TWO workgroups per CU, whose barriers and epilogues interleave (eight waves in one workgroup
// with four rows each: 57.6 us per block at 160x160x32 frames, 35 us of it with no conv at all -- one lock-step group per CU hides nothing).
// Plain bias + ReLU after conv1 only (SCRFD's form).  Same operation order per wave and item as above: [top: ST_I32 stores of the tile staged
// by the item before] [conv1's three tap columns: one piece each of the NEXT item's patch].
// ======================================================================================================================================
constexpr int NW32 = 4;
constexpr int MAX_P32 = (P_BLKS + NW32 - 1) / NW32;               // 6 pieces per wave: two per tap column of conv1
constexpr int MID32 = MW * MW * 64;                               // 16 KB
constexpr int ROWB32 = 64, CPX32 = 4;
constexpr int ST_I32 = (TO * 16 * CPX32 + NW32 * 64 - 1) / (NW32 * 64);    // 4
constexpr int STG32 = TO * 16 * ROWB32;
constexpr int OFF32_SPARE = 2 * P_BYTES, OFF32_MID = OFF32_SPARE + 1024, OFF32_STG = OFF32_MID + MID32 + 512, LDS32 = OFF32_STG + STG32;
static_assert(MAX_P32 == 6 && LDS32 <= 80 * 1024, "two workgroups per CU");

__global__ void __launch_bounds__(NW32 * 64, 2) conv_bb32(const BBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, rg = wave >> 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    if (my_items == 0) return;
    auto decode_tile = [&](int item, int &n, int &ty, int &tx) {
        if (a.rev) item = a.n_tiles - 1 - item;
        n = fastdiv(item, a.d_tpi);
        const int r = item - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.io_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.io_bytes, 0x00020000);

    int p_pk[MAX_P32];                                          // py | px << 8 | channel offset (halfs) << 16; py = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P32; k++) {
        const int j = wave + NW32 * k;
        const int row = j * 16 + (lane >> 2);
        int py = row / PW;
        const int px = row - py * PW;
        if (row >= NPIX || j >= P_BLKS) py = 255;
        p_pk[k] = py | (px << 8) | ((((lane & 3) ^ swz64(row)) * 8) << 16);
    }
    auto issue_piece = [&](int k, int n, int ty, int tx, bool live, int buf) {
        const int y0 = ty * TO - 2, x0 = tx * TO - 2;
        const int j = wave + NW32 * k;
        int pk = p_pk[k];
        asm volatile("" : "+v"(pk));
        const int py = pk & 255, iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
        const bool in = live && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && !(a.ablate & 8);
        const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * 32 + (pk >> 16)) * 2) : OOB;
        char *d = j < P_BLKS ? smem + buf * P_BYTES + j * 1024 : smem + OFF32_SPARE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)d, 16, vo, 0, 0, 0);
    };
    {
        int n, ty, tx;
        decode_tile(bid, n, ty, tx);
#pragma unroll
        for (int k = 0; k < MAX_P32; k++) issue_piece(k, n, ty, tx, true, 0);
    }
    half8 w1[9], w2[9];                                         // [dy * 3 + dx]
    {
        const char *p1 = (const char *)a.w1 + cw * 9216 + lane * 16, *p2 = (const char *)a.w2 + cw * 9216 + lane * 16;
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                w1[dy * 3 + dx] = *(const half8 *)(p1 + dx * 3072 + dy * 1024);
                w2[dy * 3 + dx] = *(const half8 *)(p2 + dx * 3072 + dy * 1024);
            }
    }
    const f32x4 bias1 = *(const f32x4 *)(a.b1 + cw * 16 + fq * 4), bias2 = *(const f32x4 *)(a.b2 + cw * 16 + fq * 4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 9; i++) asm volatile("" : "+v"(w1[i]), "+v"(w2[i]));

    int pbase[2][4];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int c = 0; c < 4; c++) pbase[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c) & 3)) << 4);

    f32x4 acc[8];
    constexpr int PD = BB_PD;
    // one conv: ROWS output rows of this wave from ROWS + 2 fragment rows x 3 tap columns (one flattened sequence, read PD rows ahead across the columns)
    auto conv_phase = [&](int base_off, auto rows_tag, auto pw_tag, const half8 *wv, auto &&col_hook) {
        constexpr int ROWS = decltype(rows_tag)::value, PWV = decltype(pw_tag)::value, PH = ROWS + 2, NQ = 3 * PH;
        int pb[2][4];
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = pbase[par][c] + base_off;
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int q) {                              // q = dx * PH + fragment row
            const int dx = q / PH, r = q - dx * PH;
            const int K = r * PWV + dx;
            pq[q % (PD + 1)] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + K * 64));
        };
#pragma unroll
        for (int q = 0; q < PD; q++) load_p(q);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int dx = q / PH, r = q - dx * PH;
            if (r == 0) col_hook(dx);
            if (q + PD < NQ) load_p(q + PD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const int mi = r - dy;
                if (mi < 0 || mi >= ROWS) continue;
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[dy * 3 + dx], pq[q % (PD + 1)], acc[mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
    // write-out of the tile staged by the item before: 16-byte slot g = i * 256 + thread = (pixel g / 4, chunk g % 4); 64 pixels = 4 tile rows per round
    auto write_out = [&](int pn, int pty, int ptx) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int q0 = t2 >> 2, c = t2 & 3;
        const int pr0 = q0 >> 4, pc = q0 & 15;
        const int oy0 = pty * TO, ox = ptx * TO + pc;
        const bool okc = pn >= 0 && pc < TO && ox < a.W && !(a.ablate & 4);
        const char *lsrc = smem + OFF32_STG + q0 * ROWB32 + (((c + pc) % CPX32) << 4);
        const unsigned g0 = (unsigned)((((pn * a.H + oy0 + pr0) * a.W + ox) * 32 + c * 8) * 2);
        const unsigned rstride = (unsigned)(a.W * 32 * 2);
#pragma unroll
        for (int i = 0; i < ST_I32; i++) {
            const int row = 4 * i + pr0;
            const bool ok = okc && row < TO && oy0 + row < a.H;
            const u32x4 v = *(const u32x4 *)(lsrc + (row < TO ? i * (64 * ROWB32) : 0));
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, ok ? g0 + (unsigned)(4 * i) * rstride : OOB, 0, 0);
        }
    };
    auto no_hook = [](int) {};

    int item = bid, pn = -1, pty = 0, ptx = 0;
    for (int it = 0; it < my_items; it++, item += gridDim.x) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();                                          // everybody's pieces; the tile of the item before is staged; its patch buffer is free
        write_out(pn, pty, ptx);
        int n, ty, tx, nn, nty, ntx;
        decode_tile(item, n, ty, tx);
        const bool nlive = it + 1 < my_items;
        decode_tile(nlive ? item + gridDim.x : 0, nn, nty, ntx);
        auto fetch_hook = [&](int pass) { issue_piece(2 * pass, nn, nty, ntx, nlive, buf ^ 1); issue_piece(2 * pass + 1, nn, nty, ntx, nlive, buf ^ 1); };   // two pieces of the next patch per tap column

        // ================= A: conv1 on the 16x16 region (rows 8 rg .. 8 rg + 7 here) =================
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(a.ablate & 1)) {
            conv_phase(buf * P_BYTES + rg * (8 * PW * 64), integral_constant<int, 8>{}, integral_constant<int, PW>{}, w1, fetch_hook);
        } else {
#pragma unroll
            for (int k = 0; k < MAX_P32; k++) issue_piece(k, nn, nty, ntx, nlive, buf ^ 1);
        }
        {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const int gx = tx * TO - 1 + fr, gy0 = ty * TO - 1 + rg * 8;
            const bool xin = (unsigned)gx < (unsigned)a.W;
            char *mp = smem + OFF32_MID + (rg * 8 * MW + fr) * 64 + (((cw * 2 + (q4 >> 1)) ^ swz64(fr)) << 4) + (q4 & 1) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                half4 h = __builtin_elementwise_max(__builtin_convertvector(acc[i] + bias1, half4), half4{0, 0, 0, 0});
                if (!(xin && (unsigned)(gy0 + i) < (unsigned)a.H)) h = half4{0, 0, 0, 0};      // outside the image: conv2's zero padding
                *(half4 *)(mp + i * (MW * 64)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // the intermediate tile is complete

        // ================= B: conv2 on the 14x14 tile (rows 7 rg .. 7 rg + 6 here) =================
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(a.ablate & 2)) conv_phase(OFF32_MID + rg * (7 * MW * 64), integral_constant<int, 7>{}, integral_constant<int, MW>{}, w2, no_hook);
        {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const int g0 = cw * 2 + (q4 >> 1);
            const char *xp = smem + buf * P_BYTES + (q4 & 1) * 8;
            char *sp = smem + OFF32_STG + fr * ROWB32 + (((cw * 2 + (q4 >> 1) + fr) % CPX32) << 4) + (q4 & 1) * 8;
            half4 rs[7];
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const int r = rg * 7 + i, lin = (r + 2) * PW + fr + 2;
                rs[i] = *(const half4 *)(xp + lin * 64 + ((g0 ^ swz64(lin)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const int r = rg * 7 + i;
                f32x4 v = acc[i] + bias2 + __builtin_convertvector(rs[i], f32x4);
                half4 h = __builtin_convertvector(v, half4);
                if (a.act2 == ACT_RELU) h = __builtin_elementwise_max(h, half4{0, 0, 0, 0});
                *(half4 *)(sp + r * (16 * ROWB32)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pn = n; pty = ty; ptx = tx;
    }
    raw_barrier();
    write_out(pn, pty, ptx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// x [B, H, W, Cp] fp16 -> out [B, H, W, Cp], Cp = 64 | 32 stored channels; w1 / w2: repack kind 2 images (147 456 B | 73 728 B each), b1 fp32 [ncls1][64], b2 fp32 [64], s1 fp32 [64] or NULL
int conv_bb_launch(fid_ctx *ctx, const void *in, const void *w1, const float *b1, int ncls1, int act1, const float *s1, const void *w2, const float *b2,
                   void *out, int B, int H, int W, int act2, int rev, int Cp) {
    FID_REQUIRE(in && w1 && w2 && b1 && b2 && out && B > 0 && H >= 3 && W >= 3, "conv_bb: bad arguments");
    FID_REQUIRE((act2 == ACT_RELU || act2 == ACT_NONE) && (act1 == ACT_RELU || (act1 == ACT_PRELU && s1)) && (ncls1 == 1 || ncls1 == 9), "conv_bb: activations %d / %d, %d bias rows", act1, act2, ncls1);
    BBArgs a{};
    a.in = in; a.w1 = w1; a.w2 = w2; a.b1 = b1; a.b2 = b2; a.out = out;
    a.H = H; a.W = W; a.act2 = act2; a.rev = rev;
    a.s1 = s1; a.act1 = act1; a.ncls1 = ncls1;
    a.tiles_x = cdiv(W, TO);
    a.tiles_per_img = a.tiles_x * cdiv(H, TO);
    a.n_tiles = B * a.tiles_per_img;
    a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    FID_REQUIRE(Cp == 64 || (Cp == 32 && ncls1 == 1 && act1 == ACT_RELU), "conv_bb: %d stored channels (%d bias rows, activation %d)", Cp, ncls1, act1);
    const size_t bytes = (size_t)B * H * W * Cp * 2;
    FID_REQUIRE(bytes <= OOB, "conv_bb: tensor larger than 2 GiB");
    a.io_bytes = (unsigned)bytes;
    static const int ablate = getenv("FID_BB_ABLATE") ? atoi(getenv("FID_BB_ABLATE")) : 0;
    a.ablate = ablate;
    if (Cp == 32) {                                             // four waves, two workgroups per CU
        FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_bb32, LDS32));
        hipLaunchKernelGGL(conv_bb32, dim3(std::min(a.n_tiles, 2 * ctx->num_cus)), dim3(NW32 * 64), LDS32, ctx->stream, a);
        FID_HIP(hipGetLastError());
        return FID_OK;
    }
    // FID_BB_V=2 (read per launch: the tests switch it): two independent four-wave workgroups per CU (conv_bb2) instead of the eight-wave form.
    // Measured (profiles/r04/ab_runs.txt): equal alone, one-lane bench step -0.8 %, two-lane step +1.0 % -> not the default.
    const char *bv = getenv("FID_BB_V");
    if (bv && atoi(bv) == 2) {
        static const int stagger = getenv("FID_BB_STAGGER") ? atoi(getenv("FID_BB_STAGGER")) : 0;
        a.stagger = stagger;
        const int grid2 = std::min(a.n_tiles, 2 * ctx->num_cus);
        if (ncls1 == 9 || act1 == ACT_PRELU) {
            FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_bb2<true>, LDS2_BYTES));
            hipLaunchKernelGGL(conv_bb2<true>, dim3(grid2), dim3(NW2 * 64), LDS2_BYTES, ctx->stream, a);
        } else {
            FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_bb2<false>, LDS2_BYTES));
            hipLaunchKernelGGL(conv_bb2<false>, dim3(grid2), dim3(NW2 * 64), LDS2_BYTES, ctx->stream, a);
        }
        FID_HIP(hipGetLastError());
        return FID_OK;
    }
    const int grid = std::min(a.n_tiles, ctx->num_cus);
    if (ncls1 == 9 || act1 == ACT_PRELU) {
        FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_bb<true>, LDS_BYTES));
        hipLaunchKernelGGL(conv_bb<true>, dim3(grid), dim3(NWT * 64), LDS_BYTES, ctx->stream, a);
    } else {
        FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_bb<false>, LDS_BYTES));
        hipLaunchKernelGGL(conv_bb<false>, dim3(grid), dim3(NWT * 64), LDS_BYTES, ctx->stream, a);
    }
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
