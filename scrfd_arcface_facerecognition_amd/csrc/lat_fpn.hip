// PAFPN lateral + output conv of one pyramid level as ONE launch (round 4):
//   lat = conv1x1(c) + bias [+ nearest-2x upsample of the coarser level's lat]          (SCRFD: neck.lat0 / neck.lat1)
//   fpn = conv3x3(lat) + bias                                                            (neck.fpn0 / neck.fpn1)
// reference models/scrfd.py:83 (nodes inside session.run).  Unfused, the lateral is a memory-bound launch of its own (lat0: 41 us for 4 GFLOP at
// 64 frames: it reads c3 and writes lat0, which the 3x3 conv then reads back) and the small levels are two launches of mostly skeleton.  Here the
// lateral is computed on the 18 x 18 region around a 16 x 16 output tile (1.27x of a stage that is 15 % of the item's matrix work) and only ever
// exists in LDS; it leaves for global memory only where a finer level's lateral adds it (lat_out; nothing else reads a lateral).
//
//   item   = a 16 x 16 tile of fpn x all 64 couts
//   waves  = 8 = (cout fragment cw: couts 16 cw .. 16 cw + 15) x (half h):
//            stage S: the 21 flattened 16-pixel fragments of the region, h = 0 takes the even, h = 1 the odd ones; the wave's 1x1 weight fragments
//                     (NCH x 4 VGPRs) are resident; c's haloed patch (all NCH chunks, 20.25 KB each, conv3x3_wr's swizzled pixel layout) arrives by
//                     LDS-DMA while the item BEFORE runs its 3x3 stage, and with it the 10 x 10 pixels of the coarser lateral under the region
//                     (region pixel (py, px) adds coarse pixel ((py + 1) >> 1, (px + 1) >> 1) of that block: an item-independent LDS address; read
//                     from global memory at the start of the item the loads were exposed: 62 -> us); bias = the accumulators' initial value;
//                     outside the image 0 (the 3x3 conv's zero padding) -> x region in LDS
//            stage C: rows 8 h .. 8 h + 7 of the tile, the 2 x 9 weight fragments of the wave's couts resident (repack kind 2), row-sharing tap order
//                     (conv_bb.hip's stage A on the lateral) -> staging -> 16-byte row stores at the start of the next item
//   LDS    = patch 63 KB + coarser lateral 13 KB + lateral 40.5 KB + staging 32 KB = 149.5 KB: one workgroup per CU.  Only full vmcnt(0) drains (no hand-counted waits).
#include <type_traits>

#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TO = 16, PW = 18, NPIX = PW * PW;                  // output tile edge, lateral region edge
constexpr int NF = (NPIX + 15) / 16;                             // 21 flattened pixel fragments of the region
constexpr int X_BYTES = NPIX * 64;                               // one 32-channel chunk of the lateral (2 chunks): 20.25 KB
constexpr int P_BLKS = (NPIX * 64 + 1023) / 1024, P_BYTES = P_BLKS * 1024;   // one 32-channel chunk of c's patch, whole DMA pieces: 21 KB
constexpr int NWT = 8, NT = NWT * 64;
constexpr int ROWB = 128, CPX = 8;
constexpr int ST_I = TO * 16 * CPX / NT;                         // 4 stores per thread and item
constexpr int STG_BYTES = TO * 16 * ROWB;
constexpr int MFH = (NF + 1) / 2;                                // fragments of a half: 11 (h = 0) / 10
constexpr int RW = 10, R_BLKS = (RW * RW * 128 + 1023) / 1024, R_BYTES = R_BLKS * 1024;   // the coarser lateral under the region: 10 x 10 pixels x 64 channels, 13 one-KB pieces

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct LFArgs {
    const void *in;           // c: fp16 [B, H, W, NCH*32]
    const void *w0;           // 1x1 weights fp16 [64][NCH*32]
    const float *b0;          // fp32 [64]
    const void *res;          // the coarser level's lateral fp16 [B, rH, rW, 64] or NULL
    const void *w1;           // 3x3: repack kind 2 image of the 64 x 9 x 64 filter bank
    const float *b1;          // fp32 [64]
    void *out;                // fpn: fp16 [B, H, W, 64]
    void *lat;                // the lateral itself [B, H, W, 64] (a finer level adds it) or NULL
    int H, W, rH, rW;
    int tiles_x, tiles_per_img, n_tiles;
    FastDiv d_tpi, d_tx;
    unsigned in_bytes, out_bytes, res_bytes;
};

template <int NCH>
__global__ void __launch_bounds__(NT, 2) lat_fpn(const LFArgs a) {
    constexpr int OFF_P = 0, OFF_R = NCH * P_BYTES, OFF_SPARE = OFF_R + R_BYTES, OFF_X = OFF_SPARE + 1024, OFF_STG = OFF_X + 2 * X_BYTES, LDS = OFF_STG + STG_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    constexpr int N_PP = NCH * P_BLKS, N_PIECES = N_PP + R_BLKS, MAX_P = (N_PIECES + NWT - 1) / NWT;    // patch pieces, then the pieces of the coarser lateral
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, hh = wave >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_tiles ? (a.n_tiles - 1 - bid) / gridDim.x + 1 : 0;
    if (my_items == 0) return;

    auto decode_tile = [&](int item, int &n, int &ty, int &tx) __attribute__((always_inline)) {
        n = fastdiv(item, a.d_tpi);
        const int r = item - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, a.out_bytes, 0x00020000);
    const auto rs_lat = __builtin_amdgcn_make_buffer_rsrc((void *)(a.lat ? a.lat : a.out), 0, a.out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res ? a.res : a.out), 0, a.res ? a.res_bytes : a.out_bytes, 0x00020000);

    // ---- my pieces of an item's patch: piece j = wave + 8 k -> chunk j / 21, pixels 16 (j % 21) .. + 15, four lanes per pixel (conv_bb.hip)
    int p_pk[MAX_P];                                            // py | px << 8 | channel offset (halfs) << 16; py = 255: nothing to fetch
#pragma unroll
    for (int k = 0; k < MAX_P; k++) {
        const int j = wave + NWT * k;
        if (j < N_PP) {
            const int ch = j / P_BLKS, blk = j - ch * P_BLKS;
            const int row = blk * 16 + (lane >> 2);
            int py = row / PW;
            const int px = row - py * PW;
            if (row >= NPIX) py = 255;
            p_pk[k] = py | (px << 8) | (((((lane & 3) ^ swz64(row)) * 8) + ch * 32) << 16);
        } else {                                                // a piece of the coarser lateral: 8 pixels x 128 B, pixel-linear, eight lanes per pixel
            const int cp = (j - N_PP) * 8 + (lane >> 3);
            int cy = cp / RW;
            const int cx = cp - cy * RW;
            if (cp >= RW * RW || j >= N_PIECES) cy = 255;
            p_pk[k] = cy | (cx << 8) | (((lane & 7) * 8) << 16);
        }
    }
    auto issue_patch = [&](int item, bool live) __attribute__((always_inline)) {   // exactly MAX_P instructions per wave
        int n, ty, tx;
        decode_tile(live ? item : 0, n, ty, tx);
        const int y0 = ty * TO - 1, x0 = tx * TO - 1, cy0 = ty * (TO / 2) - 1, cx0 = tx * (TO / 2) - 1;
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + NWT * k;
            int pk = p_pk[k];
            asm volatile("" : "+v"(pk));
            const int py = pk & 255;
            char *d = j < N_PIECES ? smem + OFF_P + (j < N_PP ? j * 1024 : OFF_R + (j - N_PP) * 1024) : smem + OFF_SPARE;
            if (j < N_PP) {
                const int iy = y0 + py, ix = x0 + ((pk >> 8) & 255);
                const bool in = live && py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((n * a.H + iy) * a.W + ix) * (NCH * 32) + (pk >> 16)) * 2) : OOB - (unsigned)k * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)d, 16, vo, 0, 0, 0);
            } else {                                            // region pixel (py, px) adds coarse pixel ((py + 1) >> 1, (px + 1) >> 1) of this 10 x 10 block
                const int iy = cy0 + py, ix = cx0 + ((pk >> 8) & 255);
                const bool in = live && a.res && py != 255 && (unsigned)iy < (unsigned)a.rH && (unsigned)ix < (unsigned)a.rW;
                const unsigned vo = in ? (unsigned)((((n * a.rH + iy) * a.rW + ix) * 64 + (pk >> 16)) * 2) : OOB - (unsigned)k * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (__attribute__((address_space(3))) void *)d, 16, vo, 0, 0, 0);
            }
        }
    };
    issue_patch(bid, true);

    // ---- weights -> registers: the 1x1 fragments of my couts (row = cout, 8 consecutive channels per lane) and the 3x3 bank's 18 fragments
    half8 w0f[NCH];
#pragma unroll
    for (int ck = 0; ck < NCH; ck++) w0f[ck] = *(const half8 *)((const _Float16 *)a.w0 + (cw * 16 + frow) * (NCH * 32) + ck * 32 + fq * 8);
    half8 w1[18];
    {
        const char *p1 = (const char *)a.w1 + cw * 9216 + lane * 16;
#pragma unroll
        for (int ck = 0; ck < 2; ck++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int dy = 0; dy < 3; dy++) w1[ck * 9 + dy * 3 + dx] = *(const half8 *)(p1 + ck * (8 * 9216) + dx * 3072 + dy * 1024);
    }
    const f32x4 bias0 = *(const f32x4 *)(a.b0 + cw * 16 + fq * 4), bias1 = *(const f32x4 *)(a.b1 + cw * 16 + fq * 4);

    // ---- stage S constants of this lane (item-independent): per fragment of my half the pixel's operand address in a patch chunk, its 8-byte
    // slot in a lateral chunk (swizzle term folded in) and its region coordinates (-1: a lane without a pixel -- only in a half's last fragment)
    int f_in[MFH], f_x[MFH], f_yx[MFH], f_r[MFH];
#pragma unroll
    for (int i = 0; i < MFH; i++) {
        const int fi = hh + 2 * i, qd = fi * 16 + frow;
        const bool ok = fi < NF && qd < NPIX;
        const int q = ok ? qd : 0, py = q / PW, px = q - py * PW;
        f_in[i] = q * 64 + ((fq ^ swz64(q)) << 4);
        f_x[i] = (cw >> 1) * X_BYTES + q * 64 + ((((cw & 1) * 2 + (fq >> 1)) ^ swz64(q)) << 4) + (fq & 1) * 8;
        f_yx[i] = ok ? (py | (px << 8)) : -1;
        f_r[i] = OFF_R + (((py + 1) >> 1) * RW + ((px + 1) >> 1)) * 128 + (cw * 16 + fq * 4) * 2;     // its 8 bytes of the coarser lateral (item-independent)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w1[i]));

    int pbase[2][4];
#pragma unroll
    for (int par = 0; par < 2; par++)
#pragma unroll
        for (int c = 0; c < 4; c++) pbase[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c) & 3)) << 4);

    f32x4 acc[MFH > 8 ? MFH : 8];
    constexpr int PD = 2;
    auto conv_phase = [&](int base_off) __attribute__((always_inline)) {     // 8 output rows from 10 fragment rows x 3 tap columns x 2 chunks
        constexpr int ROWS = 8, PH = ROWS + 2, NQ = 6 * PH;
        int pb[2][4];
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = pbase[par][c] + base_off;
                asm volatile("" : "+v"(pb[par][c]));
            }
        half8 pq[PD + 1];
        auto load_p = [&](int q) __attribute__((always_inline)) {
            const int pass = q / PH, r = q - pass * PH, ck = pass / 3, dx = pass - ck * 3;
            const int K = r * PW + dx;
            pq[q % (PD + 1)] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + (K * 64 + ck * X_BYTES)));
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
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[ck * 9 + dy * 3 + dx], pq[q % (PD + 1)], acc[mi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // write-out of a staged 16 x 16 x 64 tile: 16-byte slot g = i * 512 + thread = (pixel g / 8, chunk g % 8); 64 pixels = 4 tile rows per round
    auto write_tile = [&](const char *stage, decltype(rs_out) rs, int pn, int pty, int ptx, bool rotated) __attribute__((always_inline)) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int q0 = t2 >> 3, c = t2 & 7;
        const int pr0 = q0 >> 4, pc = q0 & 15;
        const int oy0 = pty * TO, ox = ptx * TO + pc;
        const bool okc = pn >= 0 && ox < a.W;
        const unsigned g0 = (unsigned)((((pn * a.H + oy0 + pr0) * a.W + ox) * 64 + c * 8) * 2);
        const unsigned rstride = (unsigned)(a.W * 64 * 2);
#pragma unroll
        for (int i = 0; i < ST_I; i++) {
            const int row = 4 * i + pr0;
            u32x4 v;
            if (rotated) {                                      // the staging area: pixel-linear rows of 128 B, chunk rotated by the pixel column
                v = *(const u32x4 *)(stage + (row * 16 + pc) * ROWB + (((c + pc) % CPX) << 4));
            } else {                                            // the lateral region itself: pixel (row + 1, pc + 1), chunk plane c / 4, swizzled group
                const int lin = (row + 1) * PW + pc + 1;
                v = *(const u32x4 *)(stage + (c >> 2) * X_BYTES + lin * 64 + (((c & 3) ^ swz64(lin)) << 4));
            }
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (okc && oy0 + row < a.H) ? g0 + (unsigned)(4 * i) * rstride : OOB, 0, 0);
        }
    };

    int item = bid, pn = -1, pty = 0, ptx = 0;
    for (int it = 0; it < my_items; it++, item += gridDim.x) {
        int n, ty, tx;
        decode_tile(item, n, ty, tx);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of this item's patch (requested during the item before)
        raw_barrier();                                          // B0: everybody's pieces; the tile of the item before is staged; the lateral region is free
        write_tile(smem + OFF_STG, rs_out, pn, pty, ptx, true);
        const bool interior = ty * TO >= 1 && tx * TO >= 1 && ty * TO + PW - 1 <= a.H && tx * TO + PW - 1 <= a.W;
        // ================= S: the 1x1 lateral on the 18 x 18 region =================
#pragma unroll
        for (int i = 0; i < MFH; i++) acc[i] = bias0;
#pragma unroll
        for (int ck = 0; ck < NCH; ck++) {
            half8 pf[MFH];
#pragma unroll
            for (int i = 0; i < MFH; i++) pf[i] = *(const half8 *)(smem + OFF_P + ck * P_BYTES + f_in[i]);
#pragma unroll
            for (int i = 0; i < MFH; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[ck], pf[i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MFH; i++) {
            f32x4 v = acc[i];
            if (a.res) v += __builtin_convertvector(*(const half4 *)(smem + f_r[i]), f32x4);
            half4 h = __builtin_convertvector(v, half4);
            if (!interior) {
                const int py = f_yx[i] & 255, px = (f_yx[i] >> 8) & 255;
                if (!((unsigned)(ty * TO - 1 + py) < (unsigned)a.H && (unsigned)(tx * TO - 1 + px) < (unsigned)a.W)) h = half4{0, 0, 0, 0};   // the 3x3 conv's zero padding
            }
            if (i < MFH - 1 || f_yx[i] >= 0) *(half4 *)(smem + OFF_X + f_x[i]) = h;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // B1: the lateral is complete; the patch buffer is free
        {
            const bool nlive = it + 1 < my_items;
            issue_patch(nlive ? item + gridDim.x : 0, nlive);   // the next item's patch arrives under stage C
        }
        if (a.lat) write_tile(smem + OFF_X, rs_lat, n, ty, tx, false);

        // ================= C: the 3x3 conv on rows 8 h .. 8 h + 7 of the tile =================
#pragma unroll
        for (int r = 0; r < 8; r++) acc[r] = bias1;
        conv_phase(OFF_X + hh * (8 * PW * 64));
        {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            char *sp = smem + OFF_STG + (hh * 8 * 16 + fr) * ROWB + (((cw * 2 + (q4 >> 1) + fr) % CPX) << 4) + (q4 & 1) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) *(half4 *)(sp + i * (16 * ROWB)) = __builtin_convertvector(acc[i], half4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pn = n; pty = ty; ptx = tx;
    }
    raw_barrier();                                              // the last tile is staged
    write_tile(smem + OFF_STG, rs_out, pn, pty, ptx, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (surplus pieces target this workgroup's LDS: drain before exit)
}

}  // namespace

// c [B, H, W, Cin_p] fp16 (Cin_p = 64 | 96), res [B, rH, rW, 64] or NULL -> out [B, H, W, 64] (+ lat [B, H, W, 64] or NULL)
bool lat_fpn_applicable(int Cin_p, int H, int W, int rH, int rW, bool has_res) {
    // the coarser lateral is read at (y >> 1, x >> 1): only an exact 2x level is the reference PAFPN's nearest interpolation
    return (Cin_p == 64 || Cin_p == 96) && H >= 3 && W >= 3 && (!has_res || (H == 2 * rH && W == 2 * rW));
}
int lat_fpn_launch(fid_ctx *ctx, const void *in, int B, int H, int W, int Cin_p, const void *w0, const float *b0, const void *res, int rH, int rW, const void *w1,
                   const float *b1, void *out, void *lat) {
    FID_REQUIRE(in && w0 && b0 && w1 && b1 && out && B > 0 && lat_fpn_applicable(Cin_p, H, W, rH, rW, res != nullptr), "lat_fpn: bad arguments (%d channels, %d x %d)", Cin_p, H, W);
    LFArgs a{};
    a.in = in; a.w0 = w0; a.b0 = b0; a.res = res; a.w1 = w1; a.b1 = b1; a.out = out; a.lat = lat;
    a.H = H; a.W = W; a.rH = rH; a.rW = rW;
    a.tiles_x = cdiv(W, TO);
    a.tiles_per_img = a.tiles_x * cdiv(H, TO);
    a.n_tiles = B * a.tiles_per_img;
    a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    const size_t ib = (size_t)B * H * W * Cin_p * 2, ob = (size_t)B * H * W * 64 * 2, rb = (size_t)B * rH * rW * 64 * 2;
    FID_REQUIRE(ib <= OOB - 4096 && ob <= OOB && rb <= OOB, "lat_fpn: tensor larger than 2 GiB");
    a.in_bytes = (unsigned)ib; a.out_bytes = (unsigned)ob; a.res_bytes = (unsigned)rb;
    const int grid = std::min(a.n_tiles, ctx->num_cus);
    if (Cin_p == 96) {
        constexpr int LDS = 3 * P_BYTES + R_BYTES + 1024 + 2 * X_BYTES + STG_BYTES;
        FID_TRY(ensure_dyn_lds(ctx, (const void *)lat_fpn<3>, LDS));
        hipLaunchKernelGGL(lat_fpn<3>, dim3(grid), dim3(NT), LDS, ctx->stream, a);
    } else {
        constexpr int LDS = 2 * P_BYTES + R_BYTES + 1024 + 2 * X_BYTES + STG_BYTES;
        FID_TRY(ensure_dyn_lds(ctx, (const void *)lat_fpn<2>, LDS));
        hipLaunchKernelGGL(lat_fpn<2>, dim3(grid), dim3(NT), LDS, ctx->stream, a);
    }
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
