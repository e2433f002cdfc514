// Fused SCRFD "deep stem", second design: uint8 frame -> conv3x3/s2+ReLU -> conv3x3+ReLU -> conv3x3+ReLU -> maxpool3x3/s2 in one
// kernel with ROW-structured tiles, all filter banks in registers and the pooling done in registers.
//
// What bounded the first design (stem_fused.hip: 8x8 pooled tiles, every stage on FLATTENED 16-pixel fragments): a flattened fragment
// has no geometry, so each of the 9 taps of the two big stages re-reads its pixel operand AND its weight fragments from LDS -- 7 (5)
// ds_read_b128 per 12 (6) MFMAs, 146-213 B/clk/CU against the 256 B/clk the LDS delivers, 38 % of its cycles lost to bank conflicts
// (row wraps inside a fragment break the XOR swizzle) -- and every stage runs on all 8 waves in lock-step behind five barriers per
// tile: MFMA-busy 30 %, 680 us per 64 frames (17 % of the MFMA peak for the algorithmic work).
//
// Here a workgroup (4 waves) produces a 6-wide x PY-high tile of POOLED pixels:
//   conv2 region (2PY+1) x 13, conv1 region (2PY+3) x 15, conv0 region (2PY+5) x 17 pixels of the stride-2 maps: conv1 / conv2 rows are
//   ONE 16-pixel MFMA fragment each, so the row-sharing tap order of conv_chunked.hip applies (the fragment of map row r shifted by
//   dx feeds output rows r - dy: 3 LDS reads per output row instead of 9) and there are no row wraps inside a fragment;
//   all weights live in registers (conv0: 2 fragments, conv1 / conv2: the 9 taps of the wave's 16 couts = 36 VGPRs each) -- no weight
//   reads from LDS at all;
//   conv0 (K = 27) gathers its pixel operand from the uint8 patch (stored as exact integers 2p-255 in fp16, dword-window layout) as
//   4 aligned ds_read_b32 per lane: K is ordered k = 10 dy + e with the 10 halfs of a patch row's window starting one half early
//   (weight 0 there), which makes every run dword aligned; it stays on flattened fragments (17 columns do not fit one fragment and
//   it has no taps to share);
//   the 3x3/s2 max-pool runs on the accumulators: rows in registers (a wave holds every conv2 row of its 16 couts), columns with two
//   DPP row shifts -- conv2's output never goes to LDS; only the pooled tile is staged for 16-byte row stores.
// Two workgroups per CU (61 KB of LDS each at PY = 8) run out of phase, so one's matrix phases overlap the other's conversion / gather /
// pooling / store phases.  Halo recompute: 1.27x the MACs of the unfused stem (1.23x before), 20.9 MFMAs per pooled pixel (18.0 before).
// Semantics as in stem_fused.hip: map positions outside the real feature map are written as 0 (the next conv's zero padding; harmless
// for the pool because every value is post-ReLU >= 0).  reference models/scrfd.py:76-83 (blobFromImage + the first four graph nodes).
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int PXT = 6;                         // pooled tile width
constexpr int CW2 = 13, CW1 = 15, CW0 = 17;    // region widths (stride-2 map columns) of conv2 / conv1 / conv0
constexpr int PW0 = 18, PW1 = 16;              // LDS pixel pitch of the conv0 / conv1 maps
constexpr int DROW = 27, RS = 108;             // dwords / halfs per input patch row: 35 pixels x 3 bytes + 3 bytes of lead-in

struct StemRArgs {
    const uint8_t *img;       // [B, H, W, 3] BGR
    const _Float16 *w0;       // [32][32]  (k = tap*3 + c, zero padded), scale/2 + BN folded
    const float *b0;
    const _Float16 *w1;       // [32][9][32]
    const float *b1;
    const _Float16 *w2;       // [C2P][9][32]
    const float *b2;
    _Float16 *out;            // [B, Hp, Wp, C2P]
    int H, W, H1, W1, Hp, Wp; // frame, stride-2 maps, pooled map
    int tiles_x, tiles_y, n_tiles;
    int stagger;              // start delay of the second half of the grid in units of 64 cycles (FID_STEM_STAGGER; default: half a tile)
    int ablate;               // FID_STEM_ABLATE timing experiments (wrong results): 1 conv0, 2 conv1, 4 conv2 + pool, 8 stores, 16 input
};

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
// lane i of a 16-lane row <- lane i + N of the same row (0 past the row's end)
template <int N>
__device__ __forceinline__ float row_shl(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + N, 0xF, 0xF, true));
}

#ifndef STEM_WPS
#define STEM_WPS 2
#endif
// -DSTEM_RELOAD_W1=1 (with -DSTEM_WPS=3 and FID_STEM_PY=6): conv1's nine weight fragments are re-read from global memory (L1 / L2 hits) after
// every tile's write-out instead of living in registers through conv2 + pooling, the phase with the most live registers -- the 36 VGPRs that
// kept twelve waves per CU from fitting under the 168-register cap without scratch (round 3: 68 B of scratch per lane, 665 -> 725 us)
#ifndef STEM_RELOAD_W1
#define STEM_RELOAD_W1 0
#endif
// (PY = 6 tiles need 50 KB of LDS: three workgroups fit a CU if the registers allow three waves per SIMD: -DSTEM_WPS=3 forces <= 168 VGPRs there)
template <int C2P, int PY>
__global__ void __launch_bounds__(256, PY == 6 ? STEM_WPS : 2) scrfd_stem_rows(const StemRArgs a) {
    constexpr int R2 = 2 * PY + 1, R1 = R2 + 2, R0 = R2 + 4, RI = 2 * R0 + 1;
    constexpr int N0 = R0 * CW0, NF0 = (N0 + 15) / 16;             // conv0: flattened pixels / fragments
    constexpr int NF2 = C2P / 16, NG2 = 4 / NF2;                   // conv2: cout fragments (4 | 2), row groups (1 | 2)
    constexpr int ROWS2 = NG2 == 1 ? R2 : PY + 1;                  // conv2 rows per wave (the two groups share row PY)
    constexpr int PR = PY / NG2;                                   // pooled rows per wave
    constexpr int ROWS1 = (R1 + 1) / 2;                            // conv1 rows per wave: 2 cout fragments x 2 row groups
    constexpr int ROWB2 = C2P * 2, CPP = ROWB2 / 16;               // bytes / 16-byte chunks of a pooled pixel
    constexpr int IN_BYTES = (RI * RS * 2 + 16 + 255) / 256 * 256;
    constexpr int C0_BYTES = (R0 + 1) * PW0 * 64, C1_BYTES = (R1 + 1) * PW1 * 64 + 256, STG_BYTES = PY * PXT * ROWB2;
    constexpr int OFF_IN = 0, OFF_C0 = IN_BYTES, OFF_C1 = OFF_C0 + C0_BYTES, OFF_STG = OFF_C1 + C1_BYTES, OFF_BIAS = OFF_STG + STG_BYTES;
    constexpr int MF0 = (NF0 + 3) / 4;                             // conv0 fragments per wave
    constexpr int OFF_TAB = OFF_BIAS + 512, TAB_DW = MF0;      // per-thread constants of conv0's gather ([entry][thread] dwords), see below
    constexpr int ZD = RI * RS / 2;                                // dword index of the zero dword behind the patch
    static_assert(PY % 2 == 0 && OFF_TAB + TAB_DW * 1024 <= 80 * 1024, "two workgroups per CU");
    static_assert(PY != 6 || STEM_WPS < 3 || OFF_TAB + TAB_DW * 1024 <= 53 * 1024, "three workgroups per CU");
    constexpr int NDW = RI * DROW, DPT = (NDW + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // ---- weights -> registers (MFMA A fragments: lane = (cout & 15, 8-channel group)) ----
    half8 w0f[2], w1f[9], w2f[9];
    const int f1 = wave & 1, g1 = wave >> 1, f2 = wave % NF2, g2 = wave / NF2;
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = fq * 8 + j, dy = k / 10, e = k - dy * 10;              // K order of conv0: 10 halfs per patch row, the first one before the window
            w0f[f][j] = (k < 30 && e >= 1) ? a.w0[(f * 16 + frow) * 32 + dy * 9 + e - 1] : (_Float16)0.f;
        }
#pragma unroll
    for (int t = 0; t < 9; t++) {
        w1f[t] = *(const half8 *)(a.w1 + ((f1 * 16 + frow) * 9 + t) * 32 + fq * 8);
        w2f[t] = *(const half8 *)(a.w2 + ((f2 * 16 + frow) * 9 + t) * 32 + fq * 8);
    }
    // biases: a small LDS table read at the head of each stage's epilogue (16 registers less across the matrix phases)
    float *sB = (float *)(smem + OFF_BIAS);
    if (tid < 32) { sB[tid] = a.b0[tid]; sB[32 + tid] = a.b1[tid]; }
    if (tid < C2P) sB[64 + tid] = a.b2[tid];
    if (tid < 4) ((unsigned *)(smem + OFF_IN))[ZD + tid] = 0u;      // the "k >= 30" operand

    // conv0's gather: the lane's four dwords of a pixel's 15-dword window (3 patch rows x 5 dwords), relative to the pixel's base dword
    int rel[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int d = fq * 4 + j, dy = d / 5;
        rel[j] = dy * (RS / 2) + (d - dy * 5);
    }

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    auto decode = [&](int tile, int &n, int &ty, int &tx) {
        n = tile / tiles_per_img;
        const int r = tile - n * tiles_per_img;
        ty = r / a.tiles_x; tx = r - ty * a.tiles_x;
    };
    // ---- input patch: aligned dwords of the frame rows into registers (next tile), later converted into LDS ----
    // patch row pr = frame row 4 PY ty - 7 + pr; the dword window of a row starts at frame byte 72 tx - 24 (pixel column 24 tx - 7 minus
    // 3 bytes of lead-in): rows and the window are dword aligned (the frame width is a multiple of 4), so a dword is inside or outside as a whole
    unsigned pre[DPT];
    unsigned pre_ok = 0;
    bool pre_int = false;                                          // the prefetched patch lies inside the frame: no dword to clear (wave-uniform)
    // my dword i of a patch = patch row pr, dword dc of its window: its byte offset from the window's first byte is the same for every tile
    // (threads past the patch's last dword: far out of range).  A tile's loads are buffer loads on ITS frame (base and size in SGPRs):
    // one add per dword, and rows above / below the frame are out of range by themselves and arrive as 0.
    const int rowbytes = a.W * 3;
    int v_rel[DPT];
#pragma unroll
    for (int i = 0; i < DPT; i++) {
        const int d = tid + 256 * i;
        const int pr = d / DROW, dc = d - pr * DROW;
        v_rel[i] = d < NDW ? pr * rowbytes + dc * 4 : 0x7FFF0000;
        asm volatile("" : "+v"(v_rel[i]));
    }
    auto prefetch = [&](int tile) {
        int n, ty, tx;
        decode(tile, n, ty, tx);
        const int iy0 = 4 * PY * ty - 7, bx0 = 72 * tx - 24;
        const auto rs_img = __builtin_amdgcn_make_buffer_rsrc((void *)(a.img + (size_t)n * a.H * rowbytes), 0, a.H * rowbytes, 0x00020000);
        const int s_off = iy0 * rowbytes + bx0;
        const bool xin_all = bx0 >= 0 && bx0 + DROW * 4 <= rowbytes;       // every dword of the window lies inside the frame's rows
        pre_int = xin_all && iy0 >= 0 && iy0 + RI <= a.H;
        if (xin_all) {
#pragma unroll
            for (int i = 0; i < DPT; i++) pre[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_img, v_rel[i] + s_off, 0, 0);
        } else {                                                    // a window that sticks out of the rows (left / right tile columns): per-dword test
#pragma unroll
            for (int i = 0; i < DPT; i++) {
                int d = tid + 256 * i;
                asm volatile("" : "+v"(d));                        // (computed here, not hoisted into registers)
                const int pr = d / DROW, bx = bx0 + (d - pr * DROW) * 4;
                pre[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_img, (bx >= 0 && bx + 4 <= rowbytes) ? v_rel[i] + s_off : 0x7FFF0000, 0, 0);
            }
        }
        pre_ok = 0xFFFFFFFFu;
        if (!pre_int) {                                             // which dwords are real pixels (a real 0 is -255, the blob's padding is 0)
            pre_ok = 0;
#pragma unroll
            for (int i = 0; i < DPT; i++) {
                int d = tid + 256 * i;
                asm volatile("" : "+v"(d));
                const int pr = d / DROW, dc = d - pr * DROW;
                const int iy = iy0 + pr, bx = bx0 + dc * 4;
                const bool in = d < NDW && (unsigned)iy < (unsigned)a.H && bx >= 0 && bx + 4 <= rowbytes;
                pre_ok |= in ? (1u << i) : 0u;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            const int d = tid + 256 * i;
            if (d < NDW) {
                const int pr = d / DROW, dc = d - pr * DROW;
                const unsigned v = pre[i];
                half4 h;
                h[0] = (_Float16)fmaf((float)(v & 0xFF), 2.f, -255.f);
                h[1] = (_Float16)fmaf((float)((v >> 8) & 0xFF), 2.f, -255.f);
                h[2] = (_Float16)fmaf((float)((v >> 16) & 0xFF), 2.f, -255.f);
                h[3] = (_Float16)fmaf((float)(v >> 24), 2.f, -255.f);
                if (!pre_int && !((pre_ok >> i) & 1u)) h = half4{0, 0, 0, 0};    // outside the frame: the blob's zero padding (a real pixel 0 is -255)
                *(half4 *)(smem + OFF_IN + (pr * RS + dc * 4) * 2) = h;
            }
        }
    };

    // ---- pixel fragment addressing of the two map stages (conv3x3_wr's scheme): lin = K + frow, the lane's swizzled 16-byte group is
    // selected by K & 1 and (K >> 1) & 3; `rot` = the same bits of a runtime row offset that is not a multiple of 8 pixels
    auto make_pb = [&](int rot, int off, int (&pb)[2][4]) {
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c + rot) & 3)) << 4) + off;
                asm volatile("" : "+v"(pb[par][c]));
            }
    };
    constexpr int NACC = ROWS2 > ROWS1 ? ROWS2 : ROWS1;
    f32x4 acc[NACC];
#ifndef STEM_PD
#define STEM_PD 3
#endif
    auto conv_rows = [&](const int (&pb)[2][4], auto rows_tag, auto pw_tag, const half8 (&wv)[9]) {
        constexpr int ROWS = decltype(rows_tag)::value, PWV = decltype(pw_tag)::value, PH = ROWS + 2;
        constexpr int PD = STEM_PD;                                 // fragments read ahead (a fragment feeds 3 MFMAs = 48 cycles; an LDS read takes > 100)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            half8 pq[PD + 1];
            auto load_p = [&](int r, int set) {
                const int K = r * PWV + dx;
                pq[set] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + K * 64));
            };
#pragma unroll
            for (int r = 0; r < PD; r++) load_p(r, r % (PD + 1));
#pragma unroll
            for (int r = 0; r < PH; r++) {
                if (r + PD < PH) load_p(r + PD, (r + PD) % (PD + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int mi = r - dy;
                    if (mi < 0 || mi >= ROWS) continue;
                    acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[dy * 3 + dx], pq[r % (PD + 1)], acc[mi], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using std::integral_constant;

    // conv0's fragments (wave, wave + 4, ...) are the same pixels of every tile: the window's LDS byte address and the lane's result address
    // (cout fragment 0; fragment 1 = ^ 32: the swizzled 16-byte group index differs in bit 1) are computed ONCE -- per tile they cost ~35 VALU
    // instructions per fragment (division by 17, swizzle, bounds), a quarter of this stage.  They live in a per-thread LDS table, not in
    // registers (the matrix phases have none to spare: held in registers they are spilled to scratch).  Lanes past the region (the last
    // fragment's tail) gather pixel N0 - 1 and write to the slack row R0 of the map, which only feeds conv1's discarded row.
    static_assert(OFF_C0 + C0_BYTES < 65536 && IN_BYTES < 65536, "both addresses of a fragment share one dword");
    unsigned *sT = (unsigned *)(smem + OFF_TAB) + tid;
#pragma unroll
    for (int i = 0; i < MF0; i++) {
        const int qd = (wave + 4 * i) * 16 + frow, q = qd < N0 ? qd : N0 - 1;
        const int y = q / CW0, x = q - y * CW0;
        const int lin = qd < N0 ? y * PW0 + x : R0 * PW0 + frow;
        const int ga = OFF_IN + (y * RS + 3 * x + 1) * 4;
        const int sa = OFF_C0 + lin * 64 + ((((fq >> 1)) ^ swz64(lin)) << 4) + (fq & 1) * 8;
        sT[i * 256] = (unsigned)ga | ((unsigned)sa << 16);         // gather address | result address << 16
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { rel[j] *= 4; asm volatile("" : "+v"(rel[j])); }      // (byte offsets of the lane's four window dwords: four registers, as before)

    int tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(tile);
    // The two workgroups of a CU run the same program on tiles of equal cost: started together they stay in lock-step -- both in a matrix
    // phase (sharing the pipes), then both in a conversion / pooling phase (pipes idle).  The second half of the grid (the co-resident
    // workgroups under round-robin dispatch; speed only) starts `stagger` x 64 cycles late, about half a tile.
    if (a.stagger > 0 && (int)blockIdx.x >= (int)gridDim.x / 2) {
        for (int i = 0; i < a.stagger; i += 100) __builtin_amdgcn_s_sleep(100);
    }

    for (; tile < a.n_tiles; tile += gridDim.x) {
        int n, ty, tx;
        decode(tile, n, ty, tx);
        const int py0 = ty * PY, px0 = tx * PXT;
        const int oy0 = 2 * py0 - 3, ox0 = 2 * px0 - 3;            // conv0 region origin in the stride-2 map (conv1: +1, conv2: +2)
        // every position of the three regions lies inside the stride-2 map (no padding to write): the masks below are skipped (wave-uniform)
        const bool interior = oy0 >= 0 && ox0 >= 0 && oy0 + R0 <= a.H1 && ox0 + CW0 + 1 <= a.W1;

        if (!(a.ablate & 16)) commit();
        {
            const int next = tile + gridDim.x;
            if (next < a.n_tiles && !(a.ablate & 16)) prefetch(next);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                             // B1: the patch is in LDS (and the write-out of the tile before has read its staging)

        // ---------------- S1: conv0 (K = 27 in 30 slots), stride 2: flattened fragments wave, wave + 4, ... ----------------
        if (!(a.ablate & 1)) {
            // FB fragments at a time: their gathers (one LDS round trip), their MFMAs (the bias is the C operand), then their epilogues --
            // an MFMA's result is not touched before the batch's other MFMAs have issued
            constexpr int FB = 3;
            int lt = tid;
            asm volatile("" : "+v"(lt));                           // opaque: the table address is formed here, per tile
            const unsigned *tp = (const unsigned *)(smem + OFF_TAB) + lt;
            unsigned g_as[MF0];
#pragma unroll
            for (int i = 0; i < MF0; i++) g_as[i] = tp[i * 256];
            const int (&relb)[4] = rel;
            f32x4 bias0[2];
#pragma unroll
            for (int f = 0; f < 2; f++) bias0[f] = *(const f32x4 *)(sB + f * 16 + fq * 4);
#pragma unroll
            for (int i0 = 0; i0 < MF0; i0 += FB) {
                u32x4 pv[FB];
#pragma unroll
                for (int i = 0; i < FB; i++) {
                    if (i0 + i >= MF0) continue;
                    const int ga = (int)(g_as[i0 + i] & 0xFFFFu);
                    pv[i][0] = *(const unsigned *)(smem + (ga + relb[0]));
                    pv[i][1] = *(const unsigned *)(smem + (ga + relb[1]));
                    pv[i][2] = *(const unsigned *)(smem + (ga + relb[2]));
                    pv[i][3] = *(const unsigned *)(smem + (fq == 3 ? OFF_IN + ZD * 4 : ga + relb[3]));      // (quarter 3: the zero dword behind the patch)
                }
                f32x4 c[FB][2];
#pragma unroll
                for (int i = 0; i < FB; i++) {
                    if (i0 + i >= MF0 || wave + 4 * (i0 + i) >= NF0) continue;     // (wave-uniform)
                    const half8 pf = __builtin_bit_cast(half8, pv[i]);
#pragma unroll
                    for (int f = 0; f < 2; f++) c[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[f], pf, bias0[f], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < FB; i++) {
                    if (i0 + i >= MF0 || wave + 4 * (i0 + i) >= NF0) continue;
                    bool inside = true;
                    if (!interior) {                               // (wave-uniform: border tiles only) positions outside the stride-2 map are conv1's zero padding
                        int lo = lane;
                        asm volatile("" : "+v"(lo));               // opaque: computed here, on the few border tiles, not hoisted out of the tile loop (and spilled)
                        const int qd = (wave + 4 * (i0 + i)) * 16 + (lo & 15), q = qd < N0 ? qd : N0 - 1;
                        const int y = q / CW0, x = q - y * CW0;
                        inside = (unsigned)(oy0 + y) < (unsigned)a.H1 && (unsigned)(ox0 + x) < (unsigned)a.W1;
                    }
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        half4 h = __builtin_elementwise_max(__builtin_convertvector(c[i][f], half4), half4{0, 0, 0, 0});   // ReLU after the rounding: same result
                        if (!inside) h = half4{0, 0, 0, 0};
                        *(half4 *)(smem + ((g_as[i0 + i] >> 16) ^ (f * 32))) = h;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                             // B2: the conv0 map is complete

        // ---------------- S2: conv1 3x3, 32 -> 32: wave = (cout fragment f1, row group g1 of ROWS1 rows) ----------------
        if (!(a.ablate & 2)) {
            const int r_lo = g1 * ROWS1;
            int pb[2][4];
            make_pb(((r_lo * PW0) >> 1) & 3, OFF_C0 + r_lo * PW0 * 64, pb);
            {
                const f32x4 bias1 = *(const f32x4 *)(sB + 32 + f1 * 16 + fq * 4);      // the accumulators start from the bias: no add in the epilogue
#pragma unroll
                for (int r = 0; r < ROWS1; r++) acc[r] = bias1;
            }
            conv_rows(pb, integral_constant<int, ROWS1>{}, integral_constant<int, PW0>{}, w1f);
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const bool xin = fr < CW1 && (unsigned)(ox0 + 1 + fr) < (unsigned)a.W1;
            char *cp = smem + OFF_C1 + (r_lo * PW1 + fr) * 64 + (((f1 * 2 + (q4 >> 1)) ^ swz64(fr)) << 4) + (q4 & 1) * 8;
#pragma unroll
            for (int i = 0; i < ROWS1; i++) {
                if (r_lo + i >= R1) continue;                      // (wave-uniform: the second row group of an odd row count)
                half4 h = __builtin_elementwise_max(__builtin_convertvector(acc[i], half4), half4{0, 0, 0, 0});
                if (!interior && !(xin && (unsigned)(oy0 + 1 + r_lo + i) < (unsigned)a.H1)) h = half4{0, 0, 0, 0};
                *(half4 *)(cp + i * (PW1 * 64)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                             // B3: the conv1 map is complete

        // ---------------- S3: conv2 3x3, 32 -> C2P, then the 3x3 / stride-2 max-pool on the accumulators ----------------
        if (!(a.ablate & 4)) {
            const int r_lo = g2 * PY;                              // (a multiple of 8 rows x 16 pixels: no swizzle rotation)
            int pb[2][4];
            make_pb(0, OFF_C1 + r_lo * PW1 * 64, pb);
            {
                const f32x4 bias2 = *(const f32x4 *)(sB + 64 + f2 * 16 + fq * 4);
#pragma unroll
                for (int r = 0; r < ROWS2; r++) acc[r] = bias2;
            }
            conv_rows(pb, integral_constant<int, ROWS2>{}, integral_constant<int, PW1>{}, w2f);
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int fr = lo & 15, q4 = lo >> 4;
            const bool xin = fr < CW2 && (unsigned)(ox0 + 2 + fr) < (unsigned)a.W1;
            const int gy0 = oy0 + 2 + r_lo;
            // The pool runs on the fp32 sums (bias included: the accumulators started from it): max commutes with ReLU and the fp16 rounding
            // (both monotone), so relu(round(max)) = max of the rounded, activated values.  Positions outside the map must not win: they
            // become -inf (every window holds a real pixel); only tiles on the map border have any.
            if (!interior) {
                const float ninf = -__builtin_inff();
#pragma unroll
                for (int i = 0; i < ROWS2; i++)
                    if (!(xin && (unsigned)(gy0 + i) < (unsigned)a.H1)) acc[i] = f32x4{ninf, ninf, ninf, ninf};
            }
            const bool st = (fr & 1) == 0 && fr < 2 * PXT;
            char *sp = smem + OFF_STG + ((g2 * PR) * PXT + (fr >> 1)) * ROWB2 + (f2 * 16 + q4 * 4) * 2;
#pragma unroll
            for (int p = 0; p < PR; p++) {
                f32x4 m;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = fmaxf(fmaxf(acc[2 * p][e], acc[2 * p + 1][e]), acc[2 * p + 2][e]);     // (v_max3_f32)
                    m[e] = fmaxf(fmaxf(v, row_shl<1>(v)), row_shl<2>(v));
                }
                const half4 h = __builtin_elementwise_max(__builtin_convertvector(m, half4), half4{0, 0, 0, 0});
                if (st) *(half4 *)(sp + p * (PXT * ROWB2)) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                             // B4: the pooled tile is staged

        // ---------------- write-out: 16 bytes per lane, whole pixel rows ----------------
        if (!(a.ablate & 8)) {
#pragma unroll
            for (int i = 0; i < (PY * PXT * CPP + 255) / 256; i++) {
                const int s = tid + 256 * i;
                const int pix = s / CPP, c = s - pix * CPP;
                const int p = pix / PXT, q = pix - p * PXT;
                const int gy = py0 + p, gx = px0 + q;
                if (s < PY * PXT * CPP && gy < a.Hp && gx < a.Wp)
                    *(u32x4 *)((char *)a.out + (((size_t)n * a.Hp + gy) * a.Wp + gx) * ROWB2 + c * 16) = *(const u32x4 *)(smem + OFF_STG + pix * ROWB2 + c * 16);
            }
        }
        if (STEM_RELOAD_W1 && PY == 6) {                          // (the pointer is made opaque per tile: the loads must not be hoisted back out of the loop)
            const _Float16 *wp = a.w1;
            asm volatile("" : "+s"(wp));
#pragma unroll
            for (int t = 0; t < 9; t++) w1f[t] = *(const half8 *)(wp + ((f1 * 16 + frow) * 9 + t) * 32 + fq * 8);
        }
    }
}


// ---- the same stem with the two halves of a tile's work on DIFFERENT waves, one tile apart ------------------------------------------------
// In scrfd_stem_rows every wave walks all stages of its tile: 255 MFMAs (4 080 cycles) inside ~20 000 cycles of conversion, gather, epilogues,
// pooling and barriers, and the second workgroup of the CU -- same program, same phase -- fills the pipe to ~40 %.  Here a workgroup has 8 waves
// in two ROLES: waves 0-3 ("front") convert the input patch and run conv0 + conv1 of tile t + 1 while waves 4-7 ("back") run conv2 + the pool +
// the write-out of tile t from the conv1 map the front waves left in the other of two LDS buffers.  Each SIMD then holds one front and one back
// wave whose matrix and VALU phases interleave by construction; a front wave keeps 44 registers of weights, a back wave 36.  Two workgroup
// barriers per period, executed by every wave whether or not it has a tile (the work is guarded, never the barriers):
//   front: conv0(t+1)                      | B_a | conv1(t+1) -> C1[(t+1) & 1]; patch of t+2 -> LDS; prefetch t+3 | B_b
//   back:  stores of t-1; conv2(t) cols 0,1 | B_a | conv2(t) col 2; pool -> staging                                | B_b
template <int C2P, int PY>
__global__ void __launch_bounds__(512, 2) scrfd_stem_roles(const StemRArgs a) {
    constexpr int R2 = 2 * PY + 1, R1 = R2 + 2, R0 = R2 + 4, RI = 2 * R0 + 1;
    constexpr int N0 = R0 * CW0, NF0 = (N0 + 15) / 16;
    constexpr int NF2 = C2P / 16, NG2 = 4 / NF2;
    constexpr int ROWS2 = NG2 == 1 ? R2 : PY + 1;
    constexpr int PR = PY / NG2;
    constexpr int ROWS1 = (R1 + 1) / 2;
    constexpr int ROWB2 = C2P * 2, CPP = ROWB2 / 16;
    constexpr int IN_BYTES = (RI * RS * 2 + 16 + 255) / 256 * 256;
    constexpr int C0_BYTES = (R0 + 1) * PW0 * 64, C1_BYTES = (R1 + 1) * PW1 * 64 + 256, STG_BYTES = PY * PXT * ROWB2;
    constexpr int OFF_IN = 0, OFF_C0 = IN_BYTES, OFF_C1 = OFF_C0 + C0_BYTES, OFF_STG = OFF_C1 + 2 * C1_BYTES, OFF_BIAS = OFF_STG + STG_BYTES;
    constexpr int ZD = RI * RS / 2;
    static_assert(PY % 2 == 0 && OFF_BIAS + 512 <= 160 * 1024, "LDS budget");
    constexpr int NDW = RI * DROW, DPT = (NDW + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, gw = wave & 3, gtid = tid & 255;      // role 0 = front (conv0 + conv1), 1 = back (conv2 + pool + stores)
    const int frow = lane & 15, fq = lane >> 4;
    const int f1 = gw & 1, g1 = gw >> 1, f2 = gw % NF2, g2 = gw / NF2;

    // ---- weights -> registers: a front wave holds conv0's two fragments and conv1's nine, a back wave conv2's nine (in the same array) ----
    half8 w0f[2], wk[9];
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = fq * 8 + j, dy = k / 10, e = k - dy * 10;
            w0f[f][j] = (role == 0 && k < 30 && e >= 1) ? a.w0[(f * 16 + frow) * 32 + dy * 9 + e - 1] : (_Float16)0.f;
        }
#pragma unroll
    for (int t = 0; t < 9; t++)
        wk[t] = role == 0 ? *(const half8 *)(a.w1 + ((f1 * 16 + frow) * 9 + t) * 32 + fq * 8) : *(const half8 *)(a.w2 + ((f2 * 16 + frow) * 9 + t) * 32 + fq * 8);
    float *sB = (float *)(smem + OFF_BIAS);
    if (tid < 32) { sB[tid] = a.b0[tid]; sB[32 + tid] = a.b1[tid]; }
    if (tid < C2P) sB[64 + tid] = a.b2[tid];
    if (tid < 4) ((unsigned *)(smem + OFF_IN))[ZD + tid] = 0u;
    int rel[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int d = fq * 4 + j, dy = d / 5;
        rel[j] = dy * (RS / 2) + (d - dy * 5);
    }

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    auto decode = [&](int tile, int &n, int &ty, int &tx) {
        n = tile / tiles_per_img;
        const int r = tile - n * tiles_per_img;
        ty = r / a.tiles_x; tx = r - ty * a.tiles_x;
    };
    unsigned pre[DPT];
    unsigned pre_ok = 0;
    auto prefetch = [&](int tile) {                             // (front threads only: gtid indexes the patch)
        int n, ty, tx;
        decode(tile, n, ty, tx);
        const int iy0 = 4 * PY * ty - 7, bx0 = 72 * tx - 24, rowbytes = a.W * 3;
        const uint8_t *base = a.img + (size_t)n * a.H * rowbytes;
        pre_ok = 0;
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            const int d = gtid + 256 * i;
            const int pr = d / DROW, dc = d - pr * DROW;
            const int iy = iy0 + pr, bx = bx0 + dc * 4;
            const bool in = d < NDW && (unsigned)iy < (unsigned)a.H && bx >= 0 && bx + 4 <= rowbytes;
            pre[i] = in ? *(const unsigned *)(base + (size_t)iy * rowbytes + bx) : 0u;
            pre_ok |= in ? (1u << i) : 0u;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            const int d = gtid + 256 * i;
            if (d < NDW) {
                const int pr = d / DROW, dc = d - pr * DROW;
                const unsigned v = pre[i];
                half4 h;
                h[0] = (_Float16)fmaf((float)(v & 0xFF), 2.f, -255.f);
                h[1] = (_Float16)fmaf((float)((v >> 8) & 0xFF), 2.f, -255.f);
                h[2] = (_Float16)fmaf((float)((v >> 16) & 0xFF), 2.f, -255.f);
                h[3] = (_Float16)fmaf((float)(v >> 24), 2.f, -255.f);
                if (!((pre_ok >> i) & 1u)) h = half4{0, 0, 0, 0};
                *(half4 *)(smem + OFF_IN + (pr * RS + dc * 4) * 2) = h;
            }
        }
    };
    auto make_pb = [&](int rot, int off, int (&pb)[2][4]) {
#pragma unroll
        for (int par = 0; par < 2; par++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                pb[par][c] = frow * 64 + ((fq ^ ((((frow + par) >> 1) + c + rot) & 3)) << 4) + off;
                asm volatile("" : "+v"(pb[par][c]));
            }
    };
    constexpr int NACC = ROWS2 > ROWS1 ? ROWS2 : ROWS1;
    f32x4 acc[NACC];
    // tap columns dx0 .. dx1 - 1 of ROWS output rows
    auto conv_cols = [&](const int (&pb)[2][4], auto rows_tag, auto pw_tag, int dx0, int dx1) {
        constexpr int ROWS = decltype(rows_tag)::value, PWV = decltype(pw_tag)::value, PH = ROWS + 2;
        constexpr int PD = STEM_PD;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            if (dx < dx0 || dx >= dx1) continue;
            half8 pq[PD + 1];
            auto load_p = [&](int r, int set) {
                const int K = r * PWV + dx;
                pq[set] = *(const half8 *)(smem + (pb[K & 1][(K >> 1) & 3] + K * 64));
            };
#pragma unroll
            for (int r = 0; r < PD; r++) load_p(r, r % (PD + 1));
#pragma unroll
            for (int r = 0; r < PH; r++) {
                if (r + PD < PH) load_p(r + PD, (r + PD) % (PD + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int mi = r - dy;
                    if (mi < 0 || mi >= ROWS) continue;
                    acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[dy * 3 + dx], pq[r % (PD + 1)], acc[mi], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using std::integral_constant;

    const int my_tiles = (int)blockIdx.x < a.n_tiles ? (a.n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    auto tile_of = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    if (role == 0 && my_tiles > 0) {
        prefetch(tile_of(0));
        commit();
        if (my_tiles > 1) prefetch(tile_of(1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                            // the first patch, the bias table, the zero dword

    for (int p = 0; p <= my_tiles + 1; p++) {
        // ================= first half of the period =================
        if (role == 0) {
            if (p < my_tiles && !(a.ablate & 1)) {              // ---- conv0 of tile p (the patch was converted in the period before) ----
                int n, ty, tx;
                decode(tile_of(p), n, ty, tx);
                const int oy0 = 2 * ty * PY - 3, ox0 = 2 * tx * PXT - 3;
                constexpr int MF0 = (NF0 + 3) / 4, FB = 3;
                const unsigned *ip = (const unsigned *)(smem + OFF_IN);
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const int fr_ = lo & 15, q4 = lo >> 4;
                f32x4 bias0[2];
#pragma unroll
                for (int f = 0; f < 2; f++) bias0[f] = *(const f32x4 *)(sB + f * 16 + q4 * 4);
#pragma unroll
                for (int i0 = 0; i0 < MF0; i0 += FB) {
                    u32x4 pv[FB];
                    int lin[FB];
                    bool inside[FB];
#pragma unroll
                    for (int i = 0; i < FB; i++) {
                        const int fi = gw + 4 * (i0 + i);
                        const int qd = fi * 16 + fr_, q = qd < N0 ? qd : N0 - 1;
                        const int y = q / CW0, x = q - y * CW0;
                        const int bdw = y * RS + 3 * x + 1;
                        pv[i][0] = ip[bdw + rel[0]];
                        pv[i][1] = ip[bdw + rel[1]];
                        pv[i][2] = ip[bdw + rel[2]];
                        pv[i][3] = ip[q4 == 3 ? ZD : bdw + rel[3]];
                        inside[i] = qd < N0 && (unsigned)(oy0 + y) < (unsigned)a.H1 && (unsigned)(ox0 + x) < (unsigned)a.W1;
                        lin[i] = qd < N0 ? y * PW0 + x : -1;
                    }
#pragma unroll
                    for (int i = 0; i < FB; i++) {
                        if (i0 + i >= MF0 || gw + 4 * (i0 + i) >= NF0) continue;
                        const half8 pf = __builtin_bit_cast(half8, pv[i]);
#pragma unroll
                        for (int f = 0; f < 2; f++) {
                            f32x4 c = {0.f, 0.f, 0.f, 0.f};
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0f[f], pf, c, 0, 0, 0);
                            half4 h = __builtin_elementwise_max(__builtin_convertvector(c + bias0[f], half4), half4{0, 0, 0, 0});
                            if (!inside[i]) h = half4{0, 0, 0, 0};
                            if (lin[i] >= 0) *(half4 *)(smem + OFF_C0 + lin[i] * 64 + (((f * 2 + (q4 >> 1)) ^ swz64(lin[i])) << 4) + (q4 & 1) * 8) = h;
                        }
                    }
                }
            }
        } else {
            if (p >= 2 && !(a.ablate & 8)) {                    // ---- stores of tile p - 2 (staged in the period before) ----
                int n, ty, tx;
                decode(tile_of(p - 2), n, ty, tx);
                const int py0 = ty * PY, px0 = tx * PXT;
#pragma unroll
                for (int i = 0; i < (PY * PXT * CPP + 255) / 256; i++) {
                    const int s_ = gtid + 256 * i;
                    const int pix = s_ / CPP, c = s_ - pix * CPP;
                    const int pp = pix / PXT, q = pix - pp * PXT;
                    const int gy = py0 + pp, gx = px0 + q;
                    if (s_ < PY * PXT * CPP && gy < a.Hp && gx < a.Wp)
                        *(u32x4 *)((char *)a.out + (((size_t)n * a.Hp + gy) * a.Wp + gx) * ROWB2 + c * 16) = *(const u32x4 *)(smem + OFF_STG + pix * ROWB2 + c * 16);
                }
            }
            if (p >= 1 && p <= my_tiles && !(a.ablate & 4)) {   // ---- conv2 of tile p - 1, tap columns 0 and 1 ----
                int pb[2][4];
                make_pb(0, OFF_C1 + ((p - 1) & 1) * C1_BYTES + g2 * PY * PW1 * 64, pb);
#pragma unroll
                for (int r = 0; r < ROWS2; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                conv_cols(pb, integral_constant<int, ROWS2>{}, integral_constant<int, PW1>{}, 0, 2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // B_a: conv0's map of tile p is complete; the staged tile p - 2 has been read

        // ================= second half =================
        if (role == 0) {
            if (p < my_tiles) {
                int n, ty, tx;
                decode(tile_of(p), n, ty, tx);
                const int oy0 = 2 * ty * PY - 3, ox0 = 2 * tx * PXT - 3;
                const bool interior = oy0 >= 0 && ox0 >= 0 && oy0 + R0 <= a.H1 && ox0 + CW0 + 1 <= a.W1;
                if (!(a.ablate & 2)) {                          // ---- conv1 of tile p -> C1[p & 1] ----
                    const int r_lo = g1 * ROWS1;
                    int pb[2][4];
                    make_pb(((r_lo * PW0) >> 1) & 3, OFF_C0 + r_lo * PW0 * 64, pb);
#pragma unroll
                    for (int r = 0; r < ROWS1; r++) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    conv_cols(pb, integral_constant<int, ROWS1>{}, integral_constant<int, PW0>{}, 0, 3);
                    int lo = lane;
                    asm volatile("" : "+v"(lo));
                    const int fr = lo & 15, q4 = lo >> 4;
                    const f32x4 bias1 = *(const f32x4 *)(sB + 32 + f1 * 16 + q4 * 4);
                    const bool xin = fr < CW1 && (unsigned)(ox0 + 1 + fr) < (unsigned)a.W1;
                    char *cp = smem + OFF_C1 + (p & 1) * C1_BYTES + (r_lo * PW1 + fr) * 64 + (((f1 * 2 + (q4 >> 1)) ^ swz64(fr)) << 4) + (q4 & 1) * 8;
#pragma unroll
                    for (int i = 0; i < ROWS1; i++) {
                        if (r_lo + i >= R1) continue;
                        half4 h = __builtin_elementwise_max(__builtin_convertvector(acc[i] + bias1, half4), half4{0, 0, 0, 0});
                        if (!interior && !(xin && (unsigned)(oy0 + 1 + r_lo + i) < (unsigned)a.H1)) h = half4{0, 0, 0, 0};
                        *(half4 *)(cp + i * (PW1 * 64)) = h;
                    }
                }
            }
            if (p + 1 < my_tiles && !(a.ablate & 16)) {          // ---- the patch of tile p + 1 (conv0 of this period is done with the buffer) ----
                commit();
                if (p + 2 < my_tiles) prefetch(tile_of(p + 2));
            }
        } else {
            if (p >= 1 && p <= my_tiles && !(a.ablate & 4)) {   // ---- conv2 of tile p - 1, tap column 2; pool; staging ----
                int n, ty, tx;
                decode(tile_of(p - 1), n, ty, tx);
                const int oy0 = 2 * ty * PY - 3, ox0 = 2 * tx * PXT - 3;
                const bool interior = oy0 >= 0 && ox0 >= 0 && oy0 + R0 <= a.H1 && ox0 + CW0 + 1 <= a.W1;
                const int r_lo = g2 * PY;
                int pb[2][4];
                make_pb(0, OFF_C1 + ((p - 1) & 1) * C1_BYTES + r_lo * PW1 * 64, pb);
                conv_cols(pb, integral_constant<int, ROWS2>{}, integral_constant<int, PW1>{}, 2, 3);
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const int fr = lo & 15, q4 = lo >> 4;
                const f32x4 bias2 = *(const f32x4 *)(sB + 64 + f2 * 16 + q4 * 4);
                const bool xin = fr < CW2 && (unsigned)(ox0 + 2 + fr) < (unsigned)a.W1;
                const int gy0 = oy0 + 2 + r_lo;
                if (!interior) {
                    const float ninf = -__builtin_inff();
#pragma unroll
                    for (int i = 0; i < ROWS2; i++)
                        if (!(xin && (unsigned)(gy0 + i) < (unsigned)a.H1)) acc[i] = f32x4{ninf, ninf, ninf, ninf};
                }
                const bool st = (fr & 1) == 0 && fr < 2 * PXT;
                char *sp = smem + OFF_STG + ((g2 * PR) * PXT + (fr >> 1)) * ROWB2 + (f2 * 16 + q4 * 4) * 2;
#pragma unroll
                for (int pp = 0; pp < PR; pp++) {
                    f32x4 m;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float v = fmaxf(fmaxf(acc[2 * pp][e], acc[2 * pp + 1][e]), acc[2 * pp + 2][e]);
                        m[e] = fmaxf(fmaxf(v, row_shl<1>(v)), row_shl<2>(v));
                    }
                    const half4 h = __builtin_elementwise_max(__builtin_convertvector(m + bias2, half4), half4{0, 0, 0, 0});
                    if (st) *(half4 *)(sp + pp * (PXT * ROWB2)) = h;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();                                          // B_b: C1[p & 1] and the next patch are complete; tile p - 1 is staged
    }
}

template <int C2P, int PY>
int launch_roles(fid_ctx *ctx, StemRArgs &a) {
    constexpr int R2 = 2 * PY + 1, R1 = R2 + 2, R0 = R2 + 4, RI = 2 * R0 + 1;
    constexpr int lds = (RI * RS * 2 + 16 + 255) / 256 * 256 + (R0 + 1) * PW0 * 64 + 2 * ((R1 + 1) * PW1 * 64 + 256) + PY * PXT * C2P * 2 + 512;
    a.tiles_x = cdiv(a.Wp, PXT); a.tiles_y = cdiv(a.Hp, PY);
    a.n_tiles = (a.n_tiles) * a.tiles_x * a.tiles_y;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)scrfd_stem_roles<C2P, PY>, lds));
    const int grid = std::min(a.n_tiles, ctx->num_cus);
    hipLaunchKernelGGL((scrfd_stem_roles<C2P, PY>), dim3(grid), dim3(512), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

template <int C2P, int PY>
int launch_rows(fid_ctx *ctx, StemRArgs &a) {
    constexpr int R2 = 2 * PY + 1, R1 = R2 + 2, R0 = R2 + 4, RI = 2 * R0 + 1;
    constexpr int MF0 = ((R0 * CW0 + 15) / 16 + 3) / 4;           // + the per-thread table of conv0's gather: MF0 dwords x 256 threads
    constexpr int lds = (RI * RS * 2 + 16 + 255) / 256 * 256 + (R0 + 1) * PW0 * 64 + (R1 + 1) * PW1 * 64 + 256 + PY * PXT * C2P * 2 + 512 + MF0 * 1024;
    a.tiles_x = cdiv(a.Wp, PXT); a.tiles_y = cdiv(a.Hp, PY);
    a.n_tiles = (a.n_tiles) * a.tiles_x * a.tiles_y;              // (n_tiles holds the batch size on entry)
    FID_TRY(ensure_dyn_lds(ctx, (const void *)scrfd_stem_rows<C2P, PY>, lds));
    const int grid = std::min(a.n_tiles, ctx->num_cus * (PY == 6 ? STEM_WPS : 2));
    hipLaunchKernelGGL((scrfd_stem_rows<C2P, PY>), dim3(grid), dim3(256), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace

// img [B,H,W,3] u8 -> out [B, H/4, W/4, C2p]; same contract as stem_fused_launch
int stem_rows_launch(fid_ctx *ctx, const uint8_t *img, int B, int H, int W, const void *w0, const float *b0, const void *w1,
                     const float *b1, const void *w2, const float *b2, void *out, int C2p) {
    FID_REQUIRE(H % 4 == 0 && W % 4 == 0, "fused stem: frame %dx%d not a multiple of 4", W, H);
    StemRArgs a{};
    a.img = img; a.w0 = (const _Float16 *)w0; a.b0 = b0; a.w1 = (const _Float16 *)w1; a.b1 = b1;
    a.w2 = (const _Float16 *)w2; a.b2 = b2; a.out = (_Float16 *)out;
    a.H = H; a.W = W; a.H1 = H / 2; a.W1 = W / 2; a.Hp = H / 4; a.Wp = W / 4;
    a.n_tiles = B;
    if (const char *e = getenv("FID_STEM_ABLATE")) a.ablate = atoi(e);
    a.stagger = getenv("FID_STEM_STAGGER") ? atoi(getenv("FID_STEM_STAGGER")) : 0;
    const int py = getenv("FID_STEM_PY") ? atoi(getenv("FID_STEM_PY")) : 8;      // (read per launch: the tests switch it)
    const bool roles = getenv("FID_STEM_ROLES") != nullptr;    // (read per launch: the tests switch it)
    if (roles && C2p == 64) return launch_roles<64, 8>(ctx, a);
    if (roles && C2p == 32) return launch_roles<32, 8>(ctx, a);
    if (C2p == 64) return py == 6 ? launch_rows<64, 6>(ctx, a) : launch_rows<64, 8>(ctx, a);
    if (C2p == 32) return py == 6 ? launch_rows<32, 6>(ctx, a) : launch_rows<32, 8>(ctx, a);
    set_error("fused stem: C2p=%d unsupported", C2p);
    return FID_E_INVALID;
}

}  // namespace fid
