// Producer / consumer channel-chunked 3x3 / stride-1 convolution (autotuner generation 5).
//
// Same data flow as conv_chunked.hip (16x16-pixel tile x CB couts per item, K walked in 32-channel chunks, per chunk
// an 18x18x32 patch + a 9xCBx32 weight chunk by LDS-DMA, row-sharing tap order), different division of labour.
// The stamped profile of conv_chunked.hip: a 7000-cycle step holds 3456 cycles of MFMA per SIMD; the rest is the
// barrier, ~1100 cycles of prefetch decoding and 1000-1800 cycles in which every wave blocks in the issue of its ~10
// LDS-DMA instructions (the CU's address path takes 1 KB per 16 cycles) -- on all four SIMDs at once, because the
// barrier aligns the waves.  With the DMA and the epilogue switched off the same MFMA section runs at ~95 % of the
// matrix rate.  So here
//
//   waves 0..7   CONSUMERS: fragments + MFMA only (plus the tile's epilogue once per item)
//   waves 8..11  PRODUCERS: walk the prefetch cursors, issue every LDS-DMA of the workgroup (a quarter each) and wait
//                           for their own (counted vmcnt) before they join the step's barrier
//
// The producer blocks in its DMA issues while the consumers multiply; nothing but the barrier is left on the
// consumers' path in a plain step.  All barriers are raw s_barrier: a consumer's output stores and epilogue loads
// stay in flight across them, the producer's vmcnt wait orders the DMA'd data (reader passes a barrier after it).
//
// Steps that start a new item first write out the previous tile: the consumers stage it through the weight slot of
// the step before (free by then) while the producer issues the patch prefetch; a second barrier (F) keeps the
// producer's next weight chunk out of that slot until the staging has been read back.
// LDS: weights 2 x 9*CB*64 B + patches PD x 21 KB (CB = 64: PD = 3, 135 KB; CB = 96: PD = 2, 150 KB).
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TH = 16, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;   // 324 patch pixels
constexpr int CK = 32;                                               // input channels per chunk
constexpr int P_BLKS = 21, P_BYTES = P_BLKS * 1024;                  // 1 KB DMA blocks of a patch chunk (16 pixels x 64 B)
constexpr int N_CONS = 8, N_PROD = 4;                                // consumer waves 0..7, producer waves 8..11

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct PCArgs {
    const void *in;
    const void *w;
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, Cin_p, Cout_p;
    int act, flags, res_Cp, nsig;
    int tiles_x, tiles_per_img, n_cblk, n_items, n_chunks;
    FastDiv d_cblk, d_tpi, d_tx;
    unsigned in_bytes, w_bytes;
    int ablate;   // timing experiments only (FID_PC_ABLATE: 1 = no weight DMA, 2 = no patch DMA, 4 = weight DMA from chunk-contiguous addresses, 8 = every patch from image 0, 16 = interior-tile addressing everywhere)
};

// Diagnostic build only (make EXTRA=-DFID_PC_STAMPS): s_memtime stamps of workgroup 0 -- producer, consumer waves 0 and 7.
#ifdef FID_PC_STAMPS
constexpr int STAMP_STEPS = 16, STAMP_K = 6;
__device__ unsigned long long g_pc_stamps[3 * STAMP_STEPS * STAMP_K];
#define STAMP(who, k)                                                                                      \
    do {                                                                                                   \
        if (blockIdx.x == 0 && s < STAMP_STEPS && (wave <= N_CONS)) {                                                          \
            unsigned long long t_;                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (lane == 0) g_pc_stamps[((who) * STAMP_STEPS + s) * STAMP_K + (k)] = t_;                     \
        }                                                                                                  \
    } while (0)
#else
#define STAMP(who, k)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most n of this wave's memory operations are outstanding (n = a small run-time count, immediate operand)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
#define FID_W(k) case k: wait_vmcnt_n<k>(); break;
        FID_W(1) FID_W(2) FID_W(3) FID_W(4) FID_W(5) FID_W(6) FID_W(7) FID_W(8) FID_W(9) FID_W(10) FID_W(11) FID_W(12)
        FID_W(13) FID_W(14) FID_W(15) FID_W(16) FID_W(17) FID_W(18) FID_W(19) FID_W(20) FID_W(21) FID_W(22) FID_W(23) FID_W(24)
#undef FID_W
        default: wait_vmcnt_n<0>(); break;
    }
}

// NI: couts per consumer wave = NI*16 (workgroup CB = 2*NI*16).  WD / PD: ring depths of the weight / patch chunks; a stream
// with depth 3 is fetched two steps ahead (its youngest chunk may still be in flight at the step's barrier), depth 2 one
// step ahead.  CB = 96 only fits 2 + 2; CB = 64 fits 2 + 3 (patches, HBM, two ahead) or 3 + 2 (weights two ahead).
// F32: the detector's 32-channel head convs -- fp32 outputs with a sigmoid on the first nsig channels; the consumers store
// their accumulator rows directly (16 B = four fp32 couts per lane, 64 B per pixel and wave), no staging, no flush steps.
// RS (opt-in, FID_PC_RS=1): the producers fetch with plain buffer loads into registers and write the LDS rings themselves
// within the step, instead of LDS-DMA.  Built to test whether the DMA issue rate (stamped: ~76 cycles per 1-KB piece per CU
// however many waves issue them) bounds a step; it does not -- the loads' own return time is as long.
template <int NI, int WD, int PD, bool F32 = false, bool RS = false>
__global__ void __launch_bounds__((N_CONS + N_PROD) * 64, 3) conv3x3_pc(const PCArgs a) {
    constexpr int CB = 2 * NI * 16, MI = 4;
    constexpr int W_BLKS = 9 * CB * 64 / 1024, W_BYTES = W_BLKS * 1024;
    constexpr int AW = WD - 1, AP = PD - 1;             // look-ahead of the two streams, in steps
    static_assert(WD * W_BYTES + PD * P_BYTES <= 160 * 1024 && !(WD == 3 && PD == 3), "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sWr = smem, *sPr = smem + WD * W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // my items: bid, + gridDim.x, ...; steps = (local item, chunk) linearised.  bid = my place in XCD-major order: the
    // workgroups of one XCD work on consecutive items (the cout blocks of a tile, then the next tile of the row)
    const int bid = (a.ablate & 256) ? (int)blockIdx.x : xcd_major_id(blockIdx.x, gridDim.x);
    const int my_items = bid < a.n_items ? (a.n_items - 1 - bid) / gridDim.x + 1 : 0;
    const int n_steps = my_items * a.n_chunks;
    if (n_steps == 0) return;

    auto decode_item = [&](int item, int &n, int &ty, int &tx, int &cb) {
        const int tile = fastdiv(item, a.d_cblk);
        cb = item - tile * a.n_cblk;
        n = fastdiv(tile, a.d_tpi);
        const int r = tile - n * a.tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };

    if (wave >= N_CONS) {
        // ======================================= PRODUCERS =======================================
        // four of them: one wave issues an LDS-DMA only every ~75 cycles (stamped: 5700 cycles for the 75 blocks of a step),
        // the address path takes one per 16; producer pw owns blocks pw, pw+4, ... of every weight and patch chunk
        const int pw = wave - N_CONS;
        __builtin_assume(pw >= 0 && pw < N_PROD);
        if (a.ablate & 512) __builtin_amdgcn_s_setprio(3);   // experiment: the producers' few instructions ahead of the consumers' MFMA streams
        constexpr int MAX_W = (W_BLKS + N_PROD - 1) / N_PROD, MAX_P = (P_BLKS + N_PROD - 1) / N_PROD;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        // per-lane constants of the DMA blocks.  Weights: byte offset of my 16 B inside the cout block's [CB][9][Cin_p]
        // rows (LDS row = t*CB + co); rows past the bank's end fall outside the descriptor and read as 0.
        int w_off[MAX_W];
#pragma unroll
        for (int k = 0; k < MAX_W; k++) {
            const int j = pw + N_PROD * k;
            const int row = j * 16 + (lane >> 2);
            const int t = row / CB, co = row - t * CB;
            w_off[k] = ((co * 9 + t) * a.Cin_p + ((lane & 3) ^ swz64(row)) * 8) * 2;
            if (a.ablate & 4) w_off[k] = j * 1024 + lane * 16;   // timing experiment: chunk-contiguous weights (wrong results)
        }
        // py | px << 8 | channel offset << 16 (py = 255: padding row) of my k-th patch block; recomputed where needed (border
        // tiles only) instead of held in registers: the producers' register budget goes to the tile they carry
        constexpr bool KEEP_PK = NI <= 2 && !RS;                   // CB = 64 has the registers to keep them (small maps are all border tiles)
        auto patch_pk_calc = [&](int k) {
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int row = (pw + N_PROD * k) * 16 + (lo >> 2);
            int py = row / PW;
            const int px = row - py * PW;
            if (row >= NPIX) py = 255;
            return py | (px << 8) | ((((lo & 3) ^ swz64(row)) * 8) << 16);
        };
        int p_pk[KEEP_PK ? MAX_P : 1];
        if (KEEP_PK) {
#pragma unroll
            for (int k = 0; k < MAX_P; k++) p_pk[KEEP_PK ? k : 0] = patch_pk_calc(k);
        }
        auto patch_pk = [&](int k) {
            if (!KEEP_PK) return patch_pk_calc(k);
            int pk = p_pk[KEEP_PK ? k : 0];
            asm volatile("" : "+v"(pk));                       // opaque: unpack at the use, do not hoist three registers per block
            return pk;
        };
        const int my_p = (P_BLKS - pw + N_PROD - 1) / N_PROD;   // patch DMAs I issue per chunk (6 or 5)
        // the same blocks for tiles whose whole 18x18 patch lies inside the image: lane offset relative to the patch's
        // top-left pixel, so an issue costs 3 vector instructions instead of ~20 (the producers share their SIMDs' vector
        // issue with the consumers' epilogue arithmetic: stamped, address arithmetic tripled the steps that have one)
        int p_off[MAX_P];
#pragma unroll
        for (int k = 0; k < (RS ? 0 : MAX_P); k++) {
            const int pk = patch_pk(k), py = pk & 255, px = (pk >> 8) & 255;
            p_off[k] = py == 255 ? -1 : ((py * a.W + px) * (a.Cin_p + ((a.ablate & 32) ? 32 : 0) + ((a.ablate & 64) ? 64 : 0)) + (pk >> 16)) * 2;   // (32 / 64: timing experiments, pixel pitch + 64 / 128 B)
        }
        // output stores: the consumers stage a finished fp16 tile in the accumulator layout's transpose (wave-major,
        // [64 pixels][NI*32 B], 16-byte chunks swizzled by pixel); the producers read it back as whole 16-byte cout segments and
        // write it out -- a store issue blocks behind the DMA traffic just like a DMA issue, so it belongs to the waves
        // that block anyway.  Block b of the staging area = consumer wave b / SK, chunks (b % SK)*64 ... +63.
        constexpr int OCPP = NI * 2, SK = OCPP, S_BLKS = N_CONS * SK, MAX_S = (S_BLKS + N_PROD - 1) / N_PROD;
        constexpr int OMASK = (OCPP & (OCPP - 1)) == 0 ? OCPP - 1 : 0;
        auto stage_pk_calc = [&](int k) {                       // pixel | cout chunk << 8 | consumer wave << 16 of my k-th staging block
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int b = pw + N_PROD * k, wv = b / SK;
            const int gl = (b - wv * SK) * 64 + lo;
            const int p = gl / OCPP, c = (gl - p * OCPP) ^ (p & OMASK);
            return p | (c << 8) | (wv << 16);
        };
        int s_pk[KEEP_PK ? MAX_S : 1];
        if (KEEP_PK) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++) s_pk[KEEP_PK ? k : 0] = stage_pk_calc(k);
        }
        auto stage_pk = [&](int k) {
            if (!KEEP_PK) return stage_pk_calc(k);
            int pk = s_pk[KEEP_PK ? k : 0];
            asm volatile("" : "+v"(pk));
            return pk;
        };
        int s_off[MAX_S];                                       // byte offset of my segment relative to the tile's first output
#pragma unroll
        for (int k = 0; k < MAX_S; k++) {
            const int pk = stage_pk(k), p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
            s_off[k] = ((((wv & 3) * MI + (p >> 4)) * a.W + (p & 15)) * a.Cout_p + (wv >> 2) * NI * 16 + c * 8) * 2;
        }
        u32x4 sv[MAX_S];                                        // my share of a staged tile, between read-back and store
        auto read_tile = [&](char *slot) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++)
                if (pw + N_PROD * k < S_BLKS) sv[k] = *(const u32x4 *)(slot + (pw + N_PROD * k) * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read back before my weight DMAs refill the slot
        };
        // Residual layers: the residual tile travels the other way through the same staging blocks.  The producers load it as
        // whole 16-byte segments (8 cache lines per instruction; the consumers' accumulator-layout loads were 8 bytes per lane
        // = 16 lines per instruction, on the path that already limits the step) during the item's last chunk, drop it into
        // the staging slot at the next step's start, and the consumers read their own elements from LDS.
        const bool has_res = a.res != nullptr;
        u32x4 rv[MAX_S];
        auto load_residual = [&](int item) {
            int n, ty, tx, cb;
            decode_item(item, n, ty, tx, cb);
            if (ty * TH + TH <= a.H && tx * TW + TW <= a.W && cb * CB + CB <= a.Cout_p) {   // whole tile inside the tensor
                const char *base = (const char *)a.res + ((((size_t)n * a.H + ty * TH) * a.W + tx * TW) * a.Cout_p + cb * CB) * 2;
#pragma unroll
                for (int k = 0; k < MAX_S; k++) rv[k] = *(const u32x4 *)(base + (unsigned)s_off[k]);
                return;
            }
#pragma unroll
            for (int k = 0; k < MAX_S; k++) {
                const int pk = stage_pk(k);
                const int p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
                const int oy = ty * TH + (wv & 3) * MI + (p >> 4), ox = tx * TW + (p & 15);
                const int co = cb * CB + (wv >> 2) * NI * 16 + c * 8;
                rv[k] = u32x4{0u, 0u, 0u, 0u};
                if (oy < a.H && ox < a.W && co < a.Cout_p)
                    rv[k] = *(const u32x4 *)((const char *)a.res + ((((size_t)n * a.H + oy) * a.W + ox) * a.Cout_p + co) * 2);
            }
        };
        auto write_residual = [&](char *slot) {
#pragma unroll
            for (int k = 0; k < MAX_S; k++) *(u32x4 *)(slot + (pw + N_PROD * k) * 1024 + lane * 16) = rv[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        static_assert(S_BLKS % N_PROD == 0, "every producer stores the same number of blocks");
        auto store_tile = [&](int item) {                       // the tile of `item`, read back into sv; true: exactly MAX_S stores issued
            int n, ty, tx, cb;
            decode_item(item, n, ty, tx, cb);
            if (ty * TH + TH <= a.H && tx * TW + TW <= a.W && cb * CB + CB <= a.Cout_p) {   // whole tile inside the tensor
                char *base = (char *)a.out + ((((size_t)n * a.H + ty * TH) * a.W + tx * TW) * a.Cout_p + cb * CB) * 2;
#pragma unroll
                for (int k = 0; k < MAX_S; k++)
                    if (pw + N_PROD * k < S_BLKS) *(u32x4 *)(base + (unsigned)s_off[k]) = sv[k];
                return true;
            }
#pragma unroll
            for (int k = 0; k < MAX_S; k++) {
                if (pw + N_PROD * k >= S_BLKS) continue;
                const int pk = stage_pk(k);
                const int p = pk & 255, c = (pk >> 8) & 255, wv = pk >> 16;
                const int oy = ty * TH + (wv & 3) * MI + (p >> 4), ox = tx * TW + (p & 15);
                const int co = cb * CB + (wv >> 2) * NI * 16 + c * 8;
                if (oy < a.H && ox < a.W && co < a.Cout_p)
                    *(u32x4 *)((char *)a.out + ((((size_t)n * a.H + oy) * a.W + ox) * a.Cout_p + co) * 2) = sv[k];
            }
            return false;
        };
        struct Cursor {
            int item, ck;      // work item / chunk the NEXT issue of this stream fetches
            int w_base;        // weights: byte offset of the item's cout block (chunk 0)
            int n, y0, x0;     // patch: image, top-left input pixel of the haloed patch
        };
        auto cursor_decode = [&](Cursor &c) {
            int n, ty, tx, cb;
            decode_item(c.item, n, ty, tx, cb);
            c.w_base = cb * CB * 9 * a.Cin_p * 2;
            c.n = (a.ablate & 8) ? 0 : n; c.y0 = ty * TH - 1; c.x0 = tx * TW - 1;   // (8: timing experiment, every patch from image 0)
        };
        auto cursor_next = [&](Cursor &c) {
            if (++c.ck == a.n_chunks) {
                c.ck = 0;
                c.item += gridDim.x;
                cursor_decode(c);
            }
        };
        auto issue_weights = [&](const Cursor &c, int slot) {
            if (a.ablate & 1) return;
            const int ubase = (a.ablate & 4) ? c.ck * W_BYTES : c.w_base + c.ck * CK * 2;
            char *dst = sWr + slot * W_BYTES;
#pragma unroll
            for (int k = 0; k < MAX_W; k++) {
                const int j = pw + N_PROD * k;
                if (j < W_BLKS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                             (unsigned)(w_off[k] + ubase), 0, 0, 0);
            }
        };
        auto issue_patch = [&](const Cursor &c, int slot) {    // exactly my_p instructions (vmcnt accounting)
            if (a.ablate & 2) return;
            const int c0 = c.ck * CK;
            char *dst = sPr + slot * P_BYTES;
            if ((a.ablate & 16) || (c.y0 >= 0 && c.x0 >= 0 && c.y0 + TH + 2 <= a.H && c.x0 + TW + 2 <= a.W)) {   // interior tile (16: timing experiment)
                const int base = (((c.n * a.H + c.y0) * a.W + c.x0) * a.Cin_p + c0) * 2;
#pragma unroll
                for (int k = 0; k < MAX_P; k++) {
                    const int j = pw + N_PROD * k;
                    if (j >= P_BLKS) continue;
                    const unsigned vo = p_off[k] < 0 ? OOB : (unsigned)(p_off[k] + base);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < MAX_P; k++) {
                const int j = pw + N_PROD * k;
                if (j >= P_BLKS) continue;
                const int pk = patch_pk(k);
                const int py = pk & 255, iy = c.y0 + py, ix = c.x0 + ((pk >> 8) & 255);
                const bool in = py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((c.n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + (pk >> 16)) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        };

        // ---- register-staged variants of the two fetches (RS): load now, write the LDS image a step later ----
        u32x4 wreg[RS ? MAX_W : 1], preg[RS ? MAX_P : 1];
        auto load_weights = [&](const Cursor &c) {
            if (a.ablate & 1) return;
            const int ubase = c.w_base + c.ck * CK * 2;
#pragma unroll
            for (int k = 0; k < MAX_W; k++)
                if (pw + N_PROD * k < W_BLKS) wreg[RS ? k : 0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)(w_off[k] + ubase), 0, 0);
        };
        auto load_patch = [&](const Cursor &c) {
            if (a.ablate & 2) return;
            const int c0 = c.ck * CK;
            if (c.y0 >= 0 && c.x0 >= 0 && c.y0 + TH + 2 <= a.H && c.x0 + TW + 2 <= a.W) {   // interior tile
                const int base = (((c.n * a.H + c.y0) * a.W + c.x0) * a.Cin_p + c0) * 2;
#pragma unroll
                for (int k = 0; k < MAX_P; k++) {
                    if (pw + N_PROD * k >= P_BLKS) continue;
                    const int pk = patch_pk(k), py = pk & 255;   // (a register load is cheap to issue: no per-block offset table)
                    const unsigned vo = py == 255 ? OOB : (unsigned)(((py * a.W + ((pk >> 8) & 255)) * a.Cin_p + (pk >> 16)) * 2 + base);
                    preg[RS ? k : 0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0);
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < MAX_P; k++) {
                if (pw + N_PROD * k >= P_BLKS) continue;
                const int pk = patch_pk(k);
                const int py = pk & 255, iy = c.y0 + py, ix = c.x0 + ((pk >> 8) & 255);
                const bool in = py != 255 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned vo = in ? (unsigned)((((c.n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + (pk >> 16)) * 2) : OOB;
                preg[RS ? k : 0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0);
            }
        };
        auto commit_weights = [&](int slot) {                   // (the compiler waits for the loads where the registers are read)
            char *dst = sWr + slot * W_BYTES + lane * 16;
#pragma unroll
            for (int k = 0; k < MAX_W; k++)
                if (pw + N_PROD * k < W_BLKS) *(u32x4 *)(dst + (pw + N_PROD * k) * 1024) = wreg[RS ? k : 0];
        };
        auto commit_patch = [&](int slot) {
            char *dst = sPr + slot * P_BYTES + lane * 16;
#pragma unroll
            for (int k = 0; k < MAX_P; k++)
                if (pw + N_PROD * k < P_BLKS) *(u32x4 *)(dst + (pw + N_PROD * k) * 1024) = preg[RS ? k : 0];
        };
        auto lds_done = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

        const int my_w = (W_BLKS - pw + N_PROD - 1) / N_PROD;   // weight DMAs I issue per chunk
        // prologue: everything step 0 reads, then the two-ahead stream's chunk of step 1
        Cursor cw, cp;
        cw.item = bid; cw.ck = 0;
        cursor_decode(cw);
        cp = cw;
        if constexpr (RS) {
            // Step s (after its barrier T(s), which frees the slots step s-1 read): load W(s+AW) and P(s+AP) into registers
            // while the consumers multiply, write them into their ring slots when they arrive, and finish the LDS writes before
            // T(s+1).  The registers live inside one step; a cursor always stands on the next chunk its stream fetches.
            auto fetch = [&](bool w, int wslot, bool p_, int pslot) {
                if (p_) { load_patch(cp); cursor_next(cp); }
                if (w) { load_weights(cw); cursor_next(cw); }
                if (w) commit_weights(wslot);                  // (from L2: back first)
                if (p_) commit_patch(pslot);
            };
            fetch(true, 0, true, 0);
            if (n_steps > 1) fetch(AW == 2, 1, AP == 2, 1);
            lds_done();
            int ck = 0, item = bid;                     // chunk / item of the step the consumers are in
            for (int s = 0; s < n_steps; s++) {
                STAMP(0, 0);
                STAMP(0, 1);
                raw_barrier();                                 // T(s)
                STAMP(0, 2);
                if (!F32 && ck == 0 && s > 0) {                // the consumers write out the previous tile first
                    char *stage = sWr + ((s + WD - 1) % WD) * W_BYTES;   // step s-1's weight slot = where W(s+AW) goes
                    if (has_res) {                             // (loaded at the end of the step before)
                        write_residual(stage);
                        raw_barrier();                         // R(s): the consumers pick their residual values up
                    }
                    raw_barrier();                             // F(s): the previous tile is staged in the weight slot
                    read_tile(stage);
                    store_tile(item - gridDim.x);
                }
                STAMP(0, 3);
                fetch(s + AW < n_steps, (s + AW) % WD, s + AP < n_steps, (s + AP) % PD);
                if (has_res && ck == a.n_chunks - 1) load_residual(item);   // the item's last chunk: its residual tile
                lds_done();
                STAMP(0, 4);
                if (++ck == a.n_chunks) { ck = 0; item += gridDim.x; }
            }
            if (F32) return;
            raw_barrier();                                     // tail A: every consumer is done with the last weight slot
            if (has_res) {
                write_residual(sWr + ((n_steps - 1) % WD) * W_BYTES);
                raw_barrier();                                 // tail R
            }
            raw_barrier();                                     // tail B: the last tile is staged
            read_tile(sWr + ((n_steps - 1) % WD) * W_BYTES);
            store_tile(item - gridDim.x);
            return;
        }
        issue_weights(cw, 0);
        issue_patch(cp, 0);
        cursor_next(cw);                                       // -> step 1
        cursor_next(cp);
        int young = 0;                                         // my youngest memory operations that need not have landed at the next barrier
        if (n_steps > 1) {
            if (AP == 2) { issue_patch(cp, 1); cursor_next(cp); young = my_p; }
            if (AW == 2) { issue_weights(cw, 1); cursor_next(cw); young = my_w; }
        }
        int ck = 0, item = bid;                         // chunk / item of the step the consumers are in
        for (int s = 0; s < n_steps; s++) {
            // everything step s reads must have landed: W(s) and P(s); only `young` younger operations may stay in flight
            STAMP(0, 0);
            wait_vmcnt_dyn(young);
            STAMP(0, 1);
            raw_barrier();                                     // T(s)
            STAMP(0, 2);
            const bool flush = !F32 && ck == 0 && s > 0;       // the consumers write out the previous tile first
            const bool have_w = s + AW < n_steps, have_p = s + AP < n_steps;
            char *stage = sWr + ((s + WD - 1) % WD) * W_BYTES; // step s-1's weight slot = where W(s+AW) goes
            young = 0;
            if (flush) {
                // patch first (its slot is not involved in the staging); after the consumers have staged the previous tile
                // (barrier F) read it back, fetch the weights that go into the staging slot, then write the tile out
                if (has_res) {                                 // (loaded during the step before; the wait ahead of T(s) covered them)
                    write_residual(stage);
                    raw_barrier();                             // R(s): the consumers pick their residual values up
                }
                if (have_p) issue_patch(cp, (s + AP) % PD);
                raw_barrier();                                 // F(s)
                read_tile(stage);
                if (have_w) issue_weights(cw, (s + AW) % WD);  // the prefetch first: the stores have a whole step
                const bool all_stores = store_tile(item - gridDim.x);
                // youngest first: [stores][W(s+AW)][P(s+AP)]: the stores may always fly on, the weights if they are two ahead
                if (all_stores) young = MAX_S + ((AW == 2 && have_w) ? my_w : 0);
            } else {
                // the one-ahead stream first, the two-ahead stream's chunk is the youngest and may stay in flight
                if (AW == 2) {
                    if (have_p) issue_patch(cp, (s + AP) % PD);
                    if (have_w) { issue_weights(cw, (s + AW) % WD); young = my_w; }
                } else {
                    if (have_w) issue_weights(cw, (s + AW) % WD);
                    if (have_p) { issue_patch(cp, (s + AP) % PD); young = AP == 2 ? my_p : 0; }
                }
            }
            if (has_res && ck == a.n_chunks - 1) {             // the item's last chunk: fetch its residual tile (after the prefetches)
                load_residual(item);
                young = 0;                                     // the loads are the youngest entries and are needed right after T(s+1)
            }
            STAMP(0, 3);
            if (have_w && s + AW + 1 < n_steps) cursor_next(cw);
            if (have_p && s + AP + 1 < n_steps) cursor_next(cp);
            STAMP(0, 4);
            if (++ck == a.n_chunks) { ck = 0; item += gridDim.x; }
        }
        if (F32) return;                                       // (no staged tile to write out)
        raw_barrier();                                         // tail A: every consumer is done with the last weight slot
        if (has_res) {
            wait_vmcnt_n<0>();
            write_residual(sWr + ((n_steps - 1) % WD) * W_BYTES);
            raw_barrier();                                     // tail R
        }
        raw_barrier();                                         // tail B: the last tile is staged
        read_tile(sWr + ((n_steps - 1) % WD) * W_BYTES);
        store_tile(item - gridDim.x);
        return;
    }

    // ========================================= CONSUMERS =========================================
    const int grp = wave >> 2, wg = wave & 3;               // cout group, pixel group
    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wg * MI) * PW + frow;
    EpiArgs ep{a.bias, a.slope, a.res, a.out, a.Cout_p, a.H, a.W, a.act, a.flags, 0, a.H, a.W, a.res_Cp};

    constexpr int OROWB = NI * 32, OCPP = NI * 2;
    constexpr int OMASK = (OCPP & (OCPP - 1)) == 0 ? OCPP - 1 : 0;
    static_assert(F32 || N_CONS * 64 * OROWB <= W_BYTES, "staging must fit a weight slot");
    EpiPix px[MI];
    int co0[NI];
    EpiRegs<NI, MI> R;
    ep_half4 hv[NI][MI];
    int bias_cb = -1;                                        // cout block whose bias / slopes sit in R
    auto epi_prefetch = [&](int item) {
        int n, ty, tx, cb;
        decode_item(item, n, ty, tx, cb);
        int lo = lane;                                       // opaque lane id: keeps the per-lane address arithmetic of this block
        asm volatile("" : "+v"(lo));                         // from being hoisted out of the step loop (and spilled)
        const int frow = lo & 15, fq = lo >> 4;
        const int co_w = cb * CB + grp * NI * 16;            // first cout of this wave
        if (a.flags & CF_BORDER) {                           // pixel coordinates: only the border-class bias needs them
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int oy = ty * TH + wg * MI + mi, ox = tx * TW + frow;
                px[mi].valid = oy < a.H && ox < a.W;
                px[mi].n = n; px[mi].oy = oy; px[mi].ox = ox;
                px[mi].m = px[mi].valid ? ((long long)n * a.H + oy) * a.W + ox : 0;
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) co0[ni] = co_w + ni * 16 + fq * 4;
        // bias row / PReLU slopes only change with the cout block: most layers here have a single one
        if (cb != bias_cb) {
            bias_cb = cb;
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int c = co0[ni] < a.Cout_p ? co0[ni] : 0;
                R.bb[ni] = (a.bias != nullptr && !(a.flags & CF_BORDER)) ? *(const ep_f32x4 *)(a.bias + c) : ep_f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.act == ACT_PRELU) R.sl[ni] = *(const ep_f32x4 *)(a.slope + c);
            }
        }
    };
    f32x4 acc[NI][MI];
    const bool has_res = a.res != nullptr;
    auto epi_residual_values = [&](char *slot) {            // residual layers: my residual elements sit where my outputs will go
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int frow = lo & 15, fq = lo >> 4;
        const char *sS = slot + wave * (64 * OROWB);
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                R.rr[ni][mi] = *(const ep_half4 *)(sS + p * OROWB + ((c ^ (p & OMASK)) << 4) + (fq & 1) * 8);
            }
        epilogue_values_fast<NI, MI>(ep, acc, px, co0, R, hv);
    };
    auto epi_stage = [&](char *slot) {                       // fp16 tile: accumulator layout -> LDS, transposed; the producers store it
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int frow = lo & 15, fq = lo >> 4;
        char *sS = slot + wave * (64 * OROWB);
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                *(ep_half4 *)(sS + p * OROWB + ((c ^ (p & OMASK)) << 4) + (fq & 1) * 8) = hv[ni][mi];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staged before the barrier that hands the slot to the producers
    };

    int li = 0, ck = 0;                                   // local item index / chunk of the current step
    for (int s = 0; s < n_steps; s++) {
        const int item = bid + li * gridDim.x;
        [[maybe_unused]] const int who = wave == 0 ? 1 : 2;
        if (wave == 0 || wave == 7) STAMP(who, 0);
        raw_barrier();                                      // T(s): the producer saw W(s), P(s) land; everyone is done with step s-1
        if (wave == 0 || wave == 7) STAMP(who, 1);
        if (!F32 && ck == 0 && s > 0) {
            char *slot = sWr + ((s + WD - 1) % WD) * W_BYTES;   // step s-1's weight slot
            if (has_res) {
                raw_barrier();                              // R(s): the producers have put the residual tile there
                epi_residual_values(slot);                  // (the previous item's sums are still in acc)
            }
            epi_stage(slot);
            raw_barrier();                                  // F(s): the producers write the tile out, then refill the slot
        }
        if (wave == 0 || wave == 7) STAMP(who, 2);
        const bool last_chunk = ck == a.n_chunks - 1;
        if (last_chunk && !F32) epi_prefetch(item);        // bias / slope / residual: they return during the matrix work
        if (wave == 0 || wave == 7) STAMP(who, 3);
        if (ck == 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char *sW = sWr + (s % WD) * W_BYTES, *sP = sPr + (s % PD) * P_BYTES;
        // row-sharing tap order (conv_chunked.hip): column dx, then the 6 patch rows of this wave; a pixel fragment
        // (row r, shift dx) feeds every output row mi = r - dy; the column's three taps keep their weights in registers
        {
            int plin = lin0, wlane = ((grp * NI) * 16 + frow) * 64 + ((fq ^ swz64(frow)) << 4);
            asm volatile("" : "+v"(plin), "+v"(wlane));   // opaque: recompute the fragment addresses per step
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
        if (wave == 0 || wave == 7) STAMP(who, 4);
        if (F32) {
            if (last_chunk) {                               // bias, sigmoid on the class scores, fp32 rows straight to memory
                int n, ty, tx, cb;
                decode_item(item, n, ty, tx, cb);
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const int frow = lo & 15, fq = lo >> 4;
#pragma unroll
                for (int ni = 0; ni < NI; ni++) {
                    const int co = cb * CB + (grp * NI + ni) * 16 + fq * 4;
                    const f32x4 b = (a.bias != nullptr && co < a.Cout_p) ? *(const f32x4 *)(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) {
                        f32x4 v = acc[ni][mi] + b;
                        if (co < a.nsig) {                  // (only the lanes that hold class scores)
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (co + i < a.nsig) v[i] = 1.f / (1.f + expf(-v[i]));
                        }
                        const int oy = ty * TH + wg * MI + mi, ox = tx * TW + frow;
                        if (oy < a.H && ox < a.W && co < a.Cout_p)
                            *(f32x4 *)((float *)a.out + (((size_t)n * a.H + oy) * a.W + ox) * a.Cout_p + co) = v;
                    }
                }
            }
        } else if (last_chunk && !has_res) epilogue_values_fast<NI, MI>(ep, acc, px, co0, R, hv);   // kept in registers until the next step stages them
        if (wave == 0 || wave == 7) STAMP(who, 5);
        if (++ck == a.n_chunks) { ck = 0; li++; }
    }
    if (F32) return;
    raw_barrier();                                          // tail A: all consumers are done reading the last weight slot
    if (has_res) {
        raw_barrier();                                      // tail R
        epi_residual_values(sWr + ((n_steps - 1) % WD) * W_BYTES);
    }
    epi_stage(sWr + ((n_steps - 1) % WD) * W_BYTES);
    raw_barrier();                                          // tail B: the producers store the last tile
}

template <int NI, int WD, int PD, bool F32 = false, bool RS = false>
int launch_pc(fid_ctx *ctx, const PCArgs &a) {
    constexpr size_t lds = (size_t)WD * 9 * 2 * NI * 16 * 64 + (size_t)PD * P_BYTES;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_pc<NI, WD, PD, F32, RS>, (int)((int)lds)));
    const int grid = std::min(a.n_items, ctx->num_cus);
    hipLaunchKernelGGL((conv3x3_pc<NI, WD, PD, F32, RS>), dim3(grid), dim3((N_CONS + N_PROD) * 64), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
#ifdef FID_PC_STAMPS
    if (const char *e = getenv("FID_PC_STAMP_DUMP")) {
        static int countdown = atoi(e);                    // dump the N-th launch of this instantiation
        if (--countdown == 0) {
            FID_HIP(hipStreamSynchronize(ctx->stream));
            static unsigned long long h[3 * STAMP_STEPS * STAMP_K];
            FID_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pc_stamps), sizeof(h)));
            const int steps = std::min(STAMP_STEPS, ((a.n_items - 1) / grid + 1) * a.n_chunks);
            fprintf(stderr, "[pc-stamps] NI=%d H=%d W=%d Cin_p=%d Cout_p=%d items=%d chunks=%d\n", NI, a.H, a.W, a.Cin_p, a.Cout_p, a.n_items, a.n_chunks);
            const unsigned long long t0 = h[0];
            for (int s = 0; s < steps; s++) {
                const unsigned long long *p = h + s * STAMP_K, *c0 = h + (STAMP_STEPS + s) * STAMP_K, *c7 = h + (2 * STAMP_STEPS + s) * STAMP_K;
                fprintf(stderr, "[pc-stamps] step %2d | producer: +%6lld wait %5lld barrier %5lld issue %5lld next %4lld | cons0: arrive +%6lld barrier %5lld flush %5lld pref %5lld mfma %5lld values %5lld | cons7: arrive +%6lld barrier %5lld mfma %5lld\n",
                        s, (long long)(p[0] - t0), (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]), (long long)(p[4] - p[3]),
                        (long long)(c0[0] - t0), (long long)(c0[1] - c0[0]), (long long)(c0[2] - c0[1]), (long long)(c0[3] - c0[2]), (long long)(c0[4] - c0[3]), (long long)(c0[5] - c0[4]),
                        (long long)(c7[0] - t0), (long long)(c7[1] - c7[0]), (long long)(c7[4] - c7[3]));
            }
        }
    }
#endif
    return FID_OK;
}

}  // namespace

bool conv_pc_applicable(const ConvArgs &a) {
    if (getenv("FID_NO_PC")) return false;
    const bool shape = a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin_p % 32 == 0 && a.Cin_p >= 64 &&
                       a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 12 && a.W >= 12 && !(a.flags & (CF_RES_UP2 | CF_ARGMAX));
    if (!shape) return false;
    if (a.flags & CF_OUT_F32)   // the detector head maps: 32 fp32 channels, no activation / residual / border classes
        return a.Cout_p == 32 && a.act == ACT_NONE && a.res == nullptr && !(a.flags & CF_BORDER);
    return a.Cout_p >= 64 && a.nsig == 0 && (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo));
}

// cb: output channels per work item (64 or 96); ring: 0 = patches two steps ahead (CB = 64) / both one ahead (CB = 96),
// 1 = weights two steps ahead (CB = 64 only)
int conv_pc_launch(fid_ctx *ctx, const ConvArgs &c, int cb, int ring) {
    PCArgs a{};
    a.in = c.in; a.w = c.w; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p;
    a.act = c.act; a.flags = c.flags; a.res_Cp = c.res_Cp; a.nsig = c.nsig;
    const int B = c.M / (c.Ho * c.Wo);
    a.tiles_x = cdiv(c.W, TW);
    a.tiles_per_img = a.tiles_x * cdiv(c.H, TH);
    a.n_cblk = cdiv(c.Cout_p, cb);
    a.n_items = B * a.tiles_per_img * a.n_cblk;
    a.n_chunks = c.Cin_p / CK;
    a.d_cblk = fastdiv_make(a.n_cblk); a.d_tpi = fastdiv_make(a.tiles_per_img); a.d_tx = fastdiv_make(a.tiles_x);
    a.in_bytes = c.in_bytes;
    a.w_bytes = (unsigned)std::min<size_t>(c.w_bytes, (size_t)c.w_rows * 9 * c.Cin_p * 2);   // rows past the bank read as 0
    if (const char *e = getenv("FID_PC_ABLATE")) a.ablate = atoi(e);
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB");
    // register-staged producers (FID_PC_RS=1) were measured 5-10 % slower than the LDS-DMA ones on every layer
    const bool dma = getenv("FID_PC_RS") == nullptr;
    if (cb == 32) return dma ? launch_pc<1, 2, 3, true>(ctx, a) : launch_pc<1, 2, 3, true, true>(ctx, a);
    if (cb == 64 && ring == 1) return launch_pc<2, 3, 2>(ctx, a);
    if (cb == 64) return dma ? launch_pc<2, 2, 3>(ctx, a) : launch_pc<2, 2, 3, false, true>(ctx, a);
    if (cb == 96) return launch_pc<3, 2, 2>(ctx, a);
    set_error("producer/consumer conv: cb=%d unsupported", cb);
    return FID_E_INVALID;
}

}  // namespace fid
