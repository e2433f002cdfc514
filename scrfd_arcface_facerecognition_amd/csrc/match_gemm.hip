// Gallery scan for a BATCH of queries against a LARGE gallery: scores = G . Q^T with a running arg-max per query, 256 x 256 tiles.
//
// reference main.py:136-142 compares one embedding with every target in a python loop; fid_match (match.hip) runs it as one GEMM with a fused
// arg-max.  On the generic 128 x 128 GEMM tile (conv.hip) that scan reaches 480-620 TFLOP/s at 1 M entries: a CU takes in ~14-16 bytes per
// clock of operands whichever way they come (LDS-DMA or register-staged loads), and a 128 x 128 x 32 step needs 16 KB for 256 matrix cycles
// per SIMD -- the fill is four times as long as the arithmetic.  Twice the tile edge halves the operand bytes per flop:
//
//   tile    = 256 gallery rows x 256 queries (fp32 sums: 128 VGPRs per lane on 8 waves -- half the CU's register file, the largest that fits);
//             consumer wave (wm, wn) = gallery rows 128 wm .. + 127 x queries 64 wn .. + 63: 12 fragment reads feed 32 MFMAs per 32-column K-step
//   roles   = 8 consumer waves only read fragments and multiply; 4 LOADER waves issue every LDS-DMA piece.  The CU's address path takes ~47
//             cycles per 1-KB piece and this kernel keeps it busy all the time; a wave that sends a piece waits for it, and a consumer that
//             waits issues no MFMAs (measured on the 8-wave form where every wave sent 4 pieces per step: matrix work alone 0.33 ms, pieces
//             alone 0.36 ms, together 0.59 ms per 1 M x 512 scan -- whether the pieces went out early, late, or in different halves of a
//             step for the two waves of a SIMD; with loaders 0.51 ms).  12 waves = three per SIMD: the consumers fit 168 VGPRs with the
//             running best in LDS and lane constants re-derived where they are used
//   stream  = both operands in K-steps of 32 columns (2 x 16 KB), four slots; in the MIDDLE of step s every wave meets the others (the
//             loaders behind a counted s_waitcnt for their pieces of step s + 1): step s + 1 is then complete and slot (s + 3) % 4, last
//             read in step s - 1, is free for the loaders' next eight pieces each; the consumers read the first fragments of step s + 1
//             behind their last MFMAs of step s, so the matrix pipes do not drain at a step border.  Rows are 64 bytes in LDS with the
//             16-byte groups XOR-swizzled on the SOURCE address like the conv patches: a fragment read (16 rows x 4 groups) is conflict-free
//   work    = workgroup (qt, r): query tile qt against a contiguous range r of gallery tiles, the K-step stream running on across tile
//             borders; the two query tiles of a 500-face chunk walk the same gallery range on neighbouring workgroup ids (one XCD: the
//             second reader of a gallery line finds it in L2), so HBM sees the gallery once
//   arg-max = per tile and lane a strict-'>' scan of its 128 sums in ascending gallery order (first maximum wins, a NaN never wins: the
//             reference's `sim > max_similarity` chain), folded into a running best per query column (LDS, 16 KB); ONE packed key
//             (sortable(score) << 32 | ~index, as conv.hip's CF_ARGMAX epilogue) per query, wave and workgroup goes to memory by atomicMax
//             at the very end -- the lowest index wins among equal scores, whichever workgroup held it
//
// Rows past the gallery / the query batch read as zeros (buffer bounds): a zero score can never be a match (fid_match needs > max(0, thresh)).
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TG = 256, TQ = 256, CK = 32;               // tile, K-step
constexpr int OP_BYTES = TG * CK * 2, ST_BYTES = 2 * OP_BYTES;   // 16 KB per operand and step
constexpr int MI = 8, NI = 4;                            // fragments of a consumer wave: 128 gallery rows x 64 queries
constexpr int NLD = 4, PPL = 32 / NLD, NSR = 4;          // loader waves, their pieces per step (32 pieces of 1 KB), ring slots
constexpr int OFF_BEST = NSR * ST_BYTES;                 // [score | index][consumer wave][ni][lane]: the running best (16 KB)
constexpr unsigned OOB = 0xFFFFFF00u;

struct MGArgs {
    const void *q, *g;
    unsigned long long *amax;
    int n, Gp, dim, col0;
    int n_qt, n_gt, gt_per_wg, ks;
    unsigned q_bytes, g_bytes;
};

__device__ __forceinline__ unsigned sortable_f(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

__global__ void __launch_bounds__(768, 3) match_scan256(const MGArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int qt = bid % a.n_qt, r = bid / a.n_qt;
    const int gt0 = r * a.gt_per_wg, gt1 = min(a.n_gt, gt0 + a.gt_per_wg);
    if (gt0 >= gt1) return;
    const int n_steps = (gt1 - gt0) * a.ks;
    if (wave >= 8) {
        // ================= loader =================
        const int lw = wave - 8;
        const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void *)a.g, 0, a.g_bytes, 0x00020000);
        const auto rs_q = __builtin_amdgcn_make_buffer_rsrc((void *)a.q, 0, a.q_bytes, 0x00020000);
        const unsigned rowb = (unsigned)a.dim * 2u;
        const unsigned q_base = (unsigned)qt * TQ * rowb;
        const int ll = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (re-derived: nothing of the loader stays live in the consumers' registers)
        const unsigned lo = (unsigned)(ll >> 2) * rowb + (unsigned)(((ll & 3) ^ ((ll >> 3) & 3)) * 16);
        auto issue = [&](int s) __attribute__((always_inline)) {       // exactly PPL instructions: pieces j = lw + 4 k, k < 4: gallery, else queries
            const bool live = s < n_steps;
            const int t = live ? s / a.ks : 0, kk = live ? s - t * a.ks : 0;
            const unsigned g_base = (unsigned)(gt0 + t) * TG * rowb + (unsigned)kk * (CK * 2);
            const unsigned qb = q_base + (unsigned)kk * (CK * 2);
            char *dst = smem + (s % NSR) * ST_BYTES;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const int j = lw + NLD * k;                                 // 0 .. 31
                unsigned po = lo + (unsigned)(16 * (j & 15)) * rowb;
                asm volatile("" : "+v"(po));
                const unsigned vo = live ? (j < 16 ? g_base : qb) + po : OOB - (unsigned)k * 16u;
                if (k < PPL / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            }
        };
        issue(0); issue(1); issue(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPL) : "memory");
        raw_barrier();
        for (int s = 0; s < n_steps; s++) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPL) : "memory");       // my pieces of step s + 1 are in (those of s + 2 may fly on)
            raw_barrier();
            issue(s + 3);                                                 // slot (s + 3) % 4 was last read in step s - 1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ================= consumer =================
    const int frow = lane & 15, fq = lane >> 4, wm = wave & 1, wn = wave >> 1;
    const int grp = (fq ^ ((frow >> 1) & 3)) * 16;
    const int a_off = (wm * 128 + frow) * 64 + grp, b_off = OP_BYTES + (wn * 64 + frow) * 64 + grp;
    f32x4 acc[MI][NI];
    // the running best per query column lives in LDS (8 VGPRs the 168-register cap does not have): [score | index][wave][ni][lane]
    {
        float *best_s = (float *)(smem + OFF_BEST) + wave * (NI * 64) + lane;
        int *best_i = (int *)(smem + OFF_BEST + 8192) + wave * (NI * 64) + lane;
#pragma unroll
        for (int ni = 0; ni < NI; ni++) { best_s[ni * 64] = -__builtin_inff(); best_i[ni * 64] = -1; }
    }
    raw_barrier();                                                        // step 0 is in
    half8 bf[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) bf[ni] = *(const half8 *)(smem + b_off + ni * 1024);
    int slot = 0;
    auto step = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char *st = smem + slot * ST_BYTES;
        slot = slot + 1 == NSR ? 0 : slot + 1;
        const char *stn = smem + slot * ST_BYTES;
        constexpr int MB = 2;
        half8 af = *(const half8 *)(st + a_off);
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            if (mi == MB) raw_barrier();                                  // step s + 1 is complete; the loaders go on to step s + 3
            half8 afn = af;
            if (mi + 1 < MI) afn = *(const half8 *)(st + a_off + (mi + 1) * 1024);
            __builtin_amdgcn_sched_barrier(0);                            // (one fragment ahead, not eight: the 168-register cap)
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[ni], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            af = afn;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) bf[ni] = *(const half8 *)(stn + b_off + ni * 1024);
    };
    for (int t = gt0; t < gt1; t++) {
        step(std::integral_constant<bool, true>{});
        for (int kk = 1; kk < a.ks; kk++) step(std::integral_constant<bool, false>{});
        // (lane-derived constants are re-derived here, not kept through the matrix loop: the 168-register cap)
        int lane_t;                                                       // (asm volatile: a plain mbcnt is hoisted out of the tile loop and spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
        const int tbase = t * TG + wm * 128 + (lane_t >> 4) * 4;
        float *best_s = (float *)(smem + OFF_BEST) + wave * (NI * 64) + lane_t;
        int *best_i = (int *)(smem + OFF_BEST + 8192) + wave * (NI * 64) + lane_t;
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            float ts = -__builtin_inff();
            int ti = 0;
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float v = acc[mi][ni][j];
                    const bool up = v > ts;
                    ti = up ? mi * 16 + j : ti;
                    ts = up ? v : ts;
                }
            if (ts > best_s[ni * 64]) { best_s[ni * 64] = ts; best_i[ni * 64] = tbase + ti; }
        }
    }
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const float *best_s = (const float *)(smem + OFF_BEST) + wave * (NI * 64) + lane_e;
    const int *best_i = (const int *)(smem + OFF_BEST + 8192) + wave * (NI * 64) + lane_e;
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        const float bs_ = best_s[ni * 64];
        const int bi_ = best_i[ni * 64];
        unsigned long long key = bi_ >= 0 ? (((unsigned long long)sortable_f(bs_) << 32) | (unsigned)(~(unsigned)(a.col0 + bi_))) : 0ull;
        auto xchg = [&](unsigned long long k, int m) __attribute__((always_inline)) {        // (the lane id re-derived above, not the one kept from the start)
            const int src = (lane_e ^ m) << 2;
            const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)k);
            const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(k >> 32));
            return ((unsigned long long)hi << 32) | lo;
        };
        unsigned long long o = xchg(key, 16);
        key = o > key ? o : key;
        o = xchg(key, 32);
        key = o > key ? o : key;
        const int qi = qt * TQ + wn * 64 + ni * 16 + (lane_e & 15);
        if ((lane_e >> 4) == 0 && qi < a.n && key != 0ull) atomicMax(a.amax + qi, key);
    }
}

}  // namespace

// the 256 x 256 scan serves query batches of more than 128 rows against galleries of at least one tile per CU
bool match_scan256_applicable(int n, int Gp, int dim, int num_cus) {
    if (getenv("FID_NO_MATCH256")) return false;          // (read per call: tests compare the two paths in one process)
    return n > 128 && dim % CK == 0 && dim >= 64 && (long long)cdiv(Gp, TG) * cdiv(n, TQ) >= num_cus;
}

int match_scan256_launch(fid_ctx *ctx, const void *q, const void *g, int n, int Gp, int dim, int col0, unsigned long long *amax) {
    MGArgs a{};
    a.q = q; a.g = g; a.amax = amax;
    a.n = n; a.Gp = Gp; a.dim = dim; a.col0 = col0;
    a.n_qt = cdiv(n, TQ); a.n_gt = cdiv(Gp, TG); a.ks = dim / CK;
    const int ranges = std::max(1, std::min(a.n_gt, ctx->num_cus / a.n_qt));
    a.gt_per_wg = cdiv(a.n_gt, ranges);
    const int R = cdiv(a.n_gt, a.gt_per_wg);
    FID_REQUIRE((size_t)n * dim * 2 < 0xFFFFFF00ull && (size_t)Gp * dim * 2 + (size_t)TG * dim * 2 < 0xFFFFFF00ull, "match: operand larger than 4 GiB");
    a.q_bytes = (unsigned)((size_t)n * dim * 2);
    a.g_bytes = (unsigned)((size_t)Gp * dim * 2);
    constexpr int LDS = OFF_BEST + 16384;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)match_scan256, LDS));
    hipLaunchKernelGGL(match_scan256, dim3(a.n_qt * R), dim3(768), LDS, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
