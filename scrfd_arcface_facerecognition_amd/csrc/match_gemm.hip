// Gallery scan for a BATCH of queries against a LARGE gallery: scores = G . Q^T with a running arg-max per query, 256 x 256 tiles.
//
// reference main.py:136-142 compares one embedding with every target in a python loop; fid_match (match.hip) runs it as one GEMM with a fused
// arg-max.  On the generic 128 x 128 GEMM tile (conv.hip) that scan reaches 480-620 TFLOP/s at 1 M entries: a CU takes in ~14-16 bytes per
// clock of operands whichever way they come (LDS-DMA or register-staged loads), and a 128 x 128 x 32 step needs 16 KB for 256 matrix cycles
// per SIMD -- the fill is four times as long as the arithmetic.  Twice the tile edge halves the operand bytes per flop:
//
//   tile    = 256 gallery rows x 256 queries (fp32 sums: 128 VGPRs per lane on 8 waves -- half the CU's register file, the largest that fits);
//             wave (wm, wn) = gallery rows 128 wm .. + 127 x queries 64 wn .. + 63: 12 fragment reads feed 32 MFMAs per 32-column K-step
//   stream  = both operands by LDS-DMA in K-steps of 32 columns (2 x 16 KB), FOUR slots, pieces requested three steps ahead behind a counted
//             s_waitcnt (one workgroup per CU: nothing but the ring hides a trip to memory); rows are 64 bytes in LDS with the 16-byte groups
//             XOR-swizzled on the SOURCE address like the conv patches, so a fragment read (16 rows x 4 groups) is conflict-free
//   work    = workgroup (qt, r): query tile qt against a contiguous range r of gallery tiles, the K-step stream running on across tile
//             borders; the two query tiles of a 500-face chunk walk the same gallery range on neighbouring workgroup ids (one XCD: the
//             second reader of a gallery line finds it in L2), so HBM sees the gallery once
//   arg-max = per tile and lane a strict-'>' scan of its 128 sums in ascending gallery order (first maximum wins, a NaN never wins: the
//             reference's `sim > max_similarity` chain), folded into a running best per query column in registers; ONE packed key
//             (sortable(score) << 32 | ~index, as conv.hip's CF_ARGMAX epilogue) per query, wave and workgroup goes to memory by atomicMax
//             at the very end -- the lowest index wins among equal scores, whichever workgroup held it
//
// Rows past the gallery / the query batch read as zeros (buffer bounds): a zero score can never be a match (fid_match needs > max(0, thresh)).
#include "conv.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TG = 256, TQ = 256, CK = 32, NST = 4;      // tile, K-step, ring slots
constexpr int OP_BYTES = TG * CK * 2, ST_BYTES = 2 * OP_BYTES;   // 16 KB per operand and step
constexpr int MI = 8, NI = 4;                            // fragments of a wave: 128 gallery rows x 64 queries
constexpr int PPW = 4;                                   // DMA pieces per wave and step (32 pieces of 1 KB over 8 waves)
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

__global__ void __launch_bounds__(512, 2) match_scan256(const MGArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);
    const int qt = bid % a.n_qt, r = bid / a.n_qt;
    const int gt0 = r * a.gt_per_wg, gt1 = min(a.n_gt, gt0 + a.gt_per_wg);
    if (gt0 >= gt1) return;
    const int n_steps = (gt1 - gt0) * a.ks;
    const int frow = lane & 15, fq = lane >> 4, wm = wave & 1, wn = wave >> 1;
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc((void *)a.g, 0, a.g_bytes, 0x00020000);
    const auto rs_q = __builtin_amdgcn_make_buffer_rsrc((void *)a.q, 0, a.q_bytes, 0x00020000);

    // ---- my pieces of a step: piece j = wave + 8 k; j < 16: gallery rows 16 j .. + 15 of the tile, else query rows 16 (j - 16) .. + 15.
    // lane = (row lane >> 2, LDS group lane & 3), which holds source group (lane & 3) ^ ((row >> 1) & 3)
    const unsigned rowb = (unsigned)a.dim * 2u;
    unsigned p_off[PPW];
#pragma unroll
    for (int k = 0; k < PPW; k++) {
        const int j = (wave + 8 * k) & 15;
        p_off[k] = (unsigned)(16 * j + (lane >> 2)) * rowb + (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) * 16);
    }
    const unsigned q_base = (unsigned)qt * TQ * rowb;
    auto issue = [&](int s) __attribute__((always_inline)) {       // exactly PPW instructions; steps past the end fetch nothing (zeros into a free slot)
        const bool live = s < n_steps;
        const int t = live ? s / a.ks : 0, kk = live ? s - t * a.ks : 0;
        const unsigned g_base = (unsigned)(gt0 + t) * TG * rowb + (unsigned)kk * (CK * 2);
        const unsigned qb = q_base + (unsigned)kk * (CK * 2);
        char *dst = smem + (s % NST) * ST_BYTES;
#pragma unroll
        for (int k = 0; k < PPW; k++) {
            const int j = wave + 8 * k;
            unsigned po = p_off[k];
            asm volatile("" : "+v"(po));
            const unsigned vo = live ? (k < 2 ? g_base : qb) + po : OOB - (unsigned)k * 16u;
            if (k < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
        }
    };
    // ---- fragment addresses: row i of an operand at i * 64, its group c in slot c ^ ((i >> 1) & 3); i = 16 f + frow, so the slot is a lane constant
    const int grp = (fq ^ ((frow >> 1) & 3)) * 16;
    const int a_off = (wm * 128 + frow) * 64 + grp, b_off = OP_BYTES + (wn * 64 + frow) * 64 + grp;

    f32x4 acc[MI][NI];
    float bs[NI];
    int bi[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++) { bs[ni] = -__builtin_inff(); bi[ni] = -1; }

    issue(0); issue(1); issue(2);
    int s = 0;
    auto step = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");   // younger than my pieces of step s: those of steps s + 1 and s + 2
        raw_barrier();                                                    // everybody's pieces of step s are in; slot (s + 3) % 4 has been read
        issue(s + 3);
        const char *st = smem + (s % NST) * ST_BYTES;
        half8 bf[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ni++) bf[ni] = *(const half8 *)(st + b_off + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const half8 af = *(const half8 *)(st + a_off + mi * 1024);
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[ni], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mi][ni], 0, 0, 0);
        }
        s++;
    };
    for (int t = gt0; t < gt1; t++) {
        step(std::integral_constant<bool, true>{});
        for (int kk = 1; kk < a.ks; kk++) step(std::integral_constant<bool, false>{});
        // the tile's 128 sums of this lane per query column: gallery row t*256 + wm*128 + 16 mi + 4 fq + j, ascending in (mi, j)
        const int tbase = t * TG + wm * 128 + fq * 4;
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
            const bool up = ts > bs[ni];
            bi[ni] = up ? tbase + ti : bi[ni];
            bs[ni] = up ? ts : bs[ni];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the surplus pieces target this workgroup's LDS: drain before exit)
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
        unsigned long long key = bi[ni] >= 0 ? (((unsigned long long)sortable_f(bs[ni]) << 32) | (unsigned)(~(unsigned)(a.col0 + bi[ni]))) : 0ull;
        unsigned long long o = __shfl_xor(key, 16);
        key = o > key ? o : key;
        o = __shfl_xor(key, 32);
        key = o > key ? o : key;
        const int qi = qt * TQ + wn * 64 + ni * 16 + frow;
        if (fq == 0 && qi < a.n && key != 0ull) atomicMax(a.amax + qi, key);
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
    constexpr int LDS = NST * ST_BYTES;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)match_scan256, LDS));
    hipLaunchKernelGGL(match_scan256, dim3(a.n_qt * R), dim3(512), LDS, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
