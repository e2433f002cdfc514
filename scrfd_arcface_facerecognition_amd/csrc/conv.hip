// Implicit-GEMM convolution / GEMM on gfx950 matrix cores (v_mfma_f32_16x16x32_f16).
//
// This one kernel family is every conv of SCRFD and ArcFace (what onnxruntime's Conv/Gemm nodes do
// inside session.run, reference models/scrfd.py:83 and models/arcface.py:51), the ArcFace FC layer
// and -- with the arg-max epilogue -- the gallery cosine match (reference main.py:136-142).
//
// Data layout: activations NHWC fp16 with channels padded to a multiple of 32, weights
// [Cout][tap][Cin] fp16, accumulation fp32.  GEMM view: rows = output pixels (M = B*Ho*Wo),
// columns = output channels, K = taps x Cin walked tap-major in BK-channel steps.
//
// Workgroup = 256 threads = 4 wavefronts (one per SIMD), 2 workgroups per CU.  Per K-step the block
// stages a [BM pixels x BK] activation tile and a [BN couts x BK] weight tile through registers
// into LDS (buffer loads: out-of-image taps and out-of-range weight rows read as 0 through the
// buffer descriptor's bounds check -- no branches), XOR-swizzled in 16-byte chunks so that both the
// ds_write_b128 and the ds_read_b128 fragment reads are bank-conflict free (checked lane by lane
// against the gfx950 bank map).  Loads for step k+1 are in flight while step k is multiplied.
//
// MFMA operand roles are swapped on purpose: A = weights (rows = couts), B = pixels, so a lane's
// 4 accumulator registers are 4 CONSECUTIVE OUTPUT CHANNELS of ONE pixel -> the epilogue applies
// bias / residual / activation on float4s and stores 8 contiguous bytes per lane into NHWC.
//
// Epilogue (fused, nothing else touches the tensor): + bias (optionally one of 9 border classes:
// the exact fold of a BatchNorm that sits in front of a zero-padded conv), + residual (optionally
// nearest-2x upsampled), ReLU / PReLU, sigmoid on the first nsig channels, fp16 or fp32 store;
// or split-K partial slabs; or arg-max over columns.
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;  // beyond any descriptor's num_records -> the load returns 0

template <int BK>
__device__ __forceinline__ int swz(int row) {
    // 16-byte-chunk XOR pattern; conflict-free for ds_read_b128 / ds_write_b128
    return BK == 64 ? (row & 7) : ((-(row >> 2)) & 3);
}

__device__ __forceinline__ unsigned sortable(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Pix {
    int n, oy, ox;
};

__device__ __forceinline__ Pix decompose(const ConvArgs &a, int m) {
    Pix p;
    const int hw = a.Ho * a.Wo;
    p.n = m / hw;
    const int r = m - p.n * hw;
    p.oy = r / a.Wo;
    p.ox = r - p.oy * a.Wo;
    return p;
}

// bias + residual + activation + store for 4 consecutive output channels of one pixel
__device__ __forceinline__ void epilogue4(const ConvArgs &a, int m, int co0, f32x4 v, const Pix &p) {
    if (a.bias) {
        int cls = 0;
        if (a.flags & CF_BORDER) {
            const int yc = p.oy == 0 ? 0 : (p.oy == a.Ho - 1 ? 2 : 1);
            const int xc = p.ox == 0 ? 0 : (p.ox == a.Wo - 1 ? 2 : 1);
            cls = yc * 3 + xc;
        }
        const f32x4 b = *(const f32x4 *)(a.bias + (size_t)cls * a.Cout_p + co0);
        v += b;
    }
    if (a.res) {
        size_t roff;
        if (a.flags & CF_RES_UP2)
            roff = ((size_t)(p.n * a.res_H + (p.oy >> 1)) * a.res_W + (p.ox >> 1)) * a.res_Cp + co0;
        else
            roff = (size_t)m * a.res_Cp + co0;
        const half4 r = *(const half4 *)((const _Float16 *)a.res + roff);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
    }
    if (a.act == ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.f);
    } else if (a.act == ACT_PRELU) {
        const f32x4 s = *(const f32x4 *)(a.slope + co0);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = v[i] > 0.f ? v[i] : v[i] * s[i];
    }
    if (a.nsig > 0) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (co0 + i < a.nsig) v[i] = 1.f / (1.f + expf(-v[i]));
    }
    if (a.flags & CF_OUT_F32) {
        *(f32x4 *)((float *)a.out + (size_t)m * a.Cout_p + co0) = v;
    } else {
        half4 h;
        h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
        *(half4 *)((_Float16 *)a.out + (size_t)m * a.Cout_p + co0) = h;
    }
}

// ---- tile epilogue shared by both kernel generations: lane holds couts co0..co0+3 (rows of D) of pixel m
// (column of D); `smem` is the (idle) staging LDS, at least LDS_BYTES large ----
template <int BM, int BN, int WM, int WN, int LDS_BYTES>
__device__ __forceinline__ void tile_epilogue(const ConvArgs &a, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], int tm, int tn, int split,
                                              char *smem) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    // ---- epilogue: lane holds couts co0..co0+3 (rows of D) of pixel `m` (column of D) ----
    if (a.flags & CF_ARGMAX) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int m = tm * BM + wm * TM + mi * 16 + frow;
            unsigned long long best = 0ull;
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int co0 = tn * BN + wn * TN + ni * 16 + fq * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (co0 + j < a.Cout_p) {
                        const float sv = acc[ni][mi][j];   // a NaN score can never win (`nan > best` is False in the reference scan)
                        const unsigned long long key = ((unsigned long long)(sv == sv ? sortable(sv) : 0u) << 32) | (unsigned)(~(unsigned)(a.amax_col0 + co0 + j));
                        best = key > best ? key : best;
                    }
                }
            }
            // the 4 lanes l, l^16, l^32, l^48 hold the same pixel
            unsigned long long o = __shfl_xor(best, 16);
            best = o > best ? o : best;
            o = __shfl_xor(best, 32);
            best = o > best ? o : best;
            if (fq == 0 && m < a.M) atomicMax(a.amax + m, best);
        }
        return;
    }
    if (a.ksplit > 1) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int m = tm * BM + wm * TM + mi * 16 + frow;
            if (m >= a.M) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int co0 = tn * BN + wn * TN + ni * 16 + fq * 4;
                if (co0 < a.Cout_p) *(f32x4 *)(a.partial + ((size_t)split * a.M + m) * a.Cout_p + co0) = acc[ni][mi];
            }
        }
        return;
    }
    const bool need_pix = (a.flags & (CF_BORDER | CF_RES_UP2)) != 0;
    EpiArgs ep{a.bias, a.slope, a.res, a.out, a.Cout_p, a.Ho, a.Wo, a.act, a.flags, a.nsig, a.res_H, a.res_W, a.res_Cp};
    EpiPix px[MI];
    int co0[NI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        const int m = tm * BM + wm * TM + mi * 16 + frow;
        px[mi].valid = m < a.M;
        px[mi].m = px[mi].valid ? m : 0;
        px[mi].n = px[mi].oy = px[mi].ox = 0;
        if (need_pix && px[mi].valid) {
            const Pix p = decompose(a, m);
            px[mi].n = p.n; px[mi].oy = p.oy; px[mi].ox = p.ox;
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ni++) co0[ni] = tn * BN + wn * TN + ni * 16 + fq * 4;
    if (a.flags & CF_OUT_F32) {
        epilogue_tile<NI, MI>(ep, acc, px, co0);
        return;
    }
    // fp16 output: transpose the wave's TM x TN tile through the (now idle) staging LDS so that every lane
    // stores 16 contiguous bytes and a wave instruction covers whole pixel rows -- the 8-byte-per-lane
    // accumulator layout touches 16 cache lines per store and made the stores the bottleneck
    constexpr int OROWB = TN * 2, OCPP = TN / 8, PPI = 64 / OCPP;
    constexpr int OMASK = (OCPP & (OCPP - 1)) == 0 ? OCPP - 1 : 0;
    static_assert(4 * TM * OROWB <= LDS_BYTES, "staging LDS too small for the output transpose");
    ep_half4 hv[NI][MI];
    epilogue_values<NI, MI>(ep, acc, px, co0, hv);
    char *sS = smem + wave * (TM * OROWB);
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
            *(ep_half4 *)(sS + p * OROWB + ((c ^ (p & OMASK)) << 4) + (fq & 1) * 8) = hv[ni][mi];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int m_base = tm * BM + wm * TM, co_base = tn * BN + wn * TN;
#pragma unroll
    for (int s2 = 0; s2 < (TM + PPI - 1) / PPI; s2++) {
        const int p = s2 * PPI + lane / OCPP, c = lane % OCPP;
        if (lane < PPI * OCPP && p < TM) {
            const u32x4 v = *(const u32x4 *)(sS + p * OROWB + ((c ^ (p & OMASK)) << 4));
            const int m = m_base + p;
            if (m < a.M && co_base + c * 8 < a.Cout_p)
                *(u32x4 *)((char *)a.out + ((size_t)m * a.Cout_p + co_base) * 2 + c * 16) = v;
        }
    }
}

template <int BM, int BN, int BK, int WM, int WN>
__global__ void __launch_bounds__(256, 2) conv_mfma_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int CPR = BK / 8;     // 16-byte chunks per staged row
    constexpr int RPP = 256 / CPR;  // rows staged per pass of the 256 threads
    constexpr int A_LD = BM / RPP, B_LD = (BN + RPP - 1) / RPP;
    constexpr bool B_PARTIAL = (BN % RPP) != 0;  // fewer weight rows than one staging pass (BN=32, BK=32)
    static_assert(WM * WN == 4 && BM % RPP == 0 && TN % 16 == 0 && TM % 16 == 0 && MI >= 1 && NI >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *sA = (_Float16 *)smem;            // [2][BM][BK] pixels
    _Float16 *sB = sA + 2 * BM * BK;            // [2][BN][BK] weights

    // XCD-aware tile order: the dispatcher deals consecutive block ids round-robin over the 8 XCDs;
    // re-label so that each XCD walks a contiguous run of tiles (all column tiles of a row tile
    // first) and re-uses that row tile's activations from its own L2.  Bijective for any grid size.
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = a.tm_fast ? L / a.tiles_m : L % a.tiles_n, tm = a.tm_fast ? L % a.tiles_m : L / a.tiles_n;
    const int split = blockIdx.y;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid / CPR, lch = tid % CPR;

    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);

    // ---- per-thread staging addresses (fixed rows for the whole K loop) ----
    int a_off[A_LD];
    unsigned a_mask[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
        const int m = tm * BM + i * RPP + lrow;
        a_off[i] = 0;
        a_mask[i] = 0;
        if (m < a.M) {
            const Pix p = decompose(a, m);
            const int iy0 = p.oy * a.stride - a.pad, ix0 = p.ox * a.stride - a.pad;
            a_off[i] = ((p.n * a.H + iy0) * a.W + ix0) * a.Cin_p + lch * 8;
            unsigned mask = 0;
            for (int t = 0; t < a.T; t++) {
                const int dy = t / a.kw, dx = t - dy * a.kw;
                if ((unsigned)(iy0 + dy) < (unsigned)a.H && (unsigned)(ix0 + dx) < (unsigned)a.W) mask |= 1u << t;
            }
            a_mask[i] = mask;
        }
    }
    int b_off[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; i++) {
        const int co = tn * BN + i * RPP + lrow;
        b_off[i] = (co < a.w_rows && (!B_PARTIAL || i * RPP + lrow < BN)) ? co * a.T * a.Cin_p + lch * 8 : -1;
    }

    const int ks_begin = split * a.ksteps_per_split;
    const int ks_end = min(a.ksteps, ks_begin + a.ksteps_per_split);
    int tap = ks_begin / a.nchunk, ch = ks_begin - tap * a.nchunk;
    int tdy = tap / a.kw, tdx = tap - tdy * a.kw;

    u32x4 ra[A_LD], rb[B_LD];
    auto issue_loads = [&]() {
        const int adelta = (tdy * a.W + tdx) * a.Cin_p + ch * BK;
        const int bdelta = tap * a.Cin_p + ch * BK;
#pragma unroll
        for (int i = 0; i < A_LD; i++) {
            const unsigned vo = ((a_mask[i] >> tap) & 1u) ? (unsigned)(a_off[i] + adelta) * 2u : OOB;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LD; i++) {
            const unsigned vo = b_off[i] >= 0 ? (unsigned)(b_off[i] + bdelta) * 2u : OOB;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; i++) {
            const int row = i * RPP + lrow;
            *(u32x4 *)(sA + (buf * BM + row) * BK + ((lch ^ swz<BK>(row)) * 8)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LD; i++) {
            const int row = i * RPP + lrow;
            if (!B_PARTIAL || row < BN) *(u32x4 *)(sB + (buf * BN + row) * BK + ((lch ^ swz<BK>(row)) * 8)) = rb[i];
        }
    };
    auto advance = [&]() {
        if (++ch == a.nchunk) {
            ch = 0;
            ++tap;
            if (++tdx == a.kw) { tdx = 0; ++tdy; }
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (ks_begin < ks_end) {
        issue_loads();
        store_tiles(0);
    }
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4;
    for (int ks = ks_begin; ks < ks_end; ks++) {
        const int cur = (ks - ks_begin) & 1;
        const bool more = ks + 1 < ks_end;
        if (more) {
            advance();
            issue_loads();
        }
        const _Float16 *cA = sA + cur * BM * BK, *cB = sB + cur * BN * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 32; kk++) {
            half8 wf[NI], pf[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int row = wn * TN + ni * 16 + frow;
                wf[ni] = *(const half8 *)(cB + row * BK + (((kk * 4 + fq) ^ swz<BK>(row)) * 8));
            }
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int row = wm * TM + mi * 16 + frow;
                pf[mi] = *(const half8 *)(cA + row * BK + (((kk * 4 + fq) ^ swz<BK>(row)) * 8));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[mi], acc[ni][mi], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
    }

    tile_epilogue<BM, BN, WM, WN, 2 * (BM + BN) * BK * 2>(a, acc, tm, tn, split, smem);
}

// ================================================================================================
// Generation 2: LDS-DMA ring.  Same GEMM view and fragment layout as conv_mfma_kernel, but
//   * tiles go global -> LDS directly (buffer_load ... lds, 1 KB per wave instruction: no VGPR staging, no
//     ds_write), swizzle applied to each lane's SOURCE address, destination lane-linear;
//   * NS ring slots, NS-1 K-steps in flight behind a COUNTED s_waitcnt vmcnt (never 0 in the loop) and one raw
//     s_barrier per K-step.  With one K-step of prefetch (generation 1) a K-step costs one L2 round trip
//     (~1.1 us measured: 14 % MFMA utilisation on the 14x14 layers); with 3 in flight the trip is amortised;
//   * all per-K-step addressing is wave-uniform scalar work (tap, channel chunk) plus one select per row.
// ================================================================================================
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int BK, int NS, int WM, int WN, bool PF = false>
__global__ void __launch_bounds__(256, (NS * (BM + (BN + 256 / (BK / 8) - 1) / (256 / (BK / 8)) * (256 / (BK / 8))) * BK * 2 <= 80 * 1024) ? 2 : 1)
    conv_mfma_dma_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int ROWB = BK * 2, CPR = BK / 8;       // bytes / 16-byte chunks per staged row
    constexpr int RPW = 64 / CPR, RPP = 4 * RPW;     // rows per 1 KB wave instruction / per pass of the 4 waves
    constexpr int BN_ALLOC = (BN + RPP - 1) / RPP * RPP;
    constexpr int A_LD = BM / RPP, B_LD = BN_ALLOC / RPP, L = A_LD + B_LD;   // DMA instructions per wave per stage
    constexpr int STAGE = (BM + BN_ALLOC) * ROWB;
    static_assert(WM * WN == 4 && BM % RPP == 0 && NS >= 2 && NS <= 4 && NS * STAGE <= 160 * 1024, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    const int Lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = a.tm_fast ? Lid / a.tiles_m : Lid % a.tiles_n, tm = a.tm_fast ? Lid % a.tiles_m : Lid / a.tiles_n;
    const int split = blockIdx.y;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = wave * RPW + lane / CPR;                     // row inside a pass
    const int my_chunk = (lane % CPR) ^ swz<BK>(lrow);            // K chunk this lane fetches: LDS slot ^ swizzle(row)

    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    const auto rs_in2 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in2 ? a.in2 : a.in), 0, a.in2 ? a.in2_bytes : a.in_bytes, 0x00020000);
    const int krow = a.in2 ? a.krow : a.T * a.Cin_p;            // halfs per weight row

    int a_off[A_LD], a_off2[A_LD];                               // (a_off2: the pixel's first shortcut tap in the second tensor; -1: no pixel)
    unsigned a_mask[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
        const int m = tm * BM + i * RPP + lrow;
        a_off[i] = 0;
        a_off2[i] = -1;
        a_mask[i] = 0;
        if (m < a.M) {
            const Pix p = decompose(a, m);
            const int iy0 = p.oy * a.stride - a.pad, ix0 = p.ox * a.stride - a.pad;
            a_off[i] = ((p.n * a.H + iy0) * a.W + ix0) * a.Cin_p + my_chunk * 8;
            if (a.in2) a_off2[i] = ((p.n * a.H2 + p.oy * a.s2) * a.W2 + p.ox * a.s2) * a.Cin2_p + my_chunk * 8;
            unsigned mask = 0;
            for (int t = 0; t < a.T; t++) {
                const int dy = t / a.kw, dx = t - dy * a.kw;
                if ((unsigned)(iy0 + dy) < (unsigned)a.H && (unsigned)(ix0 + dx) < (unsigned)a.W) mask |= 1u << t;
            }
            a_mask[i] = mask;
        }
    }
    int b_off[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; i++) {
        const int row = i * RPP + lrow, co = tn * BN + row;
        b_off[i] = (row < BN && co < a.w_rows) ? co * krow + my_chunk * 8 : -1;
    }

    const int ks_begin = split * a.ksteps_per_split;
    const int ks_end = min(a.ksteps, ks_begin + a.ksteps_per_split);
    const int nk = ks_end - ks_begin;
    // wave-uniform position of the next stage to issue: tap (dy, dx) and channel chunk
    // (the shortcut's K-steps follow the main taps: tap = T + t2)
    const int ks_main = a.T * a.nchunk;
    int tap = ks_begin < ks_main ? ks_begin / a.nchunk : a.T + (ks_begin - ks_main) / a.nchunk2;
    int ch = ks_begin < ks_main ? ks_begin - tap * a.nchunk : (ks_begin - ks_main) - (tap - a.T) * a.nchunk2;
    int tdy = tap / a.kw, tdx = tap - tdy * a.kw;

    // one DMA instruction (1 KB) of the stage being issued: j < A_LD -> activation rows, else weight rows
    int adelta = 0, bdelta = 0;
    auto stage_begin = [&]() {
        if (tap < a.T) {
            adelta = (tdy * a.W + tdx) * a.Cin_p + ch * BK;
            bdelta = tap * a.Cin_p + ch * BK;
        } else {                                                // a shortcut tap: (t2 / kw2, t2 % kw2) in the second tensor
            const int t2 = tap - a.T, dy2 = t2 / a.kw2, dx2 = t2 - dy2 * a.kw2;
            adelta = (dy2 * a.W2 + dx2) * a.Cin2_p + ch * BK;
            bdelta = a.T * a.Cin_p + t2 * a.Cin2_p + ch * BK;
        }
    };
    auto stage_piece = [&](int slot, int j) {
        char *dst = smem + slot * STAGE + wave * 1024;
        if (j < A_LD) {
            if (tap < a.T) {
                const unsigned vo = ((a_mask[j] >> tap) & 1u) ? (unsigned)(a_off[j] + adelta) * 2u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 4096), 16, vo, 0, 0, 0);
            } else {                                            // (wave-uniform branch: the same number of DMA instructions either way)
                const unsigned vo = a_off2[j] >= 0 ? (unsigned)(a_off2[j] + adelta) * 2u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in2, (__attribute__((address_space(3))) void *)(dst + j * 4096), 16, vo, 0, 0, 0);
            }
        } else {
            const int i = j - A_LD;
            const unsigned vo = b_off[i] >= 0 ? (unsigned)(b_off[i] + bdelta) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(dst + BM * ROWB + i * 4096), 16, vo, 0, 0, 0);
        }
    };
    auto stage_end = [&]() {
        if (++ch == (tap < a.T ? a.nchunk : a.nchunk2)) {
            ch = 0;
            ++tap;
            if (++tdx == a.kw) { tdx = 0; ++tdy; }
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; s++)
        if (s < nk) {
            stage_begin();
#pragma unroll
            for (int j = 0; j < L; j++) stage_piece(s, j);
            stage_end();
        }

    const int frow = lane & 15, fq = lane >> 4;
    constexpr int KK = BK / 32;
    if constexpr (PF) {
        // Fragment-prefetch variant (the autotuner's "ns = 5": 4 slots).  With 64x64 tiles a K-step is only 8 MFMAs per
        // wave (128 cycles) while the barrier plus the first ds_read latency of the step cost ~300: the kernel ran at a
        // quarter of the MFMA rate even with the DMA switched off.  Here the barrier of step k also covers stage k+1, so
        // the fragments of (k+1, kk = 0) are requested during step k's last MFMA group and a step starts multiplying at
        // once.  Price: one K-step less of DMA look-ahead (NS-2 stages in flight instead of NS-1).
        static_assert(NS == 4, "fragment prefetch needs 4 ring slots");
        half8 wf[2][NI], pf[2][MI];
        auto load_frags = [&](int kstep, int kk, int set) {
            const char *cA = smem + (kstep % NS) * STAGE, *cB = cA + BM * ROWB;
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int row = wm * TM + mi * 16 + frow;
                pf[set][mi] = *(const half8 *)(cA + row * ROWB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int row = wn * TN + ni * 16 + frow;
                wf[set][ni] = *(const half8 *)(cB + row * ROWB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4));
            }
        };
        // stage 0 landed (stages 1, 2 may be in flight)
        if (nk >= 3) wait_vmcnt<2 * L>();
        else if (nk == 2) wait_vmcnt<L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0, 0);
        for (int k = 0; k < nk; k++) {
            // stage k+1 must have landed for everyone: only stage k+2 may stay in flight
            if (k + 2 < nk) wait_vmcnt<L>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();       // also: everyone is done reading slot (k-1)%NS
            const bool more = k + NS - 1 < nk;
            if (more) {
                stage_begin();
#pragma unroll
                for (int j = 0; j < L; j++) stage_piece((k + NS - 1) % NS, j);
            }
#pragma unroll
            for (int kk = 0; kk < KK; kk++) {
                const int cur = kk & 1;
                if (kk + 1 < KK) load_frags(k, kk + 1, cur ^ 1);
                else if (k + 1 < nk) load_frags(k + 1, 0, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cur][ni], pf[cur][mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) stage_end();
        }
    } else
    for (int k = 0; k < nk; k++) {
        // stage k must have landed: allow only the younger stages (at most NS-2 of them) to be outstanding
        const int younger = min(NS - 2, nk - 1 - k);
        if (younger >= 2) wait_vmcnt<(NS >= 4 ? 2 : 0) * L>();
        else if (younger == 1) wait_vmcnt<(NS >= 3 ? 1 : 0) * L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();       // everyone's DMAs of stage k landed; everyone is done reading slot (k-1)%NS
        const bool more = k + NS - 1 < nk;
        const int nslot = (k + NS - 1) % NS;
        if (more) {      // (spreading these between the MFMA groups was measured: 3-4 % slower)
            stage_begin();
#pragma unroll
            for (int j = 0; j < L; j++) stage_piece(nslot, j);
        }
        const char *cA = smem + (k % NS) * STAGE, *cB = cA + BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            half8 wf[NI], pf[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int row = wm * TM + mi * 16 + frow;
                pf[mi] = *(const half8 *)(cA + row * ROWB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int row = wn * TN + ni * 16 + frow;
                wf[ni] = *(const half8 *)(cB + row * ROWB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[mi], acc[ni][mi], 0, 0, 0);
        }
        if (more) stage_end();
    }
    __syncthreads();   // all LDS reads done before the epilogue reuses the ring as staging
    tile_epilogue<BM, BN, WM, WN, NS * STAGE>(a, acc, tm, tn, split, smem);
}

// ================================================================================================
// Generation 6: generation 2 with the work split by ROLE (the conv_pc.hip lesson applied to the implicit GEMM).
// In generation 2 every wave issues its 6-8 LDS-DMA pieces right after the K-step's barrier and only then starts to
// multiply: a wave blocks in each DMA issue while the CU's address path serves the others (~50 cycles a piece), which
// is as long as the K-step's 16-32 MFMAs.  Here waves 0..3 (consumers) only read fragments and multiply; waves 4 and 5
// (producers) issue the whole stage -- each the pieces of two "virtual" generation-2 waves, same LDS image -- and
// wait for their own DMAs (counted vmcnt) before they join the barrier.
// ================================================================================================
template <int BM, int BN, int BK, int NS, int WM, int WN>
__global__ void __launch_bounds__(384, ((NS * (BM + (BN + 256 / (BK / 8) - 1) / (256 / (BK / 8)) * (256 / (BK / 8))) * BK * 2 <= 80 * 1024) && (BM / WM / 16) * (BN / WN / 16) < 12) ? 3 : 2)
    conv_mfma_pc_kernel(const ConvArgs a) {
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int ROWB = BK * 2, CPR = BK / 8;
    constexpr int RPW = 64 / CPR, RPP = 4 * RPW;
    constexpr int BN_ALLOC = (BN + RPP - 1) / RPP * RPP;
    constexpr int A_LD = BM / RPP, B_LD = BN_ALLOC / RPP, L = A_LD + B_LD;   // pieces per virtual wave and stage
    constexpr int STAGE = (BM + BN_ALLOC) * ROWB;
    static_assert(WM * WN == 4 && BM % RPP == 0 && NS >= 3 && NS <= 4 && NS * STAGE <= 160 * 1024, "tile shape");
    static_assert((NS - 1) * 2 * L <= 63, "a producer's DMAs in flight must fit the vmcnt counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    const int Lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tn = a.tm_fast ? Lid / a.tiles_m : Lid % a.tiles_n, tm = a.tm_fast ? Lid % a.tiles_m : Lid / a.tiles_n;
    const int split = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int ks_begin = split * a.ksteps_per_split;
    const int ks_end = min(a.ksteps, ks_begin + a.ksteps_per_split);
    const int nk = ks_end - ks_begin;

    if (wave >= 4) {
        // ---------------------------------------- producers ----------------------------------------
        const int pw = wave - 4;
        const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        int a_off[2][A_LD], b_off[2][B_LD];
        unsigned a_mask[2][A_LD];
#pragma unroll
        for (int vw = 0; vw < 2; vw++) {
            const int lrow = (2 * pw + vw) * RPW + lane / CPR;
            const int my_chunk = (lane % CPR) ^ swz<BK>(lrow);
#pragma unroll
            for (int i = 0; i < A_LD; i++) {
                const int m = tm * BM + i * RPP + lrow;
                a_off[vw][i] = 0;
                a_mask[vw][i] = 0;
                if (m < a.M) {
                    const Pix p = decompose(a, m);
                    const int iy0 = p.oy * a.stride - a.pad, ix0 = p.ox * a.stride - a.pad;
                    a_off[vw][i] = ((p.n * a.H + iy0) * a.W + ix0) * a.Cin_p + my_chunk * 8;
                    unsigned mask = 0;
                    for (int t = 0; t < a.T; t++) {
                        const int dy = t / a.kw, dx = t - dy * a.kw;
                        if ((unsigned)(iy0 + dy) < (unsigned)a.H && (unsigned)(ix0 + dx) < (unsigned)a.W) mask |= 1u << t;
                    }
                    a_mask[vw][i] = mask;
                }
            }
#pragma unroll
            for (int i = 0; i < B_LD; i++) {
                const int row = i * RPP + lrow, co = tn * BN + row;
                b_off[vw][i] = (row < BN && co < a.w_rows) ? co * a.T * a.Cin_p + my_chunk * 8 : -1;
            }
        }
        int tap = ks_begin / a.nchunk, ch = ks_begin - tap * a.nchunk;
        int tdy = tap / a.kw, tdx = tap - tdy * a.kw;
        auto issue_stage = [&](int slot) {
            const int adelta = (tdy * a.W + tdx) * a.Cin_p + ch * BK, bdelta = tap * a.Cin_p + ch * BK;
#pragma unroll
            for (int vw = 0; vw < 2; vw++) {
                char *dst = smem + slot * STAGE + (2 * pw + vw) * 1024;
#pragma unroll
                for (int j = 0; j < A_LD; j++) {
                    const unsigned vo = ((a_mask[vw][j] >> tap) & 1u) ? (unsigned)(a_off[vw][j] + adelta) * 2u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 4096), 16, vo, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < B_LD; i++) {
                    const unsigned vo = b_off[vw][i] >= 0 ? (unsigned)(b_off[vw][i] + bdelta) * 2u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(dst + BM * ROWB + i * 4096), 16, vo, 0, 0, 0);
                }
            }
            if (++ch == a.nchunk) {
                ch = 0;
                ++tap;
                if (++tdx == a.kw) { tdx = 0; ++tdy; }
            }
        };
#pragma unroll
        for (int s = 0; s < NS - 1; s++)
            if (s < nk) issue_stage(s);
        for (int k = 0; k < nk; k++) {
            // stage k must have landed: only the younger stages (at most NS-2 of them) may stay in flight
            const int younger = min(NS - 2, nk - 1 - k);
            if (younger >= 2) wait_vmcnt<(NS >= 4 ? 2 : 0) * 2 * L>();
            else if (younger == 1) wait_vmcnt<2 * L>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();       // stage k landed for everyone; the consumers are done with slot (k-1)%NS
            if (k + NS - 1 < nk) issue_stage((k + NS - 1) % NS);
        }
        __builtin_amdgcn_s_barrier();           // the consumers' barrier before their epilogue
        return;
    }

    // ---------------------------------------- consumers ----------------------------------------
    const int wm = wave / WN, wn = wave % WN;
    f32x4 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    constexpr int KK = BK / 32;
    for (int k = 0; k < nk; k++) {
        __builtin_amdgcn_s_barrier();
        const char *cA = smem + (k % NS) * STAGE, *cB = cA + BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            half8 wf[NI], pf[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                const int row = wm * TM + mi * 16 + frow;
                pf[mi] = *(const half8 *)(cA + row * ROWB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int row = wn * TN + ni * 16 + frow;
                wf[ni] = *(const half8 *)(cB + row * ROWB + (((kk * 4 + fq) ^ swz<BK>(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], pf[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all LDS reads done before the epilogue reuses the ring as staging
    tile_epilogue<BM, BN, WM, WN, NS * STAGE>(a, acc, tm, tn, split, smem);
}

// second pass of a split-K conv: sum the slabs in a fixed order (bit-reproducible), then the epilogue
__global__ void __launch_bounds__(256) splitk_epilogue(const ConvArgs a) {
    const int c4 = a.Cout_p >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.M * c4) return;
    const int m = (int)(idx / c4), co0 = (int)(idx - (long long)m * c4) * 4;
    f32x4 v = *(const f32x4 *)(a.partial + (size_t)m * a.Cout_p + co0);
    for (int s = 1; s < a.ksplit; s++) v += *(const f32x4 *)(a.partial + ((size_t)s * a.M + m) * a.Cout_p + co0);
    Pix p{0, 0, 0};
    if (a.flags & (CF_BORDER | CF_RES_UP2)) p = decompose(a, m);
    epilogue4(a, m, co0, v, p);
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(fid_ctx *ctx, const ConvArgs &a) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * BK * 2;
    if (lds > 48 * 1024) FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_mfma_kernel<BM, BN, BK, WM, WN>, (int)((int)lds)));
    dim3 grid(a.tiles_m * a.tiles_n, a.ksplit);
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, BK, WM, WN>), grid, dim3(256), lds, ctx->stream, a);
    return FID_OK;
}

template <int BM, int BN, int BK, int NS, int WM, int WN, bool PF = false>
int launch_dma(fid_ctx *ctx, const ConvArgs &a) {
    constexpr int RPP = 4 * (64 / (BK / 8));
    constexpr size_t lds = (size_t)NS * (BM + (BN + RPP - 1) / RPP * RPP) * BK * 2;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_mfma_dma_kernel<BM, BN, BK, NS, WM, WN, PF>, (int)((int)lds)));
    dim3 grid(a.tiles_m * a.tiles_n, a.ksplit);
    hipLaunchKernelGGL((conv_mfma_dma_kernel<BM, BN, BK, NS, WM, WN, PF>), grid, dim3(256), lds, ctx->stream, a);
    return FID_OK;
}

template <int BM, int BN, int BK, int NS, int WM, int WN>
int launch_pcg(fid_ctx *ctx, const ConvArgs &a) {
    constexpr int RPP = 4 * (64 / (BK / 8));
    constexpr size_t lds = (size_t)NS * (BM + (BN + RPP - 1) / RPP * RPP) * BK * 2;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv_mfma_pc_kernel<BM, BN, BK, NS, WM, WN>, (int)((int)lds)));
    dim3 grid(a.tiles_m * a.tiles_n, a.ksplit);
    hipLaunchKernelGGL((conv_mfma_pc_kernel<BM, BN, BK, NS, WM, WN>), grid, dim3(384), lds, ctx->stream, a);
    return FID_OK;
}

}  // namespace

static bool dma_have(int bm, int bn, int bk) {
    if (bk == 64) return (bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64);
    return bm == 128 && (bn == 128 || bn == 96 || bn == 64 || bn == 32);
}

ConvPlan conv_plan(const ConvArgs &a, int num_cus, bool allow_split) {
    if (a.out2) { ConvPlan d{}; d.gen = 10; d.ksplit = 1; d.bm = 128; d.bn = a.w_rows; d.bk = 32; return d; }
    ConvPlan p{};
    static const bool use_v1 = getenv("FID_CONV_V1") != nullptr;
    auto tiles = [&](int bm, int bn) { return (long long)cdiv(a.M, bm) * cdiv(a.Cout_p, bn); };
    if (!use_v1 || a.in2) {
        // ---- generation 2: LDS-DMA ring, BK = 64 over the flattened (tap, channel) axis ----
        p.gen = 2; p.bk = (a.Cin_p % 64 == 0 && (!a.in2 || a.Cin2_p % 64 == 0)) ? 64 : 32; p.ns = 4;
        p.bm = 128;
        if (a.Cout_p % 128 == 0) p.bn = 128;
        else if (a.Cout_p % 96 == 0) p.bn = 96;
        else if (a.Cout_p > 64) p.bn = 128;
        else p.bn = a.Cout_p > 32 ? 64 : 32;
        // too few tiles to occupy the chip: smaller tiles (twice as many blocks, two resident per CU)
        if (tiles(p.bm, p.bn) < (long long)num_cus && p.bn == 128) p.bn = 64;
        if (tiles(p.bm, p.bn) < (long long)num_cus && p.bn == 64 && p.bk == 64) p.bm = 64;
        if (p.bk == 64 && p.bn != 128 && p.bn != 64) p.bn = a.Cout_p > 64 ? 128 : 64;
    } else {
        p.gen = 1;
        p.bk = (a.Cin_p % 64 == 0) ? 64 : 32;
        if (p.bk == 64) {
            p.bn = a.Cout_p >= 128 ? 128 : 64;
            p.bm = 128;
            if (tiles(p.bm, p.bn) < 2LL * num_cus && p.bn == 128) p.bn = 64;
            if (tiles(p.bm, p.bn) < 2LL * num_cus) p.bm = 64, p.bn = 64;
        } else {
            p.bm = 128;
            p.bn = a.Cout_p >= 128 ? 128 : (a.Cout_p > 64 ? 96 : (a.Cout_p > 32 ? 64 : 32));
            if (a.Cout_p % 96 == 0 && a.Cout_p % 128 != 0) p.bn = 96;
            if (tiles(p.bm, p.bn) < 2LL * num_cus && p.bn == 128) p.bn = 64;
        }
    }
    const int ksteps = cdiv(a.kh * a.kw * a.Cin_p, p.bk);
    const long long t = tiles(p.bm, p.bn);
    p.ksplit = 1;
    if (allow_split && !(a.flags & CF_ARGMAX) && t * 2 <= num_cus && ksteps >= 16) {
        int want = (int)((num_cus + t - 1) / t);
        p.ksplit = std::max(1, std::min(want, ksteps / 8));
    }
    // experiment hook (tools): FID_CONV_FORCE="bm,bn,ksplit[,ns]" overrides the heuristic
    if (const char *f = getenv("FID_CONV_FORCE")) {
        int bm = 0, bn = 0, ks = 0, ns = 0;
        const int n = sscanf(f, "%d,%d,%d,%d", &bm, &bn, &ks, &ns);
        if (n >= 3) {
            const bool have = p.gen == 2 ? dma_have(bm, bn, p.bk)
                              : (p.bk == 64 ? ((bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64))
                                            : (bm == 128 && (bn == 128 || bn == 96 || bn == 64 || bn == 32)));
            if (have) { p.bm = bm; p.bn = bn; }
            if (ks >= 1 && allow_split && !(a.flags & CF_ARGMAX)) p.ksplit = std::max(1, std::min(ks, ksteps / 2));
            if (n >= 4 && ns >= 3 && ns <= 4 && p.bk == 64) p.ns = ns;
        }
    }
    p.partial_bytes = p.ksplit > 1 ? (size_t)p.ksplit * a.M * a.Cout_p * 4 : 0;
    return p;
}

static std::vector<ConvPlan> conv_candidates_all(const ConvArgs &a, int num_cus, bool allow_split);

std::vector<ConvPlan> conv_candidates(const ConvArgs &a, int num_cus, bool allow_split) {
    std::vector<ConvPlan> all = conv_candidates_all(a, num_cus, allow_split);
    if (!a.in2) return all;
    std::vector<ConvPlan> only;                      // fused shortcut (extra K-steps on a second tensor): the LDS-DMA implicit GEMM ...
    for (const ConvPlan &c : all)
        if (c.gen == 2 && a.Cin2_p % c.bk == 0) only.push_back(c);
    if (conv_s2_applicable(a)) {                     // ... and the parity-plane stride-2 kernel with the shortcut as one more step per item (ns = 10)
        ConvPlan d{};
        d.gen = 10; d.ksplit = 1; d.bm = 128; d.bn = a.Cout_p; d.bk = 32; d.ns = 10;
        only.push_back(d);
    }
    return only;
}

static std::vector<ConvPlan> conv_candidates_all(const ConvArgs &a, int num_cus, bool allow_split) {
    if (a.out2) {                                   // fused shortcut + stride-2 conv: one kernel takes it
        std::vector<ConvPlan> only;
        if (conv_s2_applicable(a)) { ConvPlan d{}; d.gen = 10; d.ksplit = 1; d.bm = 128; d.bn = a.w_rows; d.bk = 32; only.push_back(d); }
        return only;
    }
    std::vector<ConvPlan> out;
    if (conv_direct_applicable(a)) { ConvPlan d{}; d.gen = 0; d.ksplit = 1; out.push_back(d); }
    if (conv_chunked_applicable(a) && !getenv("FID_NO_CHUNKED")) {
        ConvPlan d{};
        d.gen = 3; d.ksplit = 1; d.bm = 256; d.bk = 32;
        d.bn = 64; out.push_back(d);
        if (a.Cout_p % 96 == 0) { d.bn = 96; out.push_back(d); }
    }
    if (conv_pcr_applicable(a)) { ConvPlan d{}; d.gen = 7; d.ksplit = 1; d.bm = 256; d.bn = 64; d.bk = 32; out.push_back(d); }
    if (conv_pc2_applicable(a)) { ConvPlan d{}; d.gen = 8; d.ksplit = 1; d.bm = 512; d.bn = 64; d.bk = 32; out.push_back(d); }
    if (conv_s2_applicable(a)) { ConvPlan d{}; d.gen = 10; d.ksplit = 1; d.bm = 128; d.bn = a.Cout_p; d.bk = 32; out.push_back(d); }
    if (conv_gw_applicable(a)) {
        ConvPlan d{};
        d.gen = 11; d.ksplit = 1; d.bk = 32;
        for (int bm : {128, 64})
            for (int bn : {256, 128}) {
                if (bn == 256 && a.Cout_p <= 128) continue;
                d.bm = bm; d.bn = bn; out.push_back(d);
            }
    }
    if (conv_wr_applicable(a)) {
        ConvPlan d{};
        d.gen = 9; d.ksplit = 1; d.bk = 32;
        d.bm = 512; d.bn = 128; out.push_back(d);      // a pair of tiles x 128 couts
        d.bm = 256; d.bn = 64; out.push_back(d);       // one tile x 64 couts (few tiles: more items)
        d.ns = 4; out.push_back(d); d.ns = 0;          // ... with the patches three steps ahead (small batches: one workgroup per CU)
        if (conv_wr_resident_ok(a)) { d.bm = 256; d.bn = a.Cout_p; d.ns = 1; out.push_back(d); }   // the layer's weights resident in registers
        if (conv_strip_ok(a)) {                        // the same on x-packed STRIP tiles (round 5): ns = 8 streaming, 9 resident
            d.ns = 8; d.bm = 512; d.bn = 128; out.push_back(d);
            d.bm = 256; d.bn = 64; out.push_back(d);
            if (a.Cout_p > 64) { d.bm = 256; d.bn = 128; out.push_back(d); }     // one tile x 128 couts (eight waves)
            if (conv_wr_resident_ok(a)) { d.bm = 256; d.bn = a.Cout_p; d.ns = 9; out.push_back(d); }
            d.ns = 0;
        }
        if (conv_ks_applicable(a)) {                   // one tile x 64 couts, the K axis split over two wave groups (conv_ks.hip)
            d.bm = 256; d.bn = 64; d.ns = 6; out.push_back(d);
            const int items = conv_ks_items(a);        // few items: also on half the CUs with two or more items per workgroup (tile 512; fewer CUs for longer, FID_TUNE_SHARE)
            if (items >= 2 && items <= 2 * num_cus) { d.bm = 512; out.push_back(d); }
            if (conv_ks_strip_applicable(a)) { d.bm = 256; d.bn = 64; d.ns = 7; out.push_back(d); }   // ... on x-packed STRIP tiles
        }
    }
    else if (conv_ks_applicable(a)) {                  // 7x7 maps: four images per 16x16 tile (conv_ks.hip, MOSAIC)
        ConvPlan d{};
        d.gen = 9; d.ksplit = 1; d.bk = 32; d.bm = 256; d.bn = 64; d.ns = 6; out.push_back(d);
    }
    if (conv_pc_applicable(a)) {
        ConvPlan d{};
        d.gen = 5; d.ksplit = 1; d.bm = 256; d.bk = 32;
        if (a.flags & CF_OUT_F32) { d.bn = 32; d.ns = 0; out.push_back(d); }
        else {
        d.bn = 64; d.ns = 0; out.push_back(d);
        // weights two steps ahead instead of patches: measured equal or 1-2 % slower everywhere; kept for tests / experiments
        if (getenv("FID_FORCE_NS")) { d.ns = 1; out.push_back(d); }
        d.ns = 0;
        if (a.Cout_p % 96 == 0) { d.bn = 96; out.push_back(d); }
        }
    }
    if (conv_pp_applicable(a)) {
        ConvPlan d{};
        d.gen = 4; d.ksplit = 1; d.bm = 512; d.bk = 32;
        for (int cb : {32, 48, 64}) {
            if (cb > 32 && cdiv(a.Cout_p, cb) * cb > cdiv(a.Cout_p, 32) * 32) continue;   // would pad more couts than cb = 32
            d.bn = cb;
            out.push_back(d);
        }
    }
    const int bk = (a.Cin_p % 64 == 0) ? 64 : 32;
    const int ksteps = a.kh * a.kw * (a.Cin_p / bk);
    auto tiles = [&](int bm, int bn) { return (long long)cdiv(a.M, bm) * cdiv(a.Cout_p, bn); };
    auto add = [&](int gen, int bm, int bn, int ns) {
        const int smallest = bk == 64 ? 64 : 32;
        if (bn > smallest && bn >= 2 * ((a.Cout_p + 31) / 32 * 32)) return;   // mostly-empty column tile
        if (bn == 32 && a.Cout_p > 32) return;
        if (bn == 96 && a.Cout_p % 96 != 0) return;
        ConvPlan p{};
        p.gen = gen; p.bm = bm; p.bn = bn; p.bk = bk; p.ns = ns; p.ksplit = 1;
        out.push_back(p);
        const long long t = tiles(bm, bn);
        if (allow_split && !(a.flags & CF_ARGMAX) && t < num_cus && ksteps >= 16) {
            const int want = (int)((num_cus + t - 1) / t);
            const int ks = std::max(1, std::min(want, ksteps / 8));
            if (ks > 1) { p.ksplit = ks; out.push_back(p); }
        }
    };
    // Generation 6 (two producer waves per workgroup) is opt-in: measured 30-40 % SLOWER than generation 2 on the large-M
    // layers (two waves cannot issue a stage's 16-32 DMA pieces in a K-step's 256-512 MFMA cycles: ~75 cycles per piece and
    // wave) and 5-12 % faster only on the 64-face 14x14 / 7x7 layers, which the halo-patch kernels serve better anyway.
    const char *fg6 = getenv("FID_FORCE_GEN");
    const bool gemm_pc = getenv("FID_GEMM_PC") != nullptr || (fg6 && atoi(fg6) == 6);
    if (bk == 64) {
        for (int gen = 1; gen <= 2; gen++) {
            add(gen, 128, 128, 4); add(gen, 128, 64, 4); add(gen, 64, 64, 4);
        }
        add(2, 128, 128, 3); add(2, 128, 64, 3); add(2, 64, 64, 3);
        add(2, 128, 128, 5); add(2, 128, 64, 5); add(2, 64, 64, 5);   // ns = 5: 4 slots + fragment prefetch across K-steps
        if (gemm_pc) { add(6, 128, 128, 4); add(6, 128, 64, 4); add(6, 64, 64, 4); add(6, 128, 64, 3); add(6, 64, 64, 3); }
    } else {
        for (int gen = 1; gen <= 2; gen++) {
            add(gen, 128, 128, 4); add(gen, 128, 96, 4); add(gen, 128, 64, 4); add(gen, 128, 32, 4);
        }
        if (gemm_pc) { add(6, 128, 128, 4); add(6, 128, 96, 4); add(6, 128, 64, 4); }
    }
    if (out.empty()) out.push_back(conv_plan(a, num_cus, false));
    for (auto &p : out) p.partial_bytes = p.ksplit > 1 ? (size_t)p.ksplit * a.M * a.Cout_p * 4 : 0;
    return out;
}

int plan_alt_kind(const ConvPlan &plan) {
    static const bool pc2_packed = getenv("FID_PC2_PLAIN") == nullptr;
    if (plan.gen == 8) return pc2_packed ? 1 : 0;
    if (plan.gen == 9 || plan.gen == 10 || (plan.gen == 12 && plan.ns == 10)) return 2;
    if (plan.gen == 11) return 3;
    return 0;
}

// The fraction of the chip's CUs a candidate's launch occupies (1 = all of them, or a family whose grid is not modelled here).  The
// autotuner can weigh it in (FID_TUNE_SHARE, net.hip): with two batches in flight on two streams a launch that holds half the CUs
// for the same time leaves the other half to the other lane.
float conv_plan_cu_share(const ConvArgs &a, const ConvPlan &plan, int num_cus) {
    long long wgs = -1;
    if (plan.gen == 1 || plan.gen == 2 || plan.gen == 11) wgs = (long long)cdiv(a.M, plan.bm) * cdiv(a.Cout_p, plan.bn) * std::max(1, plan.ksplit);
    else if (plan.gen == 9 && plan.ns == 6) wgs = plan.bm >= 512 ? std::min(cdiv(conv_ks_items(a), 2), num_cus / 2) : conv_ks_items(a);
    else if (plan.gen == 9 && plan.ns == 7) wgs = conv_ks_items(a, true);
    else if (plan.gen == 9 && plan.ns != 1 && plan.ns != 9) {
        const long long tiles = plan.ns == 8 ? (long long)cdiv((a.M / (a.Ho * a.Wo)) * a.Wo, 16) * cdiv(a.Ho, 16)
                                             : (long long)(a.M / (a.Ho * a.Wo)) * cdiv(a.Ho, 14) * cdiv(a.Wo, 14);      // (14 / 16-row tiles: the smaller count)
        wgs = cdiv((int)tiles, plan.bm / 256) * cdiv(a.Cout_p, plan.bn);
    }
    if (wgs < 0 || wgs >= num_cus) return 1.f;
    return (float)wgs / (float)num_cus;
}

int conv_launch(fid_ctx *ctx, ConvArgs a, const ConvPlan &plan) {
    if (plan.gen == 0) return conv_direct_launch(ctx, a);
    if (plan.gen == 3) return conv_chunked_launch(ctx, a, plan.bn);
    if (plan.gen == 4) return conv_pp_launch(ctx, a, plan.bn);
    if (plan.gen == 5) return conv_pc_launch(ctx, a, plan.bn, plan.ns);
    if (plan.gen == 7) return conv_pcr_launch(ctx, a);
    if (plan.gen == 8) return conv_pc2_launch(ctx, a);
    if (plan.gen == 10) return conv_s2_launch(ctx, a);
    if (plan.gen == 11) return conv_gw_launch(ctx, a, plan.bm, plan.bn);
    if (plan.gen == 9 && plan.ns == 6) return conv_ks_launch(ctx, a, plan.bm / 256);
    if (plan.gen == 9 && plan.ns == 7) return conv_ks_launch(ctx, a, 1, true);
    if (plan.gen == 9 && (plan.ns == 8 || plan.ns == 9)) return conv_wr_launch(ctx, a, plan.bm / 256, plan.bn, plan.ns == 9, 2, true);
    if (plan.gen == 9) return conv_wr_launch(ctx, a, plan.bm / 256, plan.bn, plan.ns == 1, plan.ns == 4 ? 4 : 2);
    a.T = a.kh * a.kw;
    FID_REQUIRE(a.T >= 1 && a.T <= 25, "conv: %dx%d taps unsupported", a.kh, a.kw);
    FID_REQUIRE(a.Cin_p % 8 == 0 && a.Cout_p % 4 == 0, "conv: channel padding (Cin_p=%d Cout_p=%d)", a.Cin_p, a.Cout_p);
    FID_REQUIRE(a.Cin_p % plan.bk == 0, "conv: Cin_p=%d not a multiple of BK=%d", a.Cin_p, plan.bk);
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB; lower the batch");
    a.nchunk = a.Cin_p / plan.bk;
    a.ksteps = cdiv(a.T * a.Cin_p, plan.bk);
    if (a.in2) {                                                // fused shortcut: T2 more taps on the second tensor (generation 2 only)
        FID_REQUIRE(plan.gen == 2, "conv: the fused shortcut runs on generation 2 only (plan names %d)", plan.gen);
        FID_REQUIRE(a.T2 >= 1 && a.T + a.T2 <= 32 && a.kw2 >= 1 && a.s2 >= 1 && a.Cin2_p % plan.bk == 0 && a.in2_bytes <= OOB,
                    "conv: fused shortcut with %d taps of %d channels (BK = %d)", a.T2, a.Cin2_p, plan.bk);
        a.nchunk2 = a.Cin2_p / plan.bk;
        a.krow = a.T * a.Cin_p + a.T2 * a.Cin2_p;
        a.ksteps += a.T2 * a.nchunk2;
    }
    a.ksplit = plan.ksplit;
    a.ksteps_per_split = cdiv(a.ksteps, a.ksplit);
    a.ksplit = cdiv(a.ksteps, a.ksteps_per_split);
    a.tiles_m = cdiv(a.M, plan.bm);
    a.tiles_n = cdiv(a.Cout_p, plan.bn);
    FID_REQUIRE(a.ksplit == 1 || a.partial, "conv: split-K without a partial buffer");
    int rc = FID_E_INVALID;
    if (plan.gen == 6) {
        const int key = (plan.bm * 1000 + plan.bn) * 1000 + plan.bk * 10 + plan.ns;
        switch (key) {
            case 128128644: rc = launch_pcg<128, 128, 64, 4, 2, 2>(ctx, a); break;
            case 128064644: rc = launch_pcg<128, 64, 64, 4, 2, 2>(ctx, a); break;
            case 128064643: rc = launch_pcg<128, 64, 64, 3, 2, 2>(ctx, a); break;
            case 64064644: rc = launch_pcg<64, 64, 64, 4, 2, 2>(ctx, a); break;
            case 64064643: rc = launch_pcg<64, 64, 64, 3, 2, 2>(ctx, a); break;
            case 128128324: rc = launch_pcg<128, 128, 32, 4, 2, 2>(ctx, a); break;
            case 128096324: rc = launch_pcg<128, 96, 32, 4, 2, 2>(ctx, a); break;
            case 128064324: rc = launch_pcg<128, 64, 32, 4, 2, 2>(ctx, a); break;
            default: set_error("conv: no producer/consumer GEMM kernel for tile %dx%dx%d ns=%d", plan.bm, plan.bn, plan.bk, plan.ns); return FID_E_INVALID;
        }
    } else if (plan.gen == 2) {
        const int key = (plan.bm * 1000 + plan.bn) * 1000 + plan.bk * 10 + plan.ns;
        switch (key) {
            case 128128644: rc = launch_dma<128, 128, 64, 4, 2, 2>(ctx, a); break;
            case 128128643: rc = launch_dma<128, 128, 64, 3, 2, 2>(ctx, a); break;
            case 128128645: rc = launch_dma<128, 128, 64, 4, 2, 2, true>(ctx, a); break;
            case 128064645: rc = launch_dma<128, 64, 64, 4, 2, 2, true>(ctx, a); break;
            case 64064645: rc = launch_dma<64, 64, 64, 4, 2, 2, true>(ctx, a); break;
            case 128064644: rc = launch_dma<128, 64, 64, 4, 2, 2>(ctx, a); break;
            case 128064643: rc = launch_dma<128, 64, 64, 3, 2, 2>(ctx, a); break;
            case 64064644: rc = launch_dma<64, 64, 64, 4, 2, 2>(ctx, a); break;
            case 64064643: rc = launch_dma<64, 64, 64, 3, 2, 2>(ctx, a); break;
            case 128128324: rc = launch_dma<128, 128, 32, 4, 2, 2>(ctx, a); break;
            case 128096324: rc = launch_dma<128, 96, 32, 4, 2, 2>(ctx, a); break;
            case 128064324: rc = launch_dma<128, 64, 32, 4, 2, 2>(ctx, a); break;
            case 128032324: rc = launch_dma<128, 32, 32, 4, 4, 1>(ctx, a); break;
            default: set_error("conv: no DMA kernel for tile %dx%dx%d ns=%d", plan.bm, plan.bn, plan.bk, plan.ns); return FID_E_INVALID;
        }
    } else {
        const int key = plan.bm * 1000000 + plan.bn * 1000 + plan.bk;
        switch (key) {
            case 128128064: rc = launch_cfg<128, 128, 64, 2, 2>(ctx, a); break;
            case 128064064: rc = launch_cfg<128, 64, 64, 2, 2>(ctx, a); break;
            case 64064064: rc = launch_cfg<64, 64, 64, 2, 2>(ctx, a); break;
            case 128128032: rc = launch_cfg<128, 128, 32, 2, 2>(ctx, a); break;
            case 128096032: rc = launch_cfg<128, 96, 32, 2, 2>(ctx, a); break;
            case 128064032: rc = launch_cfg<128, 64, 32, 2, 2>(ctx, a); break;
            case 128032032: rc = launch_cfg<128, 32, 32, 4, 1>(ctx, a); break;
            default: set_error("conv: no kernel for tile %dx%dx%d", plan.bm, plan.bn, plan.bk); return FID_E_INVALID;
        }
    }
    FID_TRY(rc);
    if (a.ksplit > 1) {
        const long long n = (long long)a.M * (a.Cout_p / 4);
        hipLaunchKernelGGL(splitk_epilogue, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, ctx->stream, a);
    }
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
