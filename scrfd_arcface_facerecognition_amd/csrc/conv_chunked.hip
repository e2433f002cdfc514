// Channel-chunked direct 3x3 / stride-1 convolution for the WIDE layers (96 ... 512 channels): the same
// halo-patch idea as conv_direct.hip, for filter banks that do not fit in LDS.
//
// The implicit-GEMM kernels re-fetch every activation once per tap: a 128x128 tile streams 32 KB per 512
// MFMA-cycles = 64 B/clk/CU, more than L2->LDS delivers (~29 B/clk/CU measured), so those layers sit at <= 45 %
// of the matrix peak by construction.  Here a work item is one 16x16-pixel tile x CB output channels
// (CB = 64 or 96); the K axis is walked in chunks of 32 input channels, and per chunk the workgroup fetches
//     patch chunk   18x18 pixels x 32 ch          20.7 KB   (activations cross L2->LDS once, not 9 times)
//     weight chunk  9 taps x CB couts x 32 ch   36.9 / 55.3 KB
// by LDS-DMA into one of two buffers while the other one is multiplied: 9 taps x 4 x NI MFMAs per wave with no
// barrier inside (one barrier per chunk).  8 waves = 4 pixel groups x 2 cout groups, so both waves of a SIMD
// always have matrix work.  (item, chunk) pairs form one linear sequence per persistent workgroup: the first
// chunk of the next item is already in flight while the last chunk of the current one is multiplied and its
// tile is stored.
// Traffic per 16x16x96 item with 96 input channels: 228 KB for 42 MFLOP (184 flop/B; implicit GEMM: 64).
// Applicable to 3x3 / stride 1 / pad 1 with Cin_p % 32 == 0; the per-layer autotuner decides whether it wins
// (small maps waste tile area: 14x14 -> 77 %, 20x20 -> 39 %).
#include "epilogue.h"

namespace fid {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int TH = 16, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;   // 324 patch pixels
constexpr int CK = 32;                                               // input channels per chunk
constexpr int PATCH_BLKS = 24;                                       // 21 DMA blocks of 16 pixels x 64 B, padded to 3 per wave
constexpr int PATCH_BYTES = PATCH_BLKS * 1024;

__device__ __forceinline__ int swz64(int lin) { return (lin >> 1) & 3; }

struct ChunkArgs {
    const void *in;
    const void *w;
    const float *bias;
    const float *slope;
    const void *res;
    void *out;
    int H, W, B, Cin_p, Cout_p, w_rows;
    int act, flags, nsig;
    int res_Cp;
    int tiles_x, tiles_y, n_tiles, n_cblk, n_items, n_chunks;
    FastDiv d_cblk, d_tpi, d_tx;             // divisions by n_cblk, tiles per image, tiles_x
    unsigned in_bytes, w_bytes;
    int ablate;   // timing experiments only (FID_CHUNK_ABLATE: 1 = no MFMA/LDS reads, 2 = no DMA, 4 = no epilogue,
                  // 8 = no epilogue loads, 16 = no output stores, 32 = no flush)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt_c() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NI>   // couts per wave = NI*16; per workgroup CB = 2*NI*16
__global__ void __launch_bounds__(512, 2) conv3x3_chunked(const ChunkArgs a) {
    constexpr int CB = 2 * NI * 16, MI = 4;
    constexpr int W_BLKS = 9 * CB * 64 / 1024;          // weight-chunk DMA blocks (16 rows of 64 B each)
    constexpr int W_BYTES = W_BLKS * 1024;
    constexpr int MAX_W = (W_BLKS + 7) / 8;             // weight DMA instructions per wave per chunk
    constexpr int MAX_P = PATCH_BLKS / 8;               // patch DMA instructions per wave per chunk (exactly 3)
    // patch ring depth: patches (HBM / Infinity-Cache latency) are fetched TWO chunks ahead when LDS allows,
    // weights (L2-hot: every workgroup of a cout block reads the same ones) one chunk ahead
    constexpr int PD = (2 * W_BYTES + 3 * PATCH_BYTES <= 160 * 1024) ? 3 : 2;
    constexpr int AHEAD = PD - 1;
    static_assert(2 * W_BYTES + PD * PATCH_BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;           // cout group, pixel group
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)a.in, 0, a.in_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // LDS: [weights ring: 2 x W_BYTES][patch ring: PD x PATCH_BYTES]
    char *sWr = smem, *sPr = smem + 2 * W_BYTES;
    // ---- per-lane constants of my DMA blocks (block j = wave + 8*k of the weight / of the patch image) ----
    // weights: byte offset of my 16 B inside the cout block's [CB][9][Cin_p] rows; rows past the bank's end fall outside
    // the buffer descriptor and read as 0 (w_rows == Cout_p is a precondition of this kernel)
    int w_off[MAX_W];
#pragma unroll
    for (int k = 0; k < MAX_W; k++) {
        const int row = (wave + 8 * k) * 16 + (lane >> 2);
        const int t = row / CB, co = row - t * CB;
        w_off[k] = ((co * 9 + t) * a.Cin_p + ((lane & 3) ^ swz64(row)) * 8) * 2;
    }
    int p_y[MAX_P], p_x[MAX_P], p_c[MAX_P];
#pragma unroll
    for (int k = 0; k < MAX_P; k++) {
        const int row = (wave + 8 * k) * 16 + (lane >> 2);
        p_c[k] = ((lane & 3) ^ swz64(row)) * 8;
        p_y[k] = row / PW;
        p_x[k] = row - p_y[k] * PW;
        if (row >= NPIX) p_y[k] = -100000;                  // padding rows of the patch image: always read as 0
    }
    auto decode_item = [&](int item, int &n, int &ty, int &tx, int &cb) {
        const int tile = fastdiv(item, a.d_cblk);
        cb = item - tile * a.n_cblk;
        n = fastdiv(tile, a.d_tpi);
        const int r = tile - n * tiles_per_img;
        ty = fastdiv(r, a.d_tx); tx = r - ty * a.tiles_x;
    };
    // A prefetch stream walks the (item, chunk) sequence of this workgroup one chunk per step.  Its cursor keeps the
    // DECODED item and only re-decodes when the item changes: decoding both streams from scratch every step (six runtime
    // integer divisions) was stamped at ~1100 cycles per step, a third of the step's MFMA time, on all SIMDs at once.
    struct Cursor {
        int item, ck;      // work item / chunk the NEXT issue of this stream fetches
        int w_base;        // weights: byte offset of the item's cout block (chunk 0)
        int n, y0, x0;     // patch: image, top-left input pixel of the haloed patch
    };
    auto cursor_decode = [&](Cursor &c) {
        int n, ty, tx, cb;
        decode_item(c.item, n, ty, tx, cb);
        c.w_base = cb * CB * 9 * a.Cin_p * 2;
        c.n = n; c.y0 = ty * TH - 1; c.x0 = tx * TW - 1;
    };
    const int bid = xcd_major_id(blockIdx.x, gridDim.x);   // XCD-major item order: consecutive items in one L2 (halo overlap, cout blocks of a tile)
    auto cursor_init = [&](Cursor &c, int step) {          // position on step `step` of this workgroup (prologue only)
        const int li = step / a.n_chunks;
        c.ck = step - li * a.n_chunks;
        c.item = bid + li * gridDim.x;
        cursor_decode(c);
    };
    auto cursor_next = [&](Cursor &c) {
        if (++c.ck == a.n_chunks) {
            c.ck = 0;
            c.item += gridDim.x;
            cursor_decode(c);
        }
    };
    auto issue_weights = [&](const Cursor &c, int slot) {
        const int ubase = c.w_base + c.ck * CK * 2;          // wave-uniform part of the offset
        char *dst = sWr + slot * W_BYTES;
#pragma unroll
        for (int k = 0; k < MAX_W; k++) {
            const int j = wave + 8 * k;
            if (j < W_BLKS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16,
                                                         (unsigned)(w_off[k] + ubase), 0, 0, 0);
        }
    };
    auto issue_patch = [&](const Cursor &c, int slot) {   // exactly MAX_P instructions per wave (vmcnt accounting)
        const int c0 = c.ck * CK;
        char *dst = sPr + slot * PATCH_BYTES;
#pragma unroll
        for (int k = 0; k < MAX_P; k++) {
            const int j = wave + 8 * k;
            const int iy = c.y0 + p_y[k], ix = c.x0 + p_x[k];
            const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned vo = in ? (unsigned)((((c.n * a.H + iy) * a.W + ix) * a.Cin_p + c0 + p_c[k]) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, vo, 0, 0, 0);
        }
    };

    // my items: bid, + gridDim.x, ...; steps = (local item, chunk) linearised
    const int my_items = bid < a.n_items ? (a.n_items - 1 - bid) / gridDim.x + 1 : 0;
    const int n_steps = my_items * a.n_chunks;
    const int frow = lane & 15, fq = lane >> 4;
    const int lin0 = (wg * MI) * PW + frow;
    EpiArgs ep{a.bias, a.slope, a.res, a.out, a.Cout_p, a.H, a.W, a.act, a.flags, a.nsig, a.H, a.W, a.res_Cp};

    // ---- epilogue, split in three so that none of its memory latencies is exposed:
    //   epi_prefetch  before the LAST chunk's MFMAs: bias / slope / residual loads (they return during the matrix work)
    //   epi_values    after them: bias + residual + activation -> fp16 values kept in registers
    //   epi_flush     after the NEXT step's barrier: transposed through LDS blocks of the weight slot that step s used
    //                 (only blocks this wave refills itself, so no second barrier), written as whole 16-byte segments.
    //                 The stores are then the OLDEST entries of the wave's memory queue and have a whole step to retire.
    constexpr int OROWB = NI * 32, OCPP = NI * 2, PPI = 64 / OCPP;
    constexpr int OMASK = (OCPP & (OCPP - 1)) == 0 ? OCPP - 1 : 0;
    static_assert((64 * OROWB + 1023) / 1024 <= W_BLKS / 8, "staging must fit the wave's own DMA blocks");
    EpiPix px[MI];
    int co0[NI];
    EpiRegs<NI, MI> R;
    ep_half4 hv[NI][MI];
    int pend_n = -1, pend_ty = 0, pend_tx = 0, pend_co = 0;      // finished tile waiting for its flush
    auto epi_prefetch = [&](int item) {
        int n, ty, tx, cb;
        decode_item(item, n, ty, tx, cb);
        // opaque lane id: keeps hipcc from hoisting this block's per-lane address arithmetic out of the step loop (where it
        // would stay live across the MFMA section and spill)
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int frow = lo & 15, fq = lo >> 4;
        const int co_w = cb * CB + grp * NI * 16;            // first cout of this wave
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int oy = ty * TH + wg * MI + mi, ox = tx * TW + frow;
            px[mi].valid = oy < a.H && ox < a.W;
            px[mi].n = n; px[mi].oy = oy; px[mi].ox = ox;
            px[mi].m = px[mi].valid ? ((long long)n * a.H + oy) * a.W + ox : 0;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) co0[ni] = co_w + ni * 16 + fq * 4;
        if (!(a.ablate & 8)) epilogue_prefetch<NI, MI>(ep, px, co0, R);
        pend_ty = ty; pend_tx = tx; pend_co = co_w;
        return n;
    };
    auto stage_addr = [&](char *slot, int off) {              // wave-local staging offset -> LDS address in my own DMA blocks
        return slot + ((wave + 8 * (off >> 10)) << 10) + (off & 1023);
    };
    // (issuing the stores AFTER the step's DMAs, out of registers, measured 5 % slower than this order)
    auto epi_flush = [&](char *slot) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int frow = lo & 15, fq = lo >> 4, lane = lo;
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int p = mi * 16 + frow, c = ni * 2 + (fq >> 1);
                *(ep_half4 *)stage_addr(slot, p * OROWB + ((c ^ (p & OMASK)) << 4) + (fq & 1) * 8) = hv[ni][mi];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s2 = 0; s2 < (64 + PPI - 1) / PPI; s2++) {
            const int p = s2 * PPI + lane / OCPP, c = lane % OCPP;
            if (lane < PPI * OCPP && p < 64) {
                const u32x4 v = *(const u32x4 *)stage_addr(slot, p * OROWB + ((c ^ (p & OMASK)) << 4));
                const int oy = pend_ty * TH + wg * MI + (p >> 4), ox = pend_tx * TW + (p & 15);
                if (oy < a.H && ox < a.W && pend_co + c * 8 < a.Cout_p && !(a.ablate & 16))
                    *(u32x4 *)((char *)a.out + ((((size_t)pend_n * a.H + oy) * a.W + ox) * a.Cout_p + pend_co) * 2 + c * 16) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads are in registers before my DMAs refill the blocks
        pend_n = -1;
    };

    // prologue: weights of step 0, patches of steps 0 and 1 (issue order matters for the counted waits below:
    // per step the weights of s+1 are issued BEFORE the patch of s+2)
    Cursor cw, cp;                                        // next weight chunk / next patch chunk to fetch
    if (n_steps > 0) {
        cursor_init(cw, 0);
        cp = cw;
        issue_weights(cw, 0);
        issue_patch(cp, 0);
        cursor_next(cw);                                  // -> step 1
        cursor_next(cp);
    }
    if (AHEAD == 2 && n_steps > 1) {
        issue_patch(cp, 1);
        cursor_next(cp);                                  // -> step 2
    }
    f32x4 acc[NI][MI];
    int li = 0, ck = 0;                                   // local item index / chunk of the current step
    for (int s = 0; s < n_steps; s++) {
        const int item = bid + li * gridDim.x;
        // outstanding, oldest first: [W(s), P(s) | P(s+1)] at s = 0, else [stores | W(s) | P(s+1)] (P(s) was issued a step
        // earlier, before W(s)); everything up to W(s) must have landed, only the newest patch may stay in flight
        if (AHEAD == 2 && s + 1 < n_steps) wait_vmcnt_c<MAX_P>();
        else wait_vmcnt_c<0>();
        __syncthreads();                                    // everyone's landed; everyone is done with step s-1's buffers
        if (pend_n >= 0 && !(a.ablate & 32)) epi_flush(sWr + ((s + 1) & 1) * W_BYTES);   // step s-1's weight slot, before W(s+1) lands in it
        const bool last_chunk = ck == a.n_chunks - 1;
        int fin_n = -1;
        if (last_chunk && !(a.ablate & 4)) fin_n = epi_prefetch(item);
        if (s + 1 < n_steps) {
            if (!(a.ablate & 2)) issue_weights(cw, (s + 1) & 1);
            if (s + 2 < n_steps) cursor_next(cw);
        }
        if (s + AHEAD < n_steps) {
            if (!(a.ablate & 2)) issue_patch(cp, (s + AHEAD) % PD);
            if (s + AHEAD + 1 < n_steps) cursor_next(cp);
        }
        if (ck == 0) {
#pragma unroll
            for (int ni = 0; ni < NI; ni++)
#pragma unroll
                for (int mi = 0; mi < MI; mi++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char *sW = sWr + (s & 1) * W_BYTES, *sP = sPr + (s % PD) * PATCH_BYTES;
        // LDS-read-lean tap order.  The naive order (per tap: NI weight + 4 pixel fragments for 4*NI MFMAs) reads 63 KB per
        // wave and chunk -- 504 KB per CU against 128 B/clk of LDS bandwidth is 1.64 us, more than the 1.44 us the MFMAs
        // take.  A pixel fragment (patch row r, column shift dx) serves every output row mi = r - dy, so the loop walks
        // dx, then the 6 patch rows of this wave: 18 pixel + 27 weight fragment reads per chunk (45 KB, -29 %).
        // Weights of the column's three taps sit in registers (9 fragments); set dy is reloaded for the next column as
        // soon as its last use (row 3 + dy) has issued.  Pixel fragments are requested two row-steps ahead.
        if (!(a.ablate & 1)) {
            int plin = lin0, wlane = ((grp * NI) * 16 + frow) * 64 + ((fq ^ swz64(frow)) << 4);
            asm volatile("" : "+v"(plin), "+v"(wlane));   // opaque: recompute the fragment addresses per step instead of
                                                            // keeping 18 + 27 of them live across the whole loop
            half8 wq[3][NI], pq[3];
            auto load_w = [&](int dy, int dx) {            // weight rows (t*CB + ni*16 + frow): t*CB and ni*16 are multiples of 16,
#pragma unroll                                             // so the swizzle term only depends on frow
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
                __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of this row's MFMAs (hipcc sinks them otherwise)
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int mi = r - dy;
                    if (mi < 0 || mi >= MI) continue;
#pragma unroll
                    for (int ni = 0; ni < NI; ni++)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[dy][ni], pq[q % 3], acc[ni][mi], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dx < 2 && r >= 3) load_w(r - 3, dx + 1);   // taps (r-3, dx) are done: fetch that set for the next column
            }
        }
        if (fin_n >= 0) {
            epilogue_values_fast<NI, MI>(ep, acc, px, co0, R, hv);
            pend_n = fin_n;
        }
        if (++ck == a.n_chunks) { ck = 0; li++; }
    }
    if (pend_n >= 0) {
        __syncthreads();                                    // all waves are done reading the last step's weight slot
        epi_flush(sWr + ((n_steps - 1) & 1) * W_BYTES);
    }
}

template <int NI>
int launch_chunked(fid_ctx *ctx, const ChunkArgs &a) {
    constexpr int CB = 2 * NI * 16;
    constexpr size_t wb = (size_t)9 * CB * 64;
    constexpr size_t lds = 2 * wb + ((2 * wb + 3 * PATCH_BYTES <= 160 * 1024) ? 3 : 2) * PATCH_BYTES;
    FID_TRY(ensure_dyn_lds(ctx, (const void *)conv3x3_chunked<NI>, (int)((int)lds)));
    const int grid = std::min(a.n_items, ctx->num_cus);
    hipLaunchKernelGGL((conv3x3_chunked<NI>), dim3(grid), dim3(512), lds, ctx->stream, a);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace

bool conv_chunked_applicable(const ConvArgs &a) {
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Cin_p % 32 == 0 && a.Cin_p >= 64 && a.Cout_p >= 64 &&
           a.w_rows == a.Cout_p && a.H == a.Ho && a.W == a.Wo && a.H >= 12 && a.W >= 12 &&
           !(a.flags & (CF_RES_UP2 | CF_ARGMAX | CF_OUT_F32)) && a.nsig == 0 &&
           (a.res == nullptr || (a.res_H == a.Ho && a.res_W == a.Wo));
}

// cb: output channels per work item (64 or 96)
int conv_chunked_launch(fid_ctx *ctx, const ConvArgs &c, int cb) {
    ChunkArgs a{};
    a.in = c.in; a.w = c.w; a.bias = c.bias; a.slope = c.slope; a.res = c.res; a.out = c.out;
    a.H = c.H; a.W = c.W; a.B = c.M / (c.Ho * c.Wo); a.Cin_p = c.Cin_p; a.Cout_p = c.Cout_p; a.w_rows = c.w_rows;
    a.act = c.act; a.flags = c.flags; a.nsig = c.nsig; a.res_Cp = c.res_Cp;
    a.tiles_x = cdiv(c.W, TW); a.tiles_y = cdiv(c.H, TH);
    a.n_tiles = a.B * a.tiles_x * a.tiles_y;
    a.n_cblk = cdiv(c.Cout_p, cb);
    a.n_items = a.n_tiles * a.n_cblk;
    a.n_chunks = c.Cin_p / CK;
    a.d_cblk = fastdiv_make(a.n_cblk); a.d_tpi = fastdiv_make(a.tiles_x * a.tiles_y); a.d_tx = fastdiv_make(a.tiles_x);
    a.in_bytes = c.in_bytes;
    a.w_bytes = std::min<size_t>(c.w_bytes, (size_t)c.w_rows * 9 * c.Cin_p * 2);   // rows past the bank read as 0
    if (const char *e = getenv("FID_CHUNK_ABLATE")) a.ablate = atoi(e);
    FID_REQUIRE(a.in_bytes <= OOB && a.w_bytes <= OOB, "conv: tensor larger than 2 GiB");
    if (cb == 64) return launch_chunked<2>(ctx, a);
    if (cb == 96) return launch_chunked<3>(ctx, a);
    set_error("chunked conv: cb=%d unsupported", cb);
    return FID_E_INVALID;
}

}  // namespace fid
