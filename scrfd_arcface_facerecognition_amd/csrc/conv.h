// Implicit-GEMM convolution on gfx950 MFMA -- argument block shared by net.hip and match.hip.
#pragma once
#include "common.h"

namespace fid {

enum : int { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2 };
enum : int {
    CF_RES_UP2 = 1,   // residual is read at (oy/2, ox/2): fused nearest-2x upsample + add (PAFPN top-down)
    CF_BORDER = 2,    // bias table has 9 border classes (exact fold of a BatchNorm that precedes a zero-padded conv)
    CF_OUT_F32 = 4,   // store fp32 instead of fp16
    CF_ARGMAX = 8,    // no store: per output row, atomicMax of (value, column) -- the gallery match epilogue
};

struct ConvArgs {
    const void *in;     // fp16 [B, H, W, Cin_p]
    const void *w;      // fp16 [w_rows][T][Cin_p]   (T = kh*kw taps, tap-major then channel)
    const void *w_alt;  // the same weights in the packing plan_alt_kind(plan) names (repack.hip), or NULL
    const float *bias;  // fp32 [ncls][Cout_p]  (ncls = 9 with CF_BORDER else 1), may be NULL
    const float *slope; // fp32 [Cout_p] PReLU slopes (ACT_PRELU)
    const void *res;    // fp16 residual [B, res_H, res_W, res_Cp] or NULL
    void *out;          // fp16/fp32 [B, Ho, Wo, Cout_p]
    void *out2;         // fused shortcut + stride-2 conv (lower.py): second fp16 output [B, Ho, Wo, Cout_p] fed by weight / bias rows Cout_p .. w_rows-1 (no activation), or NULL
    float *partial;     // split-K slabs fp32 [ksplit][M][Cout_p]
    // fused shortcut (lower.py; generation 2 only): after the kh*kw taps the K axis continues with T2 taps of a SECOND tensor -- the block input x,
    // fp16 [B, H2, W2, Cin2_p] -- read at pixel (oy * s2 + t2 / kw2, ox * s2 + t2 % kw2): the 1x1 / stride-2 (T2 = 1) or average-pool + 1x1
    // (T2 = 4: a 2x2 / stride-2 kernel) shortcut conv of a residual block as extra K-steps of the conv that adds it.  Weight rows are
    // [kh*kw * Cin_p | T2 * Cin2_p] halfs (krow).  in2 = NULL: a plain conv.
    const void *in2;
    unsigned in2_bytes;
    int H2, W2, Cin2_p, T2, kw2, s2, nchunk2, krow;
    unsigned long long *amax;  // CF_ARGMAX: packed (sortable(value) << 32 | ~(amax_col0 + column)) per row
    int amax_col0;             // CF_ARGMAX: global index of column 0 (gallery shards)
    int H, W, Cin_p, Ho, Wo, Cout_p, w_rows;
    int kh, kw, stride, pad;
    int M, T, nchunk, ksteps, ksplit, ksteps_per_split;
    int act, flags, nsig;
    int res_H, res_W, res_Cp;
    int tiles_m, tiles_n;
    int rev;       // walk the work items from the LAST to the first: the executor alternates the direction from layer to layer, so a layer starts on
                   // the part of its input the producer wrote last -- still in the 256 MB Infinity Cache when the tensors (210 MB at 160x160x64
                   // x 64 frames) are too large to survive a whole layer (kernels that do not implement it ignore the flag)
    int tm_fast;   // tile order: 0 = the cout tiles of a pixel tile are consecutive (convs: few cout tiles, weights stay in L2), 1 = the pixel tiles of a cout tile are (gallery match: few query tiles, each gallery tile is fetched from HBM once)
    unsigned in_bytes, w_bytes;
};

// fills the derived fields (T, nchunk, ksteps, tiles, split-K plan) and launches; `partial_ws`
// must hold ksplit*M*Cout_p floats when the plan splits K (query with conv_plan first).
struct ConvPlan {
    int bm, bn, bk, ksplit;
    int gen;   // 0 = conv_direct, 1 = register-staged double buffer, 2 = LDS-DMA ring, 3 = conv_chunked, 4 = conv_pp, 5 = conv_pc (bn = couts per work item), 6 = LDS-DMA ring with producer waves, 7 = conv_pcr, 8 = conv_pc2, 9 = conv_wr, 10 = conv_s2, 11 = conv_gw
    int ns;    // ring slots (gen 2); 5 = 4 slots + fragment prefetch across K-steps
    size_t partial_bytes;
};
ConvPlan conv_plan(const ConvArgs &a, int num_cus, bool allow_split);
// every kernel/tile/split combination worth timing for this conv (gen 0 = conv_direct); used by the
// executor's per-layer autotuner (net.hip) -- "measure, don't guess"
#include <vector>
std::vector<ConvPlan> conv_candidates(const ConvArgs &a, int num_cus, bool allow_split);
int conv_launch(fid_ctx *ctx, ConvArgs a, const ConvPlan &plan);
float conv_plan_cu_share(const ConvArgs &a, const ConvPlan &plan, int num_cus);      // fraction of the CUs the launch occupies (1: all / not modelled)
// does the kernel `plan` names honour ConvArgs::rev?
inline bool conv_walks_reverse(const ConvPlan &plan) { return plan.gen == 9; }
// alternate weight packing a plan's kernel wants in ConvArgs::w_alt (0 = none); repack.hip builds it
int plan_alt_kind(const ConvPlan &plan);
size_t repack_bytes(int kind, int Cout_p, int Cin_p, int taps = 9);
int repack_weights(fid_ctx *ctx, int kind, const void *src, void *dst, int Cout_p, int Cin_p, int taps = 9);

// conv_direct.hip: 3x3/s1 conv with LDS-resident weights + haloed patches (<= 64 channels in and out)
bool conv_direct_applicable(const ConvArgs &a);
int conv_direct_launch(fid_ctx *ctx, const ConvArgs &a);

// conv_chunked.hip: 3x3/s1 conv, haloed patch + weight chunks of 32 input channels (wide layers); cb = 64 | 96
bool conv_chunked_applicable(const ConvArgs &a);
int conv_chunked_launch(fid_ctx *ctx, const ConvArgs &a, int cb);

// conv_pp.hip: 3x3/s1 conv, two tiles per item in ping-pong roles sharing the weight chunks; cb = 32 | 48 | 64
bool conv_pp_applicable(const ConvArgs &a);
int conv_pp_launch(fid_ctx *ctx, const ConvArgs &a, int cb);

// conv_pc.hip: conv_chunked's data flow with a dedicated producer wave (DMA issue + prefetch decode) beside 8 MFMA waves; cb = 64 | 96
bool conv_pc_applicable(const ConvArgs &a);
int conv_pc_launch(fid_ctx *ctx, const ConvArgs &a, int cb, int ring);

// conv_pcr.hip: 64 -> 64 channels, weights resident in LDS, 8 MFMA waves on one tile + 4 producer waves
// two tiles per fetched weight chunk (generation 8, conv_pc2.hip): 64-cout blocks, Cout_p a multiple of 64
bool conv_pc2_applicable(const ConvArgs &a);
int conv_pc2_launch(fid_ctx *ctx, const ConvArgs &a);
bool conv_pcr_applicable(const ConvArgs &a);
int conv_pcr_launch(fid_ctx *ctx, const ConvArgs &a);

// conv_wr.hip (generation 9): weights in registers, waves split by cout, a pair of tiles x 128 couts per item; needs w_alt (kind 2)
bool conv_wr_applicable(const ConvArgs &a);
bool conv_wr_resident_ok(const ConvArgs &a);
int conv_wr_launch(fid_ctx *ctx, const ConvArgs &a, int nt, int cb, int resident, int ring, bool strip = false);   // strip: STRIP tiles (ns = 8 streaming, 9 resident)

// mbf_block.hip: MobileFaceNet's bottleneck (1x1 -> depthwise 3x3 / stride 1 | 2 -> 1x1 [+ input]) in one launch, the expanded maps in LDS
bool mbf_block_applicable(int H, int W, int Cin_p, int Gp, int Cout_p, int stride, bool res);
int mbf_block_launch(fid_ctx *ctx, const void *x, const void *w1, const float *b1, const float *s1, int act1, const float *dww, const float *dwb,
                     const float *dws, int dw_act, const void *w2, const float *b2, const float *s2, int act2, bool res, void *out, int B, int H, int W,
                     int Cin_p, int Gp, int Cout_p, int stride);
// conv_ks.hip (generation 9, ns = 6): conv3x3_wr's one-tile x 64-cout item with the K axis split over two wave groups (few tiles: one item per CU); needs w_alt (kind 2)
bool conv_ks_applicable(const ConvArgs &a);
bool conv_ks_mosaic(const ConvArgs &a);      // 7x7 maps: four images share a 16x16 tile
int conv_ks_launch(fid_ctx *ctx, const ConvArgs &a, int per_wg = 1, bool strip = false);   // per_wg = 2: two items per workgroup (plan tile 512)
int conv_ks_items(const ConvArgs &a, bool strip = false);               // work items (tiles x 64-cout blocks) of the launch
// STRIP tiles (round 5, generation 9 with ns = 7 / 8 / 9): x-packed pixel fragments -- a tile is TH rows x 16 consecutive columns of the strip
// formed by the rows of all images side by side (8 images x 14 columns = 7 exact fragments); conv_ks.hip explains the addressing
bool conv_strip_ok(const ConvArgs &a);       // every 16-column strip tile crosses at most one image boundary
int conv_strip_rows(const ConvArgs &a);      // tile rows of a STRIP launch (10 | 14 | 16)
bool conv_ks_strip_applicable(const ConvArgs &a);

// conv_s2.hip (generation 10): 3x3 / stride 2 with parity-plane patches and resident weights (64 / 96 input channels); needs w_alt (kind 2)
bool conv_s2_applicable(const ConvArgs &a);
int conv_s2_launch(fid_ctx *ctx, const ConvArgs &a);

// conv_gw.hip (generation 11): implicit GEMM (any 1x1 / 2x2 / 3x3, any stride) with the weights in registers, only the pixel operand
// through LDS; bm = 64 | 128 output pixels x bn = 128 | 256 couts per item; needs w_alt (kind 3)
bool conv_gw_applicable(const ConvArgs &a);
int conv_gw_launch(fid_ctx *ctx, const ConvArgs &a, int bm, int bn);

// conv_bb.hip: a residual BasicBlock on 64 channels (conv3x3 + ReLU, conv3x3, + input, activation) in one launch; w1 / w2 are repack
// kind 2 images packed by lower.py
int conv_bb_launch(fid_ctx *ctx, const void *in, const void *w1, const float *b1, int ncls1, int act1, const float *s1, const void *w2, const float *b2,
                   void *out, int B, int H, int W, int act2, int rev, int Cp = 64);

// dwpw.hip: depthwise 3x3 (stride 1 | 2, pad 1) + the pointwise 1x1 conv that consumes it, one launch (MobileFaceNet bottlenecks, SCRFD-500M)
bool dwpw_applicable(int Gp, int Cout_p);
int dwpw_launch(fid_ctx *ctx, const void *in, const float *dw_w, const float *dw_b, const float *dw_s, int dw_act, const void *pw_w, const float *pw_b,
                const float *pw_s, int pw_act, const void *res, void *out, int B, int H, int W, int Ho, int Wo, int Gp, int Cout_p, int stride);

// stem_block.hip: IResNet's first two convs (u8 crop -> conv3x3 3 -> 64 + PReLU -> [BN] conv3x3 64 -> 64 + PReLU) in one launch; the first conv's
// result leaves only at the even pixels (compact second output for the block's stride-2 shortcut)
bool stem_block_applicable(int H, int W);
int stem_block_launch(fid_ctx *ctx, const uint8_t *img, int B, int H, int W, const float *w0, const float *b0, const float *s0, int act0, const void *w1,
                      const float *b1, int ncls1, const float *s1, int act1, void *out, void *xe);

// lat_fpn.hip: a PAFPN level's 1x1 lateral (+ upsampled coarser lateral) and the 3x3 conv on it in one launch; the lateral leaves the CU only when asked for
bool lat_fpn_applicable(int Cin_p, int H, int W, int rH, int rW, bool has_res);
int lat_fpn_launch(fid_ctx *ctx, const void *in, int B, int H, int W, int Cin_p, const void *w0, const float *b0, const void *res, int rH, int rW, const void *w1,
                   const float *b1, void *out, void *lat);
// match_gemm.hip: the gallery scan on 256 x 256 tiles (query batches > 128 rows against galleries of >= one tile per CU)
bool match_scan256_applicable(int n, int Gp, int dim, int num_cus);
int match_scan256_launch(fid_ctx *ctx, const void *q, const void *g, int n, int Gp, int dim, int col0, unsigned long long *amax);

// stem_fused.hip: u8 frame -> conv/s2 -> conv -> conv -> maxpool/s2 in one kernel
int stem_fused_launch(fid_ctx *ctx, const uint8_t *img, int B, int H, int W, const void *w0, const float *b0, const void *w1,
                      const float *b1, const void *w2, const float *b2, void *out, int C2p);

// stem_rows.hip: the same fused stem on row-structured tiles (6 x PY pooled pixels, weights in registers, pooling in registers); the default
int stem_rows_launch(fid_ctx *ctx, const uint8_t *img, int B, int H, int W, const void *w0, const float *b0, const void *w1,
                     const float *b1, const void *w2, const float *b2, void *out, int C2p);

}  // namespace fid
