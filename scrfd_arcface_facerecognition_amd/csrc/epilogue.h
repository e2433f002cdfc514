// Fused conv epilogue shared by conv.hip and conv_direct.hip.
// A lane holds, for each of its MI pixel columns and NI cout groups, 4 consecutive output channels of
// one pixel (rows of the MFMA D tile).  All global loads of the tile's epilogue (bias rows, residual
// values) are issued FIRST, back to back, and consumed afterwards: one memory round trip per tile
// instead of one per (pixel group, cout group).
#pragma once
#include "conv.h"

namespace fid {

typedef _Float16 ep_half4 __attribute__((ext_vector_type(4)));
typedef float ep_f32x4 __attribute__((ext_vector_type(4)));

struct EpiArgs {
    const float *bias;   // may be NULL (gallery GEMM)
    const float *slope;
    const void *res;
    void *out;
    int Cout_p, Ho, Wo;
    int act, flags, nsig;
    int res_H, res_W, res_Cp;
};

struct EpiPix {   // one output pixel column of the tile
    bool valid;
    long long m;  // linear pixel index n*Ho*Wo + oy*Wo + ox
    int n, oy, ox;
};

template <int NI, int MI>
struct EpiRegs {          // what the epilogue reads from memory ahead of time, for one tile
    ep_half4 rr[NI][MI];  // residual values
    ep_f32x4 bb[NI];      // bias row (border-class bias rows are fetched late: they differ per pixel)
    ep_f32x4 sl[NI];      // PReLU slopes
};

template <int NI, int MI>
__device__ __forceinline__ void epilogue_finish(const EpiArgs &e, ep_f32x4 (&acc)[NI][MI], const EpiPix (&px)[MI], const int (&co0)[NI],
                                                const EpiRegs<NI, MI> &R);

// issue every global load of the tile's epilogue (may be called BEFORE the K loop: the values arrive
// while the matrix cores work)
template <int NI, int MI>
__device__ __forceinline__ void epilogue_prefetch(const EpiArgs &e, const EpiPix (&px)[MI], const int (&co0)[NI], EpiRegs<NI, MI> &R) {
    const bool has_res = e.res != nullptr;
    const bool border = (e.flags & CF_BORDER) != 0;
    auto &rr = R.rr;
    auto &bb = R.bb;
    auto &sl = R.sl;
    // ---- phase 1: every load of the tile ----
    const bool has_bias = e.bias != nullptr;
#pragma unroll
    for (int ni = 0; ni < NI; ni++) bb[ni] = ep_f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_bias && !border) {
#pragma unroll
        for (int ni = 0; ni < NI; ni++) bb[ni] = *(const ep_f32x4 *)(e.bias + (co0[ni] < e.Cout_p ? co0[ni] : 0));
    }
    if (has_res) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            size_t roff = 0;
            if (px[mi].valid) {
                if (e.flags & CF_RES_UP2) roff = ((size_t)(px[mi].n * e.res_H + (px[mi].oy >> 1)) * e.res_W + (px[mi].ox >> 1)) * e.res_Cp;
                else roff = (size_t)px[mi].m * e.res_Cp;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ni++) {
                const int c = co0[ni] < e.Cout_p ? co0[ni] : 0;
                rr[ni][mi] = *(const ep_half4 *)((const _Float16 *)e.res + roff + c);
            }
        }
    }
    if (e.act == ACT_PRELU) {
#pragma unroll
        for (int ni = 0; ni < NI; ni++) sl[ni] = *(const ep_f32x4 *)(e.slope + (co0[ni] < e.Cout_p ? co0[ni] : 0));
    }
}

template <int NI, int MI>
__device__ __forceinline__ void epilogue_finish(const EpiArgs &e, ep_f32x4 (&acc)[NI][MI], const EpiPix (&px)[MI], const int (&co0)[NI],
                                                const EpiRegs<NI, MI> &R) {
    const bool has_res = e.res != nullptr;
    const auto &rr = R.rr;
    const auto &bb = R.bb;
    const auto &sl = R.sl;
    // ---- phase 2: arithmetic + stores ----
    const bool border = (e.flags & CF_BORDER) != 0;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        ep_f32x4 bm[NI];
        if (border) {   // exact fold of a BatchNorm in front of a zero-padded conv: the bias row depends on the pixel
            const int cls = (px[mi].oy == 0 ? 0 : (px[mi].oy == e.Ho - 1 ? 2 : 1)) * 3 + (px[mi].ox == 0 ? 0 : (px[mi].ox == e.Wo - 1 ? 2 : 1));
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bm[ni] = *(const ep_f32x4 *)(e.bias + (size_t)cls * e.Cout_p + (co0[ni] < e.Cout_p ? co0[ni] : 0));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            ep_f32x4 v = acc[ni][mi] + (border ? bm[ni] : bb[ni]);
            if (has_res) {
                v[0] += (float)rr[ni][mi][0]; v[1] += (float)rr[ni][mi][1];
                v[2] += (float)rr[ni][mi][2]; v[3] += (float)rr[ni][mi][3];
            }
            if (e.act == ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = fmaxf(v[i], 0.f);
            } else if (e.act == ACT_PRELU) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = v[i] > 0.f ? v[i] : v[i] * sl[ni][i];
            }
            if (e.nsig > 0) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (co0[ni] + i < e.nsig) v[i] = 1.f / (1.f + expf(-v[i]));
            }
            if (px[mi].valid && co0[ni] < e.Cout_p) {
                if (e.flags & CF_OUT_F32) {
                    *(ep_f32x4 *)((float *)e.out + (size_t)px[mi].m * e.Cout_p + co0[ni]) = v;
                } else {
                    ep_half4 h;
                    h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
                    *(ep_half4 *)((_Float16 *)e.out + (size_t)px[mi].m * e.Cout_p + co0[ni]) = h;
                }
            }
        }
    }
}

// ---- fp16 value path: bias + residual + activation in the accumulator layout, result as fp16 (no store), for
// kernels that transpose the tile through LDS and write whole 16-byte cout segments.
// Deciding activation / residual per VALUE on runtime flags makes hipcc emit scalar branches between every few
// vector instructions (measured on the chunked conv: 2.5 us per 256x96 tile, more than a third of its matrix
// time).  Here the flags select ONE of six fully unrolled bodies per tile.  ReLU is
// applied after the fp16 rounding with packed max (exactly the same result: rounding is monotone and keeps the
// sign), bias adds are 2-wide packed fp32.
template <int ACT, bool RES, int NI, int MI>
__device__ __forceinline__ void epilogue_values_body(const EpiArgs &e, ep_f32x4 (&acc)[NI][MI], const EpiPix (&px)[MI], const int (&co0)[NI],
                                                     const EpiRegs<NI, MI> &R, ep_half4 (&h)[NI][MI]) {
    const bool border = (e.flags & CF_BORDER) != 0;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        ep_f32x4 bv[NI];
        if (border) {
            const int cls = (px[mi].oy == 0 ? 0 : (px[mi].oy == e.Ho - 1 ? 2 : 1)) * 3 + (px[mi].ox == 0 ? 0 : (px[mi].ox == e.Wo - 1 ? 2 : 1));
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bv[ni] = *(const ep_f32x4 *)(e.bias + (size_t)cls * e.Cout_p + (co0[ni] < e.Cout_p ? co0[ni] : 0));
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ni++) bv[ni] = R.bb[ni];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            ep_f32x4 v = acc[ni][mi] + bv[ni];
            if (RES) v += __builtin_convertvector(R.rr[ni][mi], ep_f32x4);
            if (ACT == ACT_PRELU) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = v[i] > 0.f ? v[i] : v[i] * R.sl[ni][i];
            }
            ep_half4 r = __builtin_convertvector(v, ep_half4);
            if (ACT == ACT_RELU) r = __builtin_elementwise_max(r, ep_half4{0, 0, 0, 0});
            h[ni][mi] = r;
        }
    }
}

template <int NI, int MI>
__device__ __forceinline__ void epilogue_values_fast(const EpiArgs &e, ep_f32x4 (&acc)[NI][MI], const EpiPix (&px)[MI], const int (&co0)[NI],
                                                     const EpiRegs<NI, MI> &R, ep_half4 (&h)[NI][MI]) {
    const bool res = e.res != nullptr;
    if (e.act == ACT_RELU) {
        if (res) epilogue_values_body<ACT_RELU, true, NI, MI>(e, acc, px, co0, R, h);
        else epilogue_values_body<ACT_RELU, false, NI, MI>(e, acc, px, co0, R, h);
    } else if (e.act == ACT_PRELU) {
        if (res) epilogue_values_body<ACT_PRELU, true, NI, MI>(e, acc, px, co0, R, h);
        else epilogue_values_body<ACT_PRELU, false, NI, MI>(e, acc, px, co0, R, h);
    } else {
        if (res) epilogue_values_body<ACT_NONE, true, NI, MI>(e, acc, px, co0, R, h);
        else epilogue_values_body<ACT_NONE, false, NI, MI>(e, acc, px, co0, R, h);
    }
}

// loads + values in one call (kernels that do not prefetch)
template <int NI, int MI>
__device__ __forceinline__ void epilogue_values(const EpiArgs &e, ep_f32x4 (&acc)[NI][MI], const EpiPix (&px)[MI], const int (&co0)[NI],
                                                ep_half4 (&h)[NI][MI]) {
    EpiRegs<NI, MI> R;
    epilogue_prefetch<NI, MI>(e, px, co0, R);
    epilogue_values_fast<NI, MI>(e, acc, px, co0, R, h);
}

template <int NI, int MI>
__device__ __forceinline__ void epilogue_tile(const EpiArgs &e, ep_f32x4 (&acc)[NI][MI], const EpiPix (&px)[MI], const int (&co0)[NI]) {
    EpiRegs<NI, MI> R;
    epilogue_prefetch<NI, MI>(e, px, co0, R);
    epilogue_finish<NI, MI>(e, acc, px, co0, R);
}

}  // namespace fid
