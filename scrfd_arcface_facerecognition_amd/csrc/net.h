// Layer-table format shared with scrfd_arcface_facerecognition_amd/lower.py (keep in sync).
#pragma once
#include "conv.h"

namespace fid {

enum : int { OP_STEM = 1, OP_CONV = 2, OP_MAXPOOL = 3, OP_DWCONV = 4, OP_STEMFUSED = 5, OP_BBLOCK = 6, OP_DWPW = 7, OP_MBBLOCK = 8, OP_STEMBLOCK = 9, OP_LATFPN = 10 };

// int32 word indices inside one op record (FID_OP_WORDS = 32 words)
enum : int {
    W_TYPE = 0,
    W_SRC = 1,    // tensor id, -1 = the uint8 BGR input images
    W_DST = 2,
    W_RES = 3,    // residual tensor id or -1
    W_KH = 4,
    W_KW = 5,
    W_STRIDE = 6,
    W_PAD = 7,
    W_CIN = 8,    // logical channels (cost accounting)
    W_COUT = 9,
    W_ACT = 10,
    W_FLAGS = 11,  // CF_* bits
    W_NSIG = 12,
    W_WOFF = 13,   // byte offset of the packed weights inside the blob
    W_WBYTES = 14,
    W_BOFF = 15,   // bias table offset or -1
    W_SOFF = 16,   // PReLU slope table offset or -1
    W_WROWS = 17,  // rows (output channels) present in the packed weights
    W_GROUPS = 18, // 1, or cin for depthwise
    // OP_STEMFUSED (stem_fused.hip): blob offsets of the three convs' packed weights / biases
    W_F_W0 = 20, W_F_B0 = 21, W_F_W1 = 22, W_F_B1 = 23, W_F_W2 = 24, W_F_B2 = 25,
    W_F_MACS_LO = 26, W_F_MACS_HI = 27,   // algorithmic MACs per image of the fused group (cost accounting)
    // OP_BBLOCK (conv_bb.hip; lower.py): blob offsets of the two convs' weights in repack kind 2 order and of their biases; W_ACT = the activation
    // after the residual add; W_F_MACS_* = MACs per image of both convs
    W_B_W1 = 20, W_B_B1 = 21, W_B_W2 = 22, W_B_B2 = 23, W_B_S1 = 24, W_B_ACT1 = 25,   // (+ conv1's PReLU slopes or -1, its activation; W_FLAGS & CF_BORDER: 9 bias rows)
    // OP_DWPW (dwpw.hip; lower.py): the record is the POINTWISE conv's (weights, bias, slopes, activation, residual, W_DST) with W_SRC = the
    // depthwise layer's input and W_STRIDE = its stride; the depthwise layer's fp32 tables and activation: W_F_MACS_* = MACs of both
    W_D_WOFF = 20, W_D_BOFF = 21, W_D_SOFF = 22, W_D_ACT = 23,
    // OP_MBBLOCK (mbf_block.hip; lower.py): 1x1 -> depthwise 3x3 -> 1x1 [+ input].  The record is the SECOND pointwise conv's (weights [Cout_p][Gp],
    // bias, slopes, activation, W_DST; W_RES = W_SRC when the block adds its input) with W_SRC = the block input and W_STRIDE = the depthwise
    // stride; the first pointwise conv's fp16 weights [Gp][Cin_p] / fp32 bias / slopes / activation, the depthwise fp32 tables [9][Gp] / bias /
    // slopes / activation and the padded expanded width Gp; W_F_MACS_* = the MACs of the three layers
    W_M_W1 = 20, W_M_B1 = 21, W_M_S1 = 22, W_M_ACT1 = 23, W_M_DW = 24, W_M_DWB = 25, W_M_DWS = 28, W_M_DWACT = 29, W_M_GP = 30,
    // OP_STEMBLOCK (stem_block.hip; lower.py): the recogniser's first conv (uint8 frame -> 64 channels; W_WOFF / W_BOFF / W_SOFF = its fp32 [64][27]
    // weights, bias, PReLU slopes or -1) + the 3x3 / stride-1 conv on 64 channels that consumes it, in one launch: the second conv's repack-kind-2
    // image, bias rows (W_FLAGS & CF_BORDER: 9 border classes), PReLU slopes or -1, the FIRST conv's activation, and the tensor id + 1 of the
    // compact second output (the first conv's result at the even pixels, for the block's stride-2 shortcut; 0: none).  W_ACT = the second conv's
    // activation, W_DST its output, W_F_MACS_* = the MACs of both
    W_S_W1 = 20, W_S_B1 = 21, W_S_S1 = 22, W_S_ACT0 = 23, W_S_DST2 = 24,
    // OP_LATFPN (lat_fpn.hip; lower.py): a PAFPN level's 1x1 lateral conv (+ the nearest-2x upsampled coarser lateral: W_RES, or -1) and the 3x3 conv
    // that consumes it, in one launch.  The record is the 3x3 conv's (W_DST, bias W_BOFF; W_WOFF / W_WBYTES = ITS repack-kind-2 image); the lateral's plain
    // fp16 [64][Cin_p] weights, its bias, and the tensor id + 1 of the lateral itself when a finer level adds it (0: it never leaves the CU);
    // W_F_MACS_* = the MACs of both
    W_L_W0 = 20, W_L_B0 = 21, W_L_LAT = 22,
    // OP_CONV fused with the block's shortcut (lower.py; conv_s2.hip DUAL): second output's tensor id + 1 (0: plain conv), its activation,
    // padded couts of the first output; W_F_MACS_LO then holds the shortcut's MACs per image
    W_X_DST2 = 20, W_X_ACT2 = 21, W_X_COUT1P = 22,
    // OP_CONV fused with the block's shortcut CONV as extra K-steps on a second input tensor (lower.py; conv.hip generation 2): tensor id + 1 of
    // the block input (0: none), its taps (1: a 1x1 conv; 4: average pool + 1x1 = a 2x2 kernel), their row length, the sampling stride;
    W_X_SRC2 = 23, W_X_T2 = 24, W_X_KW2 = 25, W_X_S2 = 28,
    // ... both forms are in the table: the shortcut conv stays an op of its own (its word 29 = index + 1 of the conv that can absorb it) and the
    // absorbing conv carries a SECOND weight image with rows [kh*kw * Cin_p | taps * Cin2_p] (word 29), the summed bias row (word 30) and the
    // shortcut's op index + 1 (word 31).  The autotuner decides per batch size: a generation-12 pick = generation 2 on the second image with the
    // shortcut as extra K-steps (the shortcut op is then skipped), any other pick = the plain conv + the shortcut op.
    W_X_W2OFF = 29, W_X_B2OFF = 30, W_X_SCOP = 31,
};

// int32 word indices of one tensor record (FID_TENSOR_WORDS = 8 words)
enum : int { T_C = 0, T_CP = 1, T_H = 2, T_W = 3, T_DTYPE = 4, T_SLOT = 5, T_FLAGS = 6 };

}  // namespace fid
