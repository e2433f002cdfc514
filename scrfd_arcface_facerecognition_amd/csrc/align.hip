// 5-point alignment: Umeyama similarity estimate + fixed-point bilinear affine warp to 112x112.
// Replaces skimage SimilarityTransform.estimate and cv2.warpAffine as called by reference
// utils/helpers.py:18-59 (estimate_norm, norm_crop_image), reached from models/arcface.py:54-57.
//
//  * the estimate is the closed form of the 2-D Umeyama solution: with
//    A = dst_d^T src_d / n,  a = A00 + A11,  b = A10 - A01:  scale*R = [[a,-b],[b,a]] / var(src).
//    (The SVD form in skimage reduces to this for every rank >= 1 case, reflected inputs included,
//    because a reflection component has zero trace against any rotation.)  Evaluated in fp64 from
//    the fp32 landmarks; skimage itself runs the SVD in fp32, so agreement is ~1e-6 relative.
//  * the warp follows OpenCV's u8 INTER_LINEAR / BORDER_CONSTANT fixed-point path:
//    1/1024 coordinate grid, 1/32-pixel interpolation table, round-half-even cvRound.
//
// HBM-bound byte work: 37.6 KB written per face, <= 4 x 37.6 KB gathered from the frame (mostly L2).
#include "common.h"

namespace {

__constant__ double c_template[10] = {38.2946, 51.6963, 73.5318, 51.5014, 56.0252, 71.7366, 41.5493, 92.3655, 70.7299, 92.2041};

__device__ __forceinline__ long long cv_round(double v) {
    // cvRound == lrint (round half to even); saturate_cast<int> semantics for out-of-range / NaN
    if (!(v > -2147483648.0)) return -2147483648ll;
    if (!(v < 2147483647.0)) return 2147483647ll;
    return __double2ll_rn(v);
}

constexpr int OUT = 112;
constexpr int ROWS_PER_BLOCK = 16;

__global__ void __launch_bounds__(256) align_warp(const uint8_t *frames, int H, int W, const float *kps, const int *counts,
                                                  int cap, int F, uint8_t *crops, double *M_out) {
    const int slot = blockIdx.y;  // b * F + f
    const int b = slot / F, f = slot - b * F;
    const int y0 = blockIdx.x * ROWS_PER_BLOCK;
    uint8_t *dst = crops + (size_t)slot * OUT * OUT * 3;
    const bool valid = f < counts[b] && f < cap;
    __shared__ double sm[6];  // inverse map m00 m01 m02 m10 m11 m12
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 0;
        if (valid) {
            const float *lm = kps + ((size_t)b * cap + f) * 10;
            // c_template holds the float32 template values widened to double, like skimage sees them
            double sx[5], sy[5], dx[5], dy[5], msx = 0, msy = 0, mdx = 0, mdy = 0;
            for (int i = 0; i < 5; i++) {
                sx[i] = (double)lm[2 * i]; sy[i] = (double)lm[2 * i + 1];
                dx[i] = (double)(float)c_template[2 * i]; dy[i] = (double)(float)c_template[2 * i + 1];
                msx += sx[i]; msy += sy[i]; mdx += dx[i]; mdy += dy[i];
            }
            msx /= 5; msy /= 5; mdx /= 5; mdy /= 5;
            double a = 0, bb = 0, var = 0;
            for (int i = 0; i < 5; i++) {
                const double px = sx[i] - msx, py = sy[i] - msy, qx = dx[i] - mdx, qy = dy[i] - mdy;
                a += qx * px + qy * py;   // n * (A00 + A11)
                bb += qy * px - qx * py;  // n * (A10 - A01)
                var += px * px + py * py; // n * var
            }
            const double M00 = a / var, M01 = -bb / var, M10 = bb / var, M11 = a / var;
            const double M02 = mdx - (M00 * msx + M01 * msy), M12 = mdy - (M10 * msx + M11 * msy);
            if (M_out) {
                double *mo = M_out + (size_t)slot * 6;
                mo[0] = M00; mo[1] = M01; mo[2] = M02; mo[3] = M10; mo[4] = M11; mo[5] = M12;
            }
            // cv2.warpAffine inverts M (no WARP_INVERSE_MAP flag)
            double D = M00 * M11 - M01 * M10;
            D = D != 0 ? 1.0 / D : 0.0;
            const double A11 = M11 * D, A22 = M00 * D;
            const double m00 = A11, m01 = -M01 * D, m10 = -M10 * D, m11 = A22;
            sm[0] = m00; sm[1] = m01; sm[2] = -m00 * M02 - m01 * M12;
            sm[3] = m10; sm[4] = m11; sm[5] = -m10 * M02 - m11 * M12;
            ok = isfinite(M00) && isfinite(M01) && isfinite(M02) && isfinite(M12);
        } else if (M_out && y0 == 0) {
            double *mo = M_out + (size_t)slot * 6;
            for (int i = 0; i < 6; i++) mo[i] = 0.0;
        }
    }
    __syncthreads();
    const uint8_t *src = frames + (size_t)b * H * W * 3;
    for (int p = threadIdx.x; p < ROWS_PER_BLOCK * OUT; p += blockDim.x) {
        const int y = y0 + p / OUT, x = p % OUT;
        uint8_t *o = dst + ((size_t)y * OUT + x) * 3;
        if (!ok) { o[0] = o[1] = o[2] = 0; continue; }
        const long long X0 = cv_round((sm[1] * y + sm[2]) * 1024.0) + 16;
        const long long Y0 = cv_round((sm[4] * y + sm[5]) * 1024.0) + 16;
        const long long ad = cv_round(sm[0] * x * 1024.0), bd = cv_round(sm[3] * x * 1024.0);
        const long long X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
        long long sxl = X >> 5, syl = Y >> 5;
        sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);  // saturate_cast<short>
        syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
        const int ix = (int)sxl, iy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
        const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
        const bool x0in = ix >= 0 && ix < W, x1in = ix + 1 >= 0 && ix + 1 < W;
        const bool y0in = iy >= 0 && iy < H, y1in = iy + 1 >= 0 && iy + 1 < H;
        int acc[3] = {512, 512, 512};
        if (y0in && x0in) { const uint8_t *q = src + ((size_t)iy * W + ix) * 3; acc[0] += w00 * q[0]; acc[1] += w00 * q[1]; acc[2] += w00 * q[2]; }
        if (y0in && x1in) { const uint8_t *q = src + ((size_t)iy * W + ix + 1) * 3; acc[0] += w01 * q[0]; acc[1] += w01 * q[1]; acc[2] += w01 * q[2]; }
        if (y1in && x0in) { const uint8_t *q = src + ((size_t)(iy + 1) * W + ix) * 3; acc[0] += w10 * q[0]; acc[1] += w10 * q[1]; acc[2] += w10 * q[2]; }
        if (y1in && x1in) { const uint8_t *q = src + ((size_t)(iy + 1) * W + ix + 1) * 3; acc[0] += w11 * q[0]; acc[1] += w11 * q[1]; acc[2] += w11 * q[2]; }
        o[0] = (uint8_t)(acc[0] >> 10); o[1] = (uint8_t)(acc[1] >> 10); o[2] = (uint8_t)(acc[2] >> 10);
    }
}

// cv2.resize INTER_LINEAR u8 + zero letterbox paste (scrfd.py:135-138).
__global__ void __launch_bounds__(256) letterbox_kernel(const uint8_t *frames, int H, int W, uint8_t *out, int in_h, int in_w,
                                                        int new_h, int new_w, double scale_x, double scale_y, int area2x) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= in_w) return;
    uint8_t *o = out + (((size_t)b * in_h + y) * in_w + x) * 3;
    if (x >= new_w || y >= new_h) { o[0] = o[1] = o[2] = 0; return; }
    const uint8_t *src = frames + (size_t)b * H * W * 3;
    if (new_w == W && new_h == H) {
        const uint8_t *q = src + ((size_t)y * W + x) * 3;
        o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
        return;
    }
    if (area2x) {  // exact 2x decimation is INTER_AREA inside cv2.resize
        for (int c = 0; c < 3; c++) {
            const int s = src[((size_t)(2 * y) * W + 2 * x) * 3 + c] + src[((size_t)(2 * y) * W + 2 * x + 1) * 3 + c] +
                          src[((size_t)(2 * y + 1) * W + 2 * x) * 3 + c] + src[((size_t)(2 * y + 1) * W + 2 * x + 1) * 3 + c];
            o[c] = (uint8_t)((s + 2) >> 2);
        }
        return;
    }
    auto coeff = [](int d, double scale, int n, int &s0, int &s1, int &a0, int &a1) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= n - 1) { s = n - 1; f = 0.f; }
        s0 = s; s1 = min(s + 1, n - 1);
        a1 = (int)rintf(__fmul_rn(f, 2048.f));
        a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    };
    int sx0, sx1, a0, a1, sy0, sy1, b0, b1;
    coeff(x, scale_x, W, sx0, sx1, a0, a1);
    coeff(y, scale_y, H, sy0, sy1, b0, b1);
    for (int c = 0; c < 3; c++) {
        const int T0 = src[((size_t)sy0 * W + sx0) * 3 + c] * a0 + src[((size_t)sy0 * W + sx1) * 3 + c] * a1;
        const int T1 = src[((size_t)sy1 * W + sx0) * 3 + c] * a0 + src[((size_t)sy1 * W + sx1) * 3 + c] * a1;
        int v = (((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// The same arithmetic, FOUR consecutive output pixels per thread (in_w % 4 == 0): a source pixel is one unaligned 4-byte load instead of three
// 1-byte loads and the 12 output bytes leave as three aligned dwords -- 16 + 3 memory instructions per 4 pixels instead of 48 + 12 (32 frames of
// 1080p -> 640x640: cfg 5 step 1.548 -> 1.535 ms).  `src_bytes` = bytes of the whole frame batch: the last pixel of the batch is read bytewise.
__global__ void __launch_bounds__(256) letterbox_kernel4(const uint8_t *frames, int H, int W, uint8_t *out, int in_h, int in_w, int new_h, int new_w,
                                                         double scale_x, double scale_y, size_t src_bytes) {
    const int b = blockIdx.z;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= in_w) return;
    unsigned *o = (unsigned *)(out + (((size_t)b * in_h + y) * in_w + x4) * 3);
    unsigned char px[12];
    const size_t fbase = (size_t)b * H * W * 3;
    auto fetch = [&](size_t off) -> unsigned {                  // the 3 bytes of a source pixel (byte 3 of the word is ignored)
        if (off + 4 <= src_bytes) {
            unsigned v;
            __builtin_memcpy(&v, frames + off, 4);
            return v;
        }
        return (unsigned)frames[off] | ((unsigned)frames[off + 1] << 8) | ((unsigned)frames[off + 2] << 16);
    };
    auto coeff = [](int d, double scale, int n, int &s0, int &s1, int &a0, int &a1) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= n - 1) { s = n - 1; f = 0.f; }
        s0 = s; s1 = min(s + 1, n - 1);
        a1 = (int)rintf(__fmul_rn(f, 2048.f));
        a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    };
    int sy0 = 0, sy1 = 0, b0 = 0, b1 = 0;
    if (y < new_h) coeff(y, scale_y, H, sy0, sy1, b0, b1);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x4 + i;
        if (x >= new_w || y >= new_h) { px[3 * i] = px[3 * i + 1] = px[3 * i + 2] = 0; continue; }
        int sx0, sx1, a0, a1;
        coeff(x, scale_x, W, sx0, sx1, a0, a1);
        const unsigned p00 = fetch(fbase + ((size_t)sy0 * W + sx0) * 3), p01 = fetch(fbase + ((size_t)sy0 * W + sx1) * 3);
        const unsigned p10 = fetch(fbase + ((size_t)sy1 * W + sx0) * 3), p11 = fetch(fbase + ((size_t)sy1 * W + sx1) * 3);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int T0 = (int)((p00 >> (8 * c)) & 255u) * a0 + (int)((p01 >> (8 * c)) & 255u) * a1;
            const int T1 = (int)((p10 >> (8 * c)) & 255u) * a0 + (int)((p11 >> (8 * c)) & 255u) * a1;
            const int v = (((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2) >> 2;
            px[3 * i + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
        o[j] = (unsigned)px[4 * j] | ((unsigned)px[4 * j + 1] << 8) | ((unsigned)px[4 * j + 2] << 16) | ((unsigned)px[4 * j + 3] << 24);
}

}  // namespace

extern "C" {

int fid_align_crops(fid_ctx *ctx, const uint8_t *frames_dev, int B, int H, int W, const float *kps_dev,
                    const int32_t *counts_dev, int cap, int faces_per_frame, uint8_t *crops_dev, double *M_dev) {
    FID_REQUIRE(ctx && frames_dev && kps_dev && counts_dev && crops_dev, "NULL argument");
    FID_REQUIRE(B > 0 && H > 0 && W > 0 && cap > 0 && faces_per_frame > 0, "bad sizes");
    FID_REQUIRE((long long)B * faces_per_frame <= 65535, "too many face slots per call (%lld)", (long long)B * faces_per_frame);
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    dim3 grid(OUT / ROWS_PER_BLOCK, B * faces_per_frame);
    hipLaunchKernelGGL(align_warp, grid, dim3(256), 0, ctx->stream, frames_dev, H, W, kps_dev, counts_dev, cap,
                       faces_per_frame, crops_dev, M_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

int fid_letterbox(fid_ctx *ctx, const uint8_t *frames_dev, int B, int H, int W, uint8_t *out_dev, int in_h, int in_w,
                  double *det_scale) {
    FID_REQUIRE(ctx && frames_dev && out_dev, "NULL argument");
    FID_REQUIRE(B > 0 && H > 0 && W > 0 && in_h > 0 && in_w > 0 && B <= 65535 && in_h <= 65535, "bad sizes");
    // scrfd.py:125-134 in double, int() truncation
    const double im_ratio = (double)H / (double)W, model_ratio = (double)in_h / (double)in_w;
    int new_h, new_w;
    if (im_ratio > model_ratio) { new_h = in_h; new_w = (int)((double)new_h / im_ratio); }
    else { new_w = in_w; new_h = (int)((double)new_w * im_ratio); }
    FID_REQUIRE(new_h > 0 && new_w > 0, "degenerate letterbox %dx%d", new_w, new_h);
    if (det_scale) *det_scale = (double)new_h / (double)H;
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));            // (a thread may drive contexts on several devices)
    dim3 grid(fid::cdiv(in_w, 256), in_h, B);
    const int area2x = (W == 2 * new_w && H == 2 * new_h) ? 1 : 0;
    static const bool lb_bytes = getenv("FID_LETTERBOX_BYTES") != nullptr;      // A/B: the one-pixel-per-thread kernel everywhere
    if (!lb_bytes && !area2x && !(new_w == W && new_h == H) && in_w % 4 == 0 && ((size_t)out_dev & 3) == 0) {
        dim3 grid4(fid::cdiv(in_w / 4, 256), in_h, B);
        hipLaunchKernelGGL(letterbox_kernel4, grid4, dim3(256), 0, ctx->stream, frames_dev, H, W, out_dev, in_h, in_w, new_h, new_w,
                           (double)W / (double)new_w, (double)H / (double)new_h, (size_t)B * H * W * 3);
        FID_HIP(hipGetLastError());
        return FID_OK;
    }
    hipLaunchKernelGGL(letterbox_kernel, grid, dim3(256), 0, ctx->stream, frames_dev, H, W, out_dev, in_h, in_w, new_h, new_w,
                       (double)W / (double)new_w, (double)H / (double)new_h, area2x);
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // extern "C"
