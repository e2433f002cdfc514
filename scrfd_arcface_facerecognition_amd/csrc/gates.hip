// Face gates of the reference's product layer for every face of a batch in one launch:
//   assess_face_quality           smart_face_recognition.py:1145-1216   -> quality [overall, blur, pose, lighting, size]
//   is_side_face / pose angles    smart_face_recognition.py:1218-1297   -> side flag
//   analyze_bbox_for_side_face    smart_face_recognition.py:1299-1399   -> side score
//   best face + the rejections    smart_face_recognition.py:1473-1519   -> per frame: index of the first highest-score face, verdict
// Inputs are the post-process's own device arrays (det [B, cap, 5], kps [B, cap, 10], counts [B]): nothing goes through the host.
// Arithmetic: the reference's face fields are float32 and its constants python numbers, i.e. float32 operations with the constant rounded
// to float32 (NumPy >= 2); explicit __f*_rn operations in the reference's order (no contraction), so the results equal the reference's
// bit for bit (tests/golden/gates.npz).  Pose angles (radians, optional) are compared in degrees in float64 like math.degrees.
#include "common.h"

namespace fid {
namespace {

__device__ __forceinline__ float cap1(float x) { return x < 1.0f ? x : 1.0f; }      // python's min(1.0, x)

__device__ void quality5(const float *d, const float *k, const fid_gate_config &c, float *q) {
    const float det = d[4];
    const float area = __fmul_rn(__fsub_rn(d[2], d[0]), __fsub_rn(d[3], d[1]));
    const float size = cap1(__fdiv_rn(area, c.size_normalization));
    const float blur = cap1(__fmul_rn(det, 1.2f));
    float x0 = k[0], x1 = k[0], y0 = k[1], y1 = k[1];
    for (int i = 1; i < 5; i++) {
        x0 = fminf(x0, k[2 * i]); x1 = fmaxf(x1, k[2 * i]);
        y0 = fminf(y0, k[2 * i + 1]); y1 = fmaxf(y1, k[2 * i + 1]);
    }
    const float pose = cap1(__fdiv_rn(__fadd_rn(__fsub_rn(x1, x0), __fsub_rn(y1, y0)), 100.0f));
    const float light = cap1(__fmul_rn(det, 1.1f));
    float o = __fadd_rn(__fmul_rn(det, c.w_detection), __fmul_rn(size, c.w_size));
    o = __fadd_rn(o, __fmul_rn(blur, c.w_blur));
    o = __fadd_rn(o, __fmul_rn(pose, c.w_pose));
    o = __fadd_rn(o, __fmul_rn(light, c.w_lighting));
    q[0] = o; q[1] = blur; q[2] = pose; q[3] = light; q[4] = size;
}

__device__ int bbox_side_score(const float *d, const fid_gate_config &c) {
    const float w = __fsub_rn(d[2], d[0]), h = __fsub_rn(d[3], d[1]), left = d[0], top = d[1], det = d[4];
    if (w <= 0.f || h <= 0.f) return 0;
    const float ratio = __fdiv_rn(w, h), area = __fmul_rn(w, h), perim = __fmul_rn(2.0f, __fadd_rn(w, h));
    const float comp = perim > 0.f ? __fdiv_rn(__fmul_rn((float)(4 * 3.14159), area), __fmul_rn(perim, perim)) : 0.f;
    int s = 0;
    if (ratio < c.ar_extreme_profile) s += 4;
    else if (ratio < c.ar_very_strong_profile) s += 3;
    else if (ratio < c.ar_strong_profile) s += 2;
    else if (ratio > c.ar_very_wide) s += 3;
    else if (ratio > c.ar_wide) s += 2;
    else if (ratio > c.ar_moderately_wide) s += 1;
    if (area < c.area_extremely_small) s += 3;
    else if (area < c.area_very_small) s += 2;
    else if (area < c.area_small) s += 1;
    else if (area > c.area_very_large) s += 2;
    else if (area > c.area_large) s += 1;
    if (comp < c.compactness_very_low) s += 2;
    else if (comp < c.compactness_low) s += 1;
    if (det != 0.f && det < c.confidence_very_low) s += 2;
    else if (det != 0.f && det < c.confidence_low) s += 1;
    if (left < c.edge_position_threshold || top < c.edge_position_threshold) s += 1;
    return s;
}

// one workgroup per frame, one thread per face slot; thread 0 then picks the frame's best face
__global__ void __launch_bounds__(256) face_gates(const float *__restrict__ det, const float *__restrict__ kps, const int *__restrict__ counts,
                                                  int cap, int F, const double *__restrict__ pose, const fid_gate_config c,
                                                  float *__restrict__ quality, int *__restrict__ side, int *__restrict__ best) {
    const int b = blockIdx.x;
    const int n = min(counts[b], F);
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float q[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int sd = 0;
        if (f < n) {
            const float *d = det + ((size_t)b * cap + f) * 5;
            quality5(d, kps + ((size_t)b * cap + f) * 10, c, q);
            const int score = bbox_side_score(d, c);
            bool flag = score >= c.decision_threshold;
            if (pose) {                                         // the angles decide whenever one of them is available (non-zero)
                // math.degrees(x) = x * (180 / pi) in float64 (smart_face_recognition.py:1226-1240): the angles cross the boundary as float64
                const double yaw = fabs(pose[((size_t)b * F + f) * 2] * (180.0 / 3.14159265358979323846));
                const double pitch = fabs(pose[((size_t)b * F + f) * 2 + 1] * (180.0 / 3.14159265358979323846));
                if (yaw > 0.0 || pitch > 0.0) flag = yaw > (double)c.yaw_threshold || pitch > (double)c.pitch_threshold;
            }
            sd = score | ((int)flag << 16);
        }
        for (int i = 0; i < 5; i++) quality[((size_t)b * F + f) * 5 + i] = q[i];
        side[(size_t)b * F + f] = sd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int idx = -1, verdict = FID_GATE_NO_FACE;
        if (n > 0) {
            idx = 0;
            for (int f = 1; f < n; f++)                         // python's max(): the FIRST face with the highest score
                if (det[((size_t)b * cap + f) * 5 + 4] > det[((size_t)b * cap + idx) * 5 + 4]) idx = f;
            const float sc = det[((size_t)b * cap + idx) * 5 + 4];
            if (sc < c.confidence_threshold) verdict = FID_GATE_LOW_CONFIDENCE;
            else if (side[(size_t)b * F + idx] >> 16) verdict = FID_GATE_SIDE_FACE;
            else if (quality[((size_t)b * F + idx) * 5] < c.min_quality_threshold) verdict = FID_GATE_LOW_QUALITY;
            else verdict = FID_GATE_ACCEPT;
        }
        best[2 * b] = idx;
        best[2 * b + 1] = verdict;
    }
}

}  // namespace
}  // namespace fid

using namespace fid;

extern "C" int fid_face_gates(fid_ctx *ctx, const float *det_dev, const float *kps_dev, const int32_t *counts_dev, int B, int cap,
                              int faces_per_frame, const double *pose_dev, const fid_gate_config *cfg, float *quality_dev, int32_t *side_dev,
                              int32_t *best_dev) {
    FID_REQUIRE(ctx && det_dev && kps_dev && counts_dev && cfg && quality_dev && side_dev && best_dev, "NULL argument");
    FID_REQUIRE(B >= 0 && cap >= 1 && faces_per_frame >= 1 && faces_per_frame <= cap, "B=%d cap=%d faces_per_frame=%d", B, cap, faces_per_frame);
    if (B == 0) return FID_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    FID_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(face_gates, dim3(B), dim3(256), 0, ctx->stream, det_dev, kps_dev, counts_dev, cap, faces_per_frame, pose_dev, *cfg, quality_dev,
                       side_dev, best_dev);
    FID_HIP(hipGetLastError());
    return FID_OK;
}
