// Calibration: what the matrix cores of this box sustain with NOTHING but independent v_mfma_f32_16x16x32_f16 in flight
// (no LDS, no memory).  Usage: tools/micro/mfma_peak [waves_per_cu = 8] [iterations = 20000]
//   make -C tools/micro && gpurun -- tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters) {
    half8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];      // (never true: keeps the loop alive)
}

// the same loop on v_mfma_f32_16x16x16_f16 (K = 16): does a half-K step cost half a 16x16x32 step?  (K granularity of the 56 / 80 / 88-channel layers)
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop16(float *out, int iters) {
    half4 a, b;
    for (int e = 0; e < 4; e++) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];
}

int main(int argc, char **argv) {
    const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 8;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = cus * waves_per_cu / 4;                 // 4 waves per workgroup
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<16>), dim3(wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * 4 * iters * 16 * (16.0 * 16 * 32 * 2);
        printf("%s: %d CUs, %d waves/CU, %d x 16 MFMA 16x16x32 f16 per wave: %.3f ms = %.1f TFLOP/s (%.0f %% of 2.5 PF), implied clock %.2f GHz\n", prop.name, cus,
               waves_per_cu, iters, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500 * 100, flop / ms * 1e-9 / 2500 * 2.4);
    }
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop16<16>), dim3(wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)wgs * 4 * iters * 16 * (16.0 * 16 * 16 * 2);
        printf("16x16x16 f16: %.3f ms for the same number of MFMAs = %.1f TFLOP/s\n", ms, flop / ms * 1e-9);
    }
    return 0;
}
