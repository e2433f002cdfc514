// Calibration: the inner loop of the halo-patch convs (csrc/conv_wr.hip) reduced to its matrix part -- B fragments read from LDS with
// ds_read_b128, each feeding M back-to-back MFMAs whose A operands sit in registers -- for the two fp16 MFMA shapes, on random data,
// with the in-kernel clock (s_memtime / s_memrealtime) reported beside the rate, so that "cycles per MFMA" (issue) and "clock held
// under load" (power) can be told apart.  VERDICT r2 item 1 asks whether v_mfma_f32_32x32x16_f16 should replace 16x16x32.
//   make -C tools/micro && gpurun -- tools/micro/mfma_lds
// Usage: mfma_lds [iterations = 4000] [zero = 0]
//   shape 16: a "row step" = 1 ds_read_b128 (16 px x 32 ch) + 3 MFMA 16x16x32 per tile (NT = 2 tiles: 6 MFMAs of 16 cycles = 96 cycles per read pair)
//   shape 32: a "row step" = 1 ds_read_b128 (32 px x 16 ch) + 3 MFMA 32x32x16 (96 cycles per read)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long t0, t1, r0, r1; };

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ _Float16 rnd_half(unsigned seed, int zero) {
    if (zero) return (_Float16)0.f;
    return (_Float16)(((int)(hash32(seed) & 0xFFFF) - 32768) * (1.0f / 65536.0f));     // uniform in [-0.5, 0.5)
}

// MODE 0: operands in registers only; MODE 1: B fragments from LDS (one read per row step, prefetched PD steps ahead)
template <int SHAPE, int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 2) loop_kernel(float *out, Stamp *st, int iters, int zero) {
    constexpr int ROWS = SHAPE == 16 ? 16 : 9;        // patch rows per column: 14 + 2 / 7 + 2
    constexpr int TH = ROWS - 2;
    constexpr int NT = SHAPE == 16 ? 2 : 1;
    constexpr int PD = 2;
    __shared__ __attribute__((aligned(16))) _Float16 sP[2 * 18 * 18 * 32 + 64];     // two tiles' 18x18x32 patches (41 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 18 * 18 * 32; i += WAVES * 64) sP[i] = rnd_half(i * 7919u + blockIdx.x, zero);
    half8 w[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) w[i][e] = rnd_half((tid * 9 + i) * 8 + e + 12345u, zero);
    __syncthreads();
    // per-lane fragment base: 16-px shape: pixel = lane & 15, 8-channel group = lane >> 4 (of 4); 32-px shape: pixel = lane & 31 (16 of tile 0, 16 of tile 1), group = lane >> 5 (of 2)
    // (XOR swizzle of the 16-byte channel groups by the pixel column as in conv_wr.hip; the second tile of a 32-px fragment flips bit 0 of
    // it so that the hardware's 16-lane groups of a ds_read_b128 still touch 16 different bank quads)
    int base, base_hi;
    if (SHAPE == 16) {
        base = (lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 1) & 3)) << 4);
        base_hi = base;
    } else {
        const int t1 = (lane & 31) >> 4, sw = ((((lane & 15) >> 1) & 3) ^ t1);
        base = t1 * (18 * 18 * 64) + (lane & 15) * 64 + (((0 + (lane >> 5)) ^ sw) << 4);
        base_hi = t1 * (18 * 18 * 64) + (lane & 15) * 64 + (((2 + (lane >> 5)) ^ sw) << 4);
    }
    asm volatile("" : "+v"(base), "+v"(base_hi));
    const char *sb = (const char *)sP;

    f32x4 acc4[SHAPE == 16 ? NT * TH : 1];
    f32x16 acc16[SHAPE == 32 ? TH : 1];
#pragma unroll
    for (int i = 0; i < (SHAPE == 16 ? NT * TH : 1); i++) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (SHAPE == 32 ? TH : 1); i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc16[i][e] = 0.f;

    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            // 32-px shape: two 16-channel halves per 32-channel chunk -> the column is walked twice (the second half's fragments are 32 B further)
#pragma unroll
            for (int kh = 0; kh < (SHAPE == 32 ? 2 : 1); kh++) {
                half8 pq[PD + 1][NT];
                auto load_p = [&](int r, int set) {
                    const int K = r * 18 + dx;
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        if (MODE == 1) pq[set][t] = *(const half8 *)(sb + (kh ? base_hi : base) + K * 64 + t * (18 * 18 * 64));
                        else pq[set][t] = w[(r + t) % 9];
                    }
                };
#pragma unroll
                for (int r = 0; r < PD; r++) load_p(r, r % (PD + 1));
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    if (r + PD < ROWS) load_p(r + PD, (r + PD) % (PD + 1));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dy = 0; dy < 3; dy++) {
                        const int mi = r - dy;
                        if (mi < 0 || mi >= TH) continue;
                        if constexpr (SHAPE == 16) {
#pragma unroll
                            for (int t = 0; t < NT; t++)
                                acc4[t * TH + mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy * 3 + dx], pq[r % (PD + 1)][0 + t], acc4[t * TH + mi], 0, 0, 0);
                        } else {
                            acc16[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[dy * 3 + dx], pq[r % (PD + 1)][0], acc16[mi], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (tid == 0) st[blockIdx.x] = Stamp{t0, t1, r0, r1};
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < (SHAPE == 16 ? NT * TH : 1); i++) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < (SHAPE == 32 ? TH : 1); i++)
#pragma unroll
        for (int e = 0; e < 16; e++) s += acc16[i][e];
    if (s == 12345.678f) out[tid] = s;                 // (practically never: keeps the loop alive)
}

template <int SHAPE, int MODE, int WAVES>
static void run(const char *name, int cus, int iters, int zero, float *out, Stamp *st_dev) {
    const int wgs = cus;                               // one workgroup per CU (the convs' persistent grids)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Stamp> st(wgs);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((loop_kernel<SHAPE, MODE, WAVES>), dim3(wgs), dim3(WAVES * 64), 0, 0, out, st_dev, iters, zero);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(st.data(), st_dev, wgs * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<double> clk(wgs), cyc(wgs);
        for (int i = 0; i < wgs; i++) {
            cyc[i] = (double)(st[i].t1 - st[i].t0);
            clk[i] = cyc[i] / ((double)(st[i].r1 - st[i].r0) * 10.0);            // s_memrealtime ticks at 100 MHz -> GHz
        }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        // MFMAs per wave and iteration: 3 columns x TH rows x 3 taps x (2 tiles | 2 channel halves)
        const int TH = SHAPE == 16 ? 14 : 7;
        const double mf = 3.0 * TH * 3 * 2, flop_per = SHAPE == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
        const double flop = (double)wgs * WAVES * iters * mf * flop_per;
        const double cyc_per_mfma_simd = cyc[wgs / 2] / (iters * mf * (WAVES / 4.0));       // shader cycles per MFMA on one SIMD
        printf("%-28s %s waves/CU %d: %.3f ms = %7.1f TFLOP/s | median in-kernel clock %.3f GHz, %.2f cycles per MFMA and SIMD (pipe: %d) -> %.0f %% of the pipe rate\n", name,
               zero ? "zeros " : "random", WAVES, ms, flop / ms * 1e-9, clk[wgs / 2], cyc_per_mfma_simd, SHAPE == 16 ? 16 : 32,
               100.0 * (SHAPE == 16 ? 16 : 32) / cyc_per_mfma_simd);
    }
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int zero = argc > 2 ? atoi(argv[2]) : 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out; Stamp *st;
    hipMalloc(&out, 1 << 16);
    hipMalloc(&st, cus * sizeof(Stamp));
    printf("%s, %d CUs, %d iterations per wave\n", prop.name, cus, iters);
    for (int z = 0; z <= (zero ? 1 : 0); z++) {
        run<16, 0, 8>("16x16x32 registers only", cus, iters, z, out, st);
        run<32, 0, 8>("32x32x16 registers only", cus, iters, z, out, st);
        run<16, 1, 8>("16x16x32 LDS-fed (wr loop)", cus, iters, z, out, st);
        run<32, 1, 8>("32x32x16 LDS-fed (wr loop)", cus, iters, z, out, st);
        run<16, 1, 4>("16x16x32 LDS-fed (wr loop)", cus, iters, z, out, st);
        run<32, 1, 4>("32x32x16 LDS-fed (wr loop)", cus, iters, z, out, st);
    }
    return 0;
}
