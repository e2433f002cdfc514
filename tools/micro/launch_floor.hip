// Calibration: what a short DEPENDENT kernel costs on one stream (the regime of MobileFaceNet at 32 faces: ~50 launches of 8-15 us).
//   chain<depth>: `wgs` workgroups of 256 threads; every thread follows `depth` dependent 16-byte global loads (a pointer chase through
//   a buffer of `mb` MB: L2 / Infinity-Cache / HBM resident by size), optionally a workgroup barrier between hops, then one store.
// Prints the steady-state time per launch of 300 back-to-back launches, each consuming the previous one's output (stream order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) chain(const uint4 *__restrict__ buf, unsigned n16, int depth, int barrier, uint4 *__restrict__ out) {
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) % n16;
    uint4 v = uint4{idx, 0, 0, 0};
    for (int d = 0; d < depth; d++) {
        v = buf[v.x % n16];
        if (barrier) __syncthreads();
    }
    out[blockIdx.x * 256u + threadIdx.x] = v;
}

int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 32, mb = argc > 2 ? atoi(argv[2]) : 4, launches = 300;
    const unsigned n16 = (unsigned)mb * (1u << 20) / 16;
    std::vector<uint4> h(n16);
    unsigned s = 12345;
    for (unsigned i = 0; i < n16; i++) { s = s * 1664525u + 1013904223u; h[i] = uint4{s % n16, s, i, 0}; }
    uint4 *buf, *out;
    CK(hipMalloc(&buf, (size_t)n16 * 16)); CK(hipMalloc(&out, (size_t)wgs * 256 * 16));
    CK(hipMemcpy(buf, h.data(), (size_t)n16 * 16, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d workgroups, %d MB buffer\n depth barrier  us/launch\n", wgs, mb);
    for (int barrier = 0; barrier < 2; barrier++)
        for (int depth : {0, 1, 2, 4, 8}) {
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, st, buf, n16, depth, barrier, out);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < launches; i++) hipLaunchKernelGGL(chain, dim3(wgs), dim3(256), 0, st, buf, n16, depth, barrier, out);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %5d %7d  %8.2f\n", depth, barrier, ms * 1e3f / launches);
        }
    return 0;
}
