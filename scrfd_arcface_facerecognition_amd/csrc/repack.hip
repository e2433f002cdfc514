// Alternate weight packings for kernels whose global->LDS / global->VGPR weight fetches want every 1-KB piece CONTIGUOUS in
// memory (8 whole 128-byte lines per wave instruction instead of 16 half lines: the CU's vector-memory front end works per
// cache line, and the halo-patch kernels are bound by how fast it accepts fill pieces).  The executor's packed weights are
// fp16 [Cout_p][tap][Cin_p] (lower.py); a repacked copy is built once per conv op, on the device, the first time a kernel
// family that wants it is timed or picked.
#include "conv.h"

namespace fid {
namespace {

typedef unsigned int rp_u32x4 __attribute__((ext_vector_type(4)));

// kind 1 -- conv3x3_pc2's LDS weight slot image, chunk by chunk: [cout block of 64][32-channel chunk][row = tap*64 + co][4 x 16 B],
// the 16-byte channel groups of a row XOR-swizzled like the slot ((row >> 1) & 3)
__global__ void __launch_bounds__(256) repack_pc2(const rp_u32x4 *__restrict__ src, rp_u32x4 *__restrict__ dst, int Cout_p, int Cin_p, long long n_units) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_units) return;
    const int n_chunks = Cin_p / 32;
    const int j = (int)(u % 2304);
    const long long blk = u / 2304;
    const int ck = (int)(blk % n_chunks), cb = (int)(blk / n_chunks);
    const int row = j >> 2, c = (j & 3) ^ ((row >> 1) & 3);
    const int t = row / 64, co = cb * 64 + (row - t * 64);
    dst[u] = src[((size_t)(co * 9 + t) * Cin_p + ck * 32 + c * 8) / 8];
}

// kind 2 -- conv3x3_wr: MFMA A fragments in lane order, 1 KB each, the 9 taps of a (cout fragment, 32-channel chunk) contiguous:
// [cout block of 128][32-channel chunk][cout fragment cf 0..7][dx][dy][lane][8 halfs]; lane l holds cout cb*128 + cf*16 + (l & 15),
// channels ck*32 + (l >> 4)*8 .. +7.  Couts beyond Cout_p (last block of a 64 / 96 / 224-channel layer) are zero.
__global__ void __launch_bounds__(256) repack_wr(const rp_u32x4 *__restrict__ src, rp_u32x4 *__restrict__ dst, int Cout_p, int Cin_p, long long n_units) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_units) return;
    const int n_chunks = Cin_p / 32;
    const int lane = (int)(u & 63);
    long long r = u >> 6;
    const int dy = (int)(r % 3); r /= 3;
    const int dx = (int)(r % 3); r /= 3;
    const int tap = dy * 3 + dx;
    const int cf = (int)(r % 8); r /= 8;
    const int ck = (int)(r % n_chunks);
    const int cb = (int)(r / n_chunks);
    const int co = cb * 128 + cf * 16 + (lane & 15);
    const int ch = ck * 32 + (lane >> 4) * 8;
    dst[u] = co < Cout_p ? src[((size_t)(co * 9 + tap) * Cin_p + ch) / 8] : rp_u32x4{0u, 0u, 0u, 0u};
}

// kind 3 -- conv_gw: MFMA A fragments in lane order for a kernel of T = kh*kw taps, in the kernel's K order (tap-major, chunks inside):
// [cout block of 128][tap = dy*kw + dx][32-channel chunk][cout fragment cf 0..7][lane][8 halfs]; couts beyond Cout_p are zero.
__global__ void __launch_bounds__(256) repack_gw(const rp_u32x4 *__restrict__ src, rp_u32x4 *__restrict__ dst, int Cout_p, int Cin_p, int taps, long long n_units) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_units) return;
    const int n_chunks = Cin_p / 32;
    const int lane = (int)(u & 63);
    long long r = u >> 6;
    const int cf = (int)(r % 8); r /= 8;
    const int ck = (int)(r % n_chunks); r /= n_chunks;
    const int tap = (int)(r % taps);
    const int cb = (int)(r / taps);
    const int co = cb * 128 + cf * 16 + (lane & 15);
    const int ch = ck * 32 + (lane >> 4) * 8;
    dst[u] = co < Cout_p ? src[((size_t)(co * taps + tap) * Cin_p + ch) / 8] : rp_u32x4{0u, 0u, 0u, 0u};
}

}  // namespace

size_t repack_bytes(int kind, int Cout_p, int Cin_p, int taps) { return (size_t)(kind >= 2 ? (Cout_p + 127) / 128 * 128 : Cout_p) * (kind == 3 ? taps : 9) * Cin_p * 2; }

int repack_weights(fid_ctx *ctx, int kind, const void *src, void *dst, int Cout_p, int Cin_p, int taps) {
    const long long n_units = (long long)repack_bytes(kind, Cout_p, Cin_p, taps) / 16;
    const unsigned grid = (unsigned)cdiv64(n_units, 256);
    if (kind == 1) {
        FID_REQUIRE(Cout_p % 64 == 0 && Cin_p % 32 == 0, "repack 1: %d x %d channels", Cout_p, Cin_p);
        hipLaunchKernelGGL(repack_pc2, dim3(grid), dim3(256), 0, ctx->stream, (const rp_u32x4 *)src, (rp_u32x4 *)dst, Cout_p, Cin_p, n_units);
    } else if (kind == 2) {
        FID_REQUIRE(Cout_p % 16 == 0 && Cin_p % 32 == 0, "repack 2: %d x %d channels", Cout_p, Cin_p);
        hipLaunchKernelGGL(repack_wr, dim3(grid), dim3(256), 0, ctx->stream, (const rp_u32x4 *)src, (rp_u32x4 *)dst, Cout_p, Cin_p, n_units);
    } else if (kind == 3) {
        FID_REQUIRE(Cout_p % 16 == 0 && Cin_p % 32 == 0 && taps >= 1, "repack 3: %d x %d channels, %d taps", Cout_p, Cin_p, taps);
        hipLaunchKernelGGL(repack_gw, dim3(grid), dim3(256), 0, ctx->stream, (const rp_u32x4 *)src, (rp_u32x4 *)dst, Cout_p, Cin_p, taps, n_units);
    } else {
        set_error("repack: unknown kind %d", kind);
        return FID_E_INVALID;
    }
    FID_HIP(hipGetLastError());
    return FID_OK;
}

}  // namespace fid
