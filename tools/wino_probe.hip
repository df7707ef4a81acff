// Micro-benchmark behind DESIGN section 11 "Winograd F(2x2, 3x3) for the split-mode 3x3s": what does the INPUT TRANSFORM cost on the
// VALU when the data are hi / lo fp16 planes?  Per (2x2 output tile, input channel): read the 4 x 4 input patch in both planes from
// LDS, join to fp32, B^T d B (32 additions), split the 16 results again, write them to LDS as the A operand of 16 transform-domain
// products.  Reported: SIMD cycles per (tile, channel) with 1 / 2 waves per SIMD, next to the MFMA cycles the pair feeds
// (16 positions x N_out x 3 planes at 512 MAC per cycle per SIMD).
//     hipcc --offload-arch=gfx950 -O3 tools/wino_probe.hip -o tools/wino_probe && tools/wino_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack2(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2)); }
__device__ __forceinline__ float lo_f(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float hi_f(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }

// LDS input image: [rows of 18 px][18 px][32 ch] fp16, hi plane then lo plane (a 16 x 16 output patch = 8 x 8 tiles of 2 x 2 + halo).
// A lane owns (tile, 8-channel chunk): 64 tiles x 4 chunks = 256 lanes = the block.  Output image: [16 positions][64 tiles][32 ch] x 2 planes.
constexpr int IW = 18, CH = 32;
constexpr int IN_PLANE = IW * IW * CH * 2;            // 20736 B
constexpr int OUT_PLANE = 16 * 64 * CH * 2;           // 65536 B

__global__ __launch_bounds__(256) void wino_in_transform(int iters, unsigned long long* cycles, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* in_h = smem; char* in_l = smem + IN_PLANE;
    char* out_h = smem + 2 * IN_PLANE; char* out_l = out_h;      // (timing probe: the lo plane lands over the hi plane -- both planes do not fit 160 KiB)
    for (int i = threadIdx.x; i < 2 * IN_PLANE / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003800u + i * 2654435761u % 1024u;
    __syncthreads();
    const int tile = threadIdx.x >> 2, chunk = threadIdx.x & 3;
    const int ty = tile >> 3, tx = tile & 7;
    uint32_t acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float d[16][8];
#pragma unroll
        for (int py = 0; py < 4; ++py)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const int off = (((2 * ty + py) * IW + 2 * tx + px) * CH + chunk * 8) * 2;
                const u32x4 h = *reinterpret_cast<const u32x4*>(in_h + off), l = *reinterpret_cast<const u32x4*>(in_l + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) { d[py * 4 + px][2 * e] = lo_f(h[e]) + lo_f(l[e]); d[py * 4 + px][2 * e + 1] = hi_f(h[e]) + hi_f(l[e]); }
            }
        // B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows, then columns
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t[16];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                t[0 * 4 + px] = d[0 * 4 + px][c] - d[2 * 4 + px][c];
                t[1 * 4 + px] = d[1 * 4 + px][c] + d[2 * 4 + px][c];
                t[2 * 4 + px] = d[2 * 4 + px][c] - d[1 * 4 + px][c];
                t[3 * 4 + px] = d[1 * 4 + px][c] - d[3 * 4 + px][c];
            }
#pragma unroll
            for (int py = 0; py < 4; ++py) {
                d[py * 4 + 0][c] = t[py * 4 + 0] - t[py * 4 + 2];
                d[py * 4 + 1][c] = t[py * 4 + 1] + t[py * 4 + 2];
                d[py * 4 + 2][c] = t[py * 4 + 2] - t[py * 4 + 1];
                d[py * 4 + 3][c] = t[py * 4 + 1] - t[py * 4 + 3];
            }
        }
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            u32x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = pack2(d[pos][2 * e], d[pos][2 * e + 1]);
                l[e] = pack2(d[pos][2 * e] - lo_f(h[e]), d[pos][2 * e + 1] - hi_f(h[e]));
            }
            const int off = ((pos * 64 + tile) * CH + chunk * 8) * 2;
            *reinterpret_cast<u32x4*>(out_h + off) = h;
            *reinterpret_cast<u32x4*>(out_l + off) = l;
            acc ^= h[0] ^ l[3];
        }
        __syncthreads();
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t lds = 2 * IN_PLANE + OUT_PLANE;
    hipFuncSetAttribute((const void*)wino_in_transform, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long* cyc; uint32_t* sink;
    hipMalloc(&cyc, 1024 * 8); hipMalloc(&sink, 1024 * 256 * 4);
    const int iters = 200;
    for (int blocks : {256}) {
        wino_in_transform<<<blocks, 256, lds>>>(iters, cyc, sink);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        wino_in_transform<<<blocks, 256, lds>>>(iters, cyc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
        // a block iteration transforms 64 tiles x 32 channels with 4 waves (one per SIMD)
        const double pairs_per_simd = 64.0 * 32.0 / 4.0;
        printf("blocks %d (one 4-wave block per CU, one wave per SIMD): %.1f us; %.0f shader cycles (s_memtime) per iteration = %.2f cycles per (tile, channel) per SIMD\n",
               blocks, ms * 1e3, avg / iters, avg / iters / pairs_per_simd);
        printf("  (wall: %.3f ns per (tile, channel) per SIMD = %.2f cycles at 2.4 GHz)\n", ms * 1e6 / iters / pairs_per_simd, ms * 1e6 / iters / pairs_per_simd * 2.4);
        printf("  MFMA time the pair feeds: 16 positions x N_out x 3 planes / 512 MAC per cycle per SIMD = %.0f / %.0f / %.0f cycles at N_out = 64 / 128 / 256\n",
               16 * 64 * 3 / 512.0, 16 * 128 * 3 / 512.0, 16 * 256 * 3 / 512.0);
        printf("  direct 3x3 for the same 2 x 2 outputs: 4 px x 9 taps x N_out x 3 planes / 512 = %.1f / %.1f / %.1f cycles\n", 36 * 64 * 3 / 512.0, 36 * 128 * 3 / 512.0, 36 * 256 * 3 / 512.0);
    }
    return 0;
}
