// Store-pattern microbenchmark (experiment, not product): how fast can 256 threads x 8 iterations write a
// bf16 [M, N] matrix when each block owns a 128 x 128 tile (256-B row segments at N*2-byte stride)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ char dyn_lds[];
template <int MODE>
__global__ __launch_bounds__(256) void wr(uint16_t* __restrict__ C, const uint16_t* __restrict__ R, int M, int N, int tiles_n) {
    const int tid = threadIdx.x;
    if (MODE & 8) { reinterpret_cast<volatile int*>(dyn_lds)[tid] = tid; __syncthreads(); }
    int t = blockIdx.x;
    if (MODE & 4) {   // xcd remap
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (t / tiles_n) * 128, n0 = (t % tiles_n) * 128;
    const int c8 = tid & 15, rbase = tid >> 4;
    u32x4 acc = {1u, 2u, 3u, (uint32_t)tid};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int m = m0 + rbase + u * 16;
        const size_t off = (size_t)m * N + n0 + c8 * 8;
        u32x4 v = acc;
        if (MODE & 1) { const u32x4 r4 = *reinterpret_cast<const u32x4*>(R + off); v[0] += r4[0]; v[1] ^= r4[1]; v[2] += r4[2]; v[3] ^= r4[3]; }
        if (MODE & 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(C + off));
        else *reinterpret_cast<u32x4*>(C + off) = v;
    }
}

__global__ void lin(u32x4* __restrict__ C, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) C[i] = u32x4{1, 2, 3, (uint32_t)i};
}

template <typename F> float timeit(F f, int it = 20) {
    for (int i = 0; i < 3; ++i) f();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); for (int i = 0; i < it; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / it * 1e3f;
}

int main() {
    const int M = 627200;
    for (int N : {256}) {
        uint16_t *C, *R; size_t bytes = (size_t)M * N * 2;
        hipMalloc(&C, bytes); hipMalloc(&R, bytes); hipMemset(R, 1, bytes);
        const int tiles_n = N / 128, nwg = (M / 128) * tiles_n;
        float t;
        t = timeit([&] { hipLaunchKernelGGL(lin, dim3(2048), dim3(256), 0, 0, (u32x4*)C, bytes / 16); }); printf("N=%d linear store            %7.1f us %5.2f TB/s\n", N, t, bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(wr<0>, dim3(nwg), dim3(256), 0, 0, C, R, M, N, tiles_n); }); printf("N=%d tile store              %7.1f us %5.2f TB/s\n", N, t, bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(wr<4>, dim3(nwg), dim3(256), 0, 0, C, R, M, N, tiles_n); }); printf("N=%d tile store xcd-remap    %7.1f us %5.2f TB/s\n", N, t, bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(wr<2>, dim3(nwg), dim3(256), 0, 0, C, R, M, N, tiles_n); }); printf("N=%d tile store nontemporal  %7.1f us %5.2f TB/s\n", N, t, bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(wr<6>, dim3(nwg), dim3(256), 0, 0, C, R, M, N, tiles_n); }); printf("N=%d tile store nt+remap     %7.1f us %5.2f TB/s\n", N, t, bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(wr<1>, dim3(nwg), dim3(256), 0, 0, C, R, M, N, tiles_n); }); printf("N=%d tile load+store         %7.1f us %5.2f TB/s (r+w)\n", N, t, 2 * bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(wr<5>, dim3(nwg), dim3(256), 0, 0, C, R, M, N, tiles_n); }); printf("N=%d tile load+store remap   %7.1f us %5.2f TB/s (r+w)\n", N, t, 2 * bytes / t / 1e6);
        for (int lds : {0, 32768, 65536, 98304}) {
            t = timeit([&] { hipLaunchKernelGGL(wr<9>, dim3(nwg), dim3(256), lds, 0, C, R, M, N, tiles_n); });
            printf("N=%d tile load+store, %3d KB LDS/block (%d blocks/CU) %7.1f us %5.2f TB/s (r+w)\n", N, lds / 1024, lds ? 160 * 1024 / lds : 8, t, 2 * bytes / t / 1e6);
            t = timeit([&] { hipLaunchKernelGGL(wr<8>, dim3(nwg), dim3(256), lds, 0, C, R, M, N, tiles_n); });
            printf("N=%d tile store only,  %3d KB LDS/block              %7.1f us %5.2f TB/s\n", N, lds / 1024, t, bytes / t / 1e6);
        }
        hipFree(C); hipFree(R);
    }
    return 0;
}
