// L2 -> CU staging-rate microbenchmark (experiment, not product): how many bytes per second can the CUs pull out of an
// L2-resident buffer, by request form (LDS-DMA 16 B/lane = 1 KiB per wave instruction, register loads 16 B/lane in
// fragment order = 1 KiB contiguous, register loads 16 B/lane from 64 rows at a 64-byte row pitch = "half lines"),
// waves per CU and requests in flight per wave.  Every kernel of the split mode saturates near 9-10 TB/s of CU-side
// loaded bytes; this says whether that is the hardware or the request pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/stage_probe.hip -o tools/stage_probe && tools/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

extern __shared__ __attribute__((aligned(16))) char smem[];

// FORM 0: LDS-DMA, 1 KiB contiguous per wave instruction.  FORM 1: registers, 1 KiB contiguous.  FORM 2: registers, 16 B per
// lane at a 64-byte pitch (each request touches 64 half-lines; the other halves are requested by the NEXT instruction).
// FORM 3: LDS-DMA, 64-byte row pitch pattern (4 lanes per row).
template <int FORM, int U>
__global__ __launch_bounds__(256) void stage(const char* __restrict__ buf, uint32_t span, int iters, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)span, 0x00020000);
    // every wave walks the whole buffer from its own starting point (decorrelated: no two waves of a CU ask for the same line at once)
    uint32_t pos = ((blockIdx.x * nw + wave) * 40960u) % span;
    char* my = smem + wave * (U * 1024);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t off;
            if (FORM == 0 || FORM == 1) off = pos + lane * 16;
            else off = pos + (lane >> 2) * 128 + (lane & 3) * 16 + (u & 1) * 64;      // FORM 2/3: even u = first halves, odd u = second halves of the same 64 lines
            if (off >= span) off -= span;
            if (FORM == 0 || FORM == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(my + u * 1024), 16, off, 0, 0, 0);
            else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                acc[0] ^= v[0]; acc[1] += v[1]; acc[2] ^= v[2]; acc[3] += v[3];
            }
            if (FORM == 0 || FORM == 1) pos += 1024; else pos += (u & 1) ? 8192 : 0;
            if (pos >= span) pos -= span;
        }
        if (FORM == 0 || FORM == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (FORM == 0 || FORM == 3) acc[0] = *reinterpret_cast<volatile uint32_t*>(my + lane * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <typename F> float timeit(F f, int it = 10) {
    for (int i = 0; i < 2; ++i) f();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); for (int i = 0; i < it; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / it * 1e3f;
}

template <int FORM, int U>
void run(const char* name, const char* buf, uint32_t span, uint32_t* sink, int blocks_per_cu, int threads) {
    const int iters = 4096 / U;
    const int grid = 256 * blocks_per_cu;
    const size_t lds = (size_t)(threads / 64) * U * 1024;
    hipFuncSetAttribute((const void*)stage<FORM, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float us = timeit([&] { hipLaunchKernelGGL((stage<FORM, U>), dim3(grid), dim3(threads), lds, 0, buf, span, iters, sink); });
    const double bytes = (double)grid * (threads / 64) * iters * U * 1024.0;
    printf("%-34s span %6.2f MB  %2d waves/CU  %2d in flight/wave  %8.1f us  %6.2f TB/s  (%5.1f B/clk/CU at 2.0 GHz)\n", name, span / 1048576.0,
           blocks_per_cu * threads / 64, U, us, bytes / us / 1e6, bytes / us / 1e6 * 1e12 / 256 / 2.0e9);
}

int main() {
    char* buf; uint32_t* sink;
    const uint32_t cap = 256u << 20;
    hipMalloc(&buf, cap); hipMemset(buf, 1, cap); hipMalloc(&sink, 4);
    for (uint32_t span : {1u << 20, 2u << 20, 16u << 20, 200u << 20}) {       // L2-resident (per XCD: 4 MiB), MALL-resident, HBM
        for (int bpc : {1, 2, 4}) {
            run<0, 4>("LDS-DMA 1 KiB contiguous", buf, span, sink, bpc, 256);
            run<0, 8>("LDS-DMA 1 KiB contiguous", buf, span, sink, bpc, 256);
            run<1, 8>("registers 1 KiB contiguous", buf, span, sink, bpc, 256);
            run<2, 8>("registers 16 B @ 64 B pitch halves", buf, span, sink, bpc, 256);
            run<3, 8>("LDS-DMA 64 B rows (half lines)", buf, span, sink, bpc, 256);
        }
        run<0, 8>("LDS-DMA 1 KiB contiguous, 512 thr", buf, span, sink, 2, 512);
        run<0, 16>("LDS-DMA 1 KiB contiguous", buf, span, sink, 2, 256);
        run<1, 16>("registers 1 KiB contiguous", buf, span, sink, 2, 256);
    }
    return 0;
}
