// Does vmcnt retire LOADS and STORES in issue order on gfx950?  (experiment, not product)
// A persistent GEMM wants to issue the next tile's operand loads, then the current tile's result stores, and wait with
// s_waitcnt vmcnt(<number of stores>) for the loads alone.  That is only correct if a younger store can never be counted
// as complete while an older load is still outstanding.  Here every wave issues ONE slow LDS-DMA load (a cold HBM line
// far away, 1 KiB per wave) into an LDS slot holding a sentinel, then NS fast stores (16 B per lane into lines the wave
// has just written, L2-resident), then s_waitcnt vmcnt(NS), then reads the slot: a sentinel = the wait let go early.
//   hipcc --offload-arch=gfx950 -O3 tools/vmcnt_probe.hip -o tools/vmcnt_probe && tools/vmcnt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int NS, int SLACK>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ cold, uint32_t cold_bytes, char* __restrict__ hot, uint32_t hot_bytes,
                                             unsigned long long* __restrict__ bad, int rounds) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* slot = smem + wave * 1024;
    const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, (int)cold_bytes, 0x00020000);
    const auto rsH = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, (int)hot_bytes, 0x00020000);
    const uint32_t gw = blockIdx.x * 4 + wave;
    const uint32_t hot_base = (gw * (uint32_t)(NS * 1024)) % (hot_bytes - NS * 1024);
    unsigned long long nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        // sentinel into the slot, warm the wave's hot lines (so the probe's stores hit L2)
        *reinterpret_cast<u32x4*>(slot + lane * 16) = u32x4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (uint32_t)r}, rsH, hot_base + s * 1024 + lane * 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // a cold line: a different 1 KiB of a multi-GiB buffer every round and wave (pseudo-random walk)
        const uint32_t line = (uint32_t)((gw * 2654435761u + (uint32_t)r * 40503u) % (cold_bytes / 1024));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (lds_void*)slot, 16, line * 1024u + lane * 16, 0, 0, 0);     // OLD, slow
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b128(u32x4{5u, 6u, 7u, (uint32_t)r}, rsH, hot_base + s * 1024 + lane * 16, 0, 0);   // YOUNG, fast
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + SLACK) : "memory");          // SLACK = 1: the control -- does not cover the load, MUST see sentinels
        const u32x4 v = *reinterpret_cast<volatile u32x4*>(slot + lane * 16);
        const uint32_t expect = (line * 1024u + lane * 16) / 4;          // cold[i] = i (dwords)
        if (v[0] != expect || v[1] != expect + 1) ++nbad;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const uint32_t cold_bytes = 0x7ff00000u, hot_bytes = 64u << 20;
    char *cold, *hot; unsigned long long* bad;
    hipMalloc(&cold, cold_bytes); hipMalloc(&hot, hot_bytes); hipMalloc(&bad, 8);
    {   // cold[i] = i
        std::vector<uint32_t> h(cold_bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
        hipMemcpy(cold, h.data(), cold_bytes, hipMemcpyHostToDevice);
    }
    auto run = [&](auto kern, int ns, int blocks) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, cold, cold_bytes, hot, hot_bytes, bad, 200);
        hipDeviceSynchronize();
        unsigned long long b; hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
        printf("NS=%2d stores behind one cold LDS-DMA load, %5d blocks x 4 waves x 200 rounds: %llu lanes saw the sentinel (of %llu)\n", ns, blocks, b,
               (unsigned long long)blocks * 256 * 200);
    };
    for (int blocks : {256, 2048}) {
        run(probe<1, 0>, 1, blocks); run(probe<4, 0>, 4, blocks); run(probe<16, 0>, 16, blocks); run(probe<32, 0>, 32, blocks);
        printf("control (wait one short of covering the load):\n");
        run(probe<16, 1>, 16, blocks);
        printf("control (vmcnt(0): everything retired -- must be clean):\n");
        run(probe<16, -16>, 16, blocks);
    }
    return 0;
}
