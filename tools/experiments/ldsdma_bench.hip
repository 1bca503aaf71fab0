// Micro-benchmark: what does one `buffer_load_dwordx4 ... lds` (1 KiB LDS-DMA piece) cost the wavefront that issues it?
// One workgroup per CU (or `wgs` per CU), W wavefronts, each issues N pieces back to back, then waits for all of them.
// Marks (s_memtime, wavefront 0): t0 before the burst, t1 after the last issue, t2 after s_waitcnt vmcnt(0).
// Patterns: 0 = 1 KiB contiguous per piece; 1 = 16 segments of 64 B at a 128-byte pitch; 2 = 64 B at a 1-KiB pitch;
//           3 = pattern 1 with every piece in a different 10 KiB-apart row (the k_wgrad3 window).
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_bench ldsdma_bench.hip ; run: ./ldsdma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(v4i32 srd, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_addr), "s"(srd) : "memory", "m0");
}
template <int N>
__global__ __launch_bounds__(256) void k(const char* buf, unsigned bytes, int pattern, int iters, unsigned long long* out, int spread) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned long long a = (unsigned long long)buf;
    v4i32 srd;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    srd.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    srd.z = (int)bytes; srd.w = 0x00020000;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)sm);
    unsigned lo;
    if (pattern == 0) lo = lane * 16;
    else if (pattern == 2) lo = (lane >> 2) * 1024 + (lane & 3) * 16;
    else lo = (lane >> 2) * 128 + (lane & 3) * 16;
    const unsigned piece_pitch = pattern == 0 ? 1024u : (pattern == 2 ? 16384u : (pattern == 3 ? 10240u : 2048u));
    unsigned long long acc1 = 0, acc2 = 0;
    unsigned base = (blockIdx.x * 977u % 4096u) * (unsigned)spread;           // workgroups start in different places
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const unsigned off = (base + (unsigned)((wave * N + i) * piece_pitch) + lo) % (bytes - 4096u);
            glds16(srd, lds0 + (unsigned)((wave * N + i) * 1024), off & ~15u);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (it > 0) { acc1 += t1 - t0; acc2 += t2 - t0; }
        base += (unsigned)(nw * N) * piece_pitch + 64u * 1024u * (unsigned)spread;
    }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = acc1 / (iters - 1); out[blockIdx.x * 2 + 1] = acc2 / (iters - 1); }
}
template <int N> void run(const char* buf, unsigned bytes, int pattern, int waves, int wgs, int spread, unsigned long long* dout) {
    const int iters = 9;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<N>, dim3(256 * wgs), dim3(64 * waves), waves * N * 1024, 0, buf, bytes, pattern, iters, dout, spread);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * 256 * wgs);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256 * wgs; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    a /= 256 * wgs; b /= 256 * wgs;
    printf("pattern %d spread %7d  N=%2d pieces/wave  %d waves/WG  %d WG/CU : issue %6.0f cycles (%5.0f per piece)  issue+landed %6.0f\n", pattern, spread, N, waves, wgs, a, a / N, b);
}
int main() {
    const unsigned bytes = 1u << 30;
    char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    unsigned long long* dout; hipMalloc(&dout, 2 * 8 * 4096);
    for (int spread : {0, 262144}) {                   // 0: every iteration re-reads the same few KiB (cache hits); else: streams from HBM
        for (int pattern : {0, 1, 2, 3}) {
            run<1>(buf, bytes, pattern, 1, 1, spread, dout);
            run<4>(buf, bytes, pattern, 1, 1, spread, dout);
            run<8>(buf, bytes, pattern, 1, 1, spread, dout);
            run<8>(buf, bytes, pattern, 4, 1, spread, dout);
            run<8>(buf, bytes, pattern, 4, 2, spread, dout);
            run<12>(buf, bytes, pattern, 4, 1, spread, dout);
        }
    }
    return 0;
}
