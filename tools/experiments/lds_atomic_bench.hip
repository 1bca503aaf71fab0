// What does the per-tile BatchNorm-sum reduction of the dgrad epilogues cost?  (k_gconv TILE_RED / k_gconv3 / k_dgrad_s2: DPP row sums,
// then 2 x 16 x MI fp64 LDS atomics per wavefront with 4 of 64 lanes active.)  One workgroup of 4 wavefronts per CU, s_memtime
// around (a) the DPP sums alone, (b) DPP sums + the fp64 LDS atomics, (c) DPP sums + plain 16-byte LDS stores of the same values.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic_bench tools/experiments/lds_atomic_bench.hip && /tmp/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ float row16_sum(float v) {
#define AY_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    AY_DPP_ADD(0xB1); AY_DPP_ADD(0x4E); AY_DPP_ADD(0x141); AY_DPP_ADD(0x140);
#undef AY_DPP_ADD
    return v;
}
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(const float* in, float* out, unsigned long long* cyc) {
    __shared__ double sl[512];
    __shared__ float4 sp[4][2 * NV / 4 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 512; i += 256) sl[i] = 0.0;
    float ssum[NV], ssq[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) { ssum[r] = in[(threadIdx.x * NV + r) & 4095]; ssq[r] = ssum[r] * ssum[r]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int lq = lane;
    asm volatile("" : "+v"(lq));
    float keep = 0.0f;
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const float a = row16_sum(ssum[r] + rep), b = row16_sum(ssq[r] + rep);
            if (MODE == 0) keep += a + b;
            if (MODE == 1) {
                if ((lq & 15) == 0) {
                    const int cl = (wave & 1) * NV * 2 + (r >> 2) * 8 + (r & 3) + 4 * (lq >> 5);
                    atomicAdd(&sl[cl], (double)a);
                    atomicAdd(&sl[256 + cl], (double)b);
                }
            }
            if (MODE == 2) { ssum[r] = a; ssq[r] = b; }
        }
        if (MODE == 2) {
            if ((lq & 15) == 0) {
#pragma unroll
                for (int r = 0; r < NV; r += 4) {
                    sp[wave][(r / 4) * 2 + 0] = make_float4(ssum[r], ssum[r + 1], ssum[r + 2], ssum[r + 3]);
                    sp[wave][(r / 4) * 2 + 1] = make_float4(ssq[r], ssq[r + 1], ssq[r + 2], ssq[r + 3]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / 4;
    out[blockIdx.x * 256 + threadIdx.x] = keep + (float)sl[threadIdx.x] + sp[wave][lane & 7].x + ssum[0];
}
template <int MODE, int NV> void run(const char* what, float* in, float* out, unsigned long long* cyc) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(256), 0, 0, in, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-58s NV=%2d: %7.0f cycles per tile epilogue (wave 0 of each workgroup, mean over 256 workgroups)\n", what, NV, s / 256);
}
int main() {
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(in, 0, 4096 * 4);
    run<0, 16>("DPP row sums only", in, out, cyc);           run<0, 32>("DPP row sums only", in, out, cyc);
    run<1, 16>("DPP row sums + fp64 LDS atomics (4 lanes active)", in, out, cyc); run<1, 32>("DPP row sums + fp64 LDS atomics (4 lanes active)", in, out, cyc);
    run<2, 16>("DPP row sums + plain 16-byte LDS stores", in, out, cyc);          run<2, 32>("DPP row sums + plain 16-byte LDS stores", in, out, cyc);
    return 0;
}
