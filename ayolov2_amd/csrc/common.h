// Shared helpers for libayolo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ayolo.h"

void ayolo_set_error(const char* fmt, ...);

#define AY_CHECK_ARG(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            ayolo_set_error(__VA_ARGS__);            \
            return AYOLO_EINVAL;                     \
        }                                            \
    } while (0)

#define AY_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ayolo_set_error("%s: %s", name, hipGetErrorString(e__));                \
            return AYOLO_ELAUNCH;                                                   \
        }                                                                           \
    } while (0)

#define AY_CHECK_HIP(expr)                                                          \
    do {                                                                            \
        hipError_t e__ = (expr);                                                    \
        if (e__ != hipSuccess) {                                                    \
            ayolo_set_error("%s: %s", #expr, hipGetErrorString(e__));               \
            return AYOLO_ELAUNCH;                                                   \
        }                                                                           \
    } while (0)

// Sum of the `reps` replicas of a pair of fp64 accumulators, in replica order (bit-identical to the plain loop: the running sums
// are never -0.0, so "+ 0.0" for a replica beyond `reps` changes nothing).  The plain loop `for r: s1 += a[r * rs]` compiles to ONE
// MEMORY ROUND TRIP PER REPLICA (load, s_waitcnt vmcnt(0), add, branch): with eight replicas every workgroup of a BatchNorm pass
// spent ~8 dependent L2 latencies in its prologue before it streamed its first byte -- 19 us for a 39 MB pass that needs 9
// (round 5, profiles/r05_bn_prologue.txt).  Here the loads of eight replicas are in flight together.
__device__ __forceinline__ void rep_sum2(const double* a, size_t rs, size_t off2, int reps, double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int r0 = 0; r0 < reps; r0 += 8) {
        double u[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + k < reps ? r0 + k : reps - 1;      // clamped: every load unconditional
            u[k] = a[(size_t)r * rs];
            v[k] = a[(size_t)r * rs + off2];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s1 += r0 + k < reps ? u[k] : 0.0;
            s2 += r0 + k < reps ? v[k] : 0.0;
        }
    }
}
// an optional per-channel vector (null: `dflt`), loaded without a branch around the load
__device__ __forceinline__ float opt_load(const float* v, const float* any, int i, float dflt) {
    const float t = (v ? v : any)[i];
    return v ? t : dflt;
}

// Scalar-cache prefetch of a uniform parameter block (the kernel-argument segment, or a job struct in global memory).
// hipcc loads the fields of a by-value parameter struct LAZILY, next to their first use: a 400-byte GConvP spans seven 64-byte
// lines, each first touched in another basic block in front of its own `s_waitcnt lgkmcnt(0)` -- seven DEPENDENT scalar-cache
// misses, ~3 000 cycles between a k_gconv workgroup's entry and its tap table (tools/gconv_probe.py), for every workgroup of a
// launch's first round (a launch has a fresh kernarg address).  Touching one dword of every line at entry overlaps the misses;
// the compiler's own loads then hit.  done() goes behind the kernel's first use of a parameter (the compiler waits there anyway).
template <int NB>
struct ScalarTouch {
    static constexpr int NL = (NB + 63) / 64;
    int t[NL];
    template <int I = 0>
    __device__ __forceinline__ void issue(unsigned long long base) {
        if constexpr (I < NL) {
            asm volatile("s_load_dword %0, %1, %2" : "=s"(t[I]) : "s"(base), "n"(I * 64));
            issue<I + 1>(base);
        }
    }
    template <int I = 0>
    __device__ __forceinline__ void keep() {
        if constexpr (I < NL) {
            asm volatile("" ::"s"(t[I]));      // the destination registers stay allocated until the loads have written them
            keep<I + 1>();
        }
    }
    __device__ __forceinline__ void done() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        keep<0>();
    }
};
#define AY_KERNARG_TOUCH_BYTES(name_, NB_)                                                                           \
    ScalarTouch<(NB_)> name_;                                                                                        \
    name_.issue((unsigned long long)(__attribute__((address_space(4))) const char*)__builtin_amdgcn_kernarg_segment_ptr())
#define AY_KERNARG_TOUCH(name_, P_) AY_KERNARG_TOUCH_BYTES(name_, (int)sizeof(P_))

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + expf(-u)); }
