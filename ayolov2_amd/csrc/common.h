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

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + expf(-u)); }
