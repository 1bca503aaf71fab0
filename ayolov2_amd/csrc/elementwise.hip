// HBM-bound NHWC kernels around the conv stack: training-mode BatchNorm + SiLU (forward / backward), SPPF
// max-pool, nearest 2x upsample, input packing, strided slice copy (concat / residual), YOLOHead decode and the
// head-gradient repack.  All activations move as 16-byte channel vectors (8 x fp16 / 4 x fp32 per lane).
//
// Replaces the BatchNorm2d / SiLU / MaxPool2d / Upsample / cat / sigmoid kernels torch launched for kindle's
// Conv, SPPF, UpSample, Concat and YOLOHead modules.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

typedef unsigned int v4u32_t __attribute__((ext_vector_type(4)));
template <typename T> struct VecT;
template <> struct VecT<half_t> { static constexpr int VE = 8; };
template <> struct VecT<float> { static constexpr int VE = 4; };

template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float (&v)[VecT<T>::VE]) {
    uint4 raw = *reinterpret_cast<const uint4*>(p);
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < VecT<T>::VE; ++i) v[i] = (float)e[i];
}
// the same in two halves, for the software-pipelined streaming loops: the raw 16 bytes of the NEXT iteration are requested before
// the current one is computed.  Left to its own devices hipcc sinks every load of an unrolled 4-pixel body to just in front of
// its first use (to save registers): the body then waits for one memory round trip per pixel with two loads in flight, and at
// three wavefronts per SIMD a CU has 24 KB in flight -- ~3 TB/s by Little's law, which is what k_bn_bwd_apply measured
// (profiles/r05_bn_stream_mlp.txt).
template <typename T> __device__ __forceinline__ uint4 load_raw(const T* p) { return *reinterpret_cast<const uint4*>(p); }
template <typename T>
__device__ __forceinline__ void unpack_raw(const uint4& raw, float (&v)[VecT<T>::VE]) {
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < VecT<T>::VE; ++i) v[i] = (float)e[i];
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[VecT<T>::VE]) {
    uint4 raw;
    T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int i = 0; i < VecT<T>::VE; ++i) e[i] = (T)v[i];
    *reinterpret_cast<uint4*>(p) = raw;
}

// Per-channel constants staged in LDS by a prologue and read back as ONE 8- (4-) channel group per thread: the fp16 kernels'
// threads would read 32-byte runs at a 32-byte lane stride (2- to 8-way bank conflicts: 0.68-0.73 of the LDS-active cycles of
// the BatchNorm kernels in the round-2 / round-3 SQ counters).  The two 16-byte halves of a group therefore live in two planes,
// so that consecutive lanes read consecutive 16-byte words: channel c of a C-channel array sits at pl_idx<VE>(c, C).
template <int VE> __device__ __forceinline__ int pl_idx(int c, int C) {
    if constexpr (VE == 8) return ((c >> 2) & 1) * (C >> 1) + (c >> 3) * 4 + (c & 3);
    else return c;
}
// the VE constants of channel group cg from such an array (16-byte LDS reads)
template <int VE> __device__ __forceinline__ void pl_load(const float* arr, int cg, int C, float (&out)[VE]) {
    const float4 lo = *reinterpret_cast<const float4*>(arr + cg * 4);
    out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w;
    if constexpr (VE == 8) {
        const float4 hi = *reinterpret_cast<const float4*>(arr + (C >> 1) + cg * 4);
        out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
    }
}

// SiLU and its derivative.  The fp32 (exact-parity) instantiation uses expf + IEEE division; the fp16-storage
// instantiation uses v_exp_f32 / v_rcp_f32 (error ~1e-6 relative, far below the fp16 output rounding) so these
// HBM-bound passes are not VALU-limited.
template <typename T> __device__ __forceinline__ float sigmoid_t(float u) {
    // fp16 storage: v_exp_f32 + v_rcp_f32 (1 ulp each).  NOT __frcp_rn: without fast-math that is a correctly rounded
    // division (v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup, ~10 VALU ops per element) which made the three
    // BN passes VALU-bound (28 ops per element in k_bn_bwd_reduce at 180 VGPRs).
    if constexpr (sizeof(T) == 2) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
    else return 1.0f / (1.0f + expf(-u));
}
template <typename T> __device__ __forceinline__ float silu_t(float u) { return u * sigmoid_t<T>(u); }
template <typename T> __device__ __forceinline__ float act_grad_t(float u, int act) {
    if (!act) return 1.0f;
    float sg = sigmoid_t<T>(u);
    if constexpr (sizeof(T) == 2) return sg * __builtin_fmaf(u, 1.0f - sg, 1.0f);
    else return sg * (1.0f + u * (1.0f - sg));
}

static unsigned grid_for(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;   // grid-stride beyond 32 blocks/CU
    return (unsigned)b;
}

#define DISPATCH_T(dtype, ...)                                 \
    if ((dtype) == AYOLO_F16) { typedef half_t T; __VA_ARGS__ } \
    else { typedef float T; __VA_ARGS__ }

// ---------------------------------------------------------------------------------------------------
// BN finalize
// ---------------------------------------------------------------------------------------------------
__global__ void k_bn_finalize(const double* stats, int reps, int sld, int C, double count, const float* gamma, const float* beta,
                              float eps, float momentum, float* rmean, float* rvar, float* smean, float* sinv,
                              float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1, s2;                                 // [reps][2][sld] accumulators; this layer's channels start at `stats`
    rep_sum2(stats + c, (size_t)2 * sld, (size_t)sld, reps, s1, s2);
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    if (smean) smean[c] = (float)mean;
    if (sinv) sinv[c] = invstd;
    float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (rmean) rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

extern "C" int ayolo_bn_finalize_ld(const double* stats, int stat_reps, int stat_ld, int C, double count, const float* gamma,
                                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                    float* save_mean, float* save_invstd, float* scale, float* shift, ayolo_stream s) {
    AY_CHECK_ARG(stats && scale && shift && C > 0 && count > 0 && stat_ld >= C, "bn_finalize: bad args");
    hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)s, stats, stat_reps > 0 ? stat_reps : 1, stat_ld, C, count,
                       gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift);
    AY_CHECK_LAUNCH("k_bn_finalize");
    return AYOLO_OK;
}

extern "C" int ayolo_bn_finalize(const double* stats, int stat_reps, int C, double count, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                 float* save_invstd, float* scale, float* shift, ayolo_stream s) {
    return ayolo_bn_finalize_ld(stats, stat_reps, C, C, count, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                                save_invstd, scale, shift, s);
}

__global__ void k_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, const float* cbias,
                                 float eps, int C, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = (gamma ? gamma[c] : 1.0f) / sqrtf(rv[c] + eps);
    float sh = (beta ? beta[c] : 0.0f) - rm[c] * sc;
    if (cbias) sh += cbias[c] * sc;
    scale[c] = sc;
    shift[c] = sh;
}

extern "C" int ayolo_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                    const float* conv_bias, float eps, int C, float* scale, float* shift, ayolo_stream s) {
    AY_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_eval_affine: bad args");
    hipLaunchKernelGGL(k_bn_eval_affine, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)s, gamma, beta, running_mean, running_var,
                       conv_bias, eps, C, scale, shift);
    AY_CHECK_LAUNCH("k_bn_eval_affine");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// a = act(z*scale + shift) (+ residual)
// ---------------------------------------------------------------------------------------------------
// Thread mapping shared by the three BN/SiLU passes: a thread owns ONE 16-byte channel group (its per-channel
// constants live in registers) and walks pixels with a grid stride -- no index divisions, no LDS, coalesced
// 16-byte accesses (a wave covers consecutive channel groups of consecutive pixels).  Activation / residual are
// compile-time (a runtime `act ? :` per element left 50 branches in the unrolled loop body).
template <typename T, int ACT, int RES>
__device__ __forceinline__ void affine_act_rows(const T* z, int ldz, T* a, int lda, long long npix, int cg,
                                                const float (&sc)[VecT<T>::VE], const float (&sh)[VecT<T>::VE],
                                                const T* res, int ldr, long long pix, long long stride) {
    constexpr int VE = VecT<T>::VE;
    // four pixels per iteration, software-pipelined with two register sets (see k_bn_bwd_apply and load_raw): the raw vectors
    // of the next four pixels are in flight while the current four are computed and stored
    constexpr int NPX = 2;                           // pixels per register set
    int opaque_true = 1;
    asm volatile("" : "+s"(opaque_true));
    auto request = [&](uint4 (&zq)[NPX], uint4 (&rq)[RES ? NPX : 1], long long at) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPX; ++j) zq[j] = load_raw<T>(z + (at + j * stride) * ldz + cg * VE);
        if constexpr (RES) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) rq[j] = load_raw<T>(res + (at + j * stride) * ldr + cg * VE);
        }
    };
    auto compute = [&](const uint4 (&zq)[NPX], const uint4 (&rq)[RES ? NPX : 1], long long at) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            float v[VE], r[VE];
            unpack_raw<T>(zq[j], v);
            if constexpr (RES) unpack_raw<T>(rq[j], r);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                float u;
                if constexpr (sizeof(T) == 2) u = __builtin_fmaf(v[i], sc[i], sh[i]);
                else u = v[i] * sc[i] + sh[i];
                if constexpr (ACT) u = silu_t<T>(u);
                if constexpr (RES) u += r[i];
                v[i] = u;
            }
            store_vec<T>(a + (at + j * stride) * lda + cg * VE, v);
        }
    };
    if (pix + (NPX - 1) * stride < npix) {
        uint4 zA[NPX], rA[RES ? NPX : 1], zB[NPX], rB[RES ? NPX : 1];
        long long pa = pix;
        request(zA, rA, pa);
        while (true) {
            const long long pb = pa + NPX * stride;
            const bool hb = pb + (NPX - 1) * stride < npix;
            if (opaque_true) request(zB, rB, hb ? pb : 0);              // (no next group: the tensor's first rows, cache hits for everyone)
            compute(zA, rA, pa);
            if (!hb) { pix = pb; break; }
            pa = pb + NPX * stride;
            const bool ha = pa + (NPX - 1) * stride < npix;
            if (opaque_true) request(zA, rA, ha ? pa : 0);
            compute(zB, rB, pb);
            if (!ha) { pix = pa; break; }
        }
    }
    for (; pix < npix; pix += stride) {
        float v[VE];
        load_vec<T>(z + pix * ldz + cg * VE, v);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            float u;
            if constexpr (sizeof(T) == 2) u = __builtin_fmaf(v[i], sc[i], sh[i]);
            else u = v[i] * sc[i] + sh[i];
            if constexpr (ACT) u = silu_t<T>(u);
            v[i] = u;
        }
        if constexpr (RES) {
            float r[VE];
            load_vec<T>(res + pix * ldr + cg * VE, r);
#pragma unroll
            for (int i = 0; i < VE; ++i) v[i] += r[i];
        }
        store_vec<T>(a + pix * lda + cg * VE, v);
    }
}

template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void k_affine_act(const T* z, int ldz, T* a, int lda, long long npix, int C,
                                                    const float* scale, const float* shift, const T* res, int ldr) {
    constexpr int VE = VecT<T>::VE;
    const int CG = C / VE;
    const int CGT = CG < 256 ? CG : 256;
    const int RPB = 256 / CGT;
    const int cgl = threadIdx.x % CGT, prow = threadIdx.x / CGT;
    if (prow >= RPB) return;
    for (int cg = cgl; cg < CG; cg += CGT) {
        float sc[VE], sh[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            sc[i] = scale ? scale[cg * VE + i] : 1.0f;
            sh[i] = shift ? shift[cg * VE + i] : 0.0f;
        }
        affine_act_rows<T, ACT, RES>(z, ldz, a, lda, npix, cg, sc, sh, res, ldr, (long long)blockIdx.x * RPB + prow,
                                     (long long)gridDim.x * RPB);
    }
}

// launch one of the four (ACT, RES) instantiations
#define DISPATCH_AR(act, res, ...)                                                   \
    if (act) { if (res) { constexpr int ACT = 1, RES = 1; __VA_ARGS__ } else { constexpr int ACT = 1, RES = 0; __VA_ARGS__ } } \
    else { if (res) { constexpr int ACT = 0, RES = 1; __VA_ARGS__ } else { constexpr int ACT = 0, RES = 0; __VA_ARGS__ } }

static unsigned grid_pixels(long long npix, int C, int ve, int min_iters) {
    int cg = C / ve, cgt = cg < 256 ? cg : 256, rpb = 256 / cgt;
    long long b = npix / ((long long)rpb * min_iters);
    if (b < 1) b = 1;
    if (b > 16384) b = 16384;
    return (unsigned)b;
}

// Training-mode BatchNorm + activation in ONE pass over z (k_bn_finalize folded into the prologue): every workgroup
// derives scale/shift of all channels from the replicated sum / sum-of-squares accumulators into LDS (fp64, as
// k_bn_finalize), workgroup 0 also writes save_mean / save_invstd and updates the running statistics.
template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void k_bn_train_act(const T* z, int ldz, T* a, int lda, long long npix, int C,
                                                      const double* stats, int reps, int sld, double count, const float* gamma,
                                                      const float* beta, float eps, float momentum, float* rmean, float* rvar,
                                                      float* smean, float* sinv, const T* res, int ldr) {
    AY_KERNARG_TOUCH_BYTES(kt_, 192);   // the three lines of the argument block at once (common.h)
    constexpr int VE = VecT<T>::VE;
    extern __shared__ float sc_sh[];   // [2][C]
    const int c_first_ = C;            // (first use of an argument: the compiler's wait is here anyway)
    asm volatile("" ::"s"(c_first_));
    kt_.done();
    // stats: [reps][2][sld] accumulators of the producing conv; this layer's channels start at `stats` (sld > C when the
    // conv computed several layers at once -- C3's cv1 | cv2 -- and this is one channel slice of it)
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = opt_load(gamma, reinterpret_cast<const float*>(stats), c, 1.0f), b = opt_load(beta, reinterpret_cast<const float*>(stats), c, 0.0f);
        double s1, s2;
        rep_sum2(stats + c, (size_t)2 * sld, (size_t)sld, reps, s1, s2);   // all loads of the prologue in flight together
        // the SAME expression sequence as k_bn_finalize (the per-module path): with reproducible statistics the two routes
        // then derive bit-identical scale / shift vectors, and a route comparison in fp16 is about the kernels only
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;        // the cancellation-prone step stays in fp64
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = g * invstd;
        sc_sh[pl_idx<VE>(c, C)] = sc;
        sc_sh[C + pl_idx<VE>(c, C)] = b - (float)mean * sc;
        if (blockIdx.x == 0) {
            if (smean) smean[c] = (float)mean;
            if (sinv) sinv[c] = invstd;
            if (rmean) rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)mean;
            if (rvar) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    const int CG = C / VE;
    const int CGT = CG < 256 ? CG : 256;
    const int RPB = 256 / CGT;
    const int cgl = threadIdx.x % CGT, prow = threadIdx.x / CGT;
    if (prow >= RPB) return;
    for (int cg = cgl; cg < CG; cg += CGT) {
        float sc[VE], sh[VE];
        pl_load<VE>(sc_sh, cg, C, sc);
        pl_load<VE>(sc_sh + C, cg, C, sh);
        affine_act_rows<T, ACT, RES>(z, ldz, a, lda, npix, cg, sc, sh, res, ldr, (long long)blockIdx.x * RPB + prow,
                                     (long long)gridDim.x * RPB);
    }
}

extern "C" int ayolo_bn_train_act(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C, const double* stats,
                                  int stat_reps, int stat_ld, double count, const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act,
                                  const void* residual, int ldr, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(z && a && stats && count > 0, "bn_train_act: bad args");
    if (stat_ld <= 0) stat_ld = C;
    AY_CHECK_ARG(stat_ld >= C, "bn_train_act: stat_ld=%d < C=%d", stat_ld, C);
    AY_CHECK_ARG(C > 0 && C % ve == 0 && ldz % ve == 0 && lda % ve == 0 && (!residual || ldr % ve == 0) && C <= 2048,
                 "bn_train_act: C=%d ldz=%d lda=%d", C, ldz, lda);
    if (npix == 0) return AYOLO_OK;
    // >= 16 pixels per thread, at most 4 workgroups per CU: measured best on every YOLOv5s layer (profiles/r03_bn_grid_sweep.txt)
    unsigned grid = grid_pixels(npix, C, ve, 16);
    if (grid > 1024u) grid = 1024u;
    DISPATCH_T(dtype, DISPATCH_AR(act, residual != nullptr,
               hipLaunchKernelGGL((k_bn_train_act<T, ACT, RES>), dim3(grid), dim3(256), 2 * C * sizeof(float), (hipStream_t)s,
                                  (const T*)z, ldz, (T*)a, lda, (long long)npix, C, stats, stat_reps > 0 ? stat_reps : 1,
                                  stat_ld, count, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd,
                                  (const T*)residual, ldr);))
    AY_CHECK_LAUNCH("k_bn_train_act");
    return AYOLO_OK;
}

extern "C" int ayolo_affine_act_res(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C,
                                    const float* scale, const float* shift, int act, const void* residual, int ldr,
                                    ayolo_stream s);
extern "C" int ayolo_affine_act(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C,
                                const float* scale, const float* shift, int act, ayolo_stream s) {
    return ayolo_affine_act_res(dtype, z, ldz, a, lda, npix, C, scale, shift, act, nullptr, 0, s);
}

extern "C" int ayolo_affine_act_res(int dtype, const void* z, int ldz, void* a, int lda, int64_t npix, int C,
                                    const float* scale, const float* shift, int act, const void* residual, int ldr,
                                    ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(residual == nullptr || ldr % ve == 0, "affine_act: ldr=%d", ldr);
    AY_CHECK_ARG(z && a && C > 0 && C % ve == 0 && ldz % ve == 0 && lda % ve == 0, "affine_act: C=%d ldz=%d lda=%d", C, ldz, lda);
    AY_CHECK_ARG(C <= 8192, "affine_act: C too large");
    if (npix == 0) return AYOLO_OK;
    DISPATCH_T(dtype, DISPATCH_AR(act, residual != nullptr,
               hipLaunchKernelGGL((k_affine_act<T, ACT, RES>), dim3(grid_pixels(npix, C, ve, 4)), dim3(256), 0, (hipStream_t)s,
                                  (const T*)z, ldz, (T*)a, lda, (long long)npix, C, scale, shift, (const T*)residual, ldr);))
    AY_CHECK_LAUNCH("k_affine_act");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// backward of a = act(bn(z)):  u = xhat*gamma + beta, xhat = (z-mean)*invstd, du = da * act'(u)
//   pass 1: sums[c] = sum du, sums[C+c] = sum du*xhat
//   pass 2: dz = gamma*invstd * (du - sums[c]/n - xhat*sums[C+c]/n)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float u, int act) {
    if (!act) return 1.0f;
    float sg = 1.0f / (1.0f + expf(-u));
    return sg * (1.0f + u * (1.0f - sg));
}

// Per-channel constants of the backward passes.  fp16 storage folds them so that one element costs two fmas before the
// activation derivative (u = z*A + Bc, xhat = z*invstd + nmi); the fp32 instantiation keeps the reference's operation
// order ((z - mean) * invstd, xhat * gamma + beta) for the 1e-4 parity mode.
template <typename T, int ACT>
__device__ __forceinline__ void bn_bwd_elem(float z, float da, float mu, float is, float ga, float be, float A, float Bc,
                                            float nmi, float& xh, float& du) {
    if constexpr (sizeof(T) == 2) {
        xh = __builtin_fmaf(z, is, nmi);
        du = ACT ? da * act_grad_t<T>(__builtin_fmaf(z, A, Bc), 1) : da;
    } else {
        xh = (z - mu) * is;
        du = da * act_grad_t<T>(xh * ga + be, ACT);
    }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const T* z, int ldz, const T* da, int ldda, long long npix, int C,
                                                       const float* mean, const float* invstd, const float* gamma,
                                                       const float* beta, double* sums, int reps) {
    constexpr int VE = VecT<T>::VE;
    extern __shared__ float bs[];   // [RPB][2*C] per-pixel-row partial sums (<= 16 KiB)
    const int CG = C / VE;
    const int CGT = CG < 256 ? CG : 256;
    const int RPB = 256 / CGT;
    const int cgl = threadIdx.x % CGT, prow = threadIdx.x / CGT;
    if (prow < RPB) {
        for (int cg = cgl; cg < CG; cg += CGT) {
            float mu[VE], is[VE], ga[VE], be[VE], A[VE], Bc[VE], nmi[VE], s1[VE], s2[VE];
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                const int c = cg * VE + i;
                mu[i] = mean[c]; is[i] = invstd[c];
                ga[i] = gamma ? gamma[c] : 1.0f; be[i] = beta ? beta[c] : 0.0f;
                A[i] = is[i] * ga[i]; Bc[i] = be[i] - mu[i] * A[i]; nmi[i] = -mu[i] * is[i];
                s1[i] = 0.0f; s2[i] = 0.0f;
            }
            const long long stride = (long long)gridDim.x * RPB;
            long long pix = (long long)blockIdx.x * RPB + prow;
            // four pixels per iteration: eight independent 16-byte loads in flight per lane
            for (; pix + 3 * stride < npix; pix += 4 * stride) {
                float zz[4][VE], dd[4][VE];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    load_vec<T>(z + (pix + j * stride) * ldz + cg * VE, zz[j]);
                    load_vec<T>(da + (pix + j * stride) * ldda + cg * VE, dd[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < VE; ++i) {
                        float xa, du;
                        bn_bwd_elem<T, ACT>(zz[j][i], dd[j][i], mu[i], is[i], ga[i], be[i], A[i], Bc[i], nmi[i], xa, du);
                        s1[i] += du;
                        if constexpr (sizeof(T) == 2) s2[i] = __builtin_fmaf(du, xa, s2[i]);
                        else s2[i] += du * xa;
                    }
            }
            for (; pix < npix; pix += stride) {
                float z0[VE], d0[VE];
                load_vec<T>(z + pix * ldz + cg * VE, z0);
                load_vec<T>(da + pix * ldda + cg * VE, d0);
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    float xa, du;
                    bn_bwd_elem<T, ACT>(z0[i], d0[i], mu[i], is[i], ga[i], be[i], A[i], Bc[i], nmi[i], xa, du);
                    s1[i] += du;
                    if constexpr (sizeof(T) == 2) s2[i] = __builtin_fmaf(du, xa, s2[i]);
                    else s2[i] += du * xa;
                }
            }
            float* row = bs + (size_t)prow * 2 * C + cg * VE;
#pragma unroll
            for (int i = 0; i < VE; ++i) { row[i] = s1[i]; row[C + i] = s2[i]; }
        }
    }
    __syncthreads();
    // column sums over the RPB pixel rows (plain LDS reads, no LDS atomics), one global atomic per channel sum
    // (fp64 from here on: see the note on BatchNorm accumulators in ayolo.h)
    double* dst = sums + (size_t)(blockIdx.x % (unsigned)reps) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float t = 0.0f;
        for (int r = 0; r < RPB; ++r) t += bs[(size_t)r * 2 * C + i];
        atomicAdd(&dst[i], (double)t);
    }
}

extern "C" int ayolo_bn_act_bwd_reduce(int dtype, const void* z, int ldz, const void* da, int ldda, int64_t npix, int C,
                                       const float* save_mean, const float* save_invstd, const float* gamma,
                                       const float* beta, int act, double* sums, int sum_reps, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    if (sum_reps < 1) sum_reps = 1;
    AY_CHECK_ARG(z && da && sums && save_mean && save_invstd, "bn_bwd_reduce: null pointer");
    AY_CHECK_ARG(C > 0 && C % ve == 0 && ldz % ve == 0 && ldda % ve == 0 && C <= 2048, "bn_bwd_reduce: C=%d", C);
    if (npix == 0) return AYOLO_OK;
    // >= 16 pixels per thread, at most 4 workgroups per CU (measured best): the per-workgroup tail (2*C atomics) stays small
    unsigned grid = grid_pixels(npix, C, ve, 16);
    if (grid > 1024u) grid = 1024u;
    const int cg_ = C / ve, cgt_ = cg_ < 256 ? cg_ : 256, rpb_ = 256 / cgt_;
    DISPATCH_T(dtype, DISPATCH_AR(act, false,
               (void)RES; hipLaunchKernelGGL((k_bn_bwd_reduce<T, ACT>), dim3(grid), dim3(256), (size_t)rpb_ * 2 * C * sizeof(float),
                                  (hipStream_t)s, (const T*)z, ldz, (const T*)da, ldda, (long long)npix, C, save_mean, save_invstd,
                                  gamma, beta, sums, sum_reps);))
    AY_CHECK_LAUNCH("k_bn_bwd_reduce");
    return AYOLO_OK;
}

// RESOUT: the block's output also fed a shortcut add (Bottleneck: out = x + act(bn(z))), so d(x) (+)= da.  The pass reads da
// anyway: it forwards it to the shortcut's gradient buffer `dres` instead of a separate strided copy re-reading it.
// B32: every tensor of the pass is < 2 GiB (host check), so the streaming loads / stores go through buffer descriptors with one
// 32-bit byte offset per access instead of a 64-bit pointer each: a dozen address registers and their 64-bit arithmetic less --
// 136 -> 1xx registers per lane for the fp16 SiLU instantiation, i.e. four wavefronts per SIMD instead of three.
template <typename T, int ACT, bool RESOUT = false, bool B32 = false>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T* z, int ldz, const T* da, int ldda, T* dz, int lddz,
                                                      long long npix, int C, const float* mean, const float* invstd,
                                                      const float* gamma, const float* beta, const double* sums,
                                                      int reps, float* dgamma, float* dbeta, float grad_scale,
                                                      T* dres = nullptr, int lddres = 0, int res_acc = 0) {
    AY_KERNARG_TOUCH_BYTES(kt_, 192);   // the three lines of the argument block at once (common.h)
    constexpr int VE = VecT<T>::VE;
    extern __shared__ float sh[];   // [6][C]: mean, invstd, gamma, beta, sum_du/n, sum_dux/n
    const float invn = 1.0f / (float)npix;
    kt_.done();
    // cooperative prologue: replica sums once per workgroup, then every thread keeps ITS channel group in registers
    for (int i = threadIdx.x; i < C; i += 256) {
        const int q = pl_idx<VE>(i, C);
        // every load of the prologue issued before the first use: one memory round trip, not twelve (rep_sum2, common.h)
        const float mu_ = mean[i], is_ = invstd[i], ga_ = opt_load(gamma, mean, i, 1.0f), be_ = opt_load(beta, mean, i, 0.0f);
        double d1, d2;
        rep_sum2(sums + i, (size_t)2 * C, (size_t)C, reps, d1, d2);
        sh[q] = mu_; sh[C + q] = is_;
        sh[2 * C + q] = ga_; sh[3 * C + q] = be_;
        const float s1 = (float)d1, s2 = (float)d2;
        sh[4 * C + q] = s1 * invn; sh[5 * C + q] = s2 * invn;
        if (blockIdx.x == 0) {
            if (dbeta) dbeta[i] = s1 * grad_scale;
            if (dgamma) dgamma[i] = s2 * grad_scale;
        }
    }
    __syncthreads();
    const int CG = C / VE;
    const int CGT = CG < 256 ? CG : 256;
    const int RPB = 256 / CGT;
    const int cgl = threadIdx.x % CGT, prow = threadIdx.x / CGT;
    if (prow >= RPB) return;
    const unsigned ldzb = (unsigned)ldz * sizeof(T), lddab = (unsigned)ldda * sizeof(T), lddzb = (unsigned)lddz * sizeof(T),
                   lddresb = (unsigned)lddres * sizeof(T);
    const unsigned npu = (unsigned)npix;
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(z), 0, B32 ? (int)(npu * ldzb) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDa = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(da), 0, B32 ? (int)(npu * lddab) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDz = __builtin_amdgcn_make_buffer_rsrc(dz, 0, B32 ? (int)(npu * lddzb) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsRes = __builtin_amdgcn_make_buffer_rsrc(dres, 0, (B32 && RESOUT) ? (int)(npu * lddresb) : 0, 0x00020000);
    for (int cg = cgl; cg < CG; cg += CGT) {
        // fp16 storage: dz = du*P + (xhat*Rr + Q) with P = gamma*invstd, Q = -P*m1, Rr = -P*m2 (two fmas)
        float mu[VE], is[VE], ga[VE], be[VE], m1[VE], m2[VE], A[VE], Bc[VE], nmi[VE], P[VE], Q[VE], Rr[VE];
        pl_load<VE>(sh, cg, C, mu); pl_load<VE>(sh + C, cg, C, is); pl_load<VE>(sh + 2 * C, cg, C, ga);
        pl_load<VE>(sh + 3 * C, cg, C, be); pl_load<VE>(sh + 4 * C, cg, C, m1); pl_load<VE>(sh + 5 * C, cg, C, m2);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            A[i] = is[i] * ga[i]; Bc[i] = be[i] - mu[i] * A[i]; nmi[i] = -mu[i] * is[i];
            P[i] = ga[i] * is[i]; Q[i] = -P[i] * m1[i]; Rr[i] = -P[i] * m2[i];
        }
        typedef typename std::conditional<B32, unsigned, long long>::type pix_t;
        const pix_t stride = (pix_t)gridDim.x * RPB, npx = (pix_t)npix;
        pix_t pix = (pix_t)blockIdx.x * RPB + prow;
        // accessors: pixel row -> this thread's 16-byte group of z / da / dres / dz
        const unsigned cgb = (unsigned)cg * 16u;
        auto ld_z = [&](pix_t px) __attribute__((always_inline)) -> uint4 {
            if constexpr (B32) return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, px * ldzb + cgb, 0, 0));
            else return load_raw<T>(z + px * ldz + cg * VE);
        };
        auto ld_da = [&](pix_t px) __attribute__((always_inline)) -> uint4 {
            if constexpr (B32) return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsDa, px * lddab + cgb, 0, 0));
            else return load_raw<T>(da + px * ldda + cg * VE);
        };
        auto ld_res = [&](pix_t px) __attribute__((always_inline)) -> uint4 {
            if constexpr (B32) return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsRes, px * lddresb + cgb, 0, 0));
            else return load_raw<T>(dres + px * lddres + cg * VE);
        };
        auto st16 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, T* ptr, const float (&v)[VE]) __attribute__((always_inline)) {
            if constexpr (B32) {
                uint4 raw;
                T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
                for (int i = 0; i < VE; ++i) e[i] = (T)v[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32_t, raw), rs, off, 0, 0);
            } else store_vec<T>(ptr, v);
        };
        // four pixels per iteration, software-pipelined (see load_raw): the z / da (/ dres) vectors of the next four pixels are
        // requested before the current four are computed.  Two register sets take turns (A, B: no copies on the back edge --
        // with one "current" and one "next" set the copies land right behind the loads, with a wait for them), and each request
        // group sits in a basic block of its own behind an always-true branch the compiler cannot see through, so that there
        // is nothing in the block to sink the loads behind.
        constexpr int NPX = 2;                       // pixels per register set
        int opaque_true = 1;
        asm volatile("" : "+s"(opaque_true));
        auto request = [&](uint4 (&zq)[NPX], uint4 (&dq)[NPX], uint4 (&rq)[RESOUT ? NPX : 1], pix_t at) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                zq[j] = ld_z(at + j * stride);
                dq[j] = ld_da(at + j * stride);
            }
            if constexpr (RESOUT) {
                if (res_acc) {
#pragma unroll
                    for (int j = 0; j < NPX; ++j) rq[j] = ld_res(at + j * stride);
                }
            }
        };
        auto compute = [&](const uint4 (&zq)[NPX], const uint4 (&dq)[NPX], const uint4 (&rq)[RESOUT ? NPX : 1], pix_t at) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                float zz[VE], dd[VE];
                unpack_raw<T>(zq[j], zz);
                unpack_raw<T>(dq[j], dd);
                if constexpr (RESOUT) {
                    float rv[VE];
                    if (res_acc) unpack_raw<T>(rq[j], rv);
#pragma unroll
                    for (int i = 0; i < VE; ++i) rv[i] = res_acc ? rv[i] + dd[i] : dd[i];
                    st16(rsRes, (unsigned)(at + j * stride) * lddresb + cgb, RESOUT ? dres + (long long)(at + j * stride) * lddres + cg * VE : nullptr, rv);
                }
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    float xh, du;
                    bn_bwd_elem<T, ACT>(zz[i], dd[i], mu[i], is[i], ga[i], be[i], A[i], Bc[i], nmi[i], xh, du);
                    if constexpr (sizeof(T) == 2) dd[i] = __builtin_fmaf(du, P[i], __builtin_fmaf(xh, Rr[i], Q[i]));
                    else dd[i] = ga[i] * is[i] * (du - m1[i] - xh * m2[i]);
                }
                st16(rsDz, (unsigned)(at + j * stride) * lddzb + cgb, dz + (long long)(at + j * stride) * lddz + cg * VE, dd);
            }
        };
        if (pix + (NPX - 1) * stride < npx) {
            uint4 zA[NPX], dA[NPX], rA[RESOUT ? NPX : 1], zB[NPX], dB[NPX], rB[RESOUT ? NPX : 1];
            pix_t pa = pix;
            request(zA, dA, rA, pa);
            while (true) {
                const pix_t pb = pa + NPX * stride;
                const bool hb = pb + (NPX - 1) * stride < npx;
                if (opaque_true) request(zB, dB, rB, hb ? pb : 0);      // (no next group: the tensor's first rows, cache hits for everyone)
                compute(zA, dA, rA, pa);
                if (!hb) { pix = pb; break; }
                pa = pb + NPX * stride;
                const bool ha = pa + (NPX - 1) * stride < npx;
                if (opaque_true) request(zA, dA, rA, ha ? pa : 0);
                compute(zB, dB, rB, pb);
                if (!ha) { pix = pa; break; }
            }
        }
        for (; pix < npx; pix += stride) {
            float zv[VE], dv[VE];
            unpack_raw<T>(ld_z(pix), zv);
            unpack_raw<T>(ld_da(pix), dv);
            if constexpr (RESOUT) {
                float rv[VE];
                if (res_acc) unpack_raw<T>(ld_res(pix), rv);
#pragma unroll
                for (int i = 0; i < VE; ++i) rv[i] = res_acc ? rv[i] + dv[i] : dv[i];
                st16(rsRes, (unsigned)pix * lddresb + cgb, dres + (long long)pix * lddres + cg * VE, rv);
            }
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                float xh, du;
                bn_bwd_elem<T, ACT>(zv[i], dv[i], mu[i], is[i], ga[i], be[i], A[i], Bc[i], nmi[i], xh, du);
                if constexpr (sizeof(T) == 2) dv[i] = __builtin_fmaf(du, P[i], __builtin_fmaf(xh, Rr[i], Q[i]));
                else dv[i] = ga[i] * is[i] * (du - m1[i] - xh * m2[i]);
            }
            st16(rsDz, (unsigned)pix * lddzb + cgb, dz + (long long)pix * lddz + cg * VE, dv);
        }
    }
}

extern "C" int ayolo_bn_act_bwd_apply_res(int dtype, const void* z, int ldz, const void* da, int ldda, void* dz, int lddz,
                                          int64_t npix, int C, const float* save_mean, const float* save_invstd,
                                          const float* gamma, const float* beta, int act, const double* sums, int sum_reps,
                                          float* dgamma, float* dbeta, float grad_scale, void* dres, int lddres,
                                          int res_accumulate, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    if (sum_reps < 1) sum_reps = 1;
    AY_CHECK_ARG(z && da && dz && sums, "bn_bwd_apply: null pointer");
    AY_CHECK_ARG(C > 0 && C % ve == 0 && ldz % ve == 0 && ldda % ve == 0 && lddz % ve == 0 && C <= 2048, "bn_bwd_apply: C=%d", C);
    AY_CHECK_ARG(!dres || (lddres >= C && lddres % ve == 0 && dres != da && dres != dz), "bn_bwd_apply: shortcut gradient lddres=%d", lddres);
    if (npix == 0) return AYOLO_OK;
    // the per-workgroup prologue stages (6 + 2*reps)*C floats in LDS: scale the elements per workgroup with C
    unsigned grid = grid_pixels(npix, C, ve, C >= 256 ? 16 : 8);
    if (grid > 2048u) grid = 2048u;
    // every tensor below 2 GiB: 32-bit buffer offsets (B32)
    const int64_t es = dtype == AYOLO_F16 ? 2 : 4, lim = (int64_t)1 << 31;
    const bool b32 = npix * ldz * es < lim && npix * ldda * es < lim && npix * lddz * es < lim && (!dres || npix * lddres * es < lim);
#define BWD_APPLY_LAUNCH(RO_, B32_, dres_, lddres_, racc_)                                                                           \
    DISPATCH_T(dtype, DISPATCH_AR(act, false,                                                                                        \
               (void)RES; hipLaunchKernelGGL((k_bn_bwd_apply<T, ACT, RO_, B32_>), dim3(grid), dim3(256), 6 * C * sizeof(float), (hipStream_t)s, \
                                  (const T*)z, ldz, (const T*)da, ldda, (T*)dz, lddz, (long long)npix, C, save_mean,                  \
                                  save_invstd, gamma, beta, sums, sum_reps, dgamma, dbeta, grad_scale, (T*)(dres_), lddres_, racc_);))
    if (dres) {
        if (b32) { BWD_APPLY_LAUNCH(true, true, dres, lddres, res_accumulate) } else { BWD_APPLY_LAUNCH(true, false, dres, lddres, res_accumulate) }
    } else {
        if (b32) { BWD_APPLY_LAUNCH(false, true, nullptr, 0, 0) } else { BWD_APPLY_LAUNCH(false, false, nullptr, 0, 0) }
    }
#undef BWD_APPLY_LAUNCH
    AY_CHECK_LAUNCH("k_bn_bwd_apply");
    return AYOLO_OK;
}

extern "C" int ayolo_bn_act_bwd_apply(int dtype, const void* z, int ldz, const void* da, int ldda, void* dz, int lddz,
                                      int64_t npix, int C, const float* save_mean, const float* save_invstd,
                                      const float* gamma, const float* beta, int act, const double* sums, int sum_reps,
                                      float* dgamma, float* dbeta, float grad_scale, ayolo_stream s) {
    return ayolo_bn_act_bwd_apply_res(dtype, z, ldz, da, ldda, dz, lddz, npix, C, save_mean, save_invstd, gamma, beta, act, sums,
                                      sum_reps, dgamma, dbeta, grad_scale, nullptr, 0, 0, s);
}

// ---------------------------------------------------------------------------------------------------
// k_bn_bwd_apply2 (round 6): the apply pass of TWO Conv-BN-act blocks whose pre-activations lie side by side
// in one buffer -- C3's cv1 | cv2 run as one conv (res/configs/model/yolov5s.yaml:23-52), so z and dz of the two blocks are the
// channel slices [0, C0) and [C0, C0 + C1) of shared rows -- in ONE launch over whole rows.  As two launches each pass read and
// wrote HALF rows of a double-pitch buffer (YOLOv5s at 160 x 160: 64-byte runs of a 128-byte pitch, 2.3-3.6 TB/s against 5 for the
// contiguous sibling) and every pair paid two prologues.  Each block keeps its own output gradient buffer (da: the Bottleneck
// chain's gradient for cv1, a slice of the concat gradient for cv2), saved statistics, affine parameters and sums; the arithmetic
// per element is k_bn_bwd_apply's, so dz / dgamma / dbeta equal the two-launch route bit for bit.
// ---------------------------------------------------------------------------------------------------
struct BnApply2P {
    ayolo_bn_apply_seg g[2];
};

template <typename T, int ACT, bool B32>
__global__ __launch_bounds__(256) void k_bn_bwd_apply2(const T* z, int ldz, T* dz, int lddz, long long npix, BnApply2P q, int reps,
                                                       float grad_scale) {
    AY_KERNARG_TOUCH_BYTES(kt_, 192);
    constexpr int VE = VecT<T>::VE;
    extern __shared__ float sh[];   // [6][Ct]: mean, invstd, gamma, beta, sum_du/n, sum_dux/n
    const float invn = 1.0f / (float)npix;
    kt_.done();
    const int C0 = q.g[0].C, Ct = q.g[0].C + q.g[1].C;
    for (int i = threadIdx.x; i < Ct; i += 256) {
        const int sg = i >= C0 ? 1 : 0;
        const ayolo_bn_apply_seg& g = q.g[sg];
        const int il = i - (sg ? C0 : 0), Cs = g.C;
        const int qi = pl_idx<VE>(i, Ct);
        const float mu_ = g.save_mean[il], is_ = g.save_invstd[il], ga_ = opt_load(g.gamma, g.save_mean, il, 1.0f),
                    be_ = opt_load(g.beta, g.save_mean, il, 0.0f);
        double d1, d2;
        rep_sum2(g.sums + il, (size_t)2 * Cs, (size_t)Cs, reps, d1, d2);
        sh[qi] = mu_; sh[Ct + qi] = is_;
        sh[2 * Ct + qi] = ga_; sh[3 * Ct + qi] = be_;
        const float s1 = (float)d1, s2 = (float)d2;
        sh[4 * Ct + qi] = s1 * invn; sh[5 * Ct + qi] = s2 * invn;
        if (blockIdx.x == 0) {
            if (g.dbeta) g.dbeta[il] = s1 * grad_scale;
            if (g.dgamma) g.dgamma[il] = s2 * grad_scale;
        }
    }
    __syncthreads();
    const int CG = Ct / VE;
    const int CGT = CG < 256 ? CG : 256;
    const int RPB = 256 / CGT;
    const int cgl = threadIdx.x % CGT, prow = threadIdx.x / CGT;
    if (prow >= RPB) return;
    const unsigned ldzb = (unsigned)ldz * sizeof(T), lddzb = (unsigned)lddz * sizeof(T);
    const unsigned npu = (unsigned)npix;
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(z), 0, B32 ? (int)(npu * ldzb) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDz = __builtin_amdgcn_make_buffer_rsrc(dz, 0, B32 ? (int)(npu * lddzb) : 0, 0x00020000);
    for (int cg = cgl; cg < CG; cg += CGT) {
        float mu[VE], is[VE], ga[VE], be[VE], m1[VE], m2[VE], A[VE], Bc[VE], nmi[VE], P[VE], Q[VE], Rr[VE];
        pl_load<VE>(sh, cg, Ct, mu); pl_load<VE>(sh + Ct, cg, Ct, is); pl_load<VE>(sh + 2 * Ct, cg, Ct, ga);
        pl_load<VE>(sh + 3 * Ct, cg, Ct, be); pl_load<VE>(sh + 4 * Ct, cg, Ct, m1); pl_load<VE>(sh + 5 * Ct, cg, Ct, m2);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            A[i] = is[i] * ga[i]; Bc[i] = be[i] - mu[i] * A[i]; nmi[i] = -mu[i] * is[i];
            P[i] = ga[i] * is[i]; Q[i] = -P[i] * m1[i]; Rr[i] = -P[i] * m2[i];
        }
        // this thread's 16-byte group of da: its block's own buffer (per-lane base pointer and row stride)
        const int sg = cg * VE >= C0 ? 1 : 0;
        const T* dab = static_cast<const T*>(q.g[sg].da) + (cg * VE - (sg ? C0 : 0));
        const long long ldd = q.g[sg].ldda;
        typedef typename std::conditional<B32, unsigned, long long>::type pix_t;
        const pix_t stride = (pix_t)gridDim.x * RPB, npx = (pix_t)npix;
        pix_t pix = (pix_t)blockIdx.x * RPB + prow;
        const unsigned cgb = (unsigned)cg * 16u;
        auto ld_z = [&](pix_t px) __attribute__((always_inline)) -> uint4 {
            if constexpr (B32) return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, px * ldzb + cgb, 0, 0));
            else return load_raw<T>(z + px * ldz + cg * VE);
        };
        auto ld_da = [&](pix_t px) __attribute__((always_inline)) -> uint4 { return load_raw<T>(dab + (long long)px * ldd); };
        auto st_dz = [&](pix_t px, const float (&v)[VE]) __attribute__((always_inline)) {
            if constexpr (B32) {
                uint4 raw;
                T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
                for (int i = 0; i < VE; ++i) e[i] = (T)v[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32_t, raw), rsDz, (unsigned)px * lddzb + cgb, 0, 0);
            } else store_vec<T>(dz + (long long)px * lddz + cg * VE, v);
        };
        constexpr int NPX = 2;
        int opaque_true = 1;
        asm volatile("" : "+s"(opaque_true));
        auto request = [&](uint4 (&zq)[NPX], uint4 (&dq)[NPX], pix_t at) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                zq[j] = ld_z(at + j * stride);
                dq[j] = ld_da(at + j * stride);
            }
        };
        auto compute = [&](const uint4 (&zq)[NPX], const uint4 (&dq)[NPX], pix_t at) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                float zz[VE], dd[VE];
                unpack_raw<T>(zq[j], zz);
                unpack_raw<T>(dq[j], dd);
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    float xh, du;
                    bn_bwd_elem<T, ACT>(zz[i], dd[i], mu[i], is[i], ga[i], be[i], A[i], Bc[i], nmi[i], xh, du);
                    if constexpr (sizeof(T) == 2) dd[i] = __builtin_fmaf(du, P[i], __builtin_fmaf(xh, Rr[i], Q[i]));
                    else dd[i] = ga[i] * is[i] * (du - m1[i] - xh * m2[i]);
                }
                st_dz(at + j * stride, dd);
            }
        };
        if (pix + (NPX - 1) * stride < npx) {
            uint4 zA[NPX], dA[NPX], zB[NPX], dB[NPX];
            pix_t pa = pix;
            request(zA, dA, pa);
            while (true) {
                const pix_t pb = pa + NPX * stride;
                const bool hb = pb + (NPX - 1) * stride < npx;
                if (opaque_true) request(zB, dB, hb ? pb : 0);
                compute(zA, dA, pa);
                if (!hb) { pix = pb; break; }
                pa = pb + NPX * stride;
                const bool ha = pa + (NPX - 1) * stride < npx;
                if (opaque_true) request(zA, dA, ha ? pa : 0);
                compute(zB, dB, pb);
                if (!ha) { pix = pa; break; }
            }
        }
        for (; pix < npx; pix += stride) {
            float zv[VE], dv[VE];
            unpack_raw<T>(ld_z(pix), zv);
            unpack_raw<T>(ld_da(pix), dv);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                float xh, du;
                bn_bwd_elem<T, ACT>(zv[i], dv[i], mu[i], is[i], ga[i], be[i], A[i], Bc[i], nmi[i], xh, du);
                if constexpr (sizeof(T) == 2) dv[i] = __builtin_fmaf(du, P[i], __builtin_fmaf(xh, Rr[i], Q[i]));
                else dv[i] = ga[i] * is[i] * (du - m1[i] - xh * m2[i]);
            }
            st_dz(pix, dv);
        }
    }
}

extern "C" int ayolo_bn_act_bwd_apply2(int dtype, const void* z, int ldz, void* dz, int lddz, int64_t npix, const ayolo_bn_apply_seg* seg0,
                                       const ayolo_bn_apply_seg* seg1, int act, int sum_reps, float grad_scale, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    if (sum_reps < 1) sum_reps = 1;
    AY_CHECK_ARG(z && dz && seg0 && seg1, "bn_bwd_apply2: null pointer");
    BnApply2P q;
    q.g[0] = *seg0; q.g[1] = *seg1;
    int Ct = 0;
    for (int k = 0; k < 2; ++k) {
        const ayolo_bn_apply_seg& g = q.g[k];
        AY_CHECK_ARG(g.da && g.save_mean && g.save_invstd && g.sums, "bn_bwd_apply2: null pointer in segment %d", k);
        AY_CHECK_ARG(g.C > 0 && g.C % ve == 0 && g.ldda % ve == 0 && g.ldda >= g.C, "bn_bwd_apply2: segment %d: C=%d ldda=%d", k, g.C, g.ldda);
        Ct += g.C;
    }
    AY_CHECK_ARG(ldz % ve == 0 && lddz % ve == 0 && ldz >= Ct && lddz >= Ct && Ct <= 2048, "bn_bwd_apply2: ldz=%d lddz=%d for %d channels", ldz, lddz, Ct);
    if (npix == 0) return AYOLO_OK;
    unsigned grid = grid_pixels(npix, Ct, ve, Ct >= 256 ? 16 : 8);
    if (grid > 2048u) grid = 2048u;
    const int64_t es = dtype == AYOLO_F16 ? 2 : 4, lim = (int64_t)1 << 31;
    const bool b32 = npix * ldz * es < lim && npix * lddz * es < lim;
#define BWD_APPLY2_LAUNCH(B32_)                                                                                                      \
    DISPATCH_T(dtype, DISPATCH_AR(act, false,                                                                                        \
               (void)RES; hipLaunchKernelGGL((k_bn_bwd_apply2<T, ACT, B32_>), dim3(grid), dim3(256), 6 * Ct * sizeof(float), (hipStream_t)s, \
                                  (const T*)z, ldz, (T*)dz, lddz, (long long)npix, q, sum_reps, grad_scale);))
    if (b32) { BWD_APPLY2_LAUNCH(true) } else { BWD_APPLY2_LAUNCH(false) }
#undef BWD_APPLY2_LAUNCH
    AY_CHECK_LAUNCH("k_bn_bwd_apply2");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// max-pool k x k, stride 1, pad k/2 (SPPF).  Forward records the window position of the first maximum
// (row-major scan, `val > max || isnan(val)` as torch) so backward is a 25-tap gather without atomics.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const T* x, int ldx, T* y, int ldy, unsigned char* idx, int B, int H,
                                                     int W, int C, int k) {
    constexpr int VE = VecT<T>::VE;
    const int CG = C / VE, pad = k / 2;
    const long long total = (long long)B * H * W * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int cg = (int)(t % CG);
        long long pix = t / CG;
        int w = (int)(pix % W);
        long long r = pix / W;
        int h = (int)(r % H);
        long long n = r / H;
        float best[VE];
        int bi[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) { best[i] = -INFINITY; bi[i] = 0; }
        for (int dy = 0; dy < k; ++dy) {
            int hh = h + dy - pad;
            if (hh < 0 || hh >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                int ww = w + dx - pad;
                if (ww < 0 || ww >= W) continue;
                float v[VE];
                load_vec<T>(x + ((n * H + hh) * W + ww) * ldx + cg * VE, v);
#pragma unroll
                for (int i = 0; i < VE; ++i)
                    if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; bi[i] = dy * k + dx; }
            }
        }
        store_vec<T>(y + pix * ldy + cg * VE, best);
        if (idx) {
#pragma unroll
            for (int i = 0; i < VE; ++i) idx[pix * C + cg * VE + i] = (unsigned char)bi[i];
        }
    }
}

// The SPPF window (5 x 5) in strips: a thread owns PSW adjacent output pixels of one row and one 16-byte channel group, loads
// each of the PSW + 4 input columns of a window row ONCE (the per-pixel kernel above loads every input 25 times: 20 M 16-byte
// loads for the 13 MB tensor of YOLOv5s at batch 64, 40 us at 0.8 TB/s) and scans the taps of every output in the same row-major
// order, so the recorded first maximum is the same.  The PSW x VE window positions leave as ONE 8- (4-) byte store per pixel.
// PSW = 2: 108 / 112 registers per lane (four wavefronts per SIMD); with four pixels per thread the kernels held 179 / 156 and ran
// two per SIMD, five serial window rows each: forward 41.8 -> 32.2 us, backward 32.7 -> 31.0 us (profiles/r05_pool_strip_width.txt)
#define PSW 2
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool5_fwd(const T* x, int ldx, T* y, int ldy, unsigned char* idx, int B, int H, int W, int C) {
    constexpr int VE = VecT<T>::VE, K = 5, PAD = 2, NC = PSW + K - 1;
    const int CG = C / VE, NS = (W + PSW - 1) / PSW;
    const long long total = (long long)B * H * NS * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int cg = (int)(t % CG);
        long long r = t / CG;
        const int w0 = (int)(r % NS) * PSW;
        r /= NS;
        const int h = (int)(r % H);
        const long long n = r / H;
        float best[PSW][VE];
        int bi[PSW][VE];
#pragma unroll
        for (int j = 0; j < PSW; ++j)
#pragma unroll
            for (int i = 0; i < VE; ++i) { best[j][i] = -INFINITY; bi[j][i] = 0; }
        // BRANCH-FREE on purpose: hipcc turns a conditional load into a branch + s_waitcnt vmcnt(0), i.e. one memory round trip per
        // tap (what bounds the per-pixel kernel).  Out-of-image taps load a clamped address and become -inf, which never beats the
        // running best (strict >) -- the same as skipping the tap; the NC loads of a window row are in flight together.
#pragma unroll 1
        for (int dy = 0; dy < K; ++dy) {
            const int hh = h + dy - PAD;
            const bool rok = hh >= 0 && hh < H;
            const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh);
            float v[NC][VE];
            const T* row = x + ((n * H + hc) * W) * (long long)ldx + cg * VE;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ww = w0 + c - PAD;
                const int wc = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
                load_vec<T>(row + (long long)wc * ldx, v[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ww = w0 + c - PAD;
                const bool ok = rok && ww >= 0 && ww < W;
#pragma unroll
                for (int i = 0; i < VE; ++i) v[c][i] = ok ? v[c][i] : -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < PSW; ++j)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
#pragma unroll
                    for (int i = 0; i < VE; ++i) {
                        const float val = v[j + dx][i];
                        const bool take = val > best[j][i] || val != val;
                        best[j][i] = take ? val : best[j][i];
                        bi[j][i] = take ? dy * K + dx : bi[j][i];
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < PSW; ++j) {
            if (w0 + j >= W) break;
            const long long pix = (n * H + h) * W + w0 + j;
            store_vec<T>(y + pix * ldy + cg * VE, best[j]);
            if (idx) {
                unsigned wd[VE / 4];
#pragma unroll
                for (int q = 0; q < VE / 4; ++q)
                    wd[q] = (unsigned)bi[j][4 * q] | ((unsigned)bi[j][4 * q + 1] << 8) | ((unsigned)bi[j][4 * q + 2] << 16) | ((unsigned)bi[j][4 * q + 3] << 24);
                unsigned* dst = reinterpret_cast<unsigned*>(idx + pix * C + cg * VE);       // C % VE == 0: VE-byte aligned
                if constexpr (VE == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(wd[0], wd[1]);
                else dst[0] = wd[0];
            }
        }
    }
}

extern "C" int ayolo_maxpool_fwd(int dtype, const void* x, int ldx, void* y, int ldy, unsigned char* argmax, int B, int H,
                                 int W, int C, int k, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(x && y && C % ve == 0 && ldx % ve == 0 && ldy % ve == 0 && k > 0 && k <= 15 && (k & 1), "maxpool_fwd: bad args");
    if (k == 5 && (argmax == nullptr || ((uintptr_t)argmax % 8) == 0)) {
        const long long strips = (long long)B * H * ((W + PSW - 1) / PSW) * (C / ve);
        DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool5_fwd<T>, dim3(grid_for(strips, 256)), dim3(256), 0, (hipStream_t)s,
                                             (const T*)x, ldx, (T*)y, ldy, argmax, B, H, W, C);)
        AY_CHECK_LAUNCH("k_maxpool5_fwd");
        return AYOLO_OK;
    }
    long long total = (long long)B * H * W * (C / ve);
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool_fwd<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s,
                                         (const T*)x, ldx, (T*)y, ldy, argmax, B, H, W, C, k);)
    AY_CHECK_LAUNCH("k_maxpool_fwd");
    return AYOLO_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void k_maxpool_bwd(const unsigned char* idx, const T* dy, int lddy, T* dx, int lddx, int B,
                                                     int H, int W, int C, int k, int accumulate) {
    constexpr int VE = VecT<T>::VE;
    const int CG = C / VE, pad = k / 2;
    const long long total = (long long)B * H * W * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int cg = (int)(t % CG);
        long long pix = t / CG;
        int w = (int)(pix % W);
        long long r = pix / W;
        int h = (int)(r % H);
        long long n = r / H;
        float g[VE];
        if (accumulate) load_vec<T>(dx + pix * lddx + cg * VE, g);
        else {
#pragma unroll
            for (int i = 0; i < VE; ++i) g[i] = 0.0f;
        }
        // output (oh, ow) looks at input (oh + dyy - pad, ow + dxx - pad): this input is tap (dyy,dxx) of
        // output (h - dyy + pad, w - dxx + pad)
        for (int dyy = 0; dyy < k; ++dyy) {
            int oh = h - dyy + pad;
            if (oh < 0 || oh >= H) continue;
            for (int dxx = 0; dxx < k; ++dxx) {
                int ow = w - dxx + pad;
                if (ow < 0 || ow >= W) continue;
                long long op = (n * H + oh) * W + ow;
                const unsigned char* ip = idx + op * C + cg * VE;
                float dv[VE];
                load_vec<T>(dy + op * lddy + cg * VE, dv);
                const int me = dyy * k + dxx;
#pragma unroll
                for (int i = 0; i < VE; ++i)
                    if (ip[i] == me) g[i] += dv[i];
            }
        }
        store_vec<T>(dx + pix * lddx + cg * VE, g);
    }
}

// backward of the 5 x 5 window in the same strips: the PSW input pixels of a thread are taps of the outputs in columns
// w0 - 2 .. w0 + PSW + 1 of five rows; every output's gradient vector and its VE window positions (one 8- / 4-byte word) are
// loaded once and matched against the PSW pixels (the per-pixel kernel: 25 x (VE byte loads + 1 vector load) per input pixel).
// Taps are visited in the per-pixel kernel's order (dyy ascending, dxx ascending), so the sums round identically.
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool5_bwd(const unsigned char* idx, const T* dy, int lddy, T* dx, int lddx, int B, int H, int W,
                                                      int C, int accumulate) {
    constexpr int VE = VecT<T>::VE, K = 5, PAD = 2, NC = PSW + K - 1;
    const int CG = C / VE, NS = (W + PSW - 1) / PSW;
    const long long total = (long long)B * H * NS * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const int cg = (int)(t % CG);
        long long r = t / CG;
        const int w0 = (int)(r % NS) * PSW;
        r /= NS;
        const int h = (int)(r % H);
        const long long n = r / H;
        float g[PSW][VE];
#pragma unroll
        for (int j = 0; j < PSW; ++j) {
            if (accumulate && w0 + j < W) load_vec<T>(dx + ((n * H + h) * W + w0 + j) * (long long)lddx + cg * VE, g[j]);
            else {
#pragma unroll
                for (int i = 0; i < VE; ++i) g[j][i] = 0.0f;
            }
        }
#pragma unroll 1
        for (int dyy = 0; dyy < K; ++dyy) {
            const int oh = h - dyy + PAD;
            const bool rok = oh >= 0 && oh < H;
            const int ohc = oh < 0 ? 0 : (oh >= H ? H - 1 : oh);
            const long long orow = (n * H + ohc) * W;
            // branch-free (see k_maxpool5_fwd): an output outside the image is loaded from a clamped address and its window
            // positions are replaced by 0xff, which matches no tap
            float dvs[NC][VE];
            unsigned wds[NC][VE / 4];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ow = w0 + c - PAD;
                const int owc = ow < 0 ? 0 : (ow >= W ? W - 1 : ow);
                load_vec<T>(dy + (orow + owc) * lddy + cg * VE, dvs[c]);
                const unsigned* ip = reinterpret_cast<const unsigned*>(idx + (orow + owc) * C + cg * VE);
                if constexpr (VE == 8) { const uint2 u = *reinterpret_cast<const uint2*>(ip); wds[c][0] = u.x; wds[c][1] = u.y; }
                else wds[c][0] = ip[0];
            }
            // input pixel w0 + j is tap dxx of output column w0 + j - dxx + PAD: dxx ascending = output column descending
#pragma unroll
            for (int c = NC - 1; c >= 0; --c) {
                const int ow = w0 + c - PAD;
                const bool ok = rok && ow >= 0 && ow < W;
                float dv[VE];
                unsigned wd[VE / 4];
#pragma unroll
                for (int i = 0; i < VE; ++i) dv[i] = dvs[c][i];
#pragma unroll
                for (int q = 0; q < VE / 4; ++q) wd[q] = ok ? wds[c][q] : 0xffffffffu;
#pragma unroll
                for (int j = 0; j < PSW; ++j) {
                    const int dxx = j - c + 2 * PAD;           // (w0 + j) - ow + PAD
                    if (dxx < 0 || dxx >= K) continue;
                    const unsigned me = (unsigned)(dyy * K + dxx);
#pragma unroll
                    for (int i = 0; i < VE; ++i)
                        if (((wd[i >> 2] >> (8 * (i & 3))) & 0xffu) == me) g[j][i] += dv[i];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PSW; ++j)
            if (w0 + j < W) store_vec<T>(dx + ((n * H + h) * W + w0 + j) * (long long)lddx + cg * VE, g[j]);
    }
}

extern "C" int ayolo_maxpool_bwd(int dtype, const unsigned char* argmax, const void* dy, int lddy, void* dx, int lddx, int B,
                                 int H, int W, int C, int k, int accumulate, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(argmax && dy && dx && C % ve == 0 && lddy % ve == 0 && lddx % ve == 0, "maxpool_bwd: bad args");
    if (k == 5 && ((uintptr_t)argmax % 8) == 0) {
        const long long strips = (long long)B * H * ((W + PSW - 1) / PSW) * (C / ve);
        DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool5_bwd<T>, dim3(grid_for(strips, 256)), dim3(256), 0, (hipStream_t)s, argmax,
                                             (const T*)dy, lddy, (T*)dx, lddx, B, H, W, C, accumulate);)
        AY_CHECK_LAUNCH("k_maxpool5_bwd");
        return AYOLO_OK;
    }
    long long total = (long long)B * H * W * (C / ve);
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_maxpool_bwd<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, argmax,
                                         (const T*)dy, lddy, (T*)dx, lddx, B, H, W, C, k, accumulate);)
    AY_CHECK_LAUNCH("k_maxpool_bwd");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// SPPF's pool cascade in ONE launch per direction (round 6; kindle SPPF = cv1 -> three chained
// MaxPool2d(5, 1, 2) -> concat [x, y1, y2, y3] -> cv2, res/configs/model/yolov5s.yaml:33).  As three k_maxpool5_fwd + three
// k_maxpool5_bwd launches the cascade cost 0.21 ms of the YOLOv5s step for 0.2 GB: every launch is compare-select-bound (25 taps
// x (compare + two selects + NaN test) per output channel) and the chain is serial.  Here a workgroup owns one image and NCG
// 16-byte channel groups of the 4 C-channel concat buffer, keeps the map in LDS and runs all three pools on it:
//   forward: every fp16 value becomes a 32-bit KEY = [sortable 16-bit value | 63 - h | 63 - w] with (h, w) the element's own map
//     position, so that "first maximum in row-major scan order" is ONE unsigned max over the window -- among equal values the
//     smaller row, then the smaller column wins, the scan's tie rule -- without any per-tap arithmetic, and the 5 x 5 window is
//     separable (two v_max3_u32 per channel and direction instead of 25 compare-selects).  torch's NaN rule (a NaN always replaces
//     the running maximum: the LAST NaN of the scan is recorded) is the same max with the position fields not complemented
//     (value 0xffff | h | w).  -0.0 is stored as +0.0 (float compare treats them as equal: the first of them wins either way; the
//     output value then reads +0.0 where the scan would copy -0.0 -- equal under ==).  The winner's key with the position of the
//     OUTPUT element is the next pool's input; values and window positions are written exactly as k_maxpool5_fwd would (a window
//     with no value above -inf records its first in-image tap, torch's rule; the scan kernels record tap 0).  The LDS maps carry
//     two zero columns / rows of padding (key 0 never wins), so no tap tests a bound.
//   backward: g3 = d3; g2 = d2 + scatter(g3; arg3); g1 = d1 + scatter(g2; arg2); dx = d0 + scatter(g1; arg1), every g rounded to
//     fp16 as the three launches store it.  The scatter is an fp64 LDS atomic per (output, channel): a sum of <= 26 fp16 values is
//     EXACT in fp64 (40 bits of exponent range + 11 of mantissa + 5 of count < 53), so the result does not depend on the order
//     of the atomics -- deterministic -- and equals the sequential fp32 sums of k_maxpool5_bwd whenever those are exact too
//     (always, unless a window's gradients span more than 2^13 in magnitude).  Only dx (slice 0) is written.
// ---------------------------------------------------------------------------------------------------
// key = [sortable value : 16 | row field : 6 | column field : 6].  Fields of a non-NaN value: 63 - h, 63 - w (the first maximum in
// row-major order has the largest fields); of a NaN (value 0xffff): h, w (the last NaN wins).
__device__ __forceinline__ unsigned sppf_key(unsigned b, unsigned pos_first, unsigned pos_last) {            // b: fp16 bits in [15:0]
    const unsigned mag = b & 0x7fffu;
    b = mag == 0u ? 0u : b;                                           // -0.0 -> +0.0
    const unsigned s = (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
    return mag > 0x7c00u ? (0xffff000u | pos_last) : ((s << 12) | pos_first);
}
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
    const unsigned m = a > b ? a : b;
    return m > c ? m : c;
}

// MAXI: items (pixel, channel group) per thread, H * W * NCG <= 256 * MAXI.  LDS: `cur` [2][H][W + 4][NCG] uint4 (two zero columns
// on either side of a row) and `rowk` [2][H + 4][W][NCG] uint4 (two zero rows above and below): the window never tests a bound.
template <int NCG, int MAXI, int NT>
__global__ __launch_bounds__(NT) void k_sppf_fwd(half_t* cat, int ld, unsigned char* arg, long long plane, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sppf_lds[];
    const int HW = H * W, NIT = HW * NCG, WP = W + 4;
    const int NCUR = H * WP * NCG, NROW = (H + 4) * W * NCG;
    uint4* cur = reinterpret_cast<uint4*>(sppf_lds);
    uint4* rowk = cur + 2 * NCUR;
    const int nblk = C / (8 * NCG);
    const int n = blockIdx.x / nblk, cg0 = (blockIdx.x % nblk) * NCG;
    const long long pix0 = (long long)n * HW;
    // this thread's items: every load of the kernel is requested before anything waits (one memory round trip)
    uint4 raw[MAXI];
    int hh[MAXI], ww[MAXI], gg[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int it = threadIdx.x + NT * i, itc = it < NIT ? it : NIT - 1;
        const int pix = itc / NCG;
        gg[i] = itc % NCG; hh[i] = pix / W; ww[i] = pix % W;
        raw[i] = *reinterpret_cast<const uint4*>(cat + (pix0 + pix) * ld + (cg0 + gg[i]) * 8);
    }
    for (int k = threadIdx.x; k < 2 * (NCUR + NROW); k += NT) cur[k] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        if (threadIdx.x + NT * i < NIT) {
            const unsigned pf = (unsigned)(((63 - hh[i]) << 6) | (63 - ww[i])), pl = (unsigned)((hh[i] << 6) | ww[i]);
            const unsigned w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
            unsigned k[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { k[2 * q] = sppf_key(w[q] & 0xffffu, pf, pl); k[2 * q + 1] = sppf_key(w[q] >> 16, pf, pl); }
            const int at = (hh[i] * WP + ww[i] + 2) * NCG + gg[i];
            cur[at] = make_uint4(k[0], k[1], k[2], k[3]);
            cur[NCUR + at] = make_uint4(k[4], k[5], k[6], k[7]);
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 1; j <= 3; ++j) {
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {                               // row pass: max of cur[h][w - 2 .. w + 2]
            if (threadIdx.x + NT * i < NIT) {
                const int at = (hh[i] * WP + ww[i] + 2) * NCG + gg[i];
                uint4 a[5], b[5];
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) { a[dx] = cur[at + (dx - 2) * NCG]; b[dx] = cur[NCUR + at + (dx - 2) * NCG]; }
                uint4 lo, hi;
                lo.x = umax3(umax3(a[0].x, a[1].x, a[2].x), a[3].x, a[4].x); lo.y = umax3(umax3(a[0].y, a[1].y, a[2].y), a[3].y, a[4].y);
                lo.z = umax3(umax3(a[0].z, a[1].z, a[2].z), a[3].z, a[4].z); lo.w = umax3(umax3(a[0].w, a[1].w, a[2].w), a[3].w, a[4].w);
                hi.x = umax3(umax3(b[0].x, b[1].x, b[2].x), b[3].x, b[4].x); hi.y = umax3(umax3(b[0].y, b[1].y, b[2].y), b[3].y, b[4].y);
                hi.z = umax3(umax3(b[0].z, b[1].z, b[2].z), b[3].z, b[4].z); hi.w = umax3(umax3(b[0].w, b[1].w, b[2].w), b[3].w, b[4].w);
                const int ro = ((hh[i] + 2) * W + ww[i]) * NCG + gg[i];
                rowk[ro] = lo;
                rowk[NROW + ro] = hi;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {                               // column pass: max of rowk[h - 2 .. h + 2][w]
            if (threadIdx.x + NT * i < NIT) {
                const int ro = ((hh[i] + 2) * W + ww[i]) * NCG + gg[i], st = W * NCG;
                uint4 a[5], b[5];
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) { a[dy] = rowk[ro + (dy - 2) * st]; b[dy] = rowk[NROW + ro + (dy - 2) * st]; }
                unsigned m[8];
                m[0] = umax3(umax3(a[0].x, a[1].x, a[2].x), a[3].x, a[4].x); m[1] = umax3(umax3(a[0].y, a[1].y, a[2].y), a[3].y, a[4].y);
                m[2] = umax3(umax3(a[0].z, a[1].z, a[2].z), a[3].z, a[4].z); m[3] = umax3(umax3(a[0].w, a[1].w, a[2].w), a[3].w, a[4].w);
                m[4] = umax3(umax3(b[0].x, b[1].x, b[2].x), b[3].x, b[4].x); m[5] = umax3(umax3(b[0].y, b[1].y, b[2].y), b[3].y, b[4].y);
                m[6] = umax3(umax3(b[0].z, b[1].z, b[2].z), b[3].z, b[4].z); m[7] = umax3(umax3(b[0].w, b[1].w, b[2].w), b[3].w, b[4].w);
                // winner (hw, ww') = (63 - rb, 63 - cb): tap = (hw - h + 2) * 5 + (ww' - w + 2) = c0 - (5 rb + cb)
                const unsigned pf = (unsigned)(((63 - hh[i]) << 6) | (63 - ww[i])), pl = (unsigned)((hh[i] << 6) | ww[i]);
                const unsigned c0 = (unsigned)(5 * (65 - hh[i]) + (65 - ww[i]));
                unsigned val[8], pos[8], nk[8];
                bool anynan = false;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned sk = m[c] >> 12, rb = (m[c] >> 6) & 63u, cb = m[c] & 63u;
                    pos[c] = c0 - (5u * rb + cb);
                    val[c] = sk ^ ((sk & 0x8000u) ? 0x8000u : 0xffffu);
                    nk[c] = (m[c] & 0xffff000u) | pf;
                    anynan |= sk == 0xffffu;
                }
                if (__builtin_amdgcn_ballot_w64(anynan) != 0ull) {     // NaN: fields hold (h, w) of the last one
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const unsigned sk = m[c] >> 12, rb = (m[c] >> 6) & 63u, cb = m[c] & 63u;
                        if (sk == 0xffffu) {
                            pos[c] = 5u * (rb + 2u - (unsigned)hh[i]) + (cb + 2u - (unsigned)ww[i]);
                            val[c] = 0x7e00u;
                            nk[c] = 0xffff000u | pl;
                        }
                    }
                }
                const int at = (hh[i] * WP + ww[i] + 2) * NCG + gg[i];
                cur[at] = make_uint4(nk[0], nk[1], nk[2], nk[3]);
                cur[NCUR + at] = make_uint4(nk[4], nk[5], nk[6], nk[7]);
                const long long pix = pix0 + hh[i] * W + ww[i];
                *reinterpret_cast<uint4*>(cat + pix * ld + (long long)j * C + (cg0 + gg[i]) * 8) =
                    make_uint4(val[0] | (val[1] << 16), val[2] | (val[3] << 16), val[4] | (val[5] << 16), val[6] | (val[7] << 16));
                if (arg)
                    *reinterpret_cast<uint2*>(arg + (j - 1) * plane + pix * C + (cg0 + gg[i]) * 8) =
                        make_uint2(pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24), pos[4] | (pos[5] << 8) | (pos[6] << 16) | (pos[7] << 24));
            }
        }
        __syncthreads();
    }
}

// channel groups per workgroup (0: the map does not fit): LDS = 32 bytes x NCG x (H (W + 4) + (H + 4) W) forward, 64 x NCG x H W backward.
// FOUR groups = 64-byte runs per pixel row of the concat buffer: with two (32-byte runs, four workgroups sharing every 128-byte line)
// the kernels ran at a third of the speed the instruction count allows (profiles/r06_sppf_geometry.txt).
static size_t sppf_lds_fwd(int H, int W, int ncg) { return (size_t)32 * ncg * ((size_t)H * (W + 4) + (size_t)(H + 4) * W); }
static int sppf_ncg(int H, int W, int C) {
    if (H < 1 || W < 1 || H > 64 || W > 64) return 0;                  // 6-bit position fields
    const long long hw = (long long)H * W;
    if (C % 32 == 0 && hw * 4 <= 2048 && sppf_lds_fwd(H, W, 4) <= 150 * 1024) return 4;
    if (C % 16 == 0 && hw * 2 <= 1024 && sppf_lds_fwd(H, W, 2) <= 64 * 1024) return 2;
    if (hw <= 2048 && sppf_lds_fwd(H, W, 1) <= 150 * 1024) return 1;
    return 0;
}

// (the function attribute is set once per device and instantiation, to the largest size asked for so far)
template <int NCG, int MAXI, int NT>
static void sppf_launch_fwd(unsigned grid, size_t lds, hipStream_t s, half_t* cat, int ld, unsigned char* argmax, long long plane, int H, int W, int C) {
    static size_t attr_set[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || attr_set[dev] < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sppf_fwd<NCG, MAXI, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = lds;
    }
    hipLaunchKernelGGL((k_sppf_fwd<NCG, MAXI, NT>), dim3(grid), dim3(NT), lds, s, cat, ld, argmax, plane, H, W, C);
}

extern "C" int ayolo_sppf_pool_fwd(int dtype, void* cat, int ld, unsigned char* argmax, int B, int H, int W, int C, ayolo_stream s) {
    AY_CHECK_ARG(dtype == AYOLO_F16, "sppf_pool_fwd: fp16 only (fp32 plans run three ayolo_maxpool_fwd launches)");
    AY_CHECK_ARG(cat && C % 8 == 0 && ld % 8 == 0 && ld >= 4 * C && ((uintptr_t)cat % 16) == 0, "sppf_pool_fwd: bad args");
    AY_CHECK_ARG(argmax == nullptr || ((uintptr_t)argmax % 8) == 0, "sppf_pool_fwd: argmax must be 8-byte aligned");
    const int ncg = sppf_ncg(H, W, C);
    AY_CHECK_ARG(ncg > 0, "sppf_pool_fwd: a %d x %d map does not fit a workgroup's LDS (use ayolo_maxpool_fwd)", H, W);
    if (B == 0) return AYOLO_OK;
    const size_t lds = sppf_lds_fwd(H, W, ncg);
    const long long plane = (long long)B * H * W * C;
    const unsigned grid = (unsigned)(B * (C / (8 * ncg)));
    const int nit = H * W * ncg;
    if (ncg == 4) sppf_launch_fwd<4, 4, 512>(grid, lds, (hipStream_t)s, (half_t*)cat, ld, argmax, plane, H, W, C);
    else if (ncg == 2) sppf_launch_fwd<2, 4, 256>(grid, lds, (hipStream_t)s, (half_t*)cat, ld, argmax, plane, H, W, C);
    else if (nit <= 1024) sppf_launch_fwd<1, 4, 256>(grid, lds, (hipStream_t)s, (half_t*)cat, ld, argmax, plane, H, W, C);
    else sppf_launch_fwd<1, 8, 256>(grid, lds, (hipStream_t)s, (half_t*)cat, ld, argmax, plane, H, W, C);
    AY_CHECK_LAUNCH("k_sppf_fwd");
    return AYOLO_OK;
}

// MAXI items per thread; every global load of the kernel -- the four gradient slices and the three position planes of the thread's
// items -- is requested up front (as three dependent stages of guarded loads the kernel spent ~20 serial memory round trips)
template <int NCG, int MAXI, int NT>
__global__ __launch_bounds__(NT) void k_sppf_bwd(const unsigned char* arg, long long plane, half_t* dcat, int ld, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sppf_lds[];
    const int HW = H * W, NIT = HW * NCG;
    double* acc = reinterpret_cast<double*>(sppf_lds);                 // [8][NIT]: channel c of item it at acc[c * NIT + it]
    const int nblk = C / (8 * NCG);
    const int n = blockIdx.x / nblk, cg0 = (blockIdx.x % nblk) * NCG;
    const long long pix0 = (long long)n * HW;
    uint4 dq[MAXI][4];                                                 // gradient slices 0..3 of this thread's items
    uint2 pq[MAXI][3];                                                 // window positions of pools 1..3
    int hh[MAXI], ww[MAXI], gg[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int it = threadIdx.x + NT * i, itc = it < NIT ? it : NIT - 1;
        const int pix = itc / NCG;
        gg[i] = itc % NCG; hh[i] = pix / W; ww[i] = pix % W;
        const half_t* at = dcat + (pix0 + pix) * ld + (cg0 + gg[i]) * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) dq[i][k] = *reinterpret_cast<const uint4*>(at + (long long)k * C);
#pragma unroll
        for (int k = 0; k < 3; ++k) pq[i][k] = *reinterpret_cast<const uint2*>(arg + k * plane + (pix0 + pix) * C + (cg0 + gg[i]) * 8);
    }
    for (int k = threadIdx.x; k < 8 * NIT; k += NT) acc[k] = 0.0;
    __syncthreads();
    uint4 gq[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) gq[i] = dq[i][3];
#pragma unroll
    for (int j = 3; j >= 1; --j) {
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {                               // scatter g_j through arg_j
            const int it = threadIdx.x + NT * i;
            if (it < NIT) {
                const uint2 pw = pq[i][j - 1];
                const half_t* gv = reinterpret_cast<const half_t*>(&gq[i]);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned tap = ((c < 4 ? pw.x : pw.y) >> (8 * (c & 3))) & 0xffu;
                    const int dy = (int)((tap * 13u) >> 6), dx = (int)tap - 5 * dy;        // tap / 5, tap % 5 for tap < 25
                    const int h2 = hh[i] + dy - 2, w2 = ww[i] + dx - 2;
                    if (tap < 25u && h2 >= 0 && h2 < H && w2 >= 0 && w2 < W)
                        atomicAdd(&acc[c * NIT + (h2 * W + w2) * NCG + gg[i]], (double)gv[c]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {                               // g_{j-1} = d_{j-1} + what arrived, rounded to fp16
            const int it = threadIdx.x + NT * i;
            if (it < NIT) {
                const half_t* dv = reinterpret_cast<const half_t*>(&dq[i][j - 1]);
                uint4 out;
                half_t* ov = reinterpret_cast<half_t*>(&out);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ov[c] = (half_t)(float)((double)dv[c] + acc[c * NIT + it]);
                    acc[c * NIT + it] = 0.0;
                }
                gq[i] = out;
                if (j == 1) *reinterpret_cast<uint4*>(dcat + (pix0 + hh[i] * W + ww[i]) * ld + (cg0 + gg[i]) * 8) = out;
            }
        }
        if (j > 1) __syncthreads();
    }
}

template <int NCG, int MAXI, int NT>
static void sppf_launch_bwd(unsigned grid, size_t lds, hipStream_t s, const unsigned char* argmax, long long plane, half_t* dcat, int ld, int H, int W, int C) {
    static size_t attr_set[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || attr_set[dev] < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sppf_bwd<NCG, MAXI, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = lds;
    }
    hipLaunchKernelGGL((k_sppf_bwd<NCG, MAXI, NT>), dim3(grid), dim3(NT), lds, s, argmax, plane, dcat, ld, H, W, C);
}

extern "C" int ayolo_sppf_pool_bwd(int dtype, const unsigned char* argmax, void* dcat, int ld, int B, int H, int W, int C, ayolo_stream s) {
    AY_CHECK_ARG(dtype == AYOLO_F16, "sppf_pool_bwd: fp16 only (fp32 plans run three ayolo_maxpool_bwd launches)");
    AY_CHECK_ARG(argmax && dcat && C % 8 == 0 && ld % 8 == 0 && ld >= 4 * C && ((uintptr_t)dcat % 16) == 0 && ((uintptr_t)argmax % 8) == 0,
                 "sppf_pool_bwd: bad args");
    const int ncg = sppf_ncg(H, W, C);
    AY_CHECK_ARG(ncg > 0, "sppf_pool_bwd: a %d x %d map does not fit a workgroup's LDS (use ayolo_maxpool_bwd)", H, W);
    if (B == 0) return AYOLO_OK;
    const size_t lds = (size_t)H * W * ncg * 64;
    const long long plane = (long long)B * H * W * C;
    const unsigned grid = (unsigned)(B * (C / (8 * ncg)));
    const int nit = H * W * ncg;
    if (ncg == 4) sppf_launch_bwd<4, 4, 512>(grid, lds, (hipStream_t)s, argmax, plane, (half_t*)dcat, ld, H, W, C);
    else if (ncg == 2) sppf_launch_bwd<2, 4, 256>(grid, lds, (hipStream_t)s, argmax, plane, (half_t*)dcat, ld, H, W, C);
    else if (nit <= 1024) sppf_launch_bwd<1, 4, 256>(grid, lds, (hipStream_t)s, argmax, plane, (half_t*)dcat, ld, H, W, C);
    else sppf_launch_bwd<1, 8, 256>(grid, lds, (hipStream_t)s, argmax, plane, (half_t*)dcat, ld, H, W, C);
    AY_CHECK_LAUNCH("k_sppf_bwd");
    return AYOLO_OK;
}

/* > 0 when ayolo_sppf_pool_fwd / _bwd accept this map (it fits a workgroup's LDS): the channel groups a workgroup takes -- 4 / 2 / 1,
 * i.e. 64- / 32- / 16-byte runs per pixel row; with 1 the cascade is slower than it could be (profiles/r06_sppf_geometry.txt) */
extern "C" int ayolo_sppf_pool_supported(int dtype, int H, int W, int C) {
    return dtype == AYOLO_F16 && C % 8 == 0 ? sppf_ncg(H, W, C) : 0;
}

// ---------------------------------------------------------------------------------------------------
// nearest 2x upsample
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_upsample_fwd(const T* x, int ldx, T* y, int ldy, int B, int H, int W, int C) {
    constexpr int VE = VecT<T>::VE;
    const int CG = C / VE, OW = 2 * W, OH = 2 * H;
    const long long total = (long long)B * OH * OW * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int cg = (int)(t % CG);
        long long pix = t / CG;
        int ow = (int)(pix % OW);
        long long r = pix / OW;
        int oh = (int)(r % OH);
        long long n = r / OH;
        uint4 v = *reinterpret_cast<const uint4*>(x + ((n * H + (oh >> 1)) * W + (ow >> 1)) * ldx + cg * VE);
        *reinterpret_cast<uint4*>(y + pix * ldy + cg * VE) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_upsample_bwd(const T* dy, int lddy, T* dx, int lddx, int B, int H, int W, int C,
                                                      int accumulate) {
    constexpr int VE = VecT<T>::VE;
    const int CG = C / VE, OW = 2 * W;
    const long long total = (long long)B * H * W * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int cg = (int)(t % CG);
        long long pix = t / CG;
        int w = (int)(pix % W);
        long long r = pix / W;
        int h = (int)(r % H);
        long long n = r / H;
        float g[VE];
        if (accumulate) load_vec<T>(dx + pix * lddx + cg * VE, g);
        else {
#pragma unroll
            for (int i = 0; i < VE; ++i) g[i] = 0.0f;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[VE];
                load_vec<T>(dy + ((n * 2 * H + 2 * h + a) * OW + 2 * w + b) * lddy + cg * VE, v);
#pragma unroll
                for (int i = 0; i < VE; ++i) g[i] += v[i];
            }
        store_vec<T>(dx + pix * lddx + cg * VE, g);
    }
}

extern "C" int ayolo_upsample2x_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C,
                                    ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(x && y && C % ve == 0 && ldx % ve == 0 && ldy % ve == 0, "upsample_fwd: bad args");
    long long total = (long long)B * 4 * H * W * (C / ve);
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_upsample_fwd<T>, dim3(grid_for(total, 256 * 2)), dim3(256), 0, (hipStream_t)s,
                                         (const T*)x, ldx, (T*)y, ldy, B, H, W, C);)
    AY_CHECK_LAUNCH("k_upsample_fwd");
    return AYOLO_OK;
}

extern "C" int ayolo_upsample2x_bwd(int dtype, const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C,
                                    int accumulate, ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(dy && dx && C % ve == 0 && lddy % ve == 0 && lddx % ve == 0, "upsample_bwd: bad args");
    long long total = (long long)B * H * W * (C / ve);
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_upsample_bwd<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s,
                                         (const T*)dy, lddy, (T*)dx, lddx, B, H, W, C, accumulate);)
    AY_CHECK_LAUNCH("k_upsample_bwd");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// input packing: NCHW fp32 -> NHWC (Cpad channels, zero filled)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_pack_input(const float* x, int B, int C, int H, int W, T* y, int Cpad) {
    const long long hw = (long long)H * W, total = (long long)B * hw;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        long long n = t / hw, p = t - n * hw;
        T* o = y + t * Cpad;
        for (int c = 0; c < Cpad; ++c) o[c] = c < C ? (T)x[(n * C + c) * hw + p] : (T)0.0f;
    }
}

extern "C" int ayolo_pack_input(const float* x, int B, int C, int H, int W, int dtype, void* y, int Cpad, ayolo_stream s) {
    AY_CHECK_ARG(x && y && Cpad >= C && Cpad <= 16, "pack_input: bad args");
    long long total = (long long)B * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_pack_input<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, x, B, C,
                                         H, W, (T*)y, Cpad);)
    AY_CHECK_LAUNCH("k_pack_input");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// strided slice copy / add
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_copy2d(const T* x, int ldx, T* y, int ldy, long long npix, int C, int accumulate) {
    constexpr int VE = VecT<T>::VE;
    const int CG = C / VE;
    const long long total = npix * CG;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        long long pix = t / CG;
        int cg = (int)(t - pix * CG);
        if (accumulate) {
            float a[VE], b[VE];
            load_vec<T>(x + pix * ldx + cg * VE, a);
            load_vec<T>(y + pix * ldy + cg * VE, b);
#pragma unroll
            for (int i = 0; i < VE; ++i) b[i] += a[i];
            store_vec<T>(y + pix * ldy + cg * VE, b);
        } else {
            *reinterpret_cast<uint4*>(y + pix * ldy + cg * VE) = *reinterpret_cast<const uint4*>(x + pix * ldx + cg * VE);
        }
    }
}

extern "C" int ayolo_copy2d(int dtype, const void* x, int ldx, void* y, int ldy, int64_t npix, int C, int accumulate,
                            ayolo_stream s) {
    const int ve = dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(x && y && C % ve == 0 && ldx % ve == 0 && ldy % ve == 0, "copy2d: bad args");
    if (npix == 0) return AYOLO_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_copy2d<T>, dim3(grid_for(npix * (C / ve), 256 * 4)), dim3(256), 0,
                                         (hipStream_t)s, (const T*)x, ldx, (T*)y, ldy, (long long)npix, C, accumulate);)
    AY_CHECK_LAUNCH("k_copy2d");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// YOLOHead: eval decode and gradient repack
// ---------------------------------------------------------------------------------------------------
// AUG: the decode of one augmented forward of test-time augmentation (ayolo_head_decode_aug) -- the augmentation's inverse is
// applied to the value on its way out (xywh / scale as a true division, then the flip about the ORIGINAL image extent) and a
// row is stored only if its destination index lies inside [win_lo, win_hi): the tail clip is a window, not a copy
template <bool AUG>
__global__ __launch_bounds__(256) void k_head_decode(const float* raw, long long sb, long long sa, long long sy, long long sx,
                                                     int B, int na, int ny, int nx, int no,
                                                     const float* anchors_px, float stride, float* out,
                                                     long long rows_total, long long row_off,
                                                     float aug_scale, int aug_flip, float aug_extent, long long win_lo, long long win_hi) {
    const long long per_img = (long long)na * ny * nx;
    const long long total = (long long)B * per_img * no;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        int o = (int)(t % no);
        long long cell = t / no;                     // ((b*na + a)*ny + y)*nx + x
        int x = (int)(cell % nx);
        long long r = cell / nx;
        int y = (int)(r % ny);
        r /= ny;
        int a = (int)(r % na);
        long long b = r / na;
        long long row = row_off + ((long long)a * ny + y) * nx + x;
        if constexpr (AUG) {
            if (row < win_lo || row >= win_hi) continue;
        }
        float sg = 1.0f / (1.0f + expf(-raw[b * sb + a * sa + y * sy + x * sx + o]));
        float v = sg;
        if (o == 0) v = (sg * 2.0f - 0.5f + (float)x) * stride;
        else if (o == 1) v = (sg * 2.0f - 0.5f + (float)y) * stride;
        else if (o == 2 || o == 3) { float q = sg * 2.0f; v = q * q * anchors_px[a * 2 + (o - 2)]; }
        if constexpr (AUG) {
            if (o < 4) v = v / aug_scale;
            if (o == 0 && aug_flip == 3) v = aug_extent - v;
            if (o == 1 && aug_flip == 2) v = aug_extent - v;
        }
        out[(b * rows_total + row) * no + o] = v;
    }
}

// The same decode with one (image, anchor) plane per blockIdx.y and 32-bit index arithmetic inside the plane: the element kernel
// above spends five 64-bit divisions on every output (o, x, y, a, b from the flat index: ~300 VALU instructions, 115 us per level
// at 1.6 TB/s for YOLOv5x at 1280^2).  Here o and x come from two multiply-high divisions by per-launch magic numbers
// (exact while plane elements * no < 2^32 and plane pixels * nx < 2^32: the host checks and falls back), the values from the
// SAME float expressions.
template <bool AUG>
__global__ __launch_bounds__(256) void k_head_decode_plane(const float* raw, long long sb, long long sa, long long sy, long long sx,
                                                           int na, int ny, int nx, int no, unsigned m_no, unsigned m_nx,
                                                           const float* anchors_px, float stride, float* out,
                                                           long long rows_total, long long row_off,
                                                           float aug_scale, int aug_flip, float aug_extent, long long win_lo, long long win_hi) {
    const unsigned plane = blockIdx.y;
    const unsigned b = plane / (unsigned)na, a = plane - b * (unsigned)na;
    const unsigned plane_n = (unsigned)ny * (unsigned)nx * (unsigned)no;
    const float* rp = raw + (long long)b * sb + (long long)a * sa;
    const long long row0 = row_off + (long long)a * ny * nx;
    float* op = out + (long long)b * rows_total * no;
    const float aw = anchors_px[a * 2], ah = anchors_px[a * 2 + 1];
    for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < plane_n; t += gridDim.x * 256) {
        const unsigned pix = __umulhi(t, m_no), o = t - pix * (unsigned)no;
        const unsigned y = __umulhi(pix, m_nx), x = pix - y * (unsigned)nx;
        const long long row = row0 + pix;
        if constexpr (AUG) {
            if (row < win_lo || row >= win_hi) continue;
        }
        float sg = 1.0f / (1.0f + expf(-rp[(long long)y * sy + (long long)x * sx + o]));
        float v = sg;
        if (o == 0) v = (sg * 2.0f - 0.5f + (float)x) * stride;
        else if (o == 1) v = (sg * 2.0f - 0.5f + (float)y) * stride;
        else if (o == 2 || o == 3) { float q = sg * 2.0f; v = q * q * (o == 2 ? aw : ah); }
        if constexpr (AUG) {
            if (o < 4) v = v / aug_scale;
            if (o == 0 && aug_flip == 3) v = aug_extent - v;
            if (o == 1 && aug_flip == 2) v = aug_extent - v;
        }
        op[row * no + o] = v;
    }
}

static int head_decode_launch(bool aug, const float* raw, const int64_t* raw_strides, int B, int na, int ny, int nx, int no,
                              const float* anchors_px, float stride, float* out, int64_t rows_total, int64_t row_off,
                              float aug_scale, int aug_flip, float aug_extent, int64_t win_lo, int64_t win_hi, ayolo_stream s) {
    AY_CHECK_ARG(raw && out && anchors_px && no > 4, "head_decode: bad args");
    long long total = (long long)B * na * ny * nx * no;
    long long sb = (long long)na * ny * nx * no, sa = (long long)ny * nx * no, sy = (long long)nx * no, sx = no;
    if (raw_strides) { sb = raw_strides[0]; sa = raw_strides[1]; sy = raw_strides[2]; sx = raw_strides[3]; }
    if (total == 0) return AYOLO_OK;
    const unsigned long long plane_n = (unsigned long long)ny * nx * no, plane_px = (unsigned long long)ny * nx;
    // (the element kernel with 64-bit index arithmetic below only takes what these 32-bit magic divisions cannot index)
    if (plane_n * (unsigned)no < (1ull << 32) && plane_px * (unsigned)nx < (1ull << 32) && (long long)B * na < 65536 && no > 1 && nx > 1) {
        const unsigned m_no = (unsigned)(((1ull << 32) + (unsigned)no - 1) / (unsigned)no);
        const unsigned m_nx = (unsigned)(((1ull << 32) + (unsigned)nx - 1) / (unsigned)nx);
        unsigned gx = (unsigned)((plane_n + 256 * 4 - 1) / (256 * 4));
        if (gx > 4096u) gx = 4096u;
        const dim3 grid(gx, (unsigned)(B * na));
        if (aug)
            hipLaunchKernelGGL(k_head_decode_plane<true>, grid, dim3(256), 0, (hipStream_t)s, raw, sb, sa, sy, sx, na, ny, nx, no, m_no, m_nx,
                               anchors_px, stride, out, (long long)rows_total, (long long)row_off, aug_scale, aug_flip, aug_extent,
                               (long long)win_lo, (long long)win_hi);
        else
            hipLaunchKernelGGL(k_head_decode_plane<false>, grid, dim3(256), 0, (hipStream_t)s, raw, sb, sa, sy, sx, na, ny, nx, no, m_no, m_nx,
                               anchors_px, stride, out, (long long)rows_total, (long long)row_off, 1.0f, 0, 0.0f, 0LL, 0LL);
        AY_CHECK_LAUNCH("k_head_decode_plane");
        return AYOLO_OK;
    }
    if (aug)
        hipLaunchKernelGGL(k_head_decode<true>, dim3(grid_for(total, 256 * 4)), dim3(256), 0, (hipStream_t)s, raw, sb, sa, sy, sx, B, na,
                           ny, nx, no, anchors_px, stride, out, (long long)rows_total, (long long)row_off, aug_scale, aug_flip,
                           aug_extent, (long long)win_lo, (long long)win_hi);
    else
        hipLaunchKernelGGL(k_head_decode<false>, dim3(grid_for(total, 256 * 4)), dim3(256), 0, (hipStream_t)s, raw, sb, sa, sy, sx, B, na,
                           ny, nx, no, anchors_px, stride, out, (long long)rows_total, (long long)row_off, 1.0f, 0, 0.0f, 0LL, 0LL);
    AY_CHECK_LAUNCH("k_head_decode");
    return AYOLO_OK;
}

extern "C" int ayolo_head_decode(const float* raw, const int64_t* raw_strides, int B, int na, int ny, int nx, int no,
                                 const float* anchors_px, float stride, float* out, int64_t rows_total, int64_t row_off,
                                 ayolo_stream s) {
    return head_decode_launch(false, raw, raw_strides, B, na, ny, nx, no, anchors_px, stride, out, rows_total, row_off, 1.0f, 0, 0.0f, 0, 0, s);
}

extern "C" int ayolo_head_decode_aug(const float* raw, const int64_t* raw_strides, int B, int na, int ny, int nx, int no,
                                     const float* anchors_px, float stride, float* out, int64_t rows_total, int64_t row_off,
                                     float scale, int flip, float flip_extent, int64_t win_lo, int64_t win_hi, ayolo_stream s) {
    AY_CHECK_ARG(scale > 0.0f && (flip == 0 || flip == 2 || flip == 3) && win_lo >= 0 && win_hi <= rows_total,
                 "head_decode_aug: scale > 0, flip in {0, 2, 3}, window inside the output");
    return head_decode_launch(true, raw, raw_strides, B, na, ny, nx, no, anchors_px, stride, out, rows_total, row_off, scale, flip,
                              flip_extent, win_lo, win_hi, s);
}

// d(raw) (B,na,ny,nx,no) fp32 -> NHWC gradient dz[pix][ldz] (channel c = a*no + o; channels >= na*no zero) and
// dbias[c] += sum over pixels (fp32, zeroed by the caller).
template <typename T>
__global__ __launch_bounds__(256) void k_head_grad_pack(const float* draw, int B, int na, int ny, int nx, int no, T* dz,
                                                        int ldz, float* dbias) {
    // thread = one 8-channel group of the NHWC row (one 16-byte fp16 / two 16-byte fp32 stores); a workgroup covers
    // 256 / (ldz/8) pixels per iteration, pixels strided over the grid; bias partials live in registers, one global
    // atomic per (workgroup, channel)
    const unsigned hw = (unsigned)(ny * nx);
    const long long npix = (long long)B * hw;
    const int Cc = na * no;
    const int CG = ldz / 8;                      // channel groups per pixel (host: ldz % 8 == 0, CG <= 256)
    const int RPB = 256 / CG;
    const int cg = threadIdx.x % CG, prow = threadIdx.x / CG;
    int aa[8], oo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = cg * 8 + i;
        aa[i] = c < Cc ? c / no : -1;
        oo[i] = c < Cc ? c - aa[i] * no : 0;
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long pix = (long long)blockIdx.x * RPB + prow; prow < RPB && pix < npix; pix += (long long)gridDim.x * RPB) {
        const unsigned b = (unsigned)(pix / hw), p = (unsigned)(pix - (long long)b * hw);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = aa[i] >= 0 ? draw[(((long long)b * na + aa[i]) * hw + p) * no + oo[i]] : 0.0f;
            acc[i] += v[i];
        }
        T out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = (T)v[i];
        T* dst = dz + pix * ldz + cg * 8;
        if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(out);
        else {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(out);
            *reinterpret_cast<uint4*>(dst + 4) = *reinterpret_cast<const uint4*>(out + 4);
        }
    }
    if (dbias) {
        // workgroup-level sum over its pixel rows in LDS, then ONE global atomic per (workgroup, channel)
        __shared__ float sb[2048];
        for (int c = threadIdx.x; c < ldz; c += 256) sb[c] = 0.0f;
        __syncthreads();
        if (prow < RPB) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (aa[i] >= 0) atomicAdd(&sb[cg * 8 + i], acc[i]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < Cc; c += 256) atomicAdd(&dbias[c], sb[c]);
    }
}

extern "C" int ayolo_head_grad_pack(const float* draw, int B, int na, int ny, int nx, int no, int dtype, void* dz, int ldz,
                                    float* dbias, ayolo_stream s) {
    AY_CHECK_ARG(draw && dz && ldz >= na * no && ldz % 8 == 0 && ldz <= 2048, "head_grad_pack: bad args (ldz=%d)", ldz);
    long long total = (long long)B * ny * nx * ldz;
    long long npix = (long long)B * ny * nx;
    (void)total;
    DISPATCH_T(dtype, hipLaunchKernelGGL(k_head_grad_pack<T>, dim3((unsigned)(npix < 2048 ? npix : 2048)), dim3(256), 0,
                                         (hipStream_t)s, draw, B, na, ny, nx, no, (T*)dz, ldz, dbias);)
    AY_CHECK_LAUNCH("k_head_grad_pack");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// ModelEMA.update (scripts/utils/torch_utils.py:405-416): v = v*d; v += (1-d)*m  for every floating tensor of the
// state dict, all tensors in ONE launch (job table in device memory, blockIdx.y = tensor).  Same two-step rounding as
// the reference's in-place ops (this file is built with -ffp-contract=off).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ema_update(const ayolo_ema_job* jobs, float d) {
    const ayolo_ema_job J = jobs[blockIdx.y];
    const float omd = 1.0f - d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (long long)gridDim.x * 256) {
        float v = J.ema[i] * d;
        v += omd * J.src[i];
        J.ema[i] = v;
    }
}

extern "C" int ayolo_ema_update(const ayolo_ema_job* jobs_dev, int njobs, float decay, ayolo_stream s) {
    AY_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535, "ema_update: njobs=%d", njobs);
    // callers cut large tensors into jobs of a few 10^4 elements (one grid row each)
    hipLaunchKernelGGL(k_ema_update, dim3(8, (unsigned)njobs), dim3(256), 0, (hipStream_t)s, jobs_dev, decay);
    AY_CHECK_LAUNCH("k_ema_update");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void ayolo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ayolo_last_error(void) { return g_err; }
extern "C" int ayolo_version(void) { return 1; }
