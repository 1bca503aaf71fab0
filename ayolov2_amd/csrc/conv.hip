// Implicit-GEMM convolution on the gfx950 matrix cores (no im2col materialisation).
//
//   gconv  : y[pixel][n] = sum_{tap,c} x[pixel @ tap][c] * w[n][tap][c]     (forward AND dgrad)
//   wgrad  : dw[n][tap][c] += sum_{pixel} dy[pixel][n] * x[pixel @ tap][c]
//
// Layout: activations NHWC (channel stride = ld*), weights [N][taps][C] with C contiguous.  MFMA roles are
// "swapped" (A = weights, rows = output channel; B = gathered pixels), so each lane of the 32x32 accumulator
// holds ONE pixel and 4-channel runs: the epilogue stores 8 B (fp16) / 16 B (fp32) channel vectors straight to
// NHWC without an LDS transpose.
//   fp16 : v_mfma_f32_32x32x16_f16  (fp32 accumulate)
//   fp32 : v_mfma_f32_32x32x2_f32   (exact fp32 -- the 1e-4 parity mode)
// Tiles: TM in {32,64,128} output channels x 128 pixels x BK=32, 4 wavefronts, register-staged global->LDS
// double buffer with padded rows (80 B / 144 B) so ds_read_b128 fragments are bank-conflict free.
//
// Replaces kindle Conv/YOLOHead.conv forward (yolov5s.yaml:21-57) and the autograd backward torch/cuDNN ran
// (scripts/train/yolo_trainer.py:329).
#include "common.h"
#include <stdlib.h>

#define MAX_TAPS 36
#define BK 32
#define TP 128   // pixels per block tile

struct GConvP {
    const void* x; const void* w; void* y;
    int B, XH, XW, ldx;
    int OH, OW, ish, isw;
    int YH, YW, ldy, osh, osw, oah, oaw;
    int C, ntaps, K, ldw, Nout;
    int epi; const float* scale; const float* shift; float* stats;
    int head_no; int accumulate; int stat_reps;
    int x_linear, y_linear, ntn, nslots;
    long long Mtotal;
    signed char dh[MAX_TAPS], dw[MAX_TAPS], wt[MAX_TAPS];
};

template <typename T> struct Tr;
template <> struct Tr<half_t> {
    static constexpr int CE = 8;            // elements per 16-byte chunk
    static constexpr int PADE = 8;          // row padding (elements)
    typedef half8 frag;                     // 8 k-values per lane per k16 step
    typedef uint4 chunk;
};
template <> struct Tr<float> {
    static constexpr int CE = 4;
    static constexpr int PADE = 4;
    struct frag { float v[8]; };
    typedef uint4 chunk;
};

__device__ __forceinline__ void mma_step(const half8& a, const half8& b, float16v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const Tr<float>::frag& a, const Tr<float>::frag& b, float16v& acc) {
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
}
__device__ __forceinline__ half8 lds_frag(const half_t* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ Tr<float>::frag lds_frag(const float* p) {
    Tr<float>::frag f;
    float4v a = *reinterpret_cast<const float4v*>(p), b = *reinterpret_cast<const float4v*>(p + 4);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
    return f;
}

__device__ __forceinline__ float cvt_round(float v, half_t*) { return (float)(half_t)v; }
__device__ __forceinline__ float cvt_round(float v, float*) { return v; }

// EM (epilogue mode, compile time so that unused paths cost no registers):
//   0 plain store (+ optional BN statistics)   1 accumulate into y (dgrad)   2 affine / affine+SiLU   3 YOLOHead fp32
//
// Persistent, tile-pipelined implicit GEMM.  A workgroup owns output-channel tile `nt` and walks the pixel tiles of
// its XCD's band; one "step" = (pixel tile, 32-wide k slice).  The global loads of step s+2 are issued while the
// MFMAs consume step s from LDS and step s+1 waits in registers (two register stages + two LDS stages), across
// tile boundaries: the measured load->use latency under load (~3000 cycles) is hidden by ~two compute phases per
// workgroup times the resident workgroups.  BN statistics are accumulated in registers across all tiles of the
// workgroup and reduced once at the end.
template <typename T, int TM>
struct GTile {
    static constexpr int CE = Tr<T>::CE;
    static constexpr int CPR = BK / CE;
    static constexpr int LDR = BK + Tr<T>::PADE;
    static constexpr int WM = TM / 32, WP = 4 / WM, NI = TP / (32 * WP);
    static constexpr int XR = (TP * CPR) / 256;
    static constexpr int WCH = TM * CPR;
    static constexpr int WR = (WCH + 255) / 256;
};

// loader position: which (tile, k-slice) is fetched next, plus the per-row pixel decode of that tile
template <typename T, int TM>
struct GLoader {
    long long tile;
    int kt, tap, cch;
    bool valid;
    long long xbase[GTile<T, TM>::XR];
    int xh0[GTile<T, TM>::XR], xw0[GTile<T, TM>::XR];
};

template <typename T, int TM>
__device__ __forceinline__ void g_setup_rows(const GConvP& p, GLoader<T, TM>& L, int tid, int kc) {
    using G = GTile<T, TM>;
    L.tap = (kc * G::CE) / p.C;
    L.cch = (kc * G::CE) % p.C;
    const long long m0 = L.tile * TP;
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        const int row = (tid + 256 * r) / G::CPR;
        const long long m = m0 + row;
        if (m < p.Mtotal) {
            if (p.x_linear) {
                L.xbase[r] = m; L.xh0[r] = 0; L.xw0[r] = 0;
            } else {
                const unsigned mu = (unsigned)m;
                unsigned t = mu / (unsigned)p.OW;
                int ow = (int)(mu - t * (unsigned)p.OW);
                unsigned n = t / (unsigned)p.OH;
                int oh = (int)(t - n * (unsigned)p.OH);
                L.xbase[r] = (long long)n * p.XH * p.XW;
                L.xh0[r] = oh * p.ish; L.xw0[r] = ow * p.isw;
            }
        } else { L.xbase[r] = -1; L.xh0[r] = 0; L.xw0[r] = 0; }
    }
}

// advance the loader by one step (possibly into the next tile of the band) and issue its global loads
template <typename T, int TM, typename KT>
__device__ __forceinline__ void g_issue(const GConvP& p, GLoader<T, TM>& L, bool first, int nk, long long ntiles, unsigned lstride,
                                        KT ktab, const T* __restrict__ X, const T* __restrict__ Wg, int tid, int kc, int n0,
                                        uint4 (&xreg)[GTile<T, TM>::XR], uint4 (&wreg)[GTile<T, TM>::WR]) {
    using G = GTile<T, TM>;
    if (first) {
        L.valid = L.tile < ntiles;
        if (L.valid) g_setup_rows<T, TM>(p, L, tid, kc);
    } else if (L.valid) {
        ++L.kt;
        if (L.kt == nk) {
            L.kt = 0;
            L.tile += lstride;
            L.valid = L.tile < ntiles;
            if (L.valid) g_setup_rows<T, TM>(p, L, tid, kc);
        } else {
            L.cch += BK;
            while (L.cch >= p.C) { L.cch -= p.C; ++L.tap; }
        }
    }
    if (!L.valid) return;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const bool tap_ok = L.tap < p.ntaps;
    const int tq = tap_ok ? L.tap : 0;
    const int dh = ktab[tq], dw = ktab[MAX_TAPS + tq];
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        xreg[r] = zero4;
        if (p.x_linear) {
            if (tap_ok && L.xbase[r] >= 0) xreg[r] = *reinterpret_cast<const uint4*>(X + L.xbase[r] * p.ldx + L.cch);
        } else {
            int ih = L.xh0[r] + dh, iw = L.xw0[r] + dw;
            bool ok = tap_ok && L.xbase[r] >= 0 && ih >= 0 && ih < p.XH && iw >= 0 && iw < p.XW;
            if (ok) xreg[r] = *reinterpret_cast<const uint4*>(X + (L.xbase[r] + (long long)ih * p.XW + iw) * p.ldx + L.cch);
        }
    }
    const int wcol = tap_ok ? ktab[2 * MAX_TAPS + tq] * p.C + L.cch : 0;
#pragma unroll
    for (int r = 0; r < G::WR; ++r) {
        int q = tid + 256 * r;
        int row = q / G::CPR;
        wreg[r] = zero4;
        if (q < G::WCH && tap_ok && (n0 + row) < p.Nout)
            wreg[r] = *reinterpret_cast<const uint4*>(Wg + (long long)(n0 + row) * p.ldw + wcol);
    }
}

template <typename T, int TM>
__device__ __forceinline__ void g_stage(T* sW, T* sX, int buf, int tid, int kc, const uint4 (&xreg)[GTile<T, TM>::XR],
                                        const uint4 (&wreg)[GTile<T, TM>::WR]) {
    using G = GTile<T, TM>;
    T* dX = sX + buf * TP * G::LDR;
    T* dW = sW + buf * TM * G::LDR;
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        int row = (tid + 256 * r) / G::CPR;
        *reinterpret_cast<uint4*>(dX + row * G::LDR + kc * G::CE) = xreg[r];
    }
#pragma unroll
    for (int r = 0; r < G::WR; ++r) {
        int q = tid + 256 * r;
        if (q < G::WCH) *reinterpret_cast<uint4*>(dW + (q / G::CPR) * G::LDR + kc * G::CE) = wreg[r];
    }
}

template <typename T, int TM>
__device__ __forceinline__ void g_mma(const T* sW, const T* sX, int buf, int wm, int wp, int lane,
                                      float16v (&acc)[GTile<T, TM>::NI]) {
    using G = GTile<T, TM>;
    const T* cW = sW + buf * TM * G::LDR + (wm * 32 + (lane & 31)) * G::LDR + (lane >> 5) * 8;
    const T* cX = sX + buf * TP * G::LDR + (wp * G::NI * 32 + (lane & 31)) * G::LDR + (lane >> 5) * 8;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        auto a = lds_frag(cW + kk * 16);
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) {
            auto b = lds_frag(cX + ni * 32 * G::LDR + kk * 16);
            mma_step(a, b, acc[ni]);
        }
    }
}

// tile finished: acc[ni][r] holds channel = cbase + 8*(r>>2) + (r&3), pixel = m0 + wp*NI*32 + ni*32 + (lane&31)
template <typename T, int TM, int EM>
__device__ __forceinline__ void g_epilogue(const GConvP& p, long long tile, int wp, int lane, int cbase, bool want_stats,
                                           float16v (&acc)[GTile<T, TM>::NI], float (&ssum)[16], float (&ssq)[16]) {
    using G = GTile<T, TM>;
    const long long m0 = tile * TP;
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni) {
        const long long m = m0 + wp * G::NI * 32 + ni * 32 + (lane & 31);
        const bool pv = m < p.Mtotal;
        long long yo = 0;
        if (pv) {
            if (p.y_linear) yo = m * p.ldy;
            else {
                const unsigned mu = (unsigned)m;
                unsigned t = mu / (unsigned)p.OW;
                int ow = (int)(mu - t * (unsigned)p.OW);
                unsigned nn = t / (unsigned)p.OH;
                int oh = (int)(t - nn * (unsigned)p.OH);
                yo = (((long long)nn * p.YH + (oh * p.osh + p.oah)) * p.YW + (ow * p.osw + p.oaw)) * p.ldy;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cbase + 8 * g;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[ni][g * 4 + e]; acc[ni][g * 4 + e] = 0.0f; }
            if constexpr (EM == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c + e < p.Nout) {
                        float sc = p.scale ? p.scale[c + e] : 1.0f, sh = p.shift ? p.shift[c + e] : 0.0f;
                        float u = v[e] * sc + sh;
                        v[e] = (p.epi == AYOLO_EPI_AFFINE_SILU) ? silu_f(u) : u;
                    }
                }
            }
            if constexpr (EM == 3) {
                // YOLOHead: fp32 logits + bias, NHWC [pixel][ldy] (ldy = Cout rounded up to 8), 16-byte stores;
                // the (B, na, ny, nx, no) tensor the loss / decode see is a strided view of this buffer
                if (pv && c < p.ldy) {
                    float* Y = reinterpret_cast<float*>(p.y) + m * p.ldy + c;
                    float4v f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = v[e] + ((p.shift && c + e < p.Nout) ? p.shift[c + e] : 0.0f);
                    *reinterpret_cast<float4v*>(Y) = f;
                }
                continue;
            }
            if (want_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float q = pv ? cvt_round(v[e], (T*)nullptr) : 0.0f;
                    ssum[g * 4 + e] += q;
                    ssq[g * 4 + e] += q * q;
                }
            }
            if (pv) {
                T* Y = reinterpret_cast<T*>(p.y) + yo + c;
                if (c + 3 < p.Nout && (p.ldy & 3) == 0) {
                    if constexpr (EM == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)Y[e];
                    }
                    if constexpr (sizeof(T) == 2) {
                        half4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
                        *reinterpret_cast<half4*>(Y) = h;
                    } else {
                        float4v f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = v[e];
                        *reinterpret_cast<float4v*>(Y) = f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < p.Nout) Y[e] = (T)(EM == 1 ? v[e] + (float)Y[e] : v[e]);
                }
            }
        }
    }
}

template <typename T, int TM>
__device__ __forceinline__ void g_stats_flush(const GConvP& p, float* sStat, int tid, int lane, int wm, int n0, unsigned slot,
                                              const float (&ssum)[16], const float (&ssq)[16]) {
    for (int i = tid; i < 2 * TM; i += 256) sStat[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = ssum[r], b = ssq[r];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
        }
        if ((lane & 31) == 0) {
            int cl = wm * 32 + 4 * (lane >> 5) + 8 * (r >> 2) + (r & 3);
            atomicAdd(&sStat[cl], a);
            atomicAdd(&sStat[TM + cl], b);
        }
    }
    __syncthreads();
    // replicated accumulators: workgroups spread over stat_reps copies so L2 atomics do not serialise
    float* st = p.stats + (size_t)(slot % (unsigned)p.stat_reps) * 2 * p.Nout;
    for (int i = tid; i < TM; i += 256) {
        if (n0 + i < p.Nout) {
            atomicAdd(&st[n0 + i], sStat[i]);
            atomicAdd(&st[p.Nout + n0 + i], sStat[TM + i]);
        }
    }
}

template <typename T, int TM, int EM>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? (TM == 128 ? 2 : (TM == 64 ? 3 : 4)) : 1)) void k_gconv(GConvP p) {
    using G = GTile<T, TM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sW = reinterpret_cast<T*>(smem_raw);                          // [2][TM][LDR]
    T* sX = sW + 2 * TM * G::LDR;                                    // [2][TP][LDR]
    float* sStat = reinterpret_cast<float*>(sX + 2 * TP * G::LDR);   // [2][TM]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % G::WM, wp = wave / G::WM;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(p.w);

    // ---- block -> (channel tile, XCD band, slot).  Workgroups are dealt round-robin to the 8 XCDs; all channel
    // tiles of one pixel tile go to the SAME XCD back to back, and each XCD walks a contiguous band of pixel tiles.
    const unsigned Lb = blockIdx.x;
    const unsigned xcd = Lb & 7u, idx = Lb >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const int n0 = (int)nt * TM;
    const long long ntiles_all = (p.Mtotal + TP - 1) / TP;
    const long long tpx = (ntiles_all + 7) / 8;
    const long long band_lo = (long long)xcd * tpx;
    const long long ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = (unsigned)p.nslots / 8u;

    typedef __attribute__((address_space(4))) const signed char* kptr_t;
    const kptr_t ktab = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(GConvP, dh);
    const int kc = tid % G::CPR;

    float16v acc[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const bool want_stats = (EM == 0) && (p.stats != nullptr);
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
    const int cbase = n0 + wm * 32 + 4 * (lane >> 5);

    int nk = (p.K + BK - 1) / BK;
    if (nk < 1) nk = 1;                      // K == 0 (tap-less dgrad residue class): one all-zero step
    long long cur_tile = band_lo + lslot;
    if (cur_tile >= ntiles) return;
    int cur_kt = 0;

    GLoader<T, TM> L;
    L.tile = cur_tile; L.kt = 0; L.tap = 0; L.cch = 0; L.valid = true;
    // Prefetch depth: two register stages for the wide tiles; the 32-channel tile keeps one (its 4 workgroups
    // per CU hide latency better than a deeper pipeline at 3 per CU -- measured).
    constexpr bool DEEP = (TM != 32);
    uint4 xA[G::XR], wA[G::WR];

    if constexpr (DEEP) {
        uint4 xB[G::XR], wB[G::WR];
        // prologue: step 0 -> regs A -> LDS[0]; step 1 -> regs B (in flight)
        g_issue<T, TM>(p, L, true, nk, ntiles, lstride, ktab, X, Wg, tid, kc, n0, xA, wA);
        g_stage<T, TM>(sW, sX, 0, tid, kc, xA, wA);
        g_issue<T, TM>(p, L, false, nk, ntiles, lstride, ktab, X, Wg, tid, kc, n0, xB, wB);
        bool next_valid = L.valid;               // regs B hold a real step
        __syncthreads();
        // One half-iteration: compute the step in LDS[BUF]; regs `xn/wn` hold the following step (staged into
        // LDS[BUF^1] afterwards); regs `xf/wf` are free and receive the loads of the step after that.
#define G_HALF(BUF, xf, wf, xn, wn)                                                                              \
        {                                                                                                        \
            const bool have_next = next_valid;                                                                   \
            g_issue<T, TM>(p, L, false, nk, ntiles, lstride, ktab, X, Wg, tid, kc, n0, xf, wf);                  \
            next_valid = L.valid;                                                                                \
            g_mma<T, TM>(sW, sX, BUF, wm, wp, lane, acc);                                                        \
            if (cur_kt == nk - 1) g_epilogue<T, TM, EM>(p, cur_tile, wp, lane, cbase, want_stats, acc, ssum, ssq); \
            if (have_next) g_stage<T, TM>(sW, sX, (BUF) ^ 1, tid, kc, xn, wn);                                   \
            __syncthreads();                                                                                     \
            if (!have_next) break;                                                                               \
            ++cur_kt;                                                                                            \
            if (cur_kt == nk) { cur_kt = 0; cur_tile += lstride; }                                               \
        }
        while (true) {
            G_HALF(0, xA, wA, xB, wB)
            G_HALF(1, xB, wB, xA, wA)
        }
#undef G_HALF
    } else {
        g_issue<T, TM>(p, L, true, nk, ntiles, lstride, ktab, X, Wg, tid, kc, n0, xA, wA);
        g_stage<T, TM>(sW, sX, 0, tid, kc, xA, wA);
        __syncthreads();
        int buf = 0;
        while (true) {
            g_issue<T, TM>(p, L, false, nk, ntiles, lstride, ktab, X, Wg, tid, kc, n0, xA, wA);   // step s+1
            const bool have_next = L.valid;
            g_mma<T, TM>(sW, sX, buf, wm, wp, lane, acc);
            if (cur_kt == nk - 1) g_epilogue<T, TM, EM>(p, cur_tile, wp, lane, cbase, want_stats, acc, ssum, ssq);
            if (have_next) g_stage<T, TM>(sW, sX, buf ^ 1, tid, kc, xA, wA);
            __syncthreads();
            if (!have_next) break;
            buf ^= 1;
            ++cur_kt;
            if (cur_kt == nk) { cur_kt = 0; cur_tile += lstride; }
        }
    }

    if (want_stats) g_stats_flush<T, TM>(p, sStat, tid, lane, wm, n0, slot, ssum, ssq);
}

// ---------------------------------------------------------------------------------------------------
// Direct (halo-tiled) convolution for k x k > 1x1: the input patch of an 8x16 output tile (with halo) is
// staged in LDS ONCE per 32-channel chunk and all k*k taps read their MFMA fragments from it at shifted
// addresses, so global/L2 traffic for the activations drops from k*k x to ~(1 + halo) x.  Weights stream as
// [TM x 32] tiles per (chunk, tap) step; the next step's weights and a slice of the next patch are
// prefetched into registers while the MFMAs run (persistent over tiles, same epilogue as k_gconv).
// ---------------------------------------------------------------------------------------------------
#define DTH 8
#define DTW 16
#define DCK 32          // channels per LDS patch chunk
#define DSL_MAX 3       // patch chunk loads per thread per step (upper bound)

struct DConvP {
    GConvP g;            // shared fields (x, w, y, dims, epilogue, taps)
    int tyn, txn;        // tiles per image in y / x
    int PH, PW;          // patch extent (pixels)
    int dh_min, dw_min;  // smallest tap offsets
    int sl;              // patch chunk loads per thread per step
};

template <typename T, int TM, int EM>
__global__ __launch_bounds__(256) void k_dconv(DConvP dp) {
    const GConvP& p = dp.g;
    constexpr int CE = Tr<T>::CE;
    constexpr int CPR = DCK / CE;            // 16-byte chunks per patch pixel
    constexpr int LDR = DCK + Tr<T>::PADE;   // LDS row stride (elements) for both the patch and the weight tile
    constexpr int WM = TM / 32, WP = 4 / WM, NI = TP / (32 * WP);
    constexpr int WCH = TM * CPR;
    constexpr int WR = (WCH + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int patch_elems = dp.PH * dp.PW * LDR;
    T* sW = reinterpret_cast<T*>(smem_raw);                        // [2][TM][LDR]
    T* sP = sW + 2 * TM * LDR;                                     // [2][PH*PW][LDR]
    float* sStat = reinterpret_cast<float*>(sP + 2 * patch_elems); // [2][TM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wp = wave / WM;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(p.w);

    const unsigned L = blockIdx.x;
    const unsigned xcd = L & 7u, idx = L >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const unsigned nslots = (unsigned)p.nslots;
    const int n0 = (int)nt * TM;
    const long long tiles_per_img = (long long)dp.tyn * dp.txn;
    const long long ntiles_all = (long long)p.B * tiles_per_img;
    const long long tpx = (ntiles_all + 7) / 8;
    const long long band_lo = (long long)xcd * tpx;
    const long long ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = nslots / 8u;

    typedef __attribute__((address_space(4))) const signed char* kptr_t;
    const kptr_t ktab = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(GConvP, dh);

    const int nchunks = p.C / DCK;
    const int nsteps = nchunks * p.ntaps;            // steps per tile: (chunk, tap)
    const int patch_chunks = dp.PH * dp.PW * CPR;
    const int kc = tid % CPR;

    float16v acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const bool want_stats = (EM == 0) && (p.stats != nullptr);
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
    const int cbase = n0 + wm * 32 + 4 * (lane >> 5);

    // per-lane fragment geometry: pixel j of the tile -> (oy, ox)
    int frag_off[NI];     // element offset of the lane's pixel inside the patch for tap offset (dh_min, dw_min)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int j = wp * NI * 32 + ni * 32 + (lane & 31);
        const int oy = j / DTW, ox = j % DTW;
        frag_off[ni] = ((oy * p.ish) * dp.PW + ox * p.isw) * LDR + (lane >> 5) * 8;
    }

    uint4 preg[DSL_MAX], wreg[WR];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    long long cur_tile = band_lo + lslot;
    if (cur_tile >= ntiles) return;
    // loader position (one step ahead of the compute position)
    long long ld_tile = cur_tile;
    int ld_step = 0;
    int cur_step = 0;
    int wbuf = 0, pbuf = 0;      // LDS buffers holding the CURRENT step's weights / CURRENT chunk's patch
    bool first = true;

    // tile origin of the patch being loaded
    auto tile_origin = [&](long long tile, int& n, int& ih0, int& iw0) {
        const unsigned tu = (unsigned)tile;
        const unsigned nn = tu / (unsigned)tiles_per_img;
        const unsigned r = tu - nn * (unsigned)tiles_per_img;
        const unsigned ty = r / (unsigned)dp.txn, tx = r - ty * (unsigned)dp.txn;
        n = (int)nn;
        ih0 = (int)ty * DTH * p.ish + dp.dh_min;
        iw0 = (int)tx * DTW * p.isw + dp.dw_min;
    };
    // loads slice `sl_idx` (0..ntaps-1) of the patch for (tile, chunk) into preg
    auto load_patch_slice = [&](long long tile, int chunk, int slice) {
        int n, ih0, iw0;
        tile_origin(tile, n, ih0, iw0);
#pragma unroll
        for (int u = 0; u < DSL_MAX; ++u) {
            preg[u] = zero4;
            if (u < dp.sl) {
                const int q = (slice * dp.sl + u) * 256 + tid;
                if (q < patch_chunks) {
                    const int pix = q / CPR;
                    const int py = pix / dp.PW, px = pix - py * dp.PW;
                    const int ih = ih0 + py, iw = iw0 + px;
                    if (ih >= 0 && ih < p.XH && iw >= 0 && iw < p.XW)
                        preg[u] = *reinterpret_cast<const uint4*>(X + (((long long)n * p.XH + ih) * p.XW + iw) * p.ldx + chunk * DCK + kc * CE);
                }
            }
        }
    };
    auto store_patch_slice = [&](int buf, int slice) {
        T* dP = sP + buf * patch_elems;
#pragma unroll
        for (int u = 0; u < DSL_MAX; ++u) {
            if (u < dp.sl) {
                const int q = (slice * dp.sl + u) * 256 + tid;
                if (q < patch_chunks) *reinterpret_cast<uint4*>(dP + (q / CPR) * LDR + kc * CE) = preg[u];
            }
        }
    };
    auto load_w = [&](int step) {
        const int chunk = step / p.ntaps, tap = step - chunk * p.ntaps;
        const int wcol = ktab[2 * MAX_TAPS + tap] * p.C + chunk * DCK + kc * CE;
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int q = tid + 256 * r, row = q / CPR;
            wreg[r] = zero4;
            if (q < WCH && (n0 + row) < p.Nout) wreg[r] = *reinterpret_cast<const uint4*>(Wg + (long long)(n0 + row) * p.ldw + wcol);
        }
    };
    auto store_w = [&](int buf) {
        T* dW = sW + buf * TM * LDR;
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int q = tid + 256 * r;
            if (q < WCH) *reinterpret_cast<uint4*>(dW + (q / CPR) * LDR + kc * CE) = wreg[r];
        }
    };

    // ---- prologue: whole first patch (all slices) + first weight tile
    for (int sidx = 0; sidx < p.ntaps; ++sidx) {
        load_patch_slice(cur_tile, 0, sidx);
        store_patch_slice(0, sidx);
    }
    load_w(0);
    store_w(0);
    __syncthreads();

    while (true) {
        // ---------------- prefetch for the next step: its weight tile, and one slice of the NEXT chunk's patch
        int nx_step = cur_step + 1;
        long long nx_tile = cur_tile;
        if (nx_step == nsteps) { nx_step = 0; nx_tile += lstride; }
        const bool more = nx_tile < ntiles;
        const int cur_chunk = cur_step / p.ntaps, cur_tap = cur_step - cur_chunk * p.ntaps;
        // the patch that follows the current chunk
        int pn_chunk = cur_chunk + 1;
        long long pn_tile = cur_tile;
        if (pn_chunk == nchunks) { pn_chunk = 0; pn_tile += lstride; }
        const bool pmore = pn_tile < ntiles;
        if (more) load_w(nx_step);
        if (pmore) load_patch_slice(pn_tile, pn_chunk, cur_tap);

        // ---------------- MFMAs of the current step
        {
            const int toff = ((ktab[cur_tap] - dp.dh_min) * dp.PW + (ktab[MAX_TAPS + cur_tap] - dp.dw_min)) * LDR;
            const T* cW = sW + wbuf * TM * LDR + (wm * 32 + (lane & 31)) * LDR + (lane >> 5) * 8;
            const T* cP = sP + pbuf * patch_elems + toff;
#pragma unroll
            for (int kk = 0; kk < DCK / 16; ++kk) {
                auto a = lds_frag(cW + kk * 16);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    auto b = lds_frag(cP + frag_off[ni] + kk * 16);
                    mma_step(a, b, acc[ni]);
                }
            }
        }
        // ---------------- tile finished: epilogue
        if (cur_step == nsteps - 1) {
            const unsigned tu = (unsigned)cur_tile;
            const unsigned nn = tu / (unsigned)tiles_per_img;
            const unsigned rr = tu - nn * (unsigned)tiles_per_img;
            const unsigned ty = rr / (unsigned)dp.txn, tx = rr - ty * (unsigned)dp.txn;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int j = wp * NI * 32 + ni * 32 + (lane & 31);
                const int oh = (int)ty * DTH + j / DTW, ow = (int)tx * DTW + j % DTW;
                const bool pv = oh < p.OH && ow < p.OW;
                const long long yo = (((long long)nn * p.YH + (oh * p.osh + p.oah)) * p.YW + (ow * p.osw + p.oaw)) * p.ldy;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = cbase + 8 * g;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[ni][g * 4 + e]; acc[ni][g * 4 + e] = 0.0f; }
                    if constexpr (EM == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (c + e < p.Nout) {
                                float sc = p.scale ? p.scale[c + e] : 1.0f, sh = p.shift ? p.shift[c + e] : 0.0f;
                                float u = v[e] * sc + sh;
                                v[e] = (p.epi == AYOLO_EPI_AFFINE_SILU) ? silu_f(u) : u;
                            }
                        }
                    }
                    if (want_stats) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float q = pv ? cvt_round(v[e], (T*)nullptr) : 0.0f;
                            ssum[g * 4 + e] += q;
                            ssq[g * 4 + e] += q * q;
                        }
                    }
                    if (pv) {
                        T* Y = reinterpret_cast<T*>(p.y) + yo + c;
                        if (c + 3 < p.Nout && (p.ldy & 3) == 0) {
                            if constexpr (EM == 1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)Y[e];
                            }
                            if constexpr (sizeof(T) == 2) {
                                half4 h;
#pragma unroll
                                for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
                                *reinterpret_cast<half4*>(Y) = h;
                            } else {
                                float4v f;
#pragma unroll
                                for (int e = 0; e < 4; ++e) f[e] = v[e];
                                *reinterpret_cast<float4v*>(Y) = f;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c + e < p.Nout) Y[e] = (T)(EM == 1 ? v[e] + (float)Y[e] : v[e]);
                        }
                    }
                }
            }
        }
        // ---------------- stage the prefetched data
        if (more) store_w(wbuf ^ 1);
        if (pmore) store_patch_slice(pbuf ^ 1, cur_tap);
        __syncthreads();
        if (!more) break;
        wbuf ^= 1;
        if (cur_tap == p.ntaps - 1) pbuf ^= 1;       // next step starts a new chunk: its patch is complete
        cur_step = nx_step;
        cur_tile = nx_tile;
    }

    if (want_stats) {
        for (int i = tid; i < 2 * TM; i += 256) sStat[i] = 0.0f;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = ssum[r], b = ssq[r];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                a += __shfl_xor(a, off);
                b += __shfl_xor(b, off);
            }
            if ((lane & 31) == 0) {
                int cl = wm * 32 + 4 * (lane >> 5) + 8 * (r >> 2) + (r & 3);
                atomicAdd(&sStat[cl], a);
                atomicAdd(&sStat[TM + cl], b);
            }
        }
        __syncthreads();
        float* st = p.stats + (size_t)(slot % (unsigned)p.stat_reps) * 2 * p.Nout;
        for (int i = tid; i < TM; i += 256) {
            if (n0 + i < p.Nout) {
                atomicAdd(&st[n0 + i], sStat[i]);
                atomicAdd(&st[p.Nout + n0 + i], sStat[TM + i]);
            }
        }
    }
}

static int g_num_cu = 0;
static int num_cus() {
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cu = prop.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

template <typename T, int TM, int EM>
static int launch_gconv_em(GConvP p, hipStream_t s) {
    constexpr int LDR = BK + Tr<T>::PADE;
    size_t lds = (size_t)2 * (TM + TP) * LDR * sizeof(T) + 2 * TM * sizeof(float);
    const long long ntiles = (p.Mtotal + TP - 1) / TP;
    p.ntn = (p.Nout + TM - 1) / TM;
    // persistent grid: ~4 workgroups per CU in total, pixel-tile slots a multiple of the 8 XCDs
    static const int bpc = getenv("AYOLO_GCONV_BPC") ? atoi(getenv("AYOLO_GCONV_BPC")) : 4;
    long long want_slots = (long long)num_cus() * bpc / p.ntn;
    if (want_slots < 8) want_slots = 8;
    long long slots = ntiles < want_slots ? ntiles : want_slots;
    slots = (slots + 7) / 8 * 8;
    p.nslots = (int)slots;
    dim3 grid((unsigned)(slots * p.ntn));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gconv<T, TM, EM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gconv<T, TM, EM>), grid, dim3(256), lds, s, p);
    AY_CHECK_LAUNCH("k_gconv");
    return AYOLO_OK;
}

template <typename T, int TM, int EM>
static int launch_dconv_em(DConvP dp, hipStream_t s) {
    constexpr int LDR = DCK + Tr<T>::PADE;
    GConvP& p = dp.g;
    size_t lds = (size_t)2 * (TM + dp.PH * dp.PW) * LDR * sizeof(T) + 2 * TM * sizeof(float);
    const long long ntiles = (long long)p.B * dp.tyn * dp.txn;
    p.ntn = (p.Nout + TM - 1) / TM;
    int blocks_per_cu = (int)(150 * 1024 / lds);
    if (blocks_per_cu > 4) blocks_per_cu = 4;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    long long want_slots = (long long)num_cus() * blocks_per_cu / p.ntn;
    if (want_slots < 8) want_slots = 8;
    long long slots = ntiles < want_slots ? ntiles : want_slots;
    slots = (slots + 7) / 8 * 8;
    p.nslots = (int)slots;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dconv<T, TM, EM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL((k_dconv<T, TM, EM>), dim3((unsigned)(slots * p.ntn)), dim3(256), lds, s, dp);
    AY_CHECK_LAUNCH("k_dconv");
    return AYOLO_OK;
}

// Halo-tiled direct kernel is used for multi-tap convs on maps large enough to tile (>= 2 tiles per image) whose
// channel count is a multiple of the 32-channel LDS chunk; everything else goes through k_gconv.
static bool dconv_applicable(const GConvP& p, DConvP* out, int elem_size) {
    static const bool disabled = getenv("AYOLO_NO_DCONV") != nullptr;
    if (disabled || p.ntaps < 2 || p.C % DCK != 0 || p.epi == AYOLO_EPI_HEAD) return false;
    if (p.ish != 1 || p.isw != 1) return false;   // strided forward: the 17x33 halo patch leaves one workgroup per CU (measured slower)
    if (p.OH < DTH || p.OW < DTW) return false;
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int t = 0; t < p.ntaps; ++t) {
        dh_min = p.dh[t] < dh_min ? p.dh[t] : dh_min; dh_max = p.dh[t] > dh_max ? p.dh[t] : dh_max;
        dw_min = p.dw[t] < dw_min ? p.dw[t] : dw_min; dw_max = p.dw[t] > dw_max ? p.dw[t] : dw_max;
    }
    DConvP dp;
    dp.g = p;
    dp.tyn = (p.OH + DTH - 1) / DTH; dp.txn = (p.OW + DTW - 1) / DTW;
    dp.PH = (DTH - 1) * p.ish + (dh_max - dh_min) + 1;
    dp.PW = (DTW - 1) * p.isw + (dw_max - dw_min) + 1;
    dp.dh_min = dh_min; dp.dw_min = dw_min;
    const int cpr = DCK * elem_size / 16;
    const int patch_chunks = dp.PH * dp.PW * cpr;
    dp.sl = (patch_chunks + 256 * p.ntaps - 1) / (256 * p.ntaps);
    if (dp.sl > DSL_MAX) return false;
    const size_t lds_max = (size_t)2 * (128 + dp.PH * dp.PW) * (DCK + 16 / elem_size) * elem_size + 1024;
    if (lds_max > 150 * 1024) return false;
    // tile utilisation: skip when padding to 8x16 tiles wastes more than ~35 % of the MFMA work
    const double util = (double)p.OH * p.OW / ((double)dp.tyn * DTH * dp.txn * DTW);
    if (util < 0.65) return false;
    *out = dp;
    return true;
}

template <typename T, int TM>
static int launch_gconv(const GConvP& p, hipStream_t s) {
    DConvP dp;
    if (dconv_applicable(p, &dp, (int)sizeof(T))) {
        if (p.epi == AYOLO_EPI_AFFINE || p.epi == AYOLO_EPI_AFFINE_SILU) return launch_dconv_em<T, TM, 2>(dp, s);
        if (p.accumulate) return launch_dconv_em<T, TM, 1>(dp, s);
        return launch_dconv_em<T, TM, 0>(dp, s);
    }
    if (p.epi == AYOLO_EPI_HEAD) return launch_gconv_em<T, TM, 3>(p, s);
    if (p.epi == AYOLO_EPI_AFFINE || p.epi == AYOLO_EPI_AFFINE_SILU) return launch_gconv_em<T, TM, 2>(p, s);
    if (p.accumulate) return launch_gconv_em<T, TM, 1>(p, s);
    return launch_gconv_em<T, TM, 0>(p, s);
}

static int dispatch_gconv(int dtype, const GConvP& p, hipStream_t s) {
    int tm = p.Nout <= 32 ? 32 : (p.Nout <= 64 ? 64 : 128);
    if (dtype == AYOLO_F16) {
        if (tm == 32) return launch_gconv<half_t, 32>(p, s);
        if (tm == 64) return launch_gconv<half_t, 64>(p, s);
        return launch_gconv<half_t, 128>(p, s);
    } else {
        if (tm == 32) return launch_gconv<float, 32>(p, s);
        if (tm == 64) return launch_gconv<float, 64>(p, s);
        return launch_gconv<float, 128>(p, s);
    }
}

static int check_desc(const ayolo_conv_desc* d, const char* who) {
    AY_CHECK_ARG(d, "%s: null desc", who);
    AY_CHECK_ARG(d->dtype == AYOLO_F16 || d->dtype == AYOLO_F32, "%s: dtype %d", who, d->dtype);
    const int ce = d->dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(d->Cin % ce == 0 && d->ldx % ce == 0, "%s: Cin=%d ldx=%d must be multiples of %d", who, d->Cin,
                 d->ldx, ce);
    AY_CHECK_ARG(d->kh * d->kw <= MAX_TAPS && d->kh > 0 && d->kw > 0, "%s: kernel %dx%d unsupported", who, d->kh, d->kw);
    AY_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "%s: bad dims", who);
    AY_CHECK_ARG(d->ph < 64 && d->pw < 64, "%s: padding too large", who);
    AY_CHECK_ARG((long long)d->B * d->H * d->W < (1ll << 31) && (long long)d->B * d->Ho * d->Wo < (1ll << 31),
                 "%s: more than 2^31 pixels", who);
    return AYOLO_OK;
}

extern "C" int ayolo_conv_fwd(const ayolo_conv_desc* d, const void* x, const void* w, void* y, int epilogue,
                              const float* scale, const float* shift, float* stats, int stat_reps, int head_no,
                              ayolo_stream s) {
    int rc = check_desc(d, "conv_fwd");
    if (rc) return rc;
    AY_CHECK_ARG(x && w && y, "conv_fwd: null pointer");
    AY_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "conv_fwd: epilogue %d", epilogue);
    AY_CHECK_ARG(epilogue != AYOLO_EPI_HEAD || (head_no > 0 && d->Cout % head_no == 0), "conv_fwd: head_no=%d", head_no);
    AY_CHECK_ARG(stats == nullptr || epilogue == AYOLO_EPI_NONE, "conv_fwd: stats need EPI_NONE");
    GConvP p{};
    p.x = x; p.w = w; p.y = y;
    p.B = d->B; p.XH = d->H; p.XW = d->W; p.ldx = d->ldx;
    p.OH = d->Ho; p.OW = d->Wo; p.ish = d->sh; p.isw = d->sw;
    p.YH = d->Ho; p.YW = d->Wo; p.ldy = d->ldy; p.osh = 1; p.osw = 1; p.oah = 0; p.oaw = 0;
    p.C = d->Cin; p.ntaps = d->kh * d->kw; p.K = p.ntaps * p.C; p.ldw = p.K; p.Nout = d->Cout;
    p.epi = epilogue; p.scale = scale; p.shift = shift; p.stats = stats; p.head_no = head_no; p.accumulate = 0;
    p.stat_reps = stat_reps > 0 ? stat_reps : 1;
    p.y_linear = 1;
    p.x_linear = (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0) ? 1 : 0;
    p.Mtotal = (long long)d->B * d->Ho * d->Wo;
    for (int i = 0; i < d->kh; ++i)
        for (int j = 0; j < d->kw; ++j) {
            int t = i * d->kw + j;
            p.dh[t] = (signed char)(i - d->ph); p.dw[t] = (signed char)(j - d->pw); p.wt[t] = (signed char)t;
        }
    return dispatch_gconv(d->dtype, p, (hipStream_t)s);
}

// dgrad: dx[n,h,w,ci] = sum_{kh,kw,co} dy[n,(h+ph-kh)/sh,(w+pw-kw)/sw,co] * w[co,kh,kw,ci] over exact divisions.
// Each (h mod sh, w mod sw) residue class is a stride-1 gather conv over dy with its own tap subset, so no
// MFMA work is spent on structural zeros.
extern "C" int ayolo_conv_dgrad(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                                ayolo_stream s) {
    int rc = check_desc(d, "conv_dgrad");
    if (rc) return rc;
    AY_CHECK_ARG(dy && wt && dx, "conv_dgrad: null pointer");
    const int ce = d->dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(d->Cout % ce == 0 && d->ldy % ce == 0, "conv_dgrad: Cout=%d ldy=%d must be multiples of %d", d->Cout,
                 d->ldy, ce);
    for (int a = 0; a < d->sh; ++a)
        for (int b = 0; b < d->sw; ++b) {
            GConvP p{};
            p.x = dy; p.w = wt; p.y = dx;
            p.B = d->B; p.XH = d->Ho; p.XW = d->Wo; p.ldx = d->ldy;
            p.OH = (d->H - a + d->sh - 1) / d->sh; p.OW = (d->W - b + d->sw - 1) / d->sw;
            if (p.OH <= 0 || p.OW <= 0) continue;
            p.ish = 1; p.isw = 1;
            p.YH = d->H; p.YW = d->W; p.ldy = d->ldx; p.osh = d->sh; p.osw = d->sw; p.oah = a; p.oaw = b;
            p.C = d->Cout; p.ldw = d->kh * d->kw * d->Cout; p.Nout = d->Cin;
            p.epi = AYOLO_EPI_NONE; p.accumulate = accumulate; p.stat_reps = 1;
            p.y_linear = (d->sh == 1 && d->sw == 1) ? 1 : 0;
            p.x_linear = (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0) ? 1 : 0;
            p.Mtotal = (long long)d->B * p.OH * p.OW;
            int nt = 0;
            for (int i = 0; i < d->kh; ++i) {
                if ((a + d->ph - i) % d->sh != 0) continue;
                for (int j = 0; j < d->kw; ++j) {
                    if ((b + d->pw - j) % d->sw != 0) continue;
                    // floor division for negative numerators is not needed: exact multiples only
                    p.dh[nt] = (signed char)((a + d->ph - i) / d->sh);
                    p.dw[nt] = (signed char)((b + d->pw - j) / d->sw);
                    p.wt[nt] = (signed char)(i * d->kw + j);
                    ++nt;
                }
            }
            p.ntaps = nt; p.K = nt * p.C;   // nt == 0: no tap reaches this residue class -> the kernel writes zeros
            rc = dispatch_gconv(d->dtype, p, (hipStream_t)s);
            if (rc) return rc;
        }
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// wgrad
// ---------------------------------------------------------------------------------------------------
struct WGradP {
    const void* x; const void* dy; float* dw;
    int B, XH, XW, ldx, C;         // x: input activations
    int OH, OW, ldy, N;            // dy: output gradient, N = Cout
    int sh, sw, ntaps, K;          // K = ntaps*C (row length of dw)
    float alpha;
    long long P;                   // B*OH*OW
    long long chunk;               // pixels per split (multiple of 32)
    signed char dh[MAX_TAPS], dw_[MAX_TAPS];
};

#define BP_MAX 64   // pixels per reduction step: 64 (fp16) / 32 (fp32)
#define TNW 128 // dw columns per block tile

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// A/B fragments for v_mfma_f32_32x32x16_f16 out of a pixel-major LDS tile t[pixel][channel] via the gfx950
// transposing LDS read: each 16-lane group reads a 4(pixel) x 16(channel) block, lane q supplies the address of
// row q/4, channels (q%4)*4.. and receives channel q of all 4 rows.
__device__ __forceinline__ half8 tr_frag(const half_t* tile, int ldt, int k0, int c0, int lane) {
    const int q = lane & 15;
    const half_t* p0 = tile + (k0 + (q >> 2)) * ldt + c0 + (q & 3) * 4;
    fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0));
    fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0 + 4 * ldt));
    half8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <typename T, int TM>
__global__ __launch_bounds__(256) void k_wgrad(WGradP p) {
    constexpr int CE = Tr<T>::CE;
    constexpr int BP = sizeof(T) == 2 ? 64 : 32;
    constexpr int LDY = TM + Tr<T>::PADE;     // dy tile row stride
    constexpr int LDX = TNW + Tr<T>::PADE;    // x tile row stride
    constexpr int WM = TM / 32, WN = 4 / WM, NI = TNW / (32 * WN);
    constexpr int XCPR = TNW / CE, YCPR = TM / CE;
    constexpr int XR = (BP * XCPR) / 256;             // x chunks / thread (2 for f16, 4 for f32)
    constexpr int YCH = BP * YCPR;
    constexpr int YR = (YCH + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sY = reinterpret_cast<T*>(smem_raw);           // [2][BP][LDY]
    T* sX = sY + 2 * BP * LDY;                        // [2][BP][LDX]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int n0 = blockIdx.y * TM;                   // output-channel tile
    const int j0 = blockIdx.x * TNW;                  // dw column tile (tap*C + c)
    const long long pbeg = (long long)blockIdx.z * p.chunk;
    const long long pend = min(p.P, pbeg + p.chunk);
    if (pbeg >= pend) return;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

    // x loader: this thread always loads the same dw column chunk, for XR different pixel rows
    const int xcc = tid % XCPR;
    const int xcol = j0 + xcc * CE;
    const bool xcol_ok = xcol < p.K;
    const int xtap = xcol_ok ? xcol / p.C : 0;
    const int xc = xcol_ok ? xcol % p.C : 0;
    const int xdh = p.dh[xtap], xdw = p.dw_[xtap];
    const int ycc = tid % YCPR;
    const bool ycol_ok = (n0 + ycc * CE) < p.N;

    uint4 xreg[XR], yreg[YR];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // pixel coordinates of this thread's x rows, advanced incrementally (no divisions in the loop)
    int pn[XR], poh[XR], pow_[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        long long pp = pbeg + (tid + 256 * r) / XCPR;
        pow_[r] = (int)(pp % p.OW);
        long long t = pp / p.OW;
        poh[r] = (int)(t % p.OH);
        pn[r] = (int)(t / p.OH);
    }

    float16v acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int nk = (int)((pend - pbeg + BP - 1) / BP);
    for (int kt = -1; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            const long long pt = pbeg + (long long)(kt + 1) * BP;
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                int prow = (tid + 256 * r) / XCPR;
                long long pp = pt + prow;
                xreg[r] = zero4;
                if (pp < pend && xcol_ok) {
                    int ih = poh[r] * p.sh + xdh, iw = pow_[r] * p.sw + xdw;
                    if (ih >= 0 && ih < p.XH && iw >= 0 && iw < p.XW)
                        xreg[r] = *reinterpret_cast<const uint4*>(X + (((long long)pn[r] * p.XH + ih) * p.XW + iw) * p.ldx + xc);
                }
                pow_[r] += BP;
                while (pow_[r] >= p.OW) { pow_[r] -= p.OW; ++poh[r]; }
                while (poh[r] >= p.OH) { poh[r] -= p.OH; ++pn[r]; }
            }
#pragma unroll
            for (int r = 0; r < YR; ++r) {
                int q = tid + 256 * r;
                int prow = q / YCPR;
                long long pp = pt + prow;
                yreg[r] = zero4;
                if (q < YCH && pp < pend && ycol_ok)
                    yreg[r] = *reinterpret_cast<const uint4*>(DY + pp * p.ldy + n0 + ycc * CE);
            }
        }
        if (kt >= 0) {
            const int buf = kt & 1;
            const T* cY = sY + buf * BP * LDY;
            const T* cX = sX + buf * BP * LDX;
#pragma unroll
            for (int kk = 0; kk < BP / 16; ++kk) {
                const int k0 = kk * 16 + (lane >> 5) * 8;
                if constexpr (sizeof(T) == 2) {
                    const int csub = ((lane >> 4) & 1) * 16;
                    half8 a = tr_frag(cY, LDY, k0, wm * 32 + csub, lane);
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        half8 b = tr_frag(cX, LDX, k0, wn * NI * 32 + ni * 32 + csub, lane);
                        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ni], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) {
                        float a = cY[(k0 + s8) * LDY + wm * 32 + (lane & 31)];
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            float b = cX[(k0 + s8) * LDX + wn * NI * 32 + ni * 32 + (lane & 31)];
                            acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ni], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (more) {
            const int nb = (kt + 1) & 1;
            T* dX = sX + nb * BP * LDX;
            T* dY = sY + nb * BP * LDY;
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                int prow = (tid + 256 * r) / XCPR;
                *reinterpret_cast<uint4*>(dX + prow * LDX + xcc * CE) = xreg[r];
            }
#pragma unroll
            for (int r = 0; r < YR; ++r) {
                int q = tid + 256 * r;
                if (q < YCH) *reinterpret_cast<uint4*>(dY + (q / YCPR) * LDY + ycc * CE) = yreg[r];
            }
        }
        __syncthreads();
    }
    // acc[ni][r]: row (out channel) = n0 + wm*32 + 8*(r>>2) + 4*(lane>>5) + (r&3); col = j0 + wn*NI*32 + ni*32 + (lane&31)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        int col = j0 + wn * NI * 32 + ni * 32 + (lane & 31);
        if (col >= p.K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = n0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row < p.N) unsafeAtomicAdd(&p.dw[(long long)row * p.K + col], acc[ni][r] * p.alpha);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Halo-tiled wgrad for multi-tap stride-1 convs (fp16).  The generic k_wgrad re-reads dy once per 128-column tile
// of dw and x once per tap (PMC: 2.4x the algorithmic HBM bytes over a training step).  Here a workgroup owns
// (TM output channels) x (one 32-channel chunk of the input) x ALL taps: per 8x16 pixel tile it stages the dy
// tile and the x patch (with halo) in LDS once and every tap's MFMAs read shifted fragments of the same patch
// through the transposing LDS read.
// ---------------------------------------------------------------------------------------------------
#define WD_MAXT 9
struct WGradDP {
    WGradP w;
    int tyn, txn, PH, PW, dh_min, dw_min;
    long long tiles_total, tiles_per_split;
};

template <int TM>
__global__ __launch_bounds__(256) void k_wgrad_d(WGradDP dp) {
    typedef half_t T;
    const WGradP& p = dp.w;
    constexpr int LDY = TM + 8;
    constexpr int LDP = DCK + 8;
    constexpr int WM = TM / 32, WT = 4 / WM;          // waves along channels / along taps
    constexpr int TPW = (WD_MAXT + WT - 1) / WT;      // taps per wave (upper bound)
    constexpr int YCH = TP * (TM / 8);                // dy chunks per tile
    constexpr int YR = YCH / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sY = reinterpret_cast<T*>(smem_raw);           // [128][LDY]
    T* sP = sY + TP * LDY;                            // [PH*PW][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wt = wave / WM;
    const int chunk = blockIdx.x, n0 = blockIdx.y * TM;
    const long long t_beg = (long long)blockIdx.z * dp.tiles_per_split;
    const long long t_end = min(dp.tiles_total, t_beg + dp.tiles_per_split);
    if (t_beg >= t_end) return;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);
    const int tiles_per_img = dp.tyn * dp.txn;
    const int patch_chunks = dp.PH * dp.PW * 4;

    float16v acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    const int q = lane & 15;
    const int csub = ((lane >> 4) & 1) * 16;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    for (long long tile = t_beg; tile < t_end; ++tile) {
        const int n = (int)(tile / tiles_per_img);
        const int rr = (int)(tile - (long long)n * tiles_per_img);
        const int ty = rr / dp.txn, tx = rr - ty * dp.txn;
        // ---- stage dy tile: pixel j = (oy, ox), TM channels
#pragma unroll
        for (int r = 0; r < YR; ++r) {
            const int qq = tid + 256 * r;
            const int j = qq / (TM / 8), c8 = qq % (TM / 8);
            const int oh = ty * DTH + j / DTW, ow = tx * DTW + j % DTW;
            uint4 v = zero4;
            if (oh < p.OH && ow < p.OW && (n0 + c8 * 8) < p.N)
                v = *reinterpret_cast<const uint4*>(DY + (((long long)n * p.OH + oh) * p.OW + ow) * p.ldy + n0 + c8 * 8);
            *reinterpret_cast<uint4*>(sY + j * LDY + c8 * 8) = v;
        }
        // ---- stage x patch (32 channels of this chunk, halo included)
        const int ih0 = ty * DTH * p.sh + dp.dh_min, iw0 = tx * DTW * p.sw + dp.dw_min;
        for (int qq = tid; qq < patch_chunks; qq += 256) {
            const int pix = qq >> 2, kc4 = qq & 3;
            const int py = pix / dp.PW, px = pix - py * dp.PW;
            const int ih = ih0 + py, iw = iw0 + px;
            uint4 v = zero4;
            if (ih >= 0 && ih < p.XH && iw >= 0 && iw < p.XW)
                v = *reinterpret_cast<const uint4*>(X + (((long long)n * p.XH + ih) * p.XW + iw) * p.ldx + chunk * DCK + kc4 * 8);
            *reinterpret_cast<uint4*>(sP + pix * LDP + kc4 * 8) = v;
        }
        __syncthreads();
        // ---- MFMAs: k = the 128 pixels of the tile, 16 per step (one tile row oy = kk)
#pragma unroll
        for (int kk = 0; kk < TP / 16; ++kk) {
            const int k0 = kk * 16 + (lane >> 5) * 8;
            half8 a = tr_frag(sY, LDY, k0, wm * 32 + csub, lane);
            const int ox0 = (lane >> 5) * 8 + (q >> 2);            // first of this lane's 2 x 4 pixel rows
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) {
                const int t = wt + ti * WT;
                if (t < p.ntaps) {
                    const int dh = p.dh[t] - dp.dh_min, dw = p.dw_[t] - dp.dw_min;
                    const T* b0 = sP + ((kk * p.sh + dh) * dp.PW + (ox0 * p.sw + dw)) * LDP + csub + (q & 3) * 4;
                    fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(b0));
                    fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(b0 + 4 * p.sw * LDP));
                    half8 b;
                    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
                    b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
                    acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ti], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- acc[ti][r]: out channel = n0 + wm*32 + 8*(r>>2) + 4*(lane>>5) + (r&3); dw column = t*C + chunk*32 + (lane&31)
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int t = wt + ti * WT;
        if (t >= p.ntaps) continue;
        const int col = t * p.C + chunk * DCK + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = n0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row < p.N) unsafeAtomicAdd(&p.dw[(long long)row * p.K + col], acc[ti][r] * p.alpha);
        }
    }
}

template <int TM>
static int launch_wgrad_d(const WGradDP& dp, dim3 grid, hipStream_t s) {
    size_t lds = (size_t)(TP * (TM + 8) + dp.PH * dp.PW * (DCK + 8)) * sizeof(half_t);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_d<TM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL((k_wgrad_d<TM>), grid, dim3(256), lds, s, dp);
    AY_CHECK_LAUNCH("k_wgrad_d");
    return AYOLO_OK;
}

// returns 1 when the halo-tiled kernel was launched, 0 when it does not apply, < 0 on error
static int try_wgrad_d(const ayolo_conv_desc* d, const WGradP& p, hipStream_t s) {
    static const bool disabled = getenv("AYOLO_NO_WGRAD_D") != nullptr;
    // one workgroup per 32-channel chunk re-reads dy C/32 times: only a win for single-chunk layers (measured)
    if (disabled || d->dtype != AYOLO_F16 || p.ntaps < 2 || p.ntaps > WD_MAXT || p.C != DCK) return 0;
    if (p.sh != 1 || p.sw != 1 || p.OH < DTH || p.OW < DTW || p.ldy % 8 != 0) return 0;
    WGradDP dp;
    dp.w = p;
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int t = 0; t < p.ntaps; ++t) {
        dh_min = p.dh[t] < dh_min ? p.dh[t] : dh_min; dh_max = p.dh[t] > dh_max ? p.dh[t] : dh_max;
        dw_min = p.dw_[t] < dw_min ? p.dw_[t] : dw_min; dw_max = p.dw_[t] > dw_max ? p.dw_[t] : dw_max;
    }
    dp.tyn = (p.OH + DTH - 1) / DTH; dp.txn = (p.OW + DTW - 1) / DTW;
    const double util = (double)p.OH * p.OW / ((double)dp.tyn * DTH * dp.txn * DTW);
    if (util < 0.65) return 0;
    dp.PH = (DTH - 1) * p.sh + (dh_max - dh_min) + 1;
    dp.PW = (DTW - 1) * p.sw + (dw_max - dw_min) + 1;
    dp.dh_min = dh_min; dp.dw_min = dw_min;
    dp.tiles_total = (long long)p.B * dp.tyn * dp.txn;
    const int tm = p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128);
    const long long nxy = (long long)(p.C / DCK) * ((p.N + tm - 1) / tm);
    long long splits = (2048 + nxy - 1) / nxy;
    if (splits > dp.tiles_total / 2) splits = dp.tiles_total / 2;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    dp.tiles_per_split = (dp.tiles_total + splits - 1) / splits;
    splits = (dp.tiles_total + dp.tiles_per_split - 1) / dp.tiles_per_split;
    dim3 grid((unsigned)(p.C / DCK), (unsigned)((p.N + tm - 1) / tm), (unsigned)splits);
    int rc = tm == 32 ? launch_wgrad_d<32>(dp, grid, s) : (tm == 64 ? launch_wgrad_d<64>(dp, grid, s) : launch_wgrad_d<128>(dp, grid, s));
    return rc == AYOLO_OK ? 1 : rc;
}

template <typename T, int TM>
static int launch_wgrad(const WGradP& p, int splits, hipStream_t s) {
    constexpr int LDY = TM + Tr<T>::PADE, LDX = TNW + Tr<T>::PADE;
    constexpr int BP = sizeof(T) == 2 ? 64 : 32;
    size_t lds = (size_t)2 * BP * (LDY + LDX) * sizeof(T);
    dim3 grid((unsigned)((p.K + TNW - 1) / TNW), (unsigned)((p.N + TM - 1) / TM), (unsigned)splits);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, TM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_wgrad<T, TM>), grid, dim3(256), lds, s, p);
    AY_CHECK_LAUNCH("k_wgrad");
    return AYOLO_OK;
}

extern "C" int ayolo_conv_wgrad(const ayolo_conv_desc* d, const void* x, const void* dy, float* dw, float alpha,
                                ayolo_stream s) {
    int rc = check_desc(d, "conv_wgrad");
    if (rc) return rc;
    AY_CHECK_ARG(x && dy && dw, "conv_wgrad: null pointer");
    const int ce = d->dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(d->ldy % ce == 0, "conv_wgrad: ldy=%d must be a multiple of %d", d->ldy, ce);
    WGradP p{};
    p.x = x; p.dy = dy; p.dw = dw;
    p.B = d->B; p.XH = d->H; p.XW = d->W; p.ldx = d->ldx; p.C = d->Cin;
    p.OH = d->Ho; p.OW = d->Wo; p.ldy = d->ldy; p.N = d->Cout;
    p.sh = d->sh; p.sw = d->sw; p.ntaps = d->kh * d->kw; p.K = p.ntaps * p.C; p.alpha = alpha;
    p.P = (long long)d->B * d->Ho * d->Wo;
    for (int i = 0; i < d->kh; ++i)
        for (int j = 0; j < d->kw; ++j) {
            p.dh[i * d->kw + j] = (signed char)(i - d->ph);
            p.dw_[i * d->kw + j] = (signed char)(j - d->pw);
        }
    {
        int rd = try_wgrad_d(d, p, (hipStream_t)s);
        if (rd != 0) return rd < 0 ? rd : AYOLO_OK;
    }
    const int tm = p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128);
    const long long tiles = (long long)((p.K + TNW - 1) / TNW) * ((p.N + tm - 1) / tm);
    // split the pixel reduction so that ~4 blocks per CU are in flight, each with >= 8 reduction steps
    long long want = (1024 + tiles - 1) / tiles;
    long long max_splits = (p.P + 8 * BP_MAX - 1) / (8 * BP_MAX);
    long long splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    long long chunk = (p.P + splits - 1) / splits;
    chunk = (chunk + BP_MAX - 1) / BP_MAX * BP_MAX;
    splits = (p.P + chunk - 1) / chunk;
    p.chunk = chunk;
    hipStream_t st = (hipStream_t)s;
    if (d->dtype == AYOLO_F16) {
        if (tm == 32) return launch_wgrad<half_t, 32>(p, (int)splits, st);
        if (tm == 64) return launch_wgrad<half_t, 64>(p, (int)splits, st);
        return launch_wgrad<half_t, 128>(p, (int)splits, st);
    } else {
        if (tm == 32) return launch_wgrad<float, 32>(p, (int)splits, st);
        if (tm == 64) return launch_wgrad<float, 64>(p, (int)splits, st);
        return launch_wgrad<float, 128>(p, (int)splits, st);
    }
}

// ---------------------------------------------------------------------------------------------------
// weight cast: fp32 [Cout][taps][Cin] -> T [Cout_pad][taps][Cin_pad] and transposed T [Cin_pad][taps][Cout_pad]
// (padding rows / channels are zero)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_cast_weight(const float* w32, int Cout, int taps, int Cin, int Cout_pad, int Cin_pad, T* w, T* wt) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)Cout_pad * taps * Cin_pad;
    if (t >= tot) return;
    int c = (int)(t % Cin_pad);
    long long r = t / Cin_pad;
    int tap = (int)(r % taps);
    int co = (int)(r / taps);
    float v = (c < Cin && co < Cout) ? w32[((long long)co * taps + tap) * Cin + c] : 0.0f;
    if (w) w[t] = (T)v;
    if (wt) wt[((long long)c * taps + tap) * Cout_pad + co] = (T)v;
}

extern "C" int ayolo_cast_weight(const float* w32, int Cout, int kh, int kw, int Cin, int Cout_pad, int Cin_pad, int dtype,
                                 void* w, void* wt, ayolo_stream s) {
    AY_CHECK_ARG(w32 && (w || wt), "cast_weight: null pointer");
    AY_CHECK_ARG(Cin_pad >= Cin && Cout_pad >= Cout, "cast_weight: pad < size");
    long long tot = (long long)Cout_pad * kh * kw * Cin_pad;
    dim3 grid((unsigned)cdiv64(tot, 256));
    if (dtype == AYOLO_F16)
        hipLaunchKernelGGL(k_cast_weight<half_t>, grid, dim3(256), 0, (hipStream_t)s, w32, Cout, kh * kw, Cin, Cout_pad,
                           Cin_pad, (half_t*)w, (half_t*)wt);
    else
        hipLaunchKernelGGL(k_cast_weight<float>, grid, dim3(256), 0, (hipStream_t)s, w32, Cout, kh * kw, Cin, Cout_pad,
                           Cin_pad, (float*)w, (float*)wt);
    AY_CHECK_LAUNCH("k_cast_weight");
    return AYOLO_OK;
}
