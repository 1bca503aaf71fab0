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
template <typename T, int TM, int EM>
__global__ __launch_bounds__(256) void k_gconv(GConvP p) {
    // Persistent, tile-pipelined implicit GEMM.  A workgroup owns output-channel tile `nt` and walks pixel tiles
    // tile0, tile0+stride, ...; the global loads of step (tile, kt)+1 are in flight while the MFMAs consume
    // step (tile, kt) from LDS, across tile boundaries, so short-K layers (1x1 convs, K = 32..128) stream
    // instead of paying a load->use latency per 128-pixel tile.  BN statistics are accumulated in registers
    // across all tiles of the workgroup and reduced once at the end.
    constexpr int CE = Tr<T>::CE;
    constexpr int CPR = BK / CE;             // chunks per tile row
    constexpr int LDR = BK + Tr<T>::PADE;    // LDS row stride (elements)
    constexpr int WM = TM / 32;              // waves along channels
    constexpr int WP = 4 / WM;               // waves along pixels
    constexpr int NI = TP / (32 * WP);       // 32-pixel MFMA tiles per wave
    constexpr int XR = (TP * CPR) / 256;     // x chunks per thread
    constexpr int WCH = TM * CPR;            // w chunks per tile
    constexpr int WR = (WCH + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sW = reinterpret_cast<T*>(smem_raw);                 // [2][TM][LDR]
    T* sX = sW + 2 * TM * LDR;                              // [2][TP][LDR]
    float* sStat = reinterpret_cast<float*>(sX + 2 * TP * LDR);   // [2][TM]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wp = wave / WM;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(p.w);

    // ---- block -> (channel tile, first pixel tile, tile stride).  Workgroups are dealt round-robin to the 8 XCDs;
    // all channel tiles of one pixel tile are given to the SAME XCD, back to back, so the gathered input tile is
    // fetched into that XCD's L2 once.
    const unsigned L = blockIdx.x;
    const unsigned xcd = L & 7u, idx = L >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;     // pixel-tile slot of this workgroup
    const unsigned nslots = (unsigned)p.nslots;                   // = gridDim.x / ntn
    const int n0 = (int)nt * TM;
    const long long ntiles = (p.Mtotal + TP - 1) / TP;

    typedef __attribute__((address_space(4))) const signed char* kptr_t;
    const kptr_t ktab = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(GConvP, dh);

    // ---- per-thread loader state (for the tile being LOADED)
    const int kc = tid % CPR;
    int tap = 0, cch = 0;
    long long xbase[XR];
    int xh0[XR], xw0[XR];
    uint4 xreg[XR], wreg[WR];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto setup_rows = [&](long long tile) {
        tap = (kc * CE) / p.C;
        cch = (kc * CE) % p.C;
        const long long m0 = tile * TP;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int row = (tid + 256 * r) / CPR;
            const long long m = m0 + row;
            if (m < p.Mtotal) {
                if (p.x_linear) {                     // 1x1 / stride 1: input pixel == output pixel
                    xbase[r] = m; xh0[r] = 0; xw0[r] = 0;
                } else {
                    const unsigned mu = (unsigned)m;
                    unsigned t = mu / (unsigned)p.OW;
                    int ow = (int)(mu - t * (unsigned)p.OW);
                    unsigned n = t / (unsigned)p.OH;
                    int oh = (int)(t - n * (unsigned)p.OH);
                    xbase[r] = (long long)n * p.XH * p.XW;
                    xh0[r] = oh * p.ish; xw0[r] = ow * p.isw;
                }
            } else { xbase[r] = -1; xh0[r] = 0; xw0[r] = 0; }
        }
    };

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

    int nk = (p.K + BK - 1) / BK;
    if (nk < 1) nk = 1;                      // K == 0 (tap-less dgrad residue class): one all-zero step
    long long cur_tile = slot;
    if (cur_tile >= ntiles) return;
    long long ld_tile = cur_tile;
    int ld_kt = 0, cur_kt = 0, buf = 0;
    bool first = true;
    while (true) {
        // ---------------- issue the loads of the NEXT step (or of the very first one)
        bool more;
        if (first) {
            more = true;
            setup_rows(ld_tile);
        } else {
            ++ld_kt;
            if (ld_kt == nk) { ld_kt = 0; ld_tile += nslots; }
            more = ld_tile < ntiles;
            if (more) {
                if (ld_kt == 0) setup_rows(ld_tile);
                else {
                    cch += BK;
                    while (cch >= p.C) { cch -= p.C; ++tap; }
                }
            }
        }
        if (more) {
            const bool tap_ok = tap < p.ntaps;
            const int tq = tap_ok ? tap : 0;
            const int dh = ktab[tq], dw = ktab[MAX_TAPS + tq];
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                xreg[r] = zero4;
                if (p.x_linear) {
                    if (tap_ok && xbase[r] >= 0) xreg[r] = *reinterpret_cast<const uint4*>(X + xbase[r] * p.ldx + cch);
                } else {
                    int ih = xh0[r] + dh, iw = xw0[r] + dw;
                    bool ok = tap_ok && xbase[r] >= 0 && ih >= 0 && ih < p.XH && iw >= 0 && iw < p.XW;
                    if (ok) xreg[r] = *reinterpret_cast<const uint4*>(X + (xbase[r] + (long long)ih * p.XW + iw) * p.ldx + cch);
                }
            }
            const int wcol = tap_ok ? ktab[2 * MAX_TAPS + tq] * p.C + cch : 0;
#pragma unroll
            for (int r = 0; r < WR; ++r) {
                int q = tid + 256 * r;
                int row = q / CPR;
                wreg[r] = zero4;
                if (q < WCH && tap_ok && (n0 + row) < p.Nout)
                    wreg[r] = *reinterpret_cast<const uint4*>(Wg + (long long)(n0 + row) * p.ldw + wcol);
            }
        }
        // ---------------- MFMAs of the current step
        if (!first) {
            const T* cW = sW + buf * TM * LDR + (wm * 32 + (lane & 31)) * LDR + (lane >> 5) * 8;
            const T* cX = sX + buf * TP * LDR + (wp * NI * 32 + (lane & 31)) * LDR + (lane >> 5) * 8;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                auto a = lds_frag(cW + kk * 16);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    auto b = lds_frag(cX + ni * 32 * LDR + kk * 16);
                    mma_step(a, b, acc[ni]);
                }
            }
            // ------------ tile finished: epilogue.  acc[ni][r]: channel = cbase + 8*(r>>2) + (r&3), pixel = .. + (lane&31)
            if (cur_kt == nk - 1) {
                const long long m0 = cur_tile * TP;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const long long m = m0 + wp * NI * 32 + ni * 32 + (lane & 31);
                    const bool pv = m < p.Mtotal;
                    long long yo = 0;
                    int n = 0, oh = 0, ow = 0;
                    if (pv) {
                        if (p.y_linear) yo = m * p.ldy;
                        else {
                            const unsigned mu = (unsigned)m;
                            unsigned t = mu / (unsigned)p.OW;
                            ow = (int)(mu - t * (unsigned)p.OW);
                            unsigned nn = t / (unsigned)p.OH;
                            oh = (int)(t - nn * (unsigned)p.OH);
                            n = (int)nn;
                            yo = (((long long)n * p.YH + (oh * p.osh + p.oah)) * p.YW + (ow * p.osw + p.oaw)) * p.ldy;
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
        }
        // ---------------- stage the prefetched step into the other LDS buffer
        const int nb = first ? 0 : (buf ^ 1);
        if (more) {
            T* dX = sX + nb * TP * LDR;
            T* dW = sW + nb * TM * LDR;
#pragma unroll
            for (int r = 0; r < XR; ++r) {
                int row = (tid + 256 * r) / CPR;
                *reinterpret_cast<uint4*>(dX + row * LDR + kc * CE) = xreg[r];
            }
#pragma unroll
            for (int r = 0; r < WR; ++r) {
                int q = tid + 256 * r;
                if (q < WCH) *reinterpret_cast<uint4*>(dW + (q / CPR) * LDR + kc * CE) = wreg[r];
            }
        }
        __syncthreads();
        if (!more) break;
        if (first) { first = false; cur_tile = ld_tile; cur_kt = 0; buf = 0; }
        else {
            ++cur_kt;
            if (cur_kt == nk) { cur_kt = 0; cur_tile += nslots; }
            buf ^= 1;
        }
    }

    if (want_stats) {
        for (int i = tid; i < 2 * TM; i += 256) sStat[i] = 0.0f;
        __syncthreads();
        // one reduction per workgroup: over the 32 pixel lanes of each half-wave, then LDS, then one replica
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
    long long want_slots = (long long)num_cus() * 4 / p.ntn;
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

template <typename T, int TM>
static int launch_gconv(const GConvP& p, hipStream_t s) {
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
