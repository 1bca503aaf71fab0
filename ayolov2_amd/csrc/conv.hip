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
// Tiles: TM in {32,64,128} output channels x 128 pixels x BK=32 (gconv) / TM x 128 dw columns x 32 pixels (wgrad),
// 4 wavefronts, operands staged global -> LDS by LDS-DMA into three XOR-swizzled stages (see k_gconv).
//
// Replaces kindle Conv/YOLOHead.conv forward (yolov5s.yaml:21-57) and the autograd backward torch/cuDNN ran
// (scripts/train/yolo_trainer.py:329).
#include "common.h"
#include "gfx950_dma.h"
#include "wgrad3.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

#define MAX_TAPS 36
#define BK 32

struct GConvP {
    const void* x; const void* w; void* y;
    int B, XH, XW, ldx;
    int OH, OW, ish, isw;
    int YH, YW, ldy, osh, osw, oah, oaw;
    int C, ntaps, K, ldw, Nout;
    int epi; const float* scale; const float* shift; double* stats;
    int head_no; int accumulate; int stat_reps;
    int x_linear, y_linear, ntn, nslots;
    // 1x1 / stride 1 / no padding with whole 32-channel chunks (C % 32 == 0) and the weight tap at column 0: the step loop's
    // address work is ONE add per DMA piece (row offset + k offset) -- no tap decode, no tap-table read, no halo test
    int lin;
    long long Mtotal;
    signed char dh[MAX_TAPS], dw[MAX_TAPS], wt[MAX_TAPS];
    unsigned x_bytes, w_bytes, y_bytes;      // extents for the buffer descriptors (k_gconv)
    FastDiv dOW, dOH, dC;
    // "classes": sub-problems that share the pixel tiles and the x rows but have their own tap subset and output
    // offset -- the residue classes of a strided dgrad walked back to back by the same workgroup (ncls == 1 otherwise)
    int ncls;
    int ctap0[4], cnt[4], coah[4], coaw[4];  // first tap / tap count / output offsets of each class
    // k_gconv3 (3x3, stride 1, same-size maps): weight tap index of (dh, dw) = (g - 1, j - 1) at [g * 3 + j]
    int row3;
    signed char r3wt[9];
    // k_dgrad_s2 (dgrad of a 3x3 / stride 2 / pad 1 conv): weight tap of the nine (shift, class) products in the order
    // [A: shift (0,0) classes 0,1,2,3 | B: shift (0,1) classes 1,3 | C: shift (1,0) classes 2,3 | D: shift (1,1) class 3]
    int s2d;
    signed char s2wt[9];
    // k_gconv_s2f (forward 3x3 / stride 2 / pad 1): weight tap of (dh, dw) = (g - 1, j - 1) at [g * 3 + j]
    int s2f;
    signed char f2wt[9];
    // BN-backward statistics in the epilogue (BNR kernels, fp16 dgrad): the output y is the gradient da of up to two
    // Conv-BN-act blocks side by side in channels (a concat buffer); for each, z / saved statistics / affine parameters and
    // the [reps][2][C] accumulators of sum(du), sum(du * xhat) that ayolo_bn_act_bwd_reduce would fill
    int bnr, bnr_act, bnr_reps;
    ayolo_bn_seg bseg[2];
    unsigned z_bytes[2];
    // XF kernels (transform on load, ayolo_conv_fwd_xf): x is the producer's pre-activation z; the operand the MFMAs see is
    // act(z * xf_scale[c] + xf_shift[c]) -- the BatchNorm + SiLU pass that would have materialised it does not exist
    // Up to TWO input segments side by side in the conv's channels (C3's cv3 reads [last Bottleneck output | cv2 half]): segment
    // 1 = channels [xs_split, C) from `x2` (channel stride ldx2); each segment is virtual (transformed, activation xf_act bit) or a
    // plain materialised activation (xf_virt bit clear: copied as it lies)
    int xf, xf_act, xf_virt;         // bit 0: segment 0, bit 1: segment 1
    const float* xf_scale; const float* xf_shift;
    const void* x2; int ldx2, xs_split; unsigned x2_bytes;
    // store-back: the workgroups of channel tile 0 also WRITE the transformed chunks to the materialised activation `xa` (channel
    // stride ldxa, the conv's input channels side by side) -- the weight gradient then reads a plain operand
    void* xa; int ldxa; unsigned xa_bytes;
    // the virtual segments' BatchNorm finalize inside this kernel (nfin > 0): every workgroup derives scale / shift of the channels
    // from the producer's batch statistics (the expression sequence of k_bn_finalize), workgroup 0 also writes them to
    // xf_scale / xf_shift (the weight gradient reads them later) and the saved / running statistics
    int nfin;
    ayolo_xf_fin fin[2];
};

template <typename T> struct Tr;
template <> struct Tr<float> { struct frag { float v[8]; }; };   // 8 k-values per lane: 8 v_mfma_f32_32x32x2_f32

__device__ __forceinline__ void mma_step(const half8& a, const half8& b, float16v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const Tr<float>::frag& a, const Tr<float>::frag& b, float16v& acc) {
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
}
// SiLU of the inference epilogue: fp16 outputs use v_exp_f32 + v_rcp_f32 (1 ulp, far below the fp16 rounding of the
// result; silu_f's IEEE division costs ~10 VALU ops per element), the fp32 parity mode keeps expf + true division
template <typename T> __device__ __forceinline__ float silu_e(float u) {
    if constexpr (sizeof(T) == 2) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
    else return silu_f(u);
}
__device__ __forceinline__ float cvt_round(float v, half_t*) { return (float)(half_t)v; }
__device__ __forceinline__ float cvt_round(float v, float*) { return v; }

// ---------------------------------------------------------------------------------------------------
// k_gconv: persistent implicit GEMM, operands staged by LDS-DMA (buffer_load ... lds, 16 B per lane).
//
// EM (epilogue mode, compile time so that unused paths cost no registers):
//   0 plain store (+ optional BN statistics)   1 accumulate into y (dgrad)   2 affine / affine+SiLU   3 YOLOHead fp32
//   4 affine / affine+SiLU added onto y (in-place Bottleneck shortcut of the inference executor)
//
// A workgroup owns output-channel tile `nt` and walks the pixel tiles of its XCD's band; one "step" = (pixel tile,
// 32-wide k slice).  The tiles of step s+2 are in flight (global -> LDS, no VGPR staging, no ds_write pass) while the
// MFMAs consume step s, across tile boundaries: three LDS stages, ONE raw s_barrier per step and a counted
// s_waitcnt vmcnt(N) that never drains the queue.  Every load is unconditional: halo / padding / out-of-range rows get
// the buffer descriptor's out-of-range offset (hardware returns 0), so the instruction stream has no branch around a
// load and no vmcnt(0) in the loop (the branchy loader this replaces waited for every load it issued).
// The LDS image is lane-linear (what the DMA writes): row-major [row][BK] with the 16-byte chunk position XOR-swizzled
// by the row ((row / rows-per-256B) & (chunks-1)), applied on the SOURCE address, which makes the ds_read_b128 MFMA
// fragment reads bank-conflict free without padding.
// BN statistics are accumulated in registers across all tiles of the workgroup and reduced once at the end.
// ---------------------------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row, left in every lane of the row (4 VALU ops, no LDS traffic)
__device__ __forceinline__ float row16_sum(float v) {
#define AY_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    AY_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
    AY_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
    AY_DPP_ADD(0x141);   // row_half_mirror
    AY_DPP_ADD(0x140);   // row_mirror
#undef AY_DPP_ADD
    return v;
}

// Reduce-scatter of the per-lane BatchNorm partials over the 32 pixel lanes of a half wavefront.  Every lane holds NV = 16 * MI
// partial sums (value r = one channel of its half); wanted: for every r the total over the half's 32 lanes.  The epilogues used
// to reduce EVERY value over each 16-lane row (4 DPP adds per value) and send it to LDS with an fp64 atomic from 4 (2) active lanes:
// 2 * NV atomic instructions per wavefront and tile, ~50 cycles each with the four wavefronts of a workgroup arriving together
// -- 4 400 cycles per tile epilogue for the 128-channel tiles against 1 100 for the DPP sums alone
// (tools/experiments/lds_atomic_bench.hip, profiles/r05_lds_atomic_bench.txt).  A butterfly that HALVES the number of values at
// every level (the two lanes of a pair keep one value each) needs fewer adds than that, leaves value rs_index(lane) in each lane
// and the whole reduction leaves through ONE atomic instruction with all lanes active.  Level 16: v_permlane16_swap (odd rows of
// the first operand <-> even rows of the second, new on gfx950); levels 8 / 4 / 2 / 1: DPP row_ror:8, row_half_mirror, quad_perm.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float rs_step(float s0, float s1, bool hi) {
    const float keep = hi ? s1 : s0, oth = hi ? s0 : s1;
    return keep + dpp_f<CTRL>(oth);
}
template <int NV>
__device__ __forceinline__ float half_reduce_scatter(float (&v)[NV], int lq) {
    static_assert(NV == 16 || NV == 32, "16 values per 32-channel block of the wave tile");
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) {
        const v2u32 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2 * j]), __float_as_uint(v[2 * j + 1]), false, false);
        v[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);          // even rows: value 2j, odd rows: value 2j + 1
    }
    const bool b3 = (lq & 8) != 0, b2 = (lq & 4) != 0, b1 = (lq & 2) != 0, b0 = (lq & 1) != 0;
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) v[j] = rs_step<0x128>(v[2 * j], v[2 * j + 1], b3);       // row_ror:8      lane ^ 8
#pragma unroll
    for (int j = 0; j < NV / 8; ++j) v[j] = rs_step<0x141>(v[2 * j], v[2 * j + 1], b2);       // row_half_mirror: lane ^ 7 (flips bit 2)
#pragma unroll
    for (int j = 0; j < NV / 16; ++j) v[j] = rs_step<0x4E>(v[2 * j], v[2 * j + 1], b1);       // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (NV == 32) v[0] = rs_step<0xB1>(v[0], v[1], b0);                              // quad_perm [1,0,3,2]: lane ^ 1
    else v[0] += dpp_f<0xB1>(v[0]);                                    // 16 values: the lanes of a pair end with the same total
    return v[0];
}
// which value a lane holds afterwards (NV = 16: both lanes of a pair hold it -- the even one reports)
template <int NV>
__device__ __forceinline__ int rs_index(int lq) {
    int r = ((lq >> 4) & 1) + 2 * ((lq >> 3) & 1) + 4 * ((lq >> 2) & 1) + 8 * ((lq >> 1) & 1);
    if constexpr (NV == 32) r += 16 * (lq & 1);
    return r;
}
template <int NV> __device__ __forceinline__ bool rs_reports(int lq) { return NV == 32 || (lq & 1) == 0; }

#define GNS 3                      // LDS stages

// Timeline probe (tools/gconv_probe.py; built only with -DAYOLO_PROBE into ab/libayolo_probe.so, never into the product
// library): wave 0 of the first 512 workgroups records s_memtime at the marks of its first tile's step loop.
#ifdef AYOLO_PROBE
#define AY_PROBE_N 96
__device__ unsigned long long g_probe[512 * AY_PROBE_N];
extern "C" int ayolo_probe_read(void* dst, unsigned long long bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#define AY_PROBE(k_)                                                                                   \
    do {                                                                                               \
        if (probe_on && (k_) < AY_PROBE_N) {                                                           \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
            if (threadIdx.x == 0) s_probe[(k_)] = t_;   /* LDS: a global store would sit in the vmcnt queue */ \
        }                                                                                              \
    } while (0)
#else
#define AY_PROBE(k_) do { } while (0)
#endif
template <typename T, int TM, int TPX = 128, int NWX = 4>
struct GT {
    static constexpr int ES = sizeof(T);
    static constexpr int CE = 16 / ES;              // elements per 16-byte chunk
    static constexpr int ROWB = BK * ES;            // bytes per tile row: 64 / 128
    static constexpr int CPR = ROWB / 16;           // chunks per row: 4 / 8
    static constexpr int RPB = 256 / ROWB;          // rows per 256-byte bank row: 4 / 2
    static constexpr int RW = 1024 / ROWB;          // rows written by one wave-instruction: 16 / 8
    // pixels per block tile: 128, or 256 for the narrow channel tiles on multi-step reductions (K >= 128) so that a
    // wave still has 4-8 MFMAs per barrier
    static constexpr int TP = TPX;
    static constexpr int NW = NWX;                  // wavefronts per workgroup: 4, or 8 (k_gconv on the small maps, round 6)
    static constexpr int NT = NW * 64;
    static constexpr int PIECE = NW * 1024;         // bytes one DMA instruction of every wave covers
    static constexpr int XSTAGE = TP * ROWB;        // 8-16 KiB / 16-32 KiB
    static constexpr int XR = XSTAGE / PIECE;       // x DMA instructions per thread per step
    static constexpr int WSTAGE = TM * ROWB < PIECE ? PIECE : TM * ROWB;
    static constexpr int WR = WSTAGE / PIECE;
    static constexpr int STAGE = XSTAGE + WSTAGE;
    static constexpr int LPS = XR + WR;             // DMA instructions per thread per step
    // wave tile: MI x NI MFMA blocks of 32 channels x 32 pixels.  The 128-channel x 256-pixel fp16 tile gives each wave
    // 64 channels x 128 pixels (2 x 4 blocks: 6 LDS fragments per 8 MFMAs) instead of 32 x 256 (1 x 8: 9 per 8).
    static constexpr int MI = (ES == 2 && TPX == 256 && TM >= 128) ? 2 : 1;
    static constexpr int WM = TM / (32 * MI), WP = NW / WM, NI = TP / (32 * WP);
    static constexpr int NACC = MI * NI;            // accumulator blocks per wave
    static constexpr int NST = NACC * 4;            // store instructions per thread per epilogue
    // What follows the tiles in LDS, by epilogue: BatchNorm statistics [2][TM] fp64 (training forward, EM 0), the same + the
    // BNR constants [4][TM] fp32 (dgrad with the BatchNorm-backward sums), else the affine constants [2][TM] fp32.  Sized per
    // instantiation ON PURPOSE: LDS is allocated in 1 280-byte granules and three workgroups per CU need <= 53 760 bytes each --
    // the 128 x 128 tiles sit right at that edge (49 744 bytes of stages + tap table).  Rounds 2 -> 3 grew this tail to 4 KB for
    // every variant and the inference kernels of cfg 5 silently dropped to two workgroups per CU (k_gconv<128,2,128> 67 -> 88 us,
    // k_gconv3<128,4,128> 119 -> 157 us; bisected in round 4, profiles/r04_cfg5_bisect.txt).
    static constexpr size_t tail(int EM, bool BNR) {
        return BNR ? 2 * TM * sizeof(double) + 4 * TM * sizeof(float) : (EM == 0 ? 2 * TM * sizeof(double) : 2 * TM * sizeof(float));
    }
    static constexpr size_t lds(int EM, bool BNR) { return (size_t)GNS * STAGE + (MAX_TAPS + 1) * 16 + tail(EM, BNR); }
};

// Explicit MFMA-result hazard pad (see the comment in k_gconv's step loop).  The pad only works if the MFMAs stay in front of
// it: an asm statement -- even volatile with a memory clobber -- does not order register-only instructions, and hipcc was seen
// to schedule the last MFMAs of a step BEHIND the pad after an unrelated edit of the epilogue (test_head_conv then failed with
// one stale accumulator register per tile).  sched_barrier(0) on both sides pins it.
#define AY_MFMA_PAD(asm_nops)                          \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile(asm_nops ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)

// pixel decode of the loader's rows for pixel tile `tile` (invalid tile / rows beyond Mtotal -> never in range).
// Branch-free on purpose (selects only): see the header comment.
template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_setup_rows(const GConvP& p, unsigned tile, bool valid, int wave, int rowin, int kc,
                                             int (&xoff)[GT<T, TM, TPX, NWX>::XR], int (&xh0)[GT<T, TM, TPX, NWX>::XR], int (&xw0)[GT<T, TM, TPX, NWX>::XR]) {
    using G = GT<T, TM, TPX, NWX>;
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        const int row = (r * G::NW + wave) * G::RW + rowin;
        const unsigned mu = tile * G::TP + row;            // < 2^31 + TP: pixel counts are < 2^31 (host check)
        const bool ok = valid & (mu < (unsigned)p.Mtotal);
        const unsigned t = fdiv(mu, p.dOW);
        const int ow = (int)(mu - t * (unsigned)p.OW);
        const unsigned n = fdiv(t, p.dOH);
        const int oh = (int)(t - n * (unsigned)p.OH);
        const int h0 = oh * p.ish, w0 = ow * p.isw;
        // 1x1 / stride 1 / no padding ("x_linear") is the same formula: XH == OH, XW == OW
        xoff[r] = (int)(((n * (unsigned)p.XH + (unsigned)h0) * (unsigned)p.XW + (unsigned)w0) * (unsigned)p.ldx * G::ES);
        xh0[r] = ok ? h0 : -100000;
        xw0[r] = w0;
    }
}

// issue the DMA of one step (k slice `kt` of the loader's current tile) into the LDS stage at byte offset `so`
template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_issue(const GConvP& p, const int (&xoff)[GT<T, TM, TPX, NWX>::XR], const int (&xh0)[GT<T, TM, TPX, NWX>::XR],
                                        const int (&xw0)[GT<T, TM, TPX, NWX>::XR], const unsigned (&woff)[GT<T, TM, TPX, NWX>::WR], int kt,
                                        int tap0, int ntap, const int4* sTap, unsigned lds_tiles, unsigned so, v4i32 rsX,
                                        v4i32 rsW, int wave, int kc) {
    using G = GT<T, TM, TPX, NWX>;
    const unsigned k0 = (unsigned)(kt * BK + kc * G::CE);
    unsigned tap = fdiv(k0, p.dC);
    const int cb = (int)(k0 - tap * (unsigned)p.C) * G::ES;
    tap = tap < (unsigned)ntap ? (unsigned)tap0 + tap : MAX_TAPS;     // beyond the class's taps (K padding): never in range
    const int4 te = sTap[tap];
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        const unsigned ih = (unsigned)(xh0[r] + te.y), iw = (unsigned)(xw0[r] + te.z);
        const bool ok = (ih < (unsigned)p.XH) & (iw < (unsigned)p.XW);
        const unsigned off = ok ? (unsigned)(xoff[r] + te.x + cb) : G_OOB;
        glds16(rsX, lds_tiles + so + (r * G::NW + wave) * 1024, off);
    }
#pragma unroll
    for (int r = 0; r < G::WR; ++r) {
        const unsigned off = woff[r] + (unsigned)te.w + (unsigned)cb;
        glds16(rsW, lds_tiles + so + G::XSTAGE + (r * G::NW + wave) * 1024, off);
    }
}

// fp16: the fragments of a step (2 A + 2*NI B, 16 bytes each) are fetched in one go and consumed by g_mma_frags.  Left
// to itself hipcc reuses ONE register quad for every B fragment (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, eight times
// per step): the wave then pays the LDS latency per MFMA instead of once per step.  The step loop issues the fetch right
// after the barrier and puts the address arithmetic + DMA issue of step s+2 between the fetch and the MFMAs, so the LDS
// latency is covered by that scalar / VALU work instead of being waited for.
template <typename T, int TM, int TPX, int NWX = 4>
struct GFrags { half8 a[BK / 16][GT<T, TM, TPX, NWX>::MI], b[BK / 16][GT<T, TM, TPX, NWX>::NI]; };

template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_fetch_frags(const unsigned char* stage, int arow, int xrow, int swz, int lane,
                                              GFrags<T, TM, TPX, NWX>& f) {
    using G = GT<T, TM, TPX, NWX>;
    const unsigned char* bx = stage + xrow;
    const unsigned char* bw = stage + G::XSTAGE + arow;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        const int slot = ((kk * 2 + (lane >> 5)) ^ swz) * 16;
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) f.a[kk][mi] = *reinterpret_cast<const half8*>(bw + mi * 32 * G::ROWB + slot);
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) f.b[kk][ni] = *reinterpret_cast<const half8*>(bx + ni * 32 * G::ROWB + slot);
    }
}

// one 16-deep half of a step (the 256-pixel x 128-channel tile fetches and consumes a step in two halves: all of its
// 18 fragments at once would not fit the 256-register budget of 2 waves per SIMD next to the 128 accumulators; the
// eight MFMAs of the first half (512 cycles in the matrix pipe) cover the LDS latency of the second half's fetch)
template <typename T, int TM, int TPX, int NWX = 4>
struct GFragK { half8 a[GT<T, TM, TPX, NWX>::MI], b[GT<T, TM, TPX, NWX>::NI]; };

template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_fetch_k(const unsigned char* stage, int arow, int xrow, int swz, int lane, int kk,
                                          GFragK<T, TM, TPX, NWX>& f) {
    using G = GT<T, TM, TPX, NWX>;
    const unsigned char* bx = stage + xrow;
    const unsigned char* bw = stage + G::XSTAGE + arow;
    const int slot = ((kk * 2 + (lane >> 5)) ^ swz) * 16;
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi) f.a[mi] = *reinterpret_cast<const half8*>(bw + mi * 32 * G::ROWB + slot);
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni) f.b[ni] = *reinterpret_cast<const half8*>(bx + ni * 32 * G::ROWB + slot);
}

template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_mma_k(const GFragK<T, TM, TPX, NWX>& f, float16v (&acc)[GT<T, TM, TPX, NWX>::NACC]) {
    using G = GT<T, TM, TPX, NWX>;
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) mma_step(f.a[mi], f.b[ni], acc[mi * G::NI + ni]);
}

// The same step in two parts, for the interleaved schedule of the 8-block wave tiles: g_issue_prep does the address work
// (tap decode, halo test) into per-piece offsets, g_issue_piece<I> issues piece I.  A DMA instruction costs ~60 issue cycles
// among bare MFMAs but 100-185 in a block of its own next to fragment reads (MI355X_MICROARCH.md), and a wave has ~5 free
// issue slots behind every 32 x 32 x 16 MFMA: issued one per MFMA, the six pieces of a step ride in the matrix pipe's shadow
// instead of standing in front of it (probe, 512 -> 512 1x1 on 20 x 20 x 64: step body 1 300 cycles for 512 of MFMA).
template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_issue_prep(const GConvP& p, const int (&xoff)[GT<T, TM, TPX, NWX>::XR], const int (&xh0)[GT<T, TM, TPX, NWX>::XR],
                                             const int (&xw0)[GT<T, TM, TPX, NWX>::XR], const unsigned (&woff)[GT<T, TM, TPX, NWX>::WR], int kt,
                                             int tap0, int ntap, const int4* sTap, int kc, unsigned (&offs)[GT<T, TM, TPX, NWX>::LPS]) {
    using G = GT<T, TM, TPX, NWX>;
    const unsigned k0 = (unsigned)(kt * BK + kc * G::CE);
    unsigned tap = fdiv(k0, p.dC);
    const int cb = (int)(k0 - tap * (unsigned)p.C) * G::ES;
    tap = tap < (unsigned)ntap ? (unsigned)tap0 + tap : MAX_TAPS;
    const int4 te = sTap[tap];
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        const unsigned ih = (unsigned)(xh0[r] + te.y), iw = (unsigned)(xw0[r] + te.z);
        const bool ok = (ih < (unsigned)p.XH) & (iw < (unsigned)p.XW);
        offs[r] = ok ? (unsigned)(xoff[r] + te.x + cb) : G_OOB;
    }
#pragma unroll
    for (int r = 0; r < G::WR; ++r) offs[G::XR + r] = woff[r] + (unsigned)te.w + (unsigned)cb;
}
template <typename T, int TM, int TPX, int I, int NWX = 4>
__device__ __forceinline__ void g_issue_piece(const unsigned (&offs)[GT<T, TM, TPX, NWX>::LPS], unsigned lds_tiles, unsigned so, v4i32 rsX,
                                              v4i32 rsW, int wave) {
    using G = GT<T, TM, TPX, NWX>;
    if constexpr (I < G::XR) glds16(rsX, lds_tiles + so + (I * G::NW + wave) * 1024, offs[I]);
    else if constexpr (I < G::LPS) glds16(rsW, lds_tiles + so + G::XSTAGE + ((I - G::XR) * G::NW + wave) * 1024, offs[I]);
}
// first 16-deep half of a step with the step's DMA pieces interleaved: MFMA q, then piece q (q < LPS <= 8)
template <typename T, int TM, int TPX, int NWX = 4, int Q = 0>
__device__ __forceinline__ void g_mma_k_issue(const GFragK<T, TM, TPX, NWX>& f, float16v (&acc)[GT<T, TM, TPX, NWX>::NACC],
                                              const unsigned (&offs)[GT<T, TM, TPX, NWX>::LPS], unsigned lds_tiles, unsigned so, v4i32 rsX,
                                              v4i32 rsW, int wave) {
    using G = GT<T, TM, TPX, NWX>;
    static_assert(G::LPS <= G::NACC, "one DMA piece per MFMA of the first half");
    if constexpr (Q < G::NACC) {
        mma_step(f.a[Q / G::NI], f.b[Q % G::NI], acc[Q]);
        g_issue_piece<T, TM, TPX, Q, NWX>(offs, lds_tiles, so, rsX, rsW, wave);
        __builtin_amdgcn_sched_barrier(0);
        g_mma_k_issue<T, TM, TPX, NWX, Q + 1>(f, acc, offs, lds_tiles, so, rsX, rsW, wave);
    }
}

template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_mma_frags(const GFrags<T, TM, TPX, NWX>& f, float16v (&acc)[GT<T, TM, TPX, NWX>::NACC]) {
    using G = GT<T, TM, TPX, NWX>;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) mma_step(f.a[kk][mi], f.b[kk][ni], acc[mi * G::NI + ni]);
}

// whole-step variant of g_mma_k_issue (tiles with fewer than 8 accumulator blocks per wave): MFMA q of the step, then DMA
// piece q; pieces beyond the step's MFMA count (32-channel tiles) follow the last MFMA
template <typename T, int TM, int TPX, int NWX = 4, int Q = 0>
__device__ __forceinline__ void g_mma_frags_issue(const GFrags<T, TM, TPX, NWX>& f, float16v (&acc)[GT<T, TM, TPX, NWX>::NACC],
                                                  const unsigned (&offs)[GT<T, TM, TPX, NWX>::LPS], unsigned lds_tiles, unsigned so, v4i32 rsX,
                                                  v4i32 rsW, int wave) {
    using G = GT<T, TM, TPX, NWX>;
    constexpr int NMMA = (BK / 16) * G::NACC;
    if constexpr (Q < (NMMA > G::LPS ? NMMA : G::LPS)) {
        if constexpr (Q < NMMA) {
            constexpr int kk = Q / G::NACC, blk = Q % G::NACC;
            mma_step(f.a[kk][blk / G::NI], f.b[kk][blk % G::NI], acc[blk]);
        }
        g_issue_piece<T, TM, TPX, Q, NWX>(offs, lds_tiles, so, rsX, rsW, wave);
        __builtin_amdgcn_sched_barrier(0);
        g_mma_frags_issue<T, TM, TPX, NWX, Q + 1>(f, acc, offs, lds_tiles, so, rsX, rsW, wave);
    }
}

// fp32 (exact-parity mode): fragments are fetched and consumed pair by pair
template <typename T, int TM, int TPX, int NWX = 4>
__device__ __forceinline__ void g_mma(const unsigned char* stage, int arow, int xrow, int swz, int lane,
                                      float16v (&acc)[GT<T, TM, TPX, NWX>::NACC]) {
    using G = GT<T, TM, TPX, NWX>;
    static_assert(G::MI == 1, "fp32 mode keeps the 1 x NI wave tile");
    const unsigned char* bx = stage + xrow;
    const unsigned char* bw = stage + G::XSTAGE + arow;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        const int c0 = kk * 4 + (lane >> 5) * 2;
        const int s0 = (c0 ^ swz) * 16, s1 = ((c0 + 1) ^ swz) * 16;
        Tr<float>::frag a;
        {
            const float4v u = *reinterpret_cast<const float4v*>(bw + s0), v = *reinterpret_cast<const float4v*>(bw + s1);
            a.v[0] = u[0]; a.v[1] = u[1]; a.v[2] = u[2]; a.v[3] = u[3]; a.v[4] = v[0]; a.v[5] = v[1]; a.v[6] = v[2]; a.v[7] = v[3];
        }
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) {
            const unsigned char* q = bx + ni * 32 * G::ROWB;
            const float4v u = *reinterpret_cast<const float4v*>(q + s0), v = *reinterpret_cast<const float4v*>(q + s1);
            Tr<float>::frag b;
            b.v[0] = u[0]; b.v[1] = u[1]; b.v[2] = u[2]; b.v[3] = u[3]; b.v[4] = v[0]; b.v[5] = v[1]; b.v[6] = v[2]; b.v[7] = v[3];
            mma_step(a, b, acc[ni]);
        }
    }
}

// tile finished: acc[ni][r] holds channel = cbase + 8*(r>>2) + (r&3), pixel = m0 + wp*NI*32 + ni*32 + (lane&31).
// Exactly NST buffer stores per thread (invalid pixels / channel groups use the out-of-range offset and are dropped
// by the hardware), so the step loop's vmcnt arithmetic stays exact.
// BNR kernels (fp16 dgrad, see ayolo_conv_dgrad_bn): what the epilogue needs to form du = da * act'(bn(z)) and xhat for the
// values it stores.  The segment of a 32-channel block is uniform per (wave, mi) -- the host checks that the segment
// boundary is a multiple of 32 -- so each wave picks the z descriptor of its MI blocks ONCE, into SGPRs (a per-store
// select between two descriptors ends in a v_cndmask + readfirstlane waterfall loop around every load).
template <int MI>
struct BnrCtx {
    __amdgpu_buffer_rsrc_t rsZ[MI];
    unsigned ldzb[MI];         // bytes per pixel of the block's z buffer
    int c0[MI];                // first output channel of the block's segment
    const float* sBn;          // LDS [4][TM]: invstd | -mean * invstd | invstd * gamma | beta - mean * invstd * gamma
    int act;
};

// per-channel constants of this workgroup's channel tile [n0, n0 + TM) -> LDS (zeros beyond a segment / Nout, which keeps
// du = 0 there: the accumulators of padding channels are exact zeros)
template <int TM>
__device__ __forceinline__ void g_bnr_setup(const GConvP& p, float* sBn, int n0, int tid) {
    for (int i = tid; i < TM; i += (int)blockDim.x) {
        const int c = n0 + i;
        const int sg = (p.bnr > 1 && c >= p.bseg[1].c0) ? 1 : 0;
        const int cl = c - p.bseg[sg].c0;
        float is = 0.0f, nmi = 0.0f, A = 0.0f, Bc = 0.0f;
        // the four loads unconditional (clamped index) and together: behind a branch each was a memory round trip of its own
        // in front of the workgroup's first DMA (see rep_sum2, common.h)
        const bool in = c < p.Nout && cl >= 0 && cl < p.bseg[sg].C;
        const int cq = in ? cl : 0;
        const float* mi = p.bseg[sg].mean_invstd;
        const float mu = mi[cq], isv = mi[p.bseg[sg].C + cq];
        const float ga = opt_load(p.bseg[sg].gamma, mi, cq, 1.0f), be = opt_load(p.bseg[sg].beta, mi, cq, 0.0f);
        if (in) { is = isv; A = is * ga; Bc = be - mu * A; nmi = -mu * is; }
        sBn[i] = is; sBn[TM + i] = nmi; sBn[2 * TM + i] = A; sBn[3 * TM + i] = Bc;
    }
}

// cblk0: first output channel of this wave's first 32-channel block (wave-uniform)
template <int MI>
__device__ __forceinline__ BnrCtx<MI> g_bnr_ctx(const GConvP& p, const float* sBn, int cblk0) {
    BnrCtx<MI> b;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int cblk = __builtin_amdgcn_readfirstlane(cblk0 + mi * 32);
        const int sg = (p.bnr > 1 && cblk >= p.bseg[1].c0) ? 1 : 0;
        b.rsZ[mi] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg ? p.bseg[1].z : p.bseg[0].z), 0, sg ? p.z_bytes[1] : p.z_bytes[0], 0x00020000);
        b.ldzb[mi] = (unsigned)(sg ? p.bseg[1].ldz : p.bseg[0].ldz) * 2u;
        b.c0[mi] = sg ? p.bseg[1].c0 : p.bseg[0].c0;
    }
    b.sBn = sBn;
    b.act = p.bnr_act;
    return b;
}

template <typename T, int TM, int EM, int TPX, bool BNR = false, int NWX = 4>
__device__ __forceinline__ void g_epilogue(const GConvP& p, unsigned tile, int oah, int oaw, int wp, int lane, int cbase,
                                           bool want_stats, __amdgpu_buffer_rsrc_t rsY, float16v (&acc)[GT<T, TM, TPX, NWX>::NACC],
                                           float (&ssum)[16 * GT<T, TM, TPX, NWX>::MI], float (&ssq)[16 * GT<T, TM, TPX, NWX>::MI], const float* sAff, int cl0,
                                           const BnrCtx<GT<T, TM, TPX, NWX>::MI> bc = BnrCtx<GT<T, TM, TPX, NWX>::MI>{}) {
    using G = GT<T, TM, TPX, NWX>;
    constexpr int YES = (EM == 3) ? 4 : G::ES;       // bytes per output element
    static_assert(!BNR || (sizeof(T) == 2 && (EM == 0 || EM == 1)), "BN-backward statistics: fp16 dgrad epilogues only");
    const unsigned m0 = tile * G::TP;
    // pixel of this lane in pixel block ni: valid?, index in the y tensor
    auto pixel_of = [&](int ni, bool& pv_, unsigned& ypix_) {
        const unsigned m = m0 + wp * G::NI * 32 + ni * 32 + (lane & 31);
        pv_ = m < (unsigned)p.Mtotal;
        const unsigned mu = pv_ ? m : 0u;
        if (p.y_linear) ypix_ = mu;
        else {
            const unsigned t = fdiv(mu, p.dOW);
            const int ow = (int)(mu - t * (unsigned)p.OW);
            const unsigned nn = fdiv(t, p.dOH);
            const int oh = (int)(t - nn * (unsigned)p.OH);
            ypix_ = (nn * (unsigned)p.YH + (unsigned)(oh * p.osh + oah)) * (unsigned)p.YW + (unsigned)(ow * p.osw + oaw);
        }
    };
    // BNR: the z vectors of a pixel block are requested ONE BLOCK AHEAD of their use (2 * MI loads of 16 bytes per lane):
    // fetched where they are consumed, every (pixel block, channel group) would wait out a full HBM round trip
    half8 znext[BNR ? G::MI : 1][2];
    bool pv_n = false;
    unsigned ypix_n = 0;
    auto z_request = [&](bool pv_, unsigned ypix_) {
        if constexpr (BNR) {
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = cbase + mi * 32 - 4 * (lane >> 5) + 8 * (2 * j + (lane >> 5));
                    const unsigned zoff = (pv_ && c < p.Nout) ? ypix_ * bc.ldzb[mi] + (unsigned)(c - bc.c0[mi]) * 2u : G_OOB;
                    znext[mi][j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(bc.rsZ[mi], zoff, 0, 0));
                }
        }
    };
    // accumulate epilogues (fp16): the old values of a pixel block are requested one block ahead as well.  Fetched where they
    // are added, every 16-byte group is load -> wait -> add -> store with the next load stuck behind the store (same buffer: the
    // compiler must keep the order), i.e. 2 * MI * NI serial memory round trips per tile
    constexpr bool YPRE = sizeof(T) == 2 && (EM == 1 || EM == 4);
    half8 ynext[YPRE ? G::MI : 1][2];
    auto y_request = [&](bool pv_, unsigned ypix_) {
        if constexpr (YPRE) {
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = cbase + mi * 32 - 4 * (lane >> 5) + 8 * (2 * j + (lane >> 5));
                    const unsigned off = (pv_ && c < p.Nout) ? ypix_ * (unsigned)p.ldy * 2u + (unsigned)c * 2u : G_OOB;
                    ynext[mi][j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0));
                }
        }
    };
    // next to the z prefetch of a BNR kernel the old values are fetched per block (all 2 * MI groups at once, at the top of the
    // block) instead of a block ahead: both look-aheads together do not fit the register budget of the 128 x 256 tile
    constexpr bool YAHEAD = YPRE && !BNR;
    pixel_of(0, pv_n, ypix_n);
    if constexpr (YAHEAD) y_request(pv_n, ypix_n);
    z_request(pv_n, ypix_n);
#pragma unroll
    for (int ni = 0; ni < G::NI; ++ni) {
        const bool pv = pv_n;
        const unsigned ypix = ypix_n;
        const unsigned yo = ypix * (unsigned)p.ldy * YES;
        half8 zcur[BNR ? G::MI : 1][2];
        half8 ycur[YPRE ? G::MI : 1][2];
        if constexpr (BNR) {
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi) { zcur[mi][0] = znext[mi][0]; zcur[mi][1] = znext[mi][1]; }
        }
        if constexpr (YPRE && !YAHEAD) y_request(pv, ypix);
        if constexpr (YPRE) {
#pragma unroll
            for (int mi = 0; mi < G::MI; ++mi) { ycur[mi][0] = ynext[mi][0]; ycur[mi][1] = ynext[mi][1]; }
        }
        if (ni + 1 < G::NI) {
            pixel_of(ni + 1, pv_n, ypix_n);
            if constexpr (YAHEAD) y_request(pv_n, ypix_n);
            z_request(pv_n, ypix_n);
        }
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
            const int cb = cbase + mi * 32, cl = cl0 + mi * 32;
            if constexpr (sizeof(T) == 2 && EM != 3) {
                // fp16 outputs: lanes l and l+32 hold channels +0..3 / +4..7 of the SAME pixel for each 8-channel group.
                // Four v_permlane32_swap per group pair give every lane 8 consecutive channels, so the tile leaves in 16-byte
                // stores (2 per lane and pixel block instead of 4 of 8 bytes: half the requests the L2 has to take).
                float v[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[g][e] = acc[mi * G::NI + ni][g * 4 + e]; acc[mi * G::NI + ni][g * 4 + e] = 0.0f; }
                if constexpr (EM == 2 || EM == 4) {
                    // per-channel scale / shift of this workgroup's channel tile were staged in LDS once (sAff: a global load
                    // here would make hipcc drain the hidden DMA queue with vmcnt(0) in every epilogue)
                    const bool act = p.epi == AYOLO_EPI_AFFINE_SILU || p.epi == AYOLO_EPI_AFFINE_SILU_RES;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4v sc = *reinterpret_cast<const float4v*>(sAff + cl + 8 * g);
                        const float4v sh = *reinterpret_cast<const float4v*>(sAff + TM + cl + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float u = v[g][e] * sc[e] + sh[e];
                            v[g][e] = act ? silu_e<T>(u) : u;
                        }
                    }
                }
                if (want_stats) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float q = pv ? cvt_round(v[g][e], (T*)nullptr) : 0.0f;
                            ssum[mi * 16 + g * 4 + e] += q;
                            ssq[mi * 16 + g * 4 + e] += q * q;
                        }
                }
                const int hsel = lane >> 5;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const v2u32 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * j][e]), __float_as_uint(v[2 * j + 1][e]),
                                                                         false, false);
                        v[2 * j][e] = __uint_as_float(r[0]);
                        v[2 * j + 1][e] = __uint_as_float(r[1]);
                    }
                    const int c = cb - 4 * hsel + 8 * (2 * j + hsel);          // 8 channels: v[2j][0..3], v[2j+1][0..3]
                    const unsigned off = (pv && c < p.Nout) ? yo + (unsigned)c * 2u : G_OOB;   // Nout % 8 == 0 (host check)
                    float w8[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { w8[e] = v[2 * j][e]; w8[4 + e] = v[2 * j + 1][e]; }
                    if constexpr (EM == 1 || EM == 4) {
                        const half8 o = ycur[mi][j];
#pragma unroll
                        for (int e = 0; e < 8; ++e) w8[e] += (float)o[e];
                    }
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (half_t)w8[e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, h), rsY, off, 0, 0);
                    if constexpr (BNR) {
                        // du = da * act'(z*A + Bc), xhat = z*invstd + nmi for the 8 channels c .. c+7 of this lane's pixel, from
                        // the ROUNDED gradient just stored (what a separate pass would read back); sums stay in registers
                        const half8 zz = zcur[mi][j];
                        // opaque index: the constants are re-read from LDS for every pixel block -- left visible, the reads of all
                        // (mi, j) pairs are merged across the unrolled ni loop and 64-128 registers of constants stay live
                        int kqo = cl - 4 * hsel + 8 * (2 * j + hsel);
                        asm volatile("" : "+v"(kqo));
                        const float* kq = bc.sBn + kqo;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float4v kis = *reinterpret_cast<const float4v*>(kq + 4 * q);
                            const float4v knm = *reinterpret_cast<const float4v*>(kq + TM + 4 * q);
                            const float4v kA = *reinterpret_cast<const float4v*>(kq + 2 * TM + 4 * q);
                            const float4v kB = *reinterpret_cast<const float4v*>(kq + 3 * TM + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float zf = (float)zz[4 * q + e], da = (float)h[4 * q + e];
                                const float xh = __builtin_fmaf(zf, kis[e], knm[e]);
                                const float u = __builtin_fmaf(zf, kA[e], kB[e]);
                                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
                                const float gg = sg * __builtin_fmaf(u, 1.0f - sg, 1.0f);
                                const float du = bc.act ? da * gg : da;
                                ssum[mi * 16 + j * 8 + 4 * q + e] += du;
                                ssq[mi * 16 + j * 8 + 4 * q + e] = __builtin_fmaf(du, xh, ssq[mi * 16 + j * 8 + 4 * q + e]);
                            }
                        }
                        // pin the arithmetic HERE: its only consumers are the reductions after the last tile, and left alone the
                        // compiler sinks all of it there, keeping every z vector, every stored value and every constant of the
                        // epilogue alive in scratch (1.7 KB per lane) until then
#pragma unroll
                        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(ssum[mi * 16 + j * 8 + e]), "+v"(ssq[mi * 16 + j * 8 + e]));
                    }
                }
                continue;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = cb + 8 * g;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[mi * G::NI + ni][g * 4 + e]; acc[mi * G::NI + ni][g * 4 + e] = 0.0f; }
                if constexpr (EM == 3) {
                    // YOLOHead: fp32 logits + bias, NHWC [pixel][ldy] (ldy = Cout rounded up to 8), 16-byte stores;
                    // the (B, na, ny, nx, no) tensor the loss / decode see is a strided view of this buffer
                    const unsigned off = (pv && c < p.ldy) ? yo + (unsigned)c * 4u : G_OOB;
                    float4v f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = v[e] + ((p.shift && c + e < p.Nout) ? p.shift[c + e] : 0.0f);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, f), rsY, off, 0, 0);
                    continue;
                }
                if constexpr (EM == 2 || EM == 4) {
                    const bool act = p.epi == AYOLO_EPI_AFFINE_SILU || p.epi == AYOLO_EPI_AFFINE_SILU_RES;
                    const float4v sc = *reinterpret_cast<const float4v*>(sAff + cl + 8 * g);
                    const float4v sh = *reinterpret_cast<const float4v*>(sAff + TM + cl + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = v[e] * sc[e] + sh[e];
                        v[e] = act ? silu_e<T>(u) : u;
                    }
                }
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float q = pv ? cvt_round(v[e], (T*)nullptr) : 0.0f;
                        ssum[mi * 16 + g * 4 + e] += q;
                        ssq[mi * 16 + g * 4 + e] += q * q;
                    }
                }
                const unsigned off = (pv && c < p.Nout) ? yo + (unsigned)c * G::ES : G_OOB;   // Nout % 4 == 0 (host check)
                if constexpr (sizeof(T) == 2) {
                    if constexpr (EM == 1 || EM == 4) {
                        const half4 o = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(rsY, off, 0, 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)o[e];
                    }
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, h), rsY, off, 0, 0);
                } else {
                    if constexpr (EM == 1 || EM == 4) {
                        const float4v o = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rsY, off, 0, 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += o[e];
                    }
                    float4v f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = v[e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, f), rsY, off, 0, 0);
                }
            }
        }
    }
}

// channel (relative to the wave's first channel wm * 32 * MI) of statistics register r.  Forward statistics are taken
// before the epilogue's lane swap (4-channel runs of the MFMA layout), BNR sums after it (8 consecutive channels)
template <bool POST>
__device__ __forceinline__ int g_stat_chan(int r, int hsel) {
    if constexpr (POST) return (r >> 4) * 32 + 8 * (2 * ((r >> 3) & 1) + hsel) + (r & 7);
    else return (r >> 4) * 32 + 4 * hsel + 8 * ((r & 15) >> 2) + (r & 3);
}

// Statistics accumulate in DOUBLE from the workgroup level on (LDS and global atomics).  Per-lane partial sums and the wave
// reductions are fp32 in a fixed order; what is order-dependent -- which wavefront / workgroup adds first -- then rounds at
// 1e-16, far below one fp32 ulp of the mean / variance derived from the totals, so a step's BatchNorm statistics (and with
// them the fp16 rounding of every activation) repeat from run to run.  With fp32 atomics two identical fp16 train steps
// differed by cosine 0.993-0.998 in their gradients (tools/cond_explore.py), which is the noise floor every fp16 parity
// threshold had to sit under.
// workgroup totals in sStat [2][TM] -> global accumulators: forward statistics [reps][2][Nout], or (BNR) the
// [reps][2][C] accumulators of the segment each channel belongs to
template <int TM, bool BNR>
__device__ __forceinline__ void g_stats_to_global(const GConvP& p, const double* sStat, int tid, int n0, unsigned slot) {
    if constexpr (BNR) {
        for (int i = tid; i < TM; i += (int)blockDim.x) {
            const int c = n0 + i;
            const int sg = (p.bnr > 1 && c >= p.bseg[1].c0) ? 1 : 0;
            const int cl = c - p.bseg[sg].c0, C = p.bseg[sg].C;
            if (c < p.Nout && cl >= 0 && cl < C) {
                double* st = p.bseg[sg].sums + (size_t)(slot % (unsigned)p.bnr_reps) * 2 * C;
                atomicAdd(&st[cl], sStat[i]);
                atomicAdd(&st[C + cl], sStat[TM + i]);
            }
        }
    } else {
        // replicated accumulators: workgroups spread over stat_reps copies so L2 atomics do not serialise
        double* st = p.stats + (size_t)(slot % (unsigned)p.stat_reps) * 2 * p.Nout;
        for (int i = tid; i < TM; i += (int)blockDim.x) {
            if (n0 + i < p.Nout) {
                atomicAdd(&st[n0 + i], sStat[i]);
                atomicAdd(&st[p.Nout + n0 + i], sStat[TM + i]);
            }
        }
    }
}

template <typename T, int TM, int MI, bool BNR = false>
__device__ __forceinline__ void g_stats_flush(const GConvP& p, double* sStat, int tid, int lane, int wm, int n0, unsigned slot,
                                              const float (&ssum)[16 * MI], const float (&ssq)[16 * MI]) {
    for (int i = tid; i < 2 * TM; i += (int)blockDim.x) sStat[i] = 0.0;
    __syncthreads();
    {
        // 32 pixel lanes per half-wave: reduce-scatter (half_reduce_scatter), one atomic instruction per sum with every lane active
        float sa[16 * MI], sb[16 * MI];
#pragma unroll
        for (int r = 0; r < 16 * MI; ++r) { sa[r] = ssum[r]; sb[r] = ssq[r]; }
        const float a = half_reduce_scatter<16 * MI>(sa, lane), b = half_reduce_scatter<16 * MI>(sb, lane);
        if (rs_reports<16 * MI>(lane)) {
            const int cl = wm * 32 * MI + g_stat_chan<BNR>(rs_index<16 * MI>(lane), lane >> 5);
            atomicAdd(&sStat[cl], (double)a);
            atomicAdd(&sStat[TM + cl], (double)b);
        }
    }
    __syncthreads();
    g_stats_to_global<TM, BNR>(p, reinterpret_cast<const double*>(sStat), tid, n0, slot);
}

template <typename T, int TM, int EM, int TPX, bool BNR = false, bool XF = false, bool LIN = false, int NWX = 4>
__global__ __launch_bounds__((GT<T, TM, TPX, NWX>::NT), (NWX == 8 ? 4 : (sizeof(T) == 2 ? (GT<T, TM, TPX, NWX>::lds(EM, BNR) > 56 * 1024 ? 2 : ((TM == 128 || (TM == 64 && (EM == 0 || BNR))) ? 3 : 4)) : 1))) void k_gconv(GConvP p) {
    static_assert(NWX == 4 || (NWX == 8 && sizeof(T) == 2 && TPX == 128 && TM >= 64), "eight wavefronts: fp16, 128-pixel tiles (two workgroups = four wavefronts per SIMD per CU)");
    AY_KERNARG_TOUCH(kt_, GConvP);           // every line of the parameter struct requested at once (gfx950_dma.h)
    static_assert(TM == 32 || TM == 64 || TM == 128, "output-channel tiles of 32 / 64 / 128");
    static_assert(!XF || (sizeof(T) == 2 && !BNR && (EM == 0 || EM == 3)), "transform on load: fp16 forward of a 1x1 conv");
    using G = GT<T, TM, TPX, NWX>;
    // stores per thread and epilogue (the step loop's vmcnt arithmetic): fp16 tiles leave in 16-byte stores
    constexpr int NSTK = (sizeof(T) == 2 && EM != 3) ? G::NACC * 2 : G::NST;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    unsigned char* sTiles = smem_raw;                                             // [GNS][x tile | w tile]
    int4* sTap = reinterpret_cast<int4*>(smem_raw + GNS * G::STAGE);              // [MAX_TAPS + 1]
    float* sStat = reinterpret_cast<float*>(sTap + MAX_TAPS + 1);                 // [2][TM] (+ [4][TM] BNR constants)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % G::WM, wp = wave / G::WM;

    // ---- block -> (channel tile, XCD band, slot).  Workgroups are dealt round-robin to the 8 XCDs; all channel
    // tiles of one pixel tile go to the SAME XCD back to back, and each XCD walks a contiguous band of pixel tiles.
    const unsigned Lb = blockIdx.x;
#ifdef AYOLO_PROBE
    __shared__ unsigned long long s_probe[AY_PROBE_N];
    const bool probe_on = blockIdx.x < 512;
    int probe_k = 2;
    if (threadIdx.x < AY_PROBE_N) s_probe[threadIdx.x] = 0;
    __syncthreads();
    AY_PROBE(0);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 3] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, chip-wide
#endif
    const unsigned xcd = Lb & 7u, idx = Lb >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const int n0 = (int)nt * TM;
    const unsigned ntiles_all = (unsigned)((p.Mtotal + G::TP - 1) / G::TP);
    const unsigned tpx = (ntiles_all + 7) / 8;
    const unsigned band_lo = xcd * tpx;
    const unsigned ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = (unsigned)p.nslots / 8u;
    unsigned cur_tile = band_lo + lslot;
    kt_.done();
    if (cur_tile >= ntiles) return;

    // ---- tap table -> LDS: {x byte delta, dh, dw, w column byte offset}; entries >= ntaps never hit (K padding)
    // The three byte arrays come through SCALAR loads of their 27 dwords (uniform, compile-time offsets) and a per-lane select:
    // indexed per lane they are vector loads from the kernarg segment, and every workgroup then waits ~2 000 cycles for a
    // memory round trip before its first DMA can be issued (probe, tools/gconv_probe.py)
    static_assert(MAX_TAPS % 4 == 0 && offsetof(GConvP, dh) % 4 == 0 && offsetof(GConvP, dw) == offsetof(GConvP, dh) + MAX_TAPS &&
                  offsetof(GConvP, wt) == offsetof(GConvP, dh) + 2 * MAX_TAPS, "tap arrays: three packed dword-aligned byte arrays");
    typedef __attribute__((address_space(4))) const int* kiptr_t;
    typedef __attribute__((address_space(4))) const char* kcptr_t;
    const kiptr_t kwords = (kiptr_t)((kcptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(GConvP, dh));
    static_assert(MAX_TAPS / 4 == 9, "three groups of nine dwords below");
    int kw_[3 * (MAX_TAPS / 4)];
#pragma unroll
    for (int i = 0; i < 3 * (MAX_TAPS / 4); ++i) kw_[i] = kwords[i];
    // loaded HERE, all at once (wide s_loads, one wait): left alone the loads sink into 27 conditional blocks of the selects
#pragma unroll
    for (int g = 0; g < 3; ++g)
        asm volatile("" : "+s"(kw_[9 * g]), "+s"(kw_[9 * g + 1]), "+s"(kw_[9 * g + 2]), "+s"(kw_[9 * g + 3]), "+s"(kw_[9 * g + 4]),
                          "+s"(kw_[9 * g + 5]), "+s"(kw_[9 * g + 6]), "+s"(kw_[9 * g + 7]), "+s"(kw_[9 * g + 8]));
    if (tid <= MAX_TAPS) {
        int4 e;
        const int tq = tid < MAX_TAPS ? tid : 0;
        int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int i = 0; i < MAX_TAPS / 4; ++i) {
            const bool me = (tq >> 2) == i;
            s0 = me ? kw_[i] : s0;
            s1 = me ? kw_[MAX_TAPS / 4 + i] : s1;
            s2 = me ? kw_[2 * (MAX_TAPS / 4) + i] : s2;
        }
        const int sh8 = 8 * (tq & 3);
        const int dh = (int)(signed char)(s0 >> sh8), dw = (int)(signed char)(s1 >> sh8), wt = (int)(signed char)(s2 >> sh8);
        if (tid < p.ntaps) { e.x = (dh * p.XW + dw) * p.ldx * G::ES; e.y = dh; e.z = dw; e.w = wt * p.C * G::ES; }
        else { e.x = 0; e.y = -100000; e.z = 0; e.w = 0x40000000; }
        sTap[tid] = e;
    }
    AY_PROBE(AY_PROBE_N - 4);

    if constexpr (EM == 2 || EM == 4) {
        // affine epilogue constants of this channel tile -> LDS [scale | shift] (identity beyond Nout / for null pointers)
        for (int i = tid; i < TM; i += (int)blockDim.x) {
            const bool in = n0 + i < p.Nout;
            const float* any = p.scale ? p.scale : p.shift;       // both loads together (one memory round trip in the prologue)
            float sc_ = 1.0f, sh_ = 0.0f;
            if (any) {
                const float a_ = (p.scale ? p.scale : any)[in ? n0 + i : 0], b_ = (p.shift ? p.shift : any)[in ? n0 + i : 0];
                sc_ = (in && p.scale) ? a_ : 1.0f;
                sh_ = (in && p.shift) ? b_ : 0.0f;
            }
            sStat[i] = sc_;
            sStat[TM + i] = sh_;
        }
    }
    // XF: per-input-channel scale | shift behind the epilogue's tail, [2][C rounded up to 32] floats (zeros beyond C: the
    // zero-filled K padding stays zero under silu)
    float* sXf = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(sStat) + G::tail(EM, BNR));
    const int xf_cp = (p.C + BK - 1) / BK * BK;
    if constexpr (XF) {
        if (p.nfin > 0) {
            for (int i = tid; i < 2 * xf_cp; i += (int)blockDim.x) sXf[i] = 0.0f;
            __syncthreads();
            float* gsc = const_cast<float*>(p.xf_scale);
            float* gsh = const_cast<float*>(p.xf_shift);
            for (int f = 0; f < p.nfin; ++f) {
                const ayolo_xf_fin& q = p.fin[f];
                for (int c = tid; c < q.C; c += (int)blockDim.x) {
                    double s1, s2;
                    rep_sum2(q.stats + c, (size_t)2 * q.sld, (size_t)q.sld, q.reps, s1, s2);   // eight replicas in flight (common.h)
                    // the SAME expression sequence as k_bn_finalize / k_bn_train_act's prologue: bit-identical scale / shift
                    const double mean = s1 / q.count;
                    double var = s2 / q.count - mean * mean;
                    if (var < 0) var = 0;
                    const float invstd = (float)(1.0 / sqrt(var + (double)q.eps));
                    const float g = q.gamma ? q.gamma[c] : 1.0f, b = q.beta ? q.beta[c] : 0.0f;
                    const float sc = g * invstd;
                    const float sh = b - (float)mean * sc;
                    sXf[q.c0 + c] = sc;
                    sXf[xf_cp + q.c0 + c] = sh;
                    if (Lb == 0) {
                        gsc[q.c0 + c] = sc;
                        gsh[q.c0 + c] = sh;
                        if (q.save_mean) q.save_mean[c] = (float)mean;
                        if (q.save_invstd) q.save_invstd[c] = invstd;
                        if (q.running_mean) q.running_mean[c] = (1.0f - q.momentum) * q.running_mean[c] + q.momentum * (float)mean;
                        if (q.running_var) {
                            const double unb = q.count > 1.0 ? var * q.count / (q.count - 1.0) : var;
                            q.running_var[c] = (1.0f - q.momentum) * q.running_var[c] + q.momentum * (float)unb;
                        }
                    }
                }
            }
        } else {
            for (int i = tid; i < xf_cp; i += (int)blockDim.x) {
                sXf[i] = i < p.C ? p.xf_scale[i] : 0.0f;
                sXf[xf_cp + i] = i < p.C ? p.xf_shift[i] : 0.0f;
            }
        }
    }
    // (eight wavefronts: the forward statistics too -- 32 sum registers across the step loop do not fit the 128-register budget)
    constexpr bool TILE_RED = (BNR && TM >= 128) || (NWX == 8 && EM == 0);
    if constexpr (BNR) g_bnr_setup<TM>(p, sStat + 4 * TM, n0, tid);
    if constexpr (TILE_RED) {
        for (int i = tid; i < 2 * TM; i += (int)blockDim.x) reinterpret_cast<double*>(sStat)[i] = 0.0;
    }
    const BnrCtx<G::MI> bctx = BNR ? g_bnr_ctx<G::MI>(p, sStat + 4 * TM, n0 + wm * 32 * G::MI) : BnrCtx<G::MI>{};

    const v4i32 rsX = make_srd(p.x, p.x_bytes), rsW = make_srd(p.w, p.w_bytes);
    const unsigned lds_tiles = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sTiles);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // ---- loader lane geometry: this lane writes LDS bytes [piece*1024 + lane*16, +16) of a stage; the chunk it must
    // FETCH for that position is the position's chunk slot XOR the row swizzle
    const int slotc = lane & (G::CPR - 1);
    const int rowin = lane / G::CPR;
    const int lsw = sizeof(T) == 2 ? (lane >> 4) : ((((wave & 1) << 2) | (lane >> 4)) & 7);
    const int kc = slotc ^ lsw;

    int xoff[G::XR], xh0[G::XR], xw0[G::XR];
    // XF (1x1 / stride 1 only): the loader needs no pixel decode -- a row's byte offset in segment 0 (xoff) and in segment 1 (xoff2),
    // out of range for rows beyond the tensor
    unsigned xoff2[XF ? G::XR : 1];
    unsigned woff[G::WR];
#pragma unroll
    for (int r = 0; r < G::WR; ++r) {
        const int row = (r * G::NW + wave) * G::RW + rowin;
        const bool ok = (row < TM) & (n0 + row < p.Nout);
        woff[r] = ok ? (unsigned)(n0 + row) * (unsigned)p.ldw * G::ES : G_OOB;   // the k-chunk offset comes from g_issue
    }

    // ---- MFMA fragment geometry
    const int arow = (wm * 32 * G::MI + (lane & 31)) * G::ROWB;
    const int xrow = (wp * G::NI * 32 + (lane & 31)) * G::ROWB;
    const int swz = ((lane & 31) / G::RPB) & (G::CPR - 1);

    float16v acc[G::NACC];
#pragma unroll
    for (int i = 0; i < G::NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const bool want_stats = (EM == 0) && (p.stats != nullptr);
    float ssum[16 * G::MI], ssq[16 * G::MI];
#pragma unroll
    for (int r = 0; r < 16 * G::MI; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
    const int cbase = n0 + wm * 32 * G::MI + 4 * (lane >> 5);

    // steps of a class: its taps * C in 32-wide slices; a tap-less class (1x1 strided dgrad) is one all-zero step
#define G_NK(c) ((p.cnt[c] * p.C + BK - 1) / BK < 1 ? 1 : (p.cnt[c] * p.C + BK - 1) / BK)
    int cur_kt = 0, cur_cls = 0, cur_nk = G_NK(0);

    // loader cursor, two steps ahead of the compute cursor; past the last step it keeps issuing (out-of-range, zero
    // fill) so that every step has exactly LPS DMA instructions per thread
    unsigned ld_tile = cur_tile;
    int ld_kt = 0, ld_cls = 0, ld_nk = cur_nk;
    // tap range of the loader's class in registers: indexing the kernarg arrays with the runtime class cost two s_load +
    // s_waitcnt round trips in EVERY step
    int ld_tap0 = p.ctap0[0], ld_ntap = p.cnt[0];
    bool ld_valid = true;
    const v4i32 rsX2 = XF ? make_srd(p.x2 ? p.x2 : p.x, p.x2 ? p.x2_bytes : p.x_bytes) : rsX;
    const unsigned xf_ldb0 = (unsigned)p.ldx * G::ES, xf_ldb1 = (unsigned)p.ldx2 * G::ES;
    auto xf_setup_rows = [&](unsigned tile, bool valid) {
        if constexpr (XF) {
#pragma unroll
            for (int r = 0; r < G::XR; ++r) {
                const unsigned mu = tile * G::TP + (unsigned)((r * G::NW + wave) * G::RW + rowin);
                const bool ok = valid & (mu < (unsigned)p.Mtotal);
                xoff[r] = ok ? (int)(mu * xf_ldb0) : (int)G_OOB;
                xoff2[r] = ok ? mu * xf_ldb1 : G_OOB;
            }
        }
    };
    // per-piece offsets of k step kt (XF): the step lies in ONE segment (xs_split % 32 == 0); channels beyond C are K padding
    auto xf_prep = [&](int kt, unsigned (&offs)[G::LPS]) -> bool {
        const unsigned k0 = (unsigned)(kt * BK + kc * G::CE);
        const bool s1 = (unsigned)(kt * BK) >= (unsigned)p.xs_split;
        const bool kin = k0 < (unsigned)p.C;
        const unsigned cb = (k0 - (s1 ? (unsigned)p.xs_split : 0u)) * G::ES;
        if constexpr (XF) {
#pragma unroll
            for (int r = 0; r < G::XR; ++r) {
                const unsigned base = s1 ? xoff2[r] : (unsigned)xoff[r];
                offs[r] = (kin && base != G_OOB) ? base + cb : G_OOB;
            }
#pragma unroll
            for (int r = 0; r < G::WR; ++r) offs[G::XR + r] = (kin && woff[r] != G_OOB) ? woff[r] + k0 * G::ES : G_OOB;
        }
        return s1;
    };
    // lin (see GConvP::lin): a row's byte offset, out of range for rows beyond the tensor, + the k offset of the step -- an
    // out-of-range base plus a small k offset is still out of range, so the step needs no select at all.  The generic path spent
    // ~45 VALU instructions and a tap-table read per step on tap decode and halo tests a 1x1 conv does not have; a wavefront issues
    // an instruction every ~8 cycles (profiles/r05_w3_probe_*.txt), so that was a third of the step's time.
    static_assert(!LIN || (!XF && sizeof(T) == 2), "the 1x1 loader: fp16, not transform-on-load");
    constexpr bool lin = LIN;
    auto lin_rows = [&]() __attribute__((always_inline)) {
        if (lin) {
#pragma unroll
            for (int r = 0; r < G::XR; ++r) xoff[r] = xh0[r] < 0 ? (int)G_OOB : xoff[r];
        }
    };
    auto lin_prep = [&](int kt, unsigned (&offs)[G::LPS]) __attribute__((always_inline)) {
        const unsigned k0b = (unsigned)(kt * BK + kc * G::CE) * G::ES;
#pragma unroll
        for (int r = 0; r < G::XR; ++r) offs[r] = (unsigned)xoff[r] + k0b;
#pragma unroll
        for (int r = 0; r < G::WR; ++r) offs[G::XR + r] = woff[r] + k0b;
    };
    if constexpr (XF) xf_setup_rows(ld_tile, true);
    else { g_setup_rows<T, TM, TPX, NWX>(p, ld_tile, true, wave, rowin, kc, xoff, xh0, xw0); lin_rows(); }
    AY_PROBE(AY_PROBE_N - 5);
    __syncthreads();                          // tap table visible
    AY_PROBE(AY_PROBE_N - 6);
#define G_ISSUE(so)                                                                                                                  \
    if constexpr (XF) {                                                                                                              \
        unsigned offs_[G::LPS];                                                                                                      \
        const bool s1_ = xf_prep(ld_kt, offs_);                                                                                      \
        const v4i32 rs_ = s1_ ? rsX2 : rsX;                                                                                          \
        _Pragma("unroll") for (int r = 0; r < G::XR; ++r) glds16(rs_, lds_tiles + (so) + (r * G::NW + wave) * 1024, offs_[r]);       \
        _Pragma("unroll") for (int r = 0; r < G::WR; ++r) glds16(rsW, lds_tiles + (so) + G::XSTAGE + (r * G::NW + wave) * 1024, offs_[G::XR + r]); \
    } else g_issue<T, TM, TPX, NWX>(p, xoff, xh0, xw0, woff, ld_kt, ld_tap0, ld_ntap, sTap, lds_tiles, so, rsX, rsW, wave, kc);
#define G_ADVANCE()                                                                           \
    {                                                                                         \
        if (++ld_kt == ld_nk) {                                                               \
            ld_kt = 0;                                                                        \
            if (++ld_cls == p.ncls) {                                                         \
                ld_cls = 0;                                                                   \
                ld_tile += lstride;                                                           \
                ld_valid = ld_valid && ld_tile < ntiles;                                      \
                if constexpr (XF) xf_setup_rows(ld_tile, ld_valid);                           \
                else { g_setup_rows<T, TM, TPX, NWX>(p, ld_tile, ld_valid, wave, rowin, kc, xoff, xh0, xw0); lin_rows(); }   \
            }                                                                                 \
            ld_nk = G_NK(ld_cls);                                                             \
            ld_tap0 = p.ctap0[ld_cls]; ld_ntap = p.cnt[ld_cls];                               \
        }                                                                                     \
    }
    unsigned so0 = 0, so1 = G::STAGE, so2 = 2 * G::STAGE;
    G_ISSUE(so0)
    G_ADVANCE()
    G_ISSUE(so1)
    G_ADVANCE()

    // XF: the x tile of a step is transformed IN PLACE in its LDS stage, one step before it is consumed, by the lanes whose
    // DMA wrote it: lane l owns bytes [piece * 1024 + l * 16, +16) of every x piece of its wave -- 8 channels of one pixel --
    // so its own counted vmcnt is all the ordering the read needs (no barrier between landing and transform), and the step's
    // barrier publishes the result.  Same arithmetic as k_bn_train_act (fma, v_exp / v_rcp sigmoid, one fp16 rounding): the
    // operand bits equal the materialised activation's.
    const bool xst = XF && p.xa != nullptr && nt == 0;               // this workgroup also stores the activation it forms
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(XF ? p.xa : nullptr, 0, XF ? p.xa_bytes : 0, 0x00020000);
    auto xf_transform = [&](unsigned so, int kt, unsigned tile) {
        if constexpr (XF) {
            const int c0 = kt * BK + kc * G::CE;
            const float4v a0 = *reinterpret_cast<const float4v*>(sXf + c0), a1 = *reinterpret_cast<const float4v*>(sXf + c0 + 4);
            const float4v b0 = *reinterpret_cast<const float4v*>(sXf + xf_cp + c0), b1 = *reinterpret_cast<const float4v*>(sXf + xf_cp + c0 + 4);
            const int sg = (kt * BK) >= p.xs_split ? 1 : 0;
            if (!((p.xf_virt >> sg) & 1)) return;                       // a plain segment: the DMA already delivered the activation
            const bool act = ((p.xf_act >> sg) & 1) != 0;
#pragma unroll
            for (int r = 0; r < G::XR; ++r) {
                half8* q = reinterpret_cast<half8*>(sTiles + so + (r * G::NW + wave) * 1024 + lane * 16);
                half8 h = *q;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = __builtin_fmaf((float)h[e], e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? b0[e & 3] : b1[e & 3]);
                    if (act) u = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
                    h[e] = (half_t)u;
                }
                *q = h;
                if (xst) {
                    // exactly XR stores per transform of a virtual segment (rows beyond the tensor / K padding: out-of-range offset,
                    // dropped by the hardware), so that the step loop's counted waits stay exact
                    const unsigned mu = tile * G::TP + (unsigned)((r * G::NW + wave) * G::RW + rowin);
                    const bool ok = (tile < ntiles) & (mu < (unsigned)p.Mtotal) & (c0 < p.C);
                    const unsigned off = ok ? mu * ((unsigned)p.ldxa * G::ES) + (unsigned)c0 * G::ES : G_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, h), rsA, off, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // written before this wave reaches the next barrier
        }
    };
    // VMEM operations the transform of a step leaves behind (the counted waits of the loop skip over them): XR stores when this
    // workgroup stores back AND the step's segment is virtual
    auto xf_stores = [&](int kt) -> bool { return xst && ((p.xf_virt >> (((kt * BK) >= p.xs_split) ? 1 : 0)) & 1); };
    bool xf_prev_st = false;                                        // did the LAST transform issue stores?
    if constexpr (XF) {
        wait_vm<G::LPS>();                                          // step 0's pieces (the older half of the two issues)
        xf_transform(so0, 0, cur_tile);
        xf_prev_st = xf_stores(0);
    }

    bool after_epi = false;
    AY_PROBE(1);
    while (true) {
        // step s landed (this wave's part), then: everyone's part landed AND everyone finished reading step s-1
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        const bool prev_epi = after_epi;
        if constexpr (XF) {
            // behind step s's pieces: step s+1's pieces, the stores of the last transform (if it stored), an epilogue's stores
            if (xf_prev_st) { if (after_epi) wait_vm<G::LPS + G::XR + NSTK>(); else wait_vm<G::LPS + G::XR>(); }
            else { if (after_epi) wait_vm<G::LPS + NSTK>(); else wait_vm<G::LPS>(); }
        } else {
            if (after_epi) wait_vm<G::LPS + NSTK>(); else wait_vm<G::LPS>();
        }
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        __builtin_amdgcn_s_barrier();
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        if constexpr (sizeof(T) == 2 && G::NACC >= 8) {
            static_assert(BK == 32, "two 16-deep halves per step");
            GFragK<T, TM, TPX, NWX> f0, f1;
            g_fetch_k<T, TM, TPX, NWX>(sTiles + so0, arow, xrow, swz, lane, 0, f0);
            __builtin_amdgcn_sched_barrier(0);
            // address work of step s+2 (covers the LDS latency of the fetch), its DMA pieces one per MFMA of the first half
            unsigned offs[G::LPS];
            v4i32 rsXs = rsX;
            if constexpr (XF) { if (xf_prep(ld_kt, offs)) rsXs = rsX2; }
            else if (lin) lin_prep(ld_kt, offs);
            else g_issue_prep<T, TM, TPX, NWX>(p, xoff, xh0, xw0, woff, ld_kt, ld_tap0, ld_ntap, sTap, kc, offs);
            __builtin_amdgcn_sched_barrier(0);
            g_mma_k_issue<T, TM, TPX, NWX>(f0, acc, offs, lds_tiles, so2, rsXs, rsW, wave);   // -> the stage step s-1 used
            g_fetch_k<T, TM, TPX, NWX>(sTiles + so0, arow, xrow, swz, lane, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            G_ADVANCE()
            __builtin_amdgcn_sched_barrier(0);
            g_mma_k<T, TM, TPX, NWX>(f1, acc);
        } else if constexpr (sizeof(T) == 2) {
            GFrags<T, TM, TPX, NWX> fr;
            g_fetch_frags<T, TM, TPX, NWX>(sTiles + so0, arow, xrow, swz, lane, fr);
            __builtin_amdgcn_sched_barrier(0);
            unsigned offs[G::LPS];
            v4i32 rsXs = rsX;
            if constexpr (XF) { if (xf_prep(ld_kt, offs)) rsXs = rsX2; }
            else if (lin) lin_prep(ld_kt, offs);
            else g_issue_prep<T, TM, TPX, NWX>(p, xoff, xh0, xw0, woff, ld_kt, ld_tap0, ld_ntap, sTap, kc, offs);
            __builtin_amdgcn_sched_barrier(0);
            g_mma_frags_issue<T, TM, TPX, NWX>(fr, acc, offs, lds_tiles, so2, rsXs, rsW, wave);   // step s+2 -> the stage step s-1 used
            G_ADVANCE()
        } else {
            G_ISSUE(so2)
            G_ADVANCE()
            g_mma<T, TM, TPX, NWX>(sTiles + so0, arow, xrow, swz, lane, acc);
        }
        // The last MFMA's result must not be read for passes+2 wait states.  hipcc (ROCm 7.2) covers that hazard inside a
        // basic block but was seen to miss it across the loop back edge (fp32 head variant: the next iteration opened
        // with v_accvgpr_read of the accumulator's last register, which came back stale) -- pad it here, explicitly.
        if constexpr (sizeof(T) == 4) AY_MFMA_PAD("s_nop 15\n\ts_nop 3");
        after_epi = false;
        if constexpr (XF) {
            // step s+1's x pieces (issued during step s-1; behind them only the stores of an epilogue of step s-1 and this
            // step's issues): transform them now, in the shadow of the other waves' MFMAs
            if (xf_prev_st) { if (prev_epi) wait_vm<G::LPS + G::XR + NSTK>(); else wait_vm<G::LPS + G::XR>(); }
            else { if (prev_epi) wait_vm<G::LPS + NSTK>(); else wait_vm<G::LPS>(); }
            const bool wrap = cur_kt + 1 == cur_nk;
            const int nkt = wrap ? 0 : cur_kt + 1;
            xf_transform(so1, nkt, wrap ? cur_tile + lstride : cur_tile);
            xf_prev_st = xf_stores(nkt);
        }
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        if (cur_kt == cur_nk - 1) {
            if constexpr (sizeof(T) == 2) AY_MFMA_PAD("s_nop 11");   // fp16: accumulators are only read here
            g_epilogue<T, TM, EM, TPX, BNR, NWX>(p, cur_tile, p.coah[cur_cls], p.coaw[cur_cls], wp, lane, cbase, want_stats, rsY, acc, ssum, ssq,
                                            sStat, cbase - n0, bctx);
            if constexpr (TILE_RED) {
                // 128-channel tiles: the BNR sums are reduced per tile (DPP row sums + LDS atomics, as k_gconv3) instead of
                // living in 32-64 registers across the step loop next to the accumulators and the fragments
                int lq = lane;
                asm volatile("" : "+v"(lq));
                double* sl = reinterpret_cast<double*>(sStat) + wm * 32 * G::MI;
                const float a = half_reduce_scatter<16 * G::MI>(ssum, lq), b = half_reduce_scatter<16 * G::MI>(ssq, lq);
#pragma unroll
                for (int r = 0; r < 16 * G::MI; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
                if ((BNR || want_stats) && rs_reports<16 * G::MI>(lq)) {
                    const int cl = g_stat_chan<BNR>(rs_index<16 * G::MI>(lq), lq >> 5);
                    atomicAdd(&sl[cl], (double)a);
                    atomicAdd(&sl[TM + cl], (double)b);
                }
            }
            after_epi = true;
        }
        if (++cur_kt == cur_nk) {
            cur_kt = 0;
            if (++cur_cls == p.ncls) {
                cur_cls = 0;
                cur_tile += lstride;
                if (cur_tile >= ntiles) break;
            }
            cur_nk = G_NK(cur_cls);
        }
        const unsigned t = so0; so0 = so1; so1 = so2; so2 = t;
    }
#undef G_ISSUE
#undef G_NK
#undef G_ADVANCE
    wait_vm<0>();                             // the trailing zero-fill DMAs must land before this LDS is released
#ifdef AYOLO_PROBE
    AY_PROBE(AY_PROBE_N - 1);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (probe_on && threadIdx.x < AY_PROBE_N) g_probe[blockIdx.x * AY_PROBE_N + threadIdx.x] = s_probe[threadIdx.x];
#endif
    if constexpr (TILE_RED) {
        __syncthreads();
        if (BNR || want_stats) g_stats_to_global<TM, BNR>(p, reinterpret_cast<const double*>(sStat), tid, n0, slot);
    } else if constexpr (BNR) g_stats_flush<T, TM, G::MI, true>(p, reinterpret_cast<double*>(sStat), tid, lane, wm, n0, slot, ssum, ssq);
    else if (want_stats) g_stats_flush<T, TM, G::MI>(p, reinterpret_cast<double*>(sStat), tid, lane, wm, n0, slot, ssum, ssq);
}


// ---------------------------------------------------------------------------------------------------
// k_gconv3: the 3x3 / stride 1 / same-size case of k_gconv (forward of the Bottleneck 3x3 convs and their dgrad), fp16.
// In flattened (n, h, w) pixel order the input pixel of output pixel m under tap (dh, dw) is m + dh*W + dw, so the three
// taps of one kernel ROW read the same run of input pixels shifted by one: the x rows of a (32-channel chunk, dh) pair are
// DMA'd ONCE -- TP + 2 pixel rows -- and the three dw taps read their B fragments from LDS rows p, p + 1, p + 2 (the
// swizzle is a function of the row, so the shifted reads stay bank-conflict free).  x DMA bytes drop 3x, all DMA bytes of
// a 128 x 256 tile 1.75x (the step loop of these layers is bound by L2 -> LDS DMA throughput x latency).
// What the shift cannot express is the zero padding at the left / right image border (m +- 1 is the neighbouring image
// row there): lanes whose output pixel has w == 0 (dw = -1) or w == W - 1 (dw = +1) zero their B fragment instead.  The
// top / bottom border is the loader's: a row whose h + dh leaves the image gets the out-of-range offset, as in k_gconv.
// Pipeline: one iteration = one 32-channel chunk = 9 sub-steps (dh outer, dw inner), fully unrolled, so stages, tap
// offsets and wait counts are compile-time.  W tiles run 2 sub-steps ahead (3 stages, as k_gconv); x row groups run 2
// groups (6 sub-steps) ahead in 3 stages, each group issued in 3 parts, one per sub-step (the last part is the 2 halo rows,
// wave 0 only; the counted waits simply do not rely on it).
// ---------------------------------------------------------------------------------------------------
template <typename T, int TM, int TPX>
struct GT3 {
    using G = GT<T, TM, TPX>;
    static constexpr int XROWS = G::TP + 16;                 // TP + 2 used; padded to the DMA instruction's 16 rows
    static constexpr int XS = XROWS * G::ROWB;               // bytes per x stage
    static constexpr int WS = G::WSTAGE;
    static constexpr int XP = G::XR / 2;                     // DMA instructions per thread in x parts 0 and 1
    static constexpr size_t lds(int EM, bool BNR) { return 3 * (size_t)XS + 3 * (size_t)WS + G::tail(EM, BNR); }   // see GT::tail
    static_assert(G::XR % 2 == 0 && G::ES == 2, "fp16, 128- or 256-pixel tiles");
};

template <typename T, int TM, int EM, int TPX, bool BNR = false>
__global__ __launch_bounds__(256, 2) void k_gconv3(GConvP p) {
    AY_KERNARG_TOUCH(kt_, GConvP);           // every line of the parameter struct requested at once (gfx950_dma.h)
    using G = GT<T, TM, TPX>;
    using G3 = GT3<T, TM, TPX>;
    constexpr int NSTK = G::NACC * 2;
    constexpr int XR = G::XR, WR = G::WR, XP = G3::XP;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    unsigned char* sX = smem_raw;                                                 // [3][XROWS][32]
    unsigned char* sW = smem_raw + 3 * G3::XS;                                    // [3][TM][32]
    float* sStat = reinterpret_cast<float*>(sW + 3 * G3::WS);                     // [2][TM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % G::WM, wp = wave / G::WM;

    // block -> (channel tile, XCD band, slot): as k_gconv
    const unsigned Lb = blockIdx.x;
#ifdef AYOLO_PROBE
    __shared__ unsigned long long s_probe[AY_PROBE_N];   // tools/gconv_probe.py: [top | wait done | barrier passed | MFMAs issued] per sub-step
    const bool probe_on = blockIdx.x < 512;
    int probe_k = 2;
    if (threadIdx.x < AY_PROBE_N) s_probe[threadIdx.x] = 0;
    __syncthreads();
    AY_PROBE(0);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 3] = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned xcd = Lb & 7u, idx = Lb >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const int n0 = (int)nt * TM;
    const unsigned ntiles_all = (unsigned)((p.Mtotal + G::TP - 1) / G::TP);
    const unsigned tpx = (ntiles_all + 7) / 8;
    const unsigned band_lo = xcd * tpx;
    const unsigned ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = (unsigned)p.nslots / 8u;
    unsigned cur_tile = band_lo + lslot;
    kt_.done();
    if (cur_tile >= ntiles) return;

    if constexpr (EM == 2 || EM == 4) {
        for (int i = tid; i < TM; i += 256) {
            const bool in = n0 + i < p.Nout;
            const float* any = p.scale ? p.scale : p.shift;       // both loads together (one memory round trip in the prologue)
            float sc_ = 1.0f, sh_ = 0.0f;
            if (any) {
                const float a_ = (p.scale ? p.scale : any)[in ? n0 + i : 0], b_ = (p.shift ? p.shift : any)[in ? n0 + i : 0];
                sc_ = (in && p.scale) ? a_ : 1.0f;
                sh_ = (in && p.shift) ? b_ : 0.0f;
            }
            sStat[i] = sc_;
            sStat[TM + i] = sh_;
        }
    }
    if constexpr (BNR) g_bnr_setup<TM>(p, sStat + 4 * TM, n0, tid);
    const BnrCtx<G::MI> bctx = BNR ? g_bnr_ctx<G::MI>(p, sStat + 4 * TM, n0 + wm * 32 * G::MI) : BnrCtx<G::MI>{};

    const v4i32 rsX = make_srd(p.x, p.x_bytes), rsW = make_srd(p.w, p.w_bytes);
    const unsigned lds_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sX);
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sW);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // loader lane geometry (fp16: 4 chunks per row, 16 rows per wave-instruction)
    const int slotc = lane & (G::CPR - 1);
    const int rowin = lane / G::CPR;
    const int kc = slotc ^ (lane >> 4);
    const int kcb = kc * G::CE * G::ES;                     // byte offset of this lane's chunk inside a 32-channel slice

    unsigned woff[WR];
#pragma unroll
    for (int r = 0; r < WR; ++r) {
        const int row = (r * 4 + wave) * G::RW + rowin;
        const bool ok = (row < TM) & (n0 + row < p.Nout);
        woff[r] = ok ? (unsigned)(n0 + row) * (unsigned)p.ldw * G::ES + (unsigned)kcb : G_OOB;
    }
    // weight column byte offsets of the nine taps (SGPRs; indices are compile-time in the unrolled body)
    int wtap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wtap[t] = __builtin_amdgcn_readfirstlane((int)p.r3wt[t] * p.C * G::ES);
    const int dhstep = p.XW * p.ldx * G::ES;                // bytes per input image row
    const int nC = (p.C + BK - 1) / BK;                     // the last chunk may be partly beyond C (80, 48 channels ...)
    const int cmax = p.C - kc * G::CE;                     // chunk c holds channels of this lane's 16-byte slot iff c * BK < cmax

    // x rows of the loader's tile: row j of the stage <-> flattened pixel tile * TP + j - 1
    int xoff[XR + 1], xh0[XR + 1];
#define G3_SETUP(tile_, valid_)                                                                            \
    {                                                                                                      \
        _Pragma("unroll") for (int r = 0; r <= XR; ++r) {                                                  \
            const int j = r < XR ? (r * 4 + wave) * 16 + rowin : G::TP + rowin;                            \
            const unsigned mu = (tile_) * G::TP + (unsigned)j - 1u;                                        \
            const bool ok = (valid_) & (mu < (unsigned)p.Mtotal) & (r < XR || rowin < 2);                  \
            const unsigned t_ = fdiv(mu, p.dOW);                                                           \
            const int ow_ = (int)(mu - t_ * (unsigned)p.OW);                                               \
            const unsigned n_ = fdiv(t_, p.dOH);                                                           \
            const int oh_ = (int)(t_ - n_ * (unsigned)p.OH);                                               \
            xoff[r] = (int)(((n_ * (unsigned)p.XH + (unsigned)oh_) * (unsigned)p.XW + (unsigned)ow_) * (unsigned)p.ldx * G::ES) + kcb; \
            xh0[r] = ok ? oh_ : -100000;                                                                   \
        }                                                                                                  \
    }
    // part `t_` (0, 1: XP full instructions; 2: the halo rows, wave 0) of the x row group (chunk c_, dh = g_ - 1) -> x stage g_
#define G3_XPART(t_, g_, c_)                                                                               \
    {                                                                                                      \
        const int cb_ = (c_) * (BK * G::ES), dh_ = (g_) - 1;                                               \
        if ((t_) < 2) {                                                                                    \
            _Pragma("unroll") for (int q_ = 0; q_ < XP; ++q_) {                                            \
                const int r = (t_) * XP + q_;                                                              \
                const bool ok = ((unsigned)(xh0[r] + dh_) < (unsigned)p.XH) & ((c_) * BK < cmax);          \
                const unsigned off = ok ? (unsigned)(xoff[r] + dh_ * dhstep + cb_) : G_OOB;                \
                glds16(rsX, lds_x + (g_) * G3::XS + (r * 4 + wave) * 1024, off);                           \
            }                                                                                              \
        } else if (wave == 0) {                                                                            \
            const bool ok = ((unsigned)(xh0[XR] + dh_) < (unsigned)p.XH) & ((c_) * BK < cmax);             \
            const unsigned off = ok ? (unsigned)(xoff[XR] + dh_ * dhstep + cb_) : G_OOB;                   \
            glds16(rsX, lds_x + (g_) * G3::XS + XR * 4 * 1024, off);                                       \
        }                                                                                                  \
    }
    // W tile of sub-step q_ (tap q_) of chunk c_ -> W stage q_ % 3
#define G3_W(q_, c_)                                                                                       \
    {                                                                                                      \
        const unsigned col_ = (unsigned)(wtap[q_] + (c_) * (BK * G::ES)) | ((c_) * BK < cmax ? 0u : G_OOB);\
        _Pragma("unroll") for (int r = 0; r < WR; ++r)                                                     \
            glds16(rsW, lds_w + ((q_) % 3) * G3::WS + (r * 4 + wave) * 1024, (woff[r] + col_) | (col_ & G_OOB)); \
    }

    // MFMA fragment geometry
    const int arow = (wm * 32 * G::MI + (lane & 31)) * G::ROWB;
    const int swzA = ((lane & 31) / G::RPB) & (G::CPR - 1);
    int brow0 = wp * G::NI * 32 + (lane & 31) + 1;           // stage row of this lane's pixel in block ni = 0, dw = 0
    const int hi = lane >> 5;

    float16v acc[G::NACC];
#pragma unroll
    for (int i = 0; i < G::NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    // BN statistics: reduced per tile (DPP row sums + LDS atomics into sStat) instead of living in 64 registers across
    // the nine unrolled sub-steps, which is what made the forward variant spill
    const bool want_stats = !BNR && (EM == 0) && (p.stats != nullptr);
    const bool any_stats = BNR || want_stats;
    if constexpr (EM == 0 || BNR) {
        for (int i = tid; i < 2 * TM; i += 256) reinterpret_cast<double*>(sStat)[i] = 0.0;
    }
    // ... except for the narrow channel tiles (32 registers, plenty of room, and only 9-18 sub-steps per tile to amortise a
    // per-tile reduction over): those keep them in registers across tiles like k_gconv
    constexpr bool SREG = (EM == 0 || BNR) && (TM <= 64);
    float rsum[SREG ? 16 * G::MI : 1], rsq[SREG ? 16 * G::MI : 1];
#pragma unroll
    for (int r = 0; r < (SREG ? 16 * G::MI : 1); ++r) { rsum[r] = 0.0f; rsq[r] = 0.0f; }
    const int cbase = n0 + wm * 32 * G::MI + 4 * (lane >> 5);

    // loader cursor of the x rows (two row groups ahead of the compute cursor)
    unsigned x_tile = cur_tile;
    bool x_valid = true;
    G3_SETUP(x_tile, true)
    __syncthreads();                          // sAff visible
    // prologue: x row groups 0 and 1 of chunk 0, W tiles of sub-steps 0 and 1
    G3_XPART(0, 0, 0) G3_XPART(1, 0, 0) G3_XPART(2, 0, 0)
    G3_XPART(0, 1, 0) G3_XPART(1, 1, 0) G3_XPART(2, 1, 0)
    G3_W(0, 0)
    G3_W(1, 0)
    AY_PROBE(1);

    int c = 0;
    bool after_epi = false;
    unsigned mleft = 0, mright = 0;           // bit ni: this lane's pixel of block ni sits in image column 0 / W - 1
    while (true) {
        if (c == 0) {
            mleft = 0; mright = 0;
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) {
                const unsigned mu = cur_tile * G::TP + (unsigned)(wp * G::NI * 32 + ni * 32 + (lane & 31));
                const unsigned t_ = fdiv(mu, p.dOW);
                const int ow_ = (int)(mu - t_ * (unsigned)p.OW);
                mleft |= (ow_ == 0 ? 1u : 0u) << ni;
                mright |= (ow_ == p.OW - 1 ? 1u : 0u) << ni;
            }
        }
        const int cn = c + 1 == nC ? 0 : c + 1;            // chunk of the next iteration (its tile may be the next one)
        // one sub-step: wait for W(q) (and the x group, issued earlier), barrier, fetch, issue ahead, MFMA
#define G3_SUB(q_)                                                                                         \
        {                                                                                                  \
            constexpr int g_ = (q_) / 3, j_ = (q_) % 3;                                                    \
            /* keep the per-sub-step address arithmetic per sub-step: hoisted across the nine unrolled bodies it costs   \
               ~30 VGPRs (the forward variant then spills) */                                                            \
            _Pragma("unroll") for (int r = 0; r <= XR; ++r) asm volatile("" : "+v"(xoff[r]), "+v"(xh0[r]));  \
            asm volatile("" : "+v"(brow0));                                                                \
            G3_PROBE();                                                                                    \
            if ((q_) == 0) { if (after_epi) wait_vm<WR + NSTK>(); else wait_vm<WR>(); }                    \
            else if (j_ == 0) wait_vm<WR>();                                                               \
            else wait_vm<WR + XP>();                                                                       \
            G3_PROBE();                                                                                    \
            __builtin_amdgcn_s_barrier();                                                                  \
            G3_PROBE();                                                                                    \
            const unsigned char* stW = sW + ((q_) % 3) * G3::WS + arow;                                    \
            const int jr = brow0 + (j_ - 1);                                                               \
            const int swzB = (jr >> 2) & (G::CPR - 1);                                                     \
            const unsigned char* stX = sX + g_ * G3::XS + jr * G::ROWB;                                    \
            const unsigned zmask = j_ == 0 ? mleft : (j_ == 2 ? mright : 0u);                              \
            half8 fa0[G::MI], fb0[G::NI], fa1[G::MI], fb1[G::NI];                                          \
            {                                                                                              \
                const int sa = ((0 * 2 + hi) ^ swzA) * 16, sb = ((0 * 2 + hi) ^ swzB) * 16;                \
                _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi) fa0[mi] = *reinterpret_cast<const half8*>(stW + mi * 32 * G::ROWB + sa); \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) fb0[ni] = *reinterpret_cast<const half8*>(stX + ni * 32 * G::ROWB + sb); \
            }                                                                                              \
            if constexpr (G::NACC < 8) {                                                                   \
                const int sa = ((1 * 2 + hi) ^ swzA) * 16, sb = ((1 * 2 + hi) ^ swzB) * 16;                \
                _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi) fa1[mi] = *reinterpret_cast<const half8*>(stW + mi * 32 * G::ROWB + sa); \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) fb1[ni] = *reinterpret_cast<const half8*>(stX + ni * 32 * G::ROWB + sb); \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            /* ahead: part j_ of the x group two groups on, the W tile two sub-steps on.  The ADDRESS work sits here, under the \
               fragment-fetch latency; the DMA instructions themselves go one behind each of the first MFMAs (a piece costs ~60 \
               issue cycles among bare MFMAs and 100-185 in a block of its own: round 3 did this for k_gconv, g_mma_k_issue) */ \
            if ((q_) == 3) {                                                                               \
                if (c + 1 == nC) {                                                                         \
                    x_tile += lstride;                                                                     \
                    x_valid = x_valid && x_tile < ntiles;                                                  \
                    G3_SETUP(x_tile, x_valid)                                                              \
                }                                                                                          \
            }                                                                                              \
            constexpr int gx_ = g_ == 0 ? 2 : g_ - 1;                      /* x stage the part goes to */  \
            constexpr int qw_ = ((q_) + 2) % 9;                            /* tap of the W tile */         \
            constexpr int NPX_ = j_ < 2 ? XP : 1;                          /* x pieces (the halo part: one, wave 0 only) */ \
            const int cx_ = g_ == 0 ? c : cn, cw_ = (q_) + 2 < 9 ? c : cn;                                 \
            unsigned pxo_[NPX_], pwo_[WR];                                                                 \
            {                                                                                              \
                const int cb_ = cx_ * (BK * G::ES), dh_ = gx_ - 1;                                         \
                _Pragma("unroll") for (int k_ = 0; k_ < NPX_; ++k_) {                                      \
                    const int r = j_ < 2 ? j_ * XP + k_ : XR;                                              \
                    const bool ok = ((unsigned)(xh0[r] + dh_) < (unsigned)p.XH) & (cx_ * BK < cmax);       \
                    pxo_[k_] = ok ? (unsigned)(xoff[r] + dh_ * dhstep + cb_) : G_OOB;                      \
                }                                                                                          \
                const unsigned col_ = (unsigned)(wtap[qw_] + cw_ * (BK * G::ES)) | (cw_ * BK < cmax ? 0u : G_OOB); \
                _Pragma("unroll") for (int r = 0; r < WR; ++r) pwo_[r] = (woff[r] + col_) | (col_ & G_OOB); \
            }                                                                                              \
            /* piece k_ of this sub-step, in the order the counted waits assume: the x part, then the W tile */ \
            auto g3_piece = [&](int k_) __attribute__((always_inline)) {                                   \
                if (k_ < NPX_) {                                                                           \
                    if (j_ < 2) glds16(rsX, lds_x + gx_ * G3::XS + ((j_ * XP + k_) * 4 + wave) * 1024, pxo_[k_ < NPX_ ? k_ : 0]); \
                    else if (wave == 0) glds16(rsX, lds_x + gx_ * G3::XS + XR * 4 * 1024, pxo_[0]);        \
                } else if (k_ < NPX_ + WR) {                                                               \
                    const int r = k_ - NPX_;                                                               \
                    glds16(rsW, lds_w + (qw_ % 3) * G3::WS + (r * 4 + wave) * 1024, pwo_[r < WR ? r : 0]); \
                }                                                                                          \
            };                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (zmask) {                                                                                   \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni)                                       \
                    if ((zmask >> ni) & 1u) { _Pragma("unroll") for (int e = 0; e < 8; ++e) fb0[ni][e] = (_Float16)0.0f; } \
            }                                                                                              \
            _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi)                                           \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) {                                     \
                    mma_step(fa0[mi], fb0[ni], acc[mi * G::NI + ni]);                                      \
                    if (mi * G::NI + ni < NPX_ + WR) { g3_piece(mi * G::NI + ni); __builtin_amdgcn_sched_barrier(0); } \
                }                                                                                          \
            /* tiles with fewer MFMAs in the first half than pieces: the rest of the pieces behind them */ \
            _Pragma("unroll") for (int k_ = G::MI * G::NI; k_ < NPX_ + WR; ++k_) g3_piece(k_);             \
            if constexpr (G::NACC >= 8) {                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                         \
                const int sa = ((1 * 2 + hi) ^ swzA) * 16, sb = ((1 * 2 + hi) ^ swzB) * 16;                \
                _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi) fa1[mi] = *reinterpret_cast<const half8*>(stW + mi * 32 * G::ROWB + sa); \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) fb1[ni] = *reinterpret_cast<const half8*>(stX + ni * 32 * G::ROWB + sb); \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (zmask) {                                                                                   \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni)                                       \
                    if ((zmask >> ni) & 1u) { _Pragma("unroll") for (int e = 0; e < 8; ++e) fb1[ni][e] = (_Float16)0.0f; } \
            }                                                                                              \
            _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi)                                           \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) mma_step(fa1[mi], fb1[ni], acc[mi * G::NI + ni]); \
            G3_PROBE();                                                                                    \
        }
#ifdef AYOLO_PROBE
#define G3_PROBE() do { AY_PROBE(probe_k); ++probe_k; } while (0)
#else
#define G3_PROBE() do { } while (0)
#endif
        G3_SUB(0) G3_SUB(1) G3_SUB(2) G3_SUB(3) G3_SUB(4) G3_SUB(5) G3_SUB(6) G3_SUB(7) G3_SUB(8)
        after_epi = false;
        if (c == nC - 1) {
            AY_MFMA_PAD("s_nop 11");          // the accumulators are read right after the last MFMA
            float ssum[16 * G::MI], ssq[16 * G::MI];
#pragma unroll
            for (int r = 0; r < 16 * G::MI; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
            if constexpr (SREG) g_epilogue<T, TM, EM, TPX, BNR>(p, cur_tile, p.oah, p.oaw, wp, lane, cbase, want_stats, rsY, acc, rsum, rsq, sStat, cbase - n0, bctx);
            else g_epilogue<T, TM, EM, TPX, BNR>(p, cur_tile, p.oah, p.oaw, wp, lane, cbase, want_stats, rsY, acc, ssum, ssq, sStat, cbase - n0, bctx);
            if (any_stats && !SREG) {
                int lq = lane;                              // opaque: keeps the LDS addresses below from being hoisted out of the
                asm volatile("" : "+v"(lq));                // tile loop (and spilled: every reload would drain the DMA queue)
                double* sl = reinterpret_cast<double*>(sStat) + wm * 32 * G::MI;
                // (one array after the other, each with its atomic: the 128 x 256 tile has no register to spare for a second result)
                const int cl = g_stat_chan<BNR>(rs_index<16 * G::MI>(lq), lq >> 5);
                const float a = half_reduce_scatter<16 * G::MI>(ssum, lq);
                if (rs_reports<16 * G::MI>(lq)) atomicAdd(&sl[cl], (double)a);
                const float b = half_reduce_scatter<16 * G::MI>(ssq, lq);
                if (rs_reports<16 * G::MI>(lq)) atomicAdd(&sl[TM + cl], (double)b);
            }
            after_epi = true;
            cur_tile += lstride;
            if (cur_tile >= ntiles) break;
        }
        c = cn;
    }
#undef G3_SUB
#undef G3_PROBE
#undef G3_W
#undef G3_XPART
#undef G3_SETUP
    wait_vm<0>();                             // trailing DMAs must land before this LDS is released
#ifdef AYOLO_PROBE
    AY_PROBE(AY_PROBE_N - 1);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (probe_on && threadIdx.x < AY_PROBE_N) g_probe[blockIdx.x * AY_PROBE_N + threadIdx.x] = s_probe[threadIdx.x];
#endif
    if constexpr (SREG) {
        if (any_stats) g_stats_flush<T, TM, G::MI, BNR>(p, reinterpret_cast<double*>(sStat), tid, lane, wm, n0, slot, rsum, rsq);
        return;
    }
    if (any_stats) {
        __syncthreads();
        g_stats_to_global<TM, BNR>(p, reinterpret_cast<const double*>(sStat), tid, n0, slot);
    }
}


// ---------------------------------------------------------------------------------------------------
// k_dgrad_s2: data gradient of a 3x3 / stride 2 / pad 1 conv for <= 64 input channels (the first two down-sampling convs
// of the backbone: 420 / 210 MB of dx, the widest activations of the backward pass).
// Output pixel (2*oh + a, 2*ow + b) of residue class (a, b) sums the taps (i, j) with a + 1 - i and b + 1 - j even, reading
// dy at (oh + dh', ow + dw') with dh' = (a + 1 - i) / 2, dw' = (b + 1 - j) / 2: the NINE (class, tap) products of the four
// classes read only FOUR shifts (dh', dw' in {0, 1}) of the dy tile.  k_gconv walks the classes one after the other and
// DMAs the dy tile once per tap (nine times, 5 pieces per wave next to 4 MFMAs); here one workgroup keeps the accumulators
// of all four classes (4 x NI blocks), a dy row group (TP + 1 pixels, one per dh') is DMA'd once per 32-channel chunk and
// serves every product that reads it -- the k_gconv3 row-sharing idea across classes -- and the products that read the
// same shift form ONE step (one barrier, one B-fragment fetch, up to four W tiles):
//   A: shift (0,0) x classes 0,1,2,3   B: (0,1) x classes 1,3   C: (1,0) x classes 2,3   D: (1,1) x class 3
// Per chunk and wave 13.5 LDS-DMA pieces and 4 barriers instead of 45 and 9.
// The W tiles of a step land in the step's own LDS region two steps ahead.  x row groups: 3 stages,
// the group of the NEXT chunk is issued whole at the first step (A / C) of the current chunk's same group, into the stage
// that is neither being read nor in flight.  Right border (dw' = 1 at ow = OW - 1): the lane zeroes its B fragment; bottom
// border (dh' = 1 at oh = OH - 1): the loader's out-of-range offset.
// ---------------------------------------------------------------------------------------------------
template <typename T, int TM, int EM, int TPX, bool BNR = false>
__global__ __launch_bounds__(256, 2) void k_dgrad_s2(GConvP p) {
    AY_KERNARG_TOUCH(kt_, GConvP);           // every line of the parameter struct requested at once (gfx950_dma.h)
    using G = GT<T, TM, TPX>;
    static_assert(sizeof(T) == 2 && G::MI == 1 && G::NI == 2 && (TM == 32 || TM == 64), "fp16, 32- or 64-channel tiles");
    constexpr int XR = G::XR;
    constexpr int XS = (G::TP + 16) * G::ROWB;                // TP + 1 rows used, padded to the DMA instruction's 16 rows
    // W tiles: every step has its own region ([taps][TM rows][32]); a region is written two steps before its step reads it
    // and read by that one step only, so no rotation is needed (A: 4 tap tiles, B: 2, C: 2, D: 1)
    constexpr int TMB = TM * G::ROWB;
    constexpr int WOA = 0, WOB = 4 * TMB, WOC = 6 * TMB, WOD = 8 * TMB;
    // W pieces per wave for a step with 4 / 2 / 1 taps: rows = taps * TM, 16 rows per piece, 4 waves
    constexpr int W4 = 4 * TM / 64, W2 = 2 * TM / 64, W1 = TM / 64 > 0 ? TM / 64 : 1;
    constexpr int NSTK = 4 * G::NACC * 2;                     // stores per thread of the four class epilogues
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    unsigned char* sX = smem_raw;                             // [3][TP + 16][32]
    unsigned char* sW = smem_raw + 3 * XS;                    // [9 tap tiles][TM][32]
    float* sStat = reinterpret_cast<float*>(sW + (9 + (TM == 32 ? 1 : 0)) * TMB);   // BNR: sums [2][TM] + constants [4][TM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % G::WM, wp = wave / G::WM;

    const unsigned Lb = blockIdx.x;
    const unsigned xcd = Lb & 7u, idx = Lb >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const int n0 = (int)nt * TM;
    const unsigned ntiles_all = (unsigned)((p.Mtotal + G::TP - 1) / G::TP);
    const unsigned tpx = (ntiles_all + 7) / 8;
    const unsigned band_lo = xcd * tpx;
    const unsigned ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = (unsigned)p.nslots / 8u;
    unsigned cur_tile = band_lo + lslot;
    kt_.done();
    if (cur_tile >= ntiles) return;

    const v4i32 rsX = make_srd(p.x, p.x_bytes), rsW = make_srd(p.w, p.w_bytes);
    const unsigned lds_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sX);
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sW);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    if constexpr (BNR) {
        for (int i = tid; i < 2 * TM; i += 256) reinterpret_cast<double*>(sStat)[i] = 0.0;
        g_bnr_setup<TM>(p, sStat + 4 * TM, n0, tid);
        __syncthreads();
    }
    const BnrCtx<G::MI> bctx = BNR ? g_bnr_ctx<G::MI>(p, sStat + 4 * TM, n0 + wm * 32 * G::MI) : BnrCtx<G::MI>{};

    const int slotc = lane & (G::CPR - 1);
    const int rowin = lane / G::CPR;
    const int kc = slotc ^ (lane >> 4);
    const int kcb = kc * G::CE * G::ES;
    const int nC = (p.C + BK - 1) / BK;
    const int cmax = p.C - kc * G::CE;
    const int dhstep = p.XW * p.ldx * G::ES;

    // W loader: piece r of this wave covers stage rows (r * 4 + wave) * 16 + rowin = tap t * TM + row-in-tap, with
    // t = (r * 4 + wave) * 16 / TM and a row-in-tap that does not depend on r
    const int wrow = ((wave * 16) % TM) + rowin;
    const unsigned wbase = (n0 + wrow < p.Nout) ? (unsigned)(n0 + wrow) * (unsigned)p.ldw * G::ES + (unsigned)kcb : G_OOB;
    // column byte offset of the weight tap that piece r of this wave loads in steps A / B / C / D (out of range: no tap)
    int wcA[W4], wcB[W2], wcC[W2], wcD[W1];
#pragma unroll
    for (int r = 0; r < W4; ++r) wcA[r] = __builtin_amdgcn_readfirstlane((int)p.s2wt[(r * 4 + wave) * 16 / TM] * p.C * G::ES);
#pragma unroll
    for (int r = 0; r < W2; ++r) wcB[r] = __builtin_amdgcn_readfirstlane((int)p.s2wt[4 + (r * 4 + wave) * 16 / TM] * p.C * G::ES);
#pragma unroll
    for (int r = 0; r < W2; ++r) wcC[r] = __builtin_amdgcn_readfirstlane((int)p.s2wt[6 + (r * 4 + wave) * 16 / TM] * p.C * G::ES);
#pragma unroll
    for (int r = 0; r < W1; ++r) {
        const int t = (r * 4 + wave) * 16 / TM;                 // TM = 32: waves 2, 3 would be tap 1 of a one-tap step
        wcD[r] = __builtin_amdgcn_readfirstlane(t < 1 ? (int)p.s2wt[8] * p.C * G::ES : (int)G_OOB);
    }

    int xoff[XR + 1], xh0[XR + 1];
#define S2_SETUP(tile_, valid_)                                                                            \
    {                                                                                                      \
        _Pragma("unroll") for (int r = 0; r <= XR; ++r) {                                                  \
            const int j = r < XR ? (r * 4 + wave) * 16 + rowin : G::TP + rowin;                            \
            const unsigned mu = (tile_) * G::TP + (unsigned)j;                                             \
            const bool ok = (valid_) & (mu < (unsigned)p.Mtotal) & (r < XR || rowin < 1);                  \
            const unsigned t_ = fdiv(mu, p.dOW);                                                           \
            const int ow_ = (int)(mu - t_ * (unsigned)p.OW);                                               \
            const unsigned n_ = fdiv(t_, p.dOH);                                                           \
            const int oh_ = (int)(t_ - n_ * (unsigned)p.OH);                                               \
            xoff[r] = (int)(((n_ * (unsigned)p.XH + (unsigned)oh_) * (unsigned)p.XW + (unsigned)ow_) * (unsigned)p.ldx * G::ES) + kcb; \
            xh0[r] = ok ? oh_ : -100000;                                                                   \
        }                                                                                                  \
    }
    // the whole x row group dh' = g_ of chunk c_ -> x stage at byte offset so_
#define S2_X(g_, c_, so_)                                                                                  \
    {                                                                                                      \
        const int cb_ = (c_) * (BK * G::ES);                                                               \
        const bool cok_ = (c_) * BK < cmax;                                                                \
        _Pragma("unroll") for (int r = 0; r < XR; ++r) {                                                   \
            const bool ok = ((unsigned)(xh0[r] + (g_)) < (unsigned)p.XH) & cok_;                           \
            const unsigned off = ok ? (unsigned)(xoff[r] + (g_) * dhstep + cb_) : G_OOB;                   \
            glds16(rsX, lds_x + (so_) + (r * 4 + wave) * 1024, off);                                       \
        }                                                                                                  \
        if (wave == 0) {                                                                                   \
            const bool ok = ((unsigned)(xh0[XR] + (g_)) < (unsigned)p.XH) & cok_;                          \
            const unsigned off = ok ? (unsigned)(xoff[XR] + (g_) * dhstep + cb_) : G_OOB;                  \
            glds16(rsX, lds_x + (so_) + XR * 4 * 1024, off);                                               \
        }                                                                                                  \
    }
    // the W tiles of a step (column offsets wc_[0 .. n_)) of chunk c_ -> the step's region at byte offset wo_
#define S2_W(wc_, n_, c_, wo_)                                                                             \
    {                                                                                                      \
        const unsigned cbad_ = (c_) * BK < cmax ? 0u : G_OOB;                                              \
        _Pragma("unroll") for (int r = 0; r < (n_); ++r) {                                                 \
            const unsigned col_ = (unsigned)(wc_[r] + (c_) * (BK * G::ES));                                \
            glds16(rsW, lds_w + (wo_) + (r * 4 + wave) * 1024, (wbase + col_) | ((wbase | col_ | cbad_) & G_OOB)); \
        }                                                                                                  \
    }

    const int arow = (wm * 32 + (lane & 31)) * G::ROWB;
    const int swzA = ((lane & 31) / G::RPB) & (G::CPR - 1);
    int brow0 = wp * G::NI * 32 + (lane & 31);
    const int hi = lane >> 5;

    float16v acc[4][G::NACC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < G::NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][i][r] = 0.0f;
    const int cbase = n0 + wm * 32 + 4 * (lane >> 5);

    unsigned x_tile = cur_tile;
    bool x_valid = true;
    S2_SETUP(x_tile, true)
    // stage byte offsets of the current chunk's two row groups; the third stage is the one to fill next
    unsigned xg0 = 0, xg1 = XS;
    S2_X(0, 0, xg0)
    S2_X(1, 0, xg1)
    S2_W(wcA, W4, 0, WOA)
    S2_W(wcB, W2, 0, WOB)

    int c = 0;
    bool after_epi = false;
    unsigned mright = 0;
    while (true) {
        if (c == 0) {
            mright = 0;
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) {
                const unsigned mu = cur_tile * G::TP + (unsigned)(wp * G::NI * 32 + ni * 32 + (lane & 31));
                const unsigned t_ = fdiv(mu, p.dOW);
                const int ow_ = (int)(mu - t_ * (unsigned)p.OW);
                mright |= (ow_ == p.OW - 1 ? 1u : 0u) << ni;
            }
        }
        const int cn = c + 1 == nC ? 0 : c + 1;
        const unsigned xfree = 3u * XS - xg0 - xg1;        // the stage neither group of this chunk lives in
        // one step: NT_ products that read shift (dh', DW_) of row group at xs_, W stage ST_; class of product t = CLS_(t)
#define S2_STEP(WO_, NT_, DW_, xs_, WAITN_, AHEAD_, C0_, C1_, C2_, C3_)                                    \
        {                                                                                                  \
            _Pragma("unroll") for (int r = 0; r <= XR; ++r) asm volatile("" : "+v"(xoff[r]), "+v"(xh0[r])); \
            asm volatile("" : "+v"(brow0));                                                                \
            WAITN_                                                                                         \
            __builtin_amdgcn_s_barrier();                                                                  \
            const int jr = brow0 + (DW_);                                                                  \
            const int swzB = (jr >> 2) & (G::CPR - 1);                                                     \
            const unsigned char* stX = sX + (xs_) + jr * G::ROWB;                                          \
            const unsigned char* stW = sW + (WO_) + arow;                                                  \
            half8 fb[2][G::NI], fa[2][NT_];                                                                \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                             \
                const int sa = ((kk * 2 + hi) ^ swzA) * 16, sb = ((kk * 2 + hi) ^ swzB) * 16;              \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) fb[kk][ni] = *reinterpret_cast<const half8*>(stX + ni * 32 * G::ROWB + sb); \
                _Pragma("unroll") for (int t = 0; t < (NT_); ++t) fa[kk][t] = *reinterpret_cast<const half8*>(stW + t * TMB + sa); \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            AHEAD_                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if ((DW_) && mright) {                                                                         \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni)                                       \
                    if ((mright >> ni) & 1u) {                                                             \
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) { fb[0][ni][e] = (_Float16)0.0f; fb[1][ni][e] = (_Float16)0.0f; } \
                    }                                                                                      \
            }                                                                                              \
            constexpr int cls_[4] = {C0_, C1_, C2_, C3_};                                                  \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                               \
                _Pragma("unroll") for (int t = 0; t < (NT_); ++t)                                          \
                    _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) mma_step(fa[kk][t], fb[kk][ni], acc[cls_[t]][ni]); \
        }
        // step A: rows g0 of this chunk.  Ahead: the next chunk's g0 rows into the free stage (its tile may be the next one),
        // then the W tiles of step C
        S2_STEP(WOA, 4, 0, xg0,
                if (after_epi) wait_vm<W2 + NSTK>(); else wait_vm<W2>();,
                if (c + 1 == nC) { x_tile += lstride; x_valid = x_valid && x_tile < ntiles; S2_SETUP(x_tile, x_valid) }
                S2_X(0, cn, xfree) S2_W(wcC, W2, c, WOC),
                0, 1, 2, 3)
        S2_STEP(WOB, 2, 1, xg0, wait_vm<XR + W2>();, S2_W(wcD, W1, c, WOD), 1, 3, 0, 0)
        // step C: rows g1.  Ahead: the next chunk's g1 rows into this chunk's g0 stage (read for the last time in step B)
        S2_STEP(WOC, 2, 0, xg1, wait_vm<W1>();, S2_X(1, cn, xg0) S2_W(wcA, W4, cn, WOA), 2, 3, 0, 0)
        S2_STEP(WOD, 1, 1, xg1, wait_vm<XR + W4>();, S2_W(wcB, W2, cn, WOB), 3, 0, 0, 0)
        after_epi = false;
        if (c == nC - 1) {
            AY_MFMA_PAD("s_nop 11");
            float ssum[16], ssq[16];
            if constexpr (BNR) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                g_epilogue<T, TM, EM, TPX, BNR>(p, cur_tile, k >> 1, k & 1, wp, lane, cbase, false, rsY, acc[k], ssum, ssq, nullptr, cbase - n0, bctx);
            if constexpr (BNR) {
                // the four classes' pixels of this tile: reduced per tile (DPP row sums + LDS atomics), as k_gconv3 does --
                // 32 more registers living across the step loop do not fit next to the 128 accumulators
                int lq = lane;
                asm volatile("" : "+v"(lq));
                double* sl = reinterpret_cast<double*>(sStat) + wm * 32;
                const float a = half_reduce_scatter<16>(ssum, lq), b = half_reduce_scatter<16>(ssq, lq);
                if (rs_reports<16>(lq)) {
                    const int cl = g_stat_chan<true>(rs_index<16>(lq), lq >> 5);
                    atomicAdd(&sl[cl], (double)a);
                    atomicAdd(&sl[TM + cl], (double)b);
                }
            }
            after_epi = true;
            cur_tile += lstride;
            if (cur_tile >= ntiles) break;
        }
        c = cn;
        { const unsigned t0_ = xg0; xg0 = xfree; xg1 = t0_; }
    }
#undef S2_STEP
#undef S2_W
#undef S2_X
#undef S2_SETUP
    wait_vm<0>();
    if constexpr (BNR) {
        __syncthreads();
        g_stats_to_global<TM, true>(p, reinterpret_cast<const double*>(sStat), tid, n0, slot);
    }
}


// ---------------------------------------------------------------------------------------------------
// k_gconv_s2f: forward 3x3 / stride 2 / pad 1 on an
// even-sized map.  Output pixel (oh, ow) reads input row 2*oh + dh and columns 2*ow - 1, 2*ow, 2*ow + 1.  For one dh the taps
// dw = -1 / +1 read the ODD input columns: in flattened output order one run of odd columns (stage row j <-> output-aligned
// pixel m0 + j - 1, input column 2*ow' + 1) serves both -- dw = +1 reads row p + 1, dw = -1 reads row p, the lane with ow == 0
// zeroes that fragment (column -1 is padding; row p is the previous image row's last odd column) -- and the tap dw = 0 reads
// the even columns (same row geometry, column 2*ow').  Six row-group loads and six barriers per 32-channel chunk instead of
// nine tap loads and nine barriers (tools/experiments/next_round_math.py checks the identity on the CPU).
// Step s = 0..5 of a chunk: dh = s / 2 - 1, odd columns for even s (two taps), even columns for odd s (one tap).  x row
// groups: 3 stages, the group of step s + 2 is issued whole at step s; W: 3 stages of two tap tiles, two steps ahead.
// ---------------------------------------------------------------------------------------------------
template <typename T, int TM, int EM, int TPX>
__global__ __launch_bounds__(256, 2) void k_gconv_s2f(GConvP p) {
    AY_KERNARG_TOUCH(kt_, GConvP);           // every line of the parameter struct requested at once (gfx950_dma.h)
    using G = GT<T, TM, TPX>;
    static_assert(sizeof(T) == 2, "fp16");
    constexpr int XR = G::XR;
    constexpr int XS = (G::TP + 16) * G::ROWB;
    constexpr int TMB = TM * G::ROWB;
    constexpr int WSB = 2 * TMB < 4096 ? 4096 : 2 * TMB;      // W stage: two tap tiles
    constexpr int W2 = WSB / 4096, W1 = TMB / 4096 > 0 ? TMB / 4096 : 1;   // pieces per wave for two / one tap tiles
    constexpr int NSTK = G::NACC * 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    unsigned char* sX = smem_raw;                             // [3][TP + 16][32]
    unsigned char* sW = smem_raw + 3 * XS;                    // [3][2][TM][32]
    float* sStat = reinterpret_cast<float*>(sW + 3 * WSB);    // [2][TM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % G::WM, wp = wave / G::WM;

    const unsigned Lb = blockIdx.x;
    const unsigned xcd = Lb & 7u, idx = Lb >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const int n0 = (int)nt * TM;
    const unsigned ntiles_all = (unsigned)((p.Mtotal + G::TP - 1) / G::TP);
    const unsigned tpx = (ntiles_all + 7) / 8;
    const unsigned band_lo = xcd * tpx;
    const unsigned ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = (unsigned)p.nslots / 8u;
    unsigned cur_tile = band_lo + lslot;
    kt_.done();
    if (cur_tile >= ntiles) return;

    if constexpr (EM == 2 || EM == 4) {
        for (int i = tid; i < TM; i += 256) {
            const bool in = n0 + i < p.Nout;
            const float* any = p.scale ? p.scale : p.shift;       // both loads together (one memory round trip in the prologue)
            float sc_ = 1.0f, sh_ = 0.0f;
            if (any) {
                const float a_ = (p.scale ? p.scale : any)[in ? n0 + i : 0], b_ = (p.shift ? p.shift : any)[in ? n0 + i : 0];
                sc_ = (in && p.scale) ? a_ : 1.0f;
                sh_ = (in && p.shift) ? b_ : 0.0f;
            }
            sStat[i] = sc_;
            sStat[TM + i] = sh_;
        }
    }
    const bool want_stats = (EM == 0) && (p.stats != nullptr);
    if constexpr (EM == 0) {
        for (int i = tid; i < 2 * TM; i += 256) reinterpret_cast<double*>(sStat)[i] = 0.0;
    }

    const v4i32 rsX = make_srd(p.x, p.x_bytes), rsW = make_srd(p.w, p.w_bytes);
    const unsigned lds_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sX);
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sW);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    const int slotc = lane & (G::CPR - 1);
    const int rowin = lane / G::CPR;
    const int kc = slotc ^ (lane >> 4);
    const int kcb = kc * G::CE * G::ES;
    const int nC = (p.C + BK - 1) / BK;
    const int cmax = p.C - kc * G::CE;
    const int dhstep = p.XW * p.ldx * G::ES;                  // bytes per input image row
    const int colstep = p.ldx * G::ES;                        // bytes per input pixel

    // W loader: a stage holds two tap tiles; piece r of this wave covers stage rows (r * 4 + wave) * 16 + rowin
    unsigned woff[W2];
    int wtile[W2];                                            // 0 / 1: which of the two tap tiles the piece belongs to (>= 2: none)
#pragma unroll
    for (int r = 0; r < W2; ++r) {
        const int row = (r * 4 + wave) * 16 + rowin;
        const int t = row / TM, rt = row - t * TM;
        wtile[r] = __builtin_amdgcn_readfirstlane(((r * 4 + wave) * 16) / TM);
        woff[r] = (t < 2 && n0 + rt < p.Nout) ? (unsigned)(n0 + rt) * (unsigned)p.ldw * G::ES + (unsigned)kcb : G_OOB;
    }
    int wtap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wtap[t] = __builtin_amdgcn_readfirstlane((int)p.f2wt[t] * p.C * G::ES);

    // x rows of the loader's tile: stage row j <-> output-aligned pixel tile * TP + j - 1, input pixel (n, 2*oh, 2*ow)
    int xoff[XR + 1], xh0[XR + 1];
#define F2_SETUP(tile_, valid_)                                                                            \
    {                                                                                                      \
        _Pragma("unroll") for (int r = 0; r <= XR; ++r) {                                                  \
            const int j = r < XR ? (r * 4 + wave) * 16 + rowin : G::TP + rowin;                            \
            const unsigned mu = (tile_) * G::TP + (unsigned)j - 1u;                                        \
            const bool ok = (valid_) & (mu < (unsigned)p.Mtotal) & (r < XR || rowin < 1);                  \
            const unsigned t_ = fdiv(mu, p.dOW);                                                           \
            const int ow_ = (int)(mu - t_ * (unsigned)p.OW);                                               \
            const unsigned n_ = fdiv(t_, p.dOH);                                                           \
            const int oh_ = (int)(t_ - n_ * (unsigned)p.OH);                                               \
            xoff[r] = (int)(((n_ * (unsigned)p.XH + (unsigned)(2 * oh_)) * (unsigned)p.XW + (unsigned)(2 * ow_)) * (unsigned)p.ldx * G::ES) + kcb; \
            xh0[r] = ok ? 2 * oh_ : -100000;                                                               \
        }                                                                                                  \
    }
    // the row group of step s_ (dh = s_ / 2 - 1, odd columns for even s_) of chunk c_ -> x stage s_ % 3
#define F2_X(s_, c_)                                                                                       \
    {                                                                                                      \
        constexpr int dh_ = (s_) / 2 - 1, odd_ = 1 - ((s_) & 1);                                           \
        const int cb_ = (c_) * (BK * G::ES) + odd_ * colstep;                                              \
        const bool cok_ = (c_) * BK < cmax;                                                                \
        _Pragma("unroll") for (int r = 0; r < XR; ++r) {                                                   \
            const bool ok = ((unsigned)(xh0[r] + dh_) < (unsigned)p.XH) & cok_;                            \
            const unsigned off = ok ? (unsigned)(xoff[r] + dh_ * dhstep + cb_) : G_OOB;                    \
            glds16(rsX, lds_x + ((s_) % 3) * XS + (r * 4 + wave) * 1024, off);                             \
        }                                                                                                  \
        if (wave == 0) {                                                                                   \
            const bool ok = ((unsigned)(xh0[XR] + dh_) < (unsigned)p.XH) & cok_;                           \
            const unsigned off = ok ? (unsigned)(xoff[XR] + dh_ * dhstep + cb_) : G_OOB;                   \
            glds16(rsX, lds_x + ((s_) % 3) * XS + XR * 4 * 1024, off);                                     \
        }                                                                                                  \
    }
    // the W tiles of step s_ of chunk c_ -> W stage s_ % 3: odd-column steps carry taps dw = +1 (tile 0) and dw = -1 (tile 1),
    // even-column steps the tap dw = 0 (tile 0)
#define F2_W(s_, c_)                                                                                       \
    {                                                                                                      \
        constexpr int g_ = (s_) / 2, odd_ = 1 - ((s_) & 1);                                                \
        const unsigned cbad_ = (c_) * BK < cmax ? 0u : G_OOB;                                              \
        _Pragma("unroll") for (int r = 0; r < (odd_ ? W2 : W1); ++r) {                                     \
            const int tap_ = odd_ ? (wtile[r] == 0 ? g_ * 3 + 2 : g_ * 3 + 0) : g_ * 3 + 1;                \
            const unsigned none_ = (wtile[r] < (odd_ ? 2 : 1)) ? 0u : G_OOB;                              \
            const unsigned col_ = (unsigned)(wtap[tap_] + (c_) * (BK * G::ES));                            \
            glds16(rsW, lds_w + ((s_) % 3) * WSB + (r * 4 + wave) * 1024, (woff[r] + col_) | ((woff[r] | cbad_ | none_) & G_OOB)); \
        }                                                                                                  \
    }

    const int arow = (wm * 32 * G::MI + (lane & 31)) * G::ROWB;
    const int swzA = ((lane & 31) / G::RPB) & (G::CPR - 1);
    int brow0 = wp * G::NI * 32 + (lane & 31);               // stage row p of this lane's pixel in block ni = 0 (own row: p + 1)
    const int hi = lane >> 5;

    float16v acc[G::NACC];
#pragma unroll
    for (int i = 0; i < G::NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int cbase = n0 + wm * 32 * G::MI + 4 * (lane >> 5);

    unsigned x_tile = cur_tile;
    bool x_valid = true;
    F2_SETUP(x_tile, true)
    __syncthreads();
    F2_X(0, 0) F2_W(0, 0)
    F2_X(1, 0) F2_W(1, 0)

    int c = 0;
    bool after_epi = false;
    unsigned mleft = 0;
    while (true) {
        if (c == 0) {
            mleft = 0;
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) {
                const unsigned mu = cur_tile * G::TP + (unsigned)(wp * G::NI * 32 + ni * 32 + (lane & 31));
                const unsigned t_ = fdiv(mu, p.dOW);
                mleft |= ((int)(mu - t_ * (unsigned)p.OW) == 0 ? 1u : 0u) << ni;
            }
        }
        const int cn = c + 1 == nC ? 0 : c + 1;
        // pieces issued per step and wave: XR (+1 halo, wave 0: the waits do not count on it) + W2 / W1
#define F2_STEP(s_)                                                                                        \
        {                                                                                                  \
            constexpr int odd_ = 1 - ((s_) & 1);                                                           \
            _Pragma("unroll") for (int r = 0; r <= XR; ++r) asm volatile("" : "+v"(xoff[r]), "+v"(xh0[r])); \
            asm volatile("" : "+v"(brow0));                                                                \
            /* after W(s): everything step s - 1 issued = x group of s + 1 and the W tiles of s + 1 */     \
            if ((s_) == 0) { if (after_epi) wait_vm<XR + W1 + NSTK>(); else wait_vm<XR + W1>(); }          \
            else if (odd_) wait_vm<XR + W1>();                                                             \
            else wait_vm<XR + W2>();                                                                       \
            __builtin_amdgcn_s_barrier();                                                                  \
            const unsigned char* stW = sW + ((s_) % 3) * WSB + arow;                                       \
            const unsigned char* stX = sX + ((s_) % 3) * XS;                                               \
            const int j1 = brow0 + 1, j0 = brow0;                                                          \
            half8 fa[2][2][G::MI], fb[2][2][G::NI];                /* [kk][tap tile][..] */                  \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                             \
                const int sa = ((kk * 2 + hi) ^ swzA) * 16;                                                \
                const int sb1 = ((kk * 2 + hi) ^ ((j1 >> 2) & (G::CPR - 1))) * 16;                         \
                const int sb0 = ((kk * 2 + hi) ^ ((j0 >> 2) & (G::CPR - 1))) * 16;                         \
                _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi) {                                     \
                    fa[kk][0][mi] = *reinterpret_cast<const half8*>(stW + mi * 32 * G::ROWB + sa);         \
                    if (odd_) fa[kk][1][mi] = *reinterpret_cast<const half8*>(stW + TMB + mi * 32 * G::ROWB + sa); \
                }                                                                                          \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) {                                     \
                    fb[kk][0][ni] = *reinterpret_cast<const half8*>(stX + (j1 + ni * 32) * G::ROWB + sb1); \
                    if (odd_) fb[kk][1][ni] = *reinterpret_cast<const half8*>(stX + (j0 + ni * 32) * G::ROWB + sb0); \
                }                                                                                          \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if ((s_) == 4) {                                                                               \
                if (c + 1 == nC) { x_tile += lstride; x_valid = x_valid && x_tile < ntiles; F2_SETUP(x_tile, x_valid) } \
            }                                                                                              \
            if ((s_) + 2 < 6) { F2_X((s_) + 2, c) F2_W((s_) + 2, c) }                                      \
            else { F2_X(((s_) + 2) % 6, cn) F2_W(((s_) + 2) % 6, cn) }                                     \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (odd_ && mleft) {                                                                           \
                _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni)                                       \
                    if ((mleft >> ni) & 1u) {                                                              \
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) { fb[0][1][ni][e] = (_Float16)0.0f; fb[1][1][ni][e] = (_Float16)0.0f; } \
                    }                                                                                      \
            }                                                                                              \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                               \
                _Pragma("unroll") for (int t = 0; t < (odd_ ? 2 : 1); ++t)                                 \
                    _Pragma("unroll") for (int mi = 0; mi < G::MI; ++mi)                                   \
                        _Pragma("unroll") for (int ni = 0; ni < G::NI; ++ni) mma_step(fa[kk][t][mi], fb[kk][t][ni], acc[mi * G::NI + ni]); \
        }
        F2_STEP(0) F2_STEP(1) F2_STEP(2) F2_STEP(3) F2_STEP(4) F2_STEP(5)
        after_epi = false;
        if (c == nC - 1) {
            AY_MFMA_PAD("s_nop 11");
            float ssum[16 * G::MI], ssq[16 * G::MI];
#pragma unroll
            for (int r = 0; r < 16 * G::MI; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
            g_epilogue<T, TM, EM, TPX>(p, cur_tile, p.oah, p.oaw, wp, lane, cbase, want_stats, rsY, acc, ssum, ssq, sStat, cbase - n0);
            if (want_stats) {
                int lq = lane;
                asm volatile("" : "+v"(lq));
                double* sl = reinterpret_cast<double*>(sStat) + wm * 32 * G::MI + 4 * (lq >> 5);
                const float a = half_reduce_scatter<16 * G::MI>(ssum, lq), b = half_reduce_scatter<16 * G::MI>(ssq, lq);
                if (rs_reports<16 * G::MI>(lq)) {
                    const int r = rs_index<16 * G::MI>(lq);
                    const int cl = (r >> 4) * 32 + 8 * ((r & 15) >> 2) + (r & 3);
                    atomicAdd(&sl[cl], (double)a);
                    atomicAdd(&sl[TM + cl], (double)b);
                }
            }
            after_epi = true;
            cur_tile += lstride;
            if (cur_tile >= ntiles) break;
        }
        c = cn;
    }
#undef F2_STEP
#undef F2_W
#undef F2_X
#undef F2_SETUP
    wait_vm<0>();
    if (want_stats) {
        __syncthreads();
        g_stats_to_global<TM, false>(p, reinterpret_cast<const double*>(sStat), tid, n0, slot);
    }
}

// ---------------------------------------------------------------------------------------------------
// k_pw (round 6): the 1x1 / stride-1 conv -- forward AND dgrad -- of the layers with K = 128 / 256 reduction
// channels as a STREAMING kernel.
//
// Why: on the <= 40 x 40 maps k_gconv runs one 256-pixel tile per workgroup at 1-1.5 workgroups per CU.  A life of 18 k cycles
// (128 -> 128 on 40 x 40, tools/gconv_probe.py, profiles/r06_probe_small_maps_full.txt) is 4.5 k of prologue, 4 k-steps of ~1.35 k (one
// barrier + counted wait + fragment fetch + DMA issue per 32-deep step, 512 cycles of them matrix pipe) and 5.3 k of epilogue, all
// at the issue rate of one or two wavefronts per SIMD.  Eight wavefronts on the same step structure bought nothing (v1 of this
// round, profiles/r06_probe_nw8_v1.txt: a step costs ~800 cycles however few MFMAs it holds -- it is the barrier / wait / fetch
// chain, not the work).  So the step structure goes:
//   * the WEIGHTS of a wavefront's 32 output channels live in REGISTERS for the whole kernel (K / 16 A fragments of 16 bytes per
//     lane: 32 registers for K = 128, 64 for K = 256), loaded once per workgroup straight from global memory -- no W tile in LDS,
//     no W DMA per step;
//   * a pixel tile's x rows arrive WHOLE (all K channels: 256 / 512-byte rows) by LDS-DMA into a ring of three stages, two tiles
//     ahead; ONE barrier per pixel tile, then K / 16 back-to-back MFMAs per wavefront against its resident A fragments;
//   * eight wavefronts: 32 channels x 32 pixels each = ONE accumulator block (16 registers), so the epilogue handles 16 values
//     per lane and the BatchNorm sums of a tile go straight into the reduce-scatter (no sum registers across tiles);
//   * persistent: 1-2 workgroups per CU walk their XCD band's pixel tiles, so the DMA of tile t + 2 and the stores of tile t - 1
//     run under tile t's MFMAs and epilogue, and the prologue (kernel arguments, A fragments, BatchNorm constants) is paid once.
// LDS image of a stage: row-major [pixel][K] as the DMA writes it (lane-linear), the 16-byte chunk position XOR-swizzled by the
// pixel row (row & 15, on the SOURCE address): the 16 lanes of a ds_read_b128 group read 16 different rows at the same logical
// chunk = 16 different bank groups.  Epilogues are k_gconv's (g_epilogue on a one-block wave tile).
// ---------------------------------------------------------------------------------------------------
template <int TM, int KC>
struct PW {
    static constexpr int NW = 8;
    static constexpr int WM = TM / 32, WP = NW / WM, TP = 32 * WP;     // 256 channels: 32-pixel tiles; 128: 64-pixel tiles; 64: 128-pixel tiles
    static constexpr int K = KC * 16, RB = K * 2;                      // bytes per x row
    static constexpr int CPR = RB / 16;                                // 16-byte chunks per row: 16 / 32
    static constexpr int XT = TP * RB;                                 // bytes per stage
    static constexpr int PPT = XT / 1024, PPW = PPT / NW;              // DMA pieces per tile / per wavefront
    static constexpr int RPP = 1024 / RB;                              // pixel rows per piece
    // swizzle: chunk c of pixel row r sits at position c ^ swz(r), swz(r) = (r / RPB) & SWM -- rows shorter than a 256-byte bank
    // line (K = 32 / 64) share the line RPB at a time, so the row's index inside the line already separates them
    static constexpr int RPB = RB >= 256 ? 1 : 256 / RB;
    static constexpr int SWM = CPR - 1 < 15 ? CPR - 1 : 15;
    static constexpr int R = 3;                                        // ring stages
    using G = GT<half_t, TM, TP, 8>;
    static_assert(G::MI == 1 && G::NI == 1 && G::WM == WM && G::WP == WP, "one accumulator block per wavefront");
    static_assert(PPT % NW == 0 && RB <= 1024, "whole pieces per wavefront");
    static constexpr size_t lds(int EM, bool BNR) { return (size_t)R * XT + G::tail(EM, BNR); }
};

// XFM: 0 plain x; 1 / 2 = transform on load (ayolo_conv_fwd_xf) over one / two input segments: the x tile is the producer's
// pre-activation z, transformed IN PLACE in its stage -- by the lanes whose DMA wrote it, one tile ahead of the MFMAs -- with
// k_bn_train_act's arithmetic (the operand bits equal the materialised activation's; see k_gconv<.., XF>).  Two segments: a piece
// holds whole pixel rows, i.e. chunks of both buffers -- two lane-masked loads per piece (glds16_exec).
template <int TM, int KC, int EM, bool BNR, int XFM = 0>
__global__ __launch_bounds__(512, (KC <= 8 ? 4 : 2)) void k_pw(GConvP p) {
    AY_KERNARG_TOUCH(kt_, GConvP);
    using P = PW<TM, KC>;
    using G = typename P::G;
    using T = half_t;
    static_assert(XFM == 0 || (!BNR && (EM == 0 || EM == 3)), "transform on load: forward of a 1x1 conv");
    constexpr int NSTK = (EM != 3) ? G::NACC * 2 : G::NST;
    constexpr int PPW = P::PPW;
    constexpr int NP = PPW * (XFM == 2 ? 2 : 1);                                  // DMA instructions per thread and tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    unsigned char* sX = smem_raw;                                                 // [R][TP][K]
    float* sStat = reinterpret_cast<float*>(smem_raw + P::R * P::XT);             // GT::tail: statistics / BNR constants / affine
    float* sXf = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(sStat) + G::tail(EM, BNR));   // XFM: [scale | shift][K]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % P::WM, wp = wave / P::WM;

    // block -> (channel tile, XCD band, slot): as k_gconv
    const unsigned Lb = blockIdx.x;
#ifdef AYOLO_PROBE
    __shared__ unsigned long long s_probe[AY_PROBE_N];
    const bool probe_on = blockIdx.x < 512;
    int probe_k = 2;
    if (threadIdx.x < AY_PROBE_N) s_probe[threadIdx.x] = 0;
    __syncthreads();
    AY_PROBE(0);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 3] = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned xcd = Lb & 7u, idx = Lb >> 3;
    const unsigned nt = idx % (unsigned)p.ntn;
    const unsigned slot = (idx / (unsigned)p.ntn) * 8u + xcd;
    const int n0 = (int)nt * TM;
    const unsigned ntiles_all = (unsigned)((p.Mtotal + P::TP - 1) / P::TP);
    const unsigned tpx = (ntiles_all + 7) / 8;
    const unsigned band_lo = xcd * tpx;
    const unsigned ntiles = band_lo + tpx < ntiles_all ? band_lo + tpx : ntiles_all;
    const unsigned lslot = idx / (unsigned)p.ntn;
    const unsigned lstride = (unsigned)p.nslots / 8u;
    unsigned cur_tile = band_lo + lslot;
    kt_.done();
    if (cur_tile >= ntiles) return;

    const v4i32 rsX = make_srd(p.x, p.x_bytes);
    const unsigned lds_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)sX);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // ---- loader lane geometry: piece q = j * 8 + wave covers stage bytes [q * 1024, +1024) = RPP whole pixel rows; this lane
    // writes chunk position cp of row rowj[j] and must FETCH chunk cp ^ (row & 15) of that pixel
    const unsigned ldxb = (unsigned)p.ldx * 2u;
    const int rowin = lane / P::CPR, cp = lane % P::CPR;
    unsigned coff[PPW];
    int rowj[PPW];
    // two segments: channels [xs_split, K) come from x2 (channel stride ldx2); a lane's chunk belongs to one of them for good
    const v4i32 rsX2 = XFM == 2 ? make_srd(p.x2, p.x2_bytes) : rsX;
    const unsigned ldxb2 = XFM == 2 ? (unsigned)p.ldx2 * 2u : 0u;
    unsigned coff2[XFM == 2 ? PPW : 1];
    unsigned long long m2[XFM == 2 ? PPW : 1];                                    // lanes of piece j that load from segment 1
    int ch0[XFM ? PPW : 1];                                                       // first channel of this lane's chunk of piece j
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        rowj[j] = (j * P::NW + wave) * P::RPP + rowin;
        const int c = cp ^ ((rowj[j] / P::RPB) & P::SWM);
        coff[j] = (unsigned)rowj[j] * ldxb + (unsigned)(c * 16);
        if constexpr (XFM) ch0[j] = c * 8;
        if constexpr (XFM == 2) {
            const bool s1 = c * 8 >= p.xs_split;
            coff2[j] = (unsigned)rowj[j] * ldxb2 + (unsigned)(c * 8 - p.xs_split) * 2u;
            m2[j] = __builtin_amdgcn_ballot_w64(s1);
        }
    }
    // all pieces of tile `tile` (beyond the band / the tensor: out-of-range offset, zero fill) -> stage `st`
    auto issue_tile = [&](unsigned tile, int st) __attribute__((always_inline)) {
        const bool tv = tile < ntiles;
        const unsigned base = tile * (unsigned)P::TP;                             // first pixel of the tile (< 2^31, host check)
        const int rem = tv ? (int)((unsigned)p.Mtotal - base) : 0;               // pixels left from there
        const unsigned bb = base * ldxb;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const unsigned off = rowj[j] < rem ? bb + coff[j] : G_OOB;
            const unsigned la = lds_x + (unsigned)(st * P::XT + (j * P::NW + wave) * 1024);
            if constexpr (XFM == 2) {
                const unsigned off2 = rowj[j] < rem ? base * ldxb2 + coff2[j] : G_OOB;
                glds16_exec(rsX, la, off, ~m2[j]);
                glds16_exec(rsX2, la, off2, m2[j]);
            } else glds16(rsX, la, off);
        }
    };
    issue_tile(cur_tile, 0);
    issue_tile(cur_tile + lstride, 1);

    // ---- the wavefront's weights: A fragments of its 32 output channels for the whole reduction, in registers
    half8 afrag[KC];
    {
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
        const int n = n0 + wm * 32 + (lane & 31);
        const unsigned wo = n < p.Nout ? (unsigned)n * (unsigned)p.ldw * 2u + (unsigned)(lane >> 5) * 16u : G_OOB;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk)
            afrag[kk] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsW, wo + (unsigned)(kk * 32), 0, 0));
    }

    if constexpr (EM == 2 || EM == 4) {
        for (int i = tid; i < TM; i += (int)blockDim.x) {
            const bool in = n0 + i < p.Nout;
            const float* any = p.scale ? p.scale : p.shift;
            float sc_ = 1.0f, sh_ = 0.0f;
            if (any) {
                const float a_ = (p.scale ? p.scale : any)[in ? n0 + i : 0], b_ = (p.shift ? p.shift : any)[in ? n0 + i : 0];
                sc_ = (in && p.scale) ? a_ : 1.0f;
                sh_ = (in && p.shift) ? b_ : 0.0f;
            }
            sStat[i] = sc_;
            sStat[TM + i] = sh_;
        }
    }
    if constexpr (BNR) g_bnr_setup<TM>(p, sStat + 4 * TM, n0, tid);
    const bool want_stats = (EM == 0) && !BNR && (p.stats != nullptr);
    const bool any_stats = BNR || want_stats;
    if constexpr (EM == 0 || BNR) {
        for (int i = tid; i < 2 * TM; i += (int)blockDim.x) reinterpret_cast<double*>(sStat)[i] = 0.0;
    }
    const BnrCtx<1> bctx = BNR ? g_bnr_ctx<1>(p, sStat + 4 * TM, n0 + wm * 32) : BnrCtx<1>{};

    // ---- transform on load: per-input-channel scale | shift -> LDS (from the producer's batch statistics when the finalize rides
    // in this launch: k_bn_finalize's expression sequence, workgroup 0 also writes the saved / running statistics), then the
    // in-place transform of a landed tile by the lanes that own its bytes
    if constexpr (XFM != 0) {
        if (p.nfin > 0) {
            for (int i = tid; i < 2 * P::K; i += (int)blockDim.x) sXf[i] = 0.0f;
            __syncthreads();
            float* gsc = const_cast<float*>(p.xf_scale);
            float* gsh = const_cast<float*>(p.xf_shift);
            for (int f = 0; f < p.nfin; ++f) {
                const ayolo_xf_fin& q = p.fin[f];
                for (int c = tid; c < q.C; c += (int)blockDim.x) {
                    double s1, s2;
                    rep_sum2(q.stats + c, (size_t)2 * q.sld, (size_t)q.sld, q.reps, s1, s2);
                    const double mean = s1 / q.count;
                    double var = s2 / q.count - mean * mean;
                    if (var < 0) var = 0;
                    const float invstd = (float)(1.0 / sqrt(var + (double)q.eps));
                    const float g = q.gamma ? q.gamma[c] : 1.0f, b = q.beta ? q.beta[c] : 0.0f;
                    const float sc = g * invstd;
                    const float sh = b - (float)mean * sc;
                    sXf[q.c0 + c] = sc;
                    sXf[P::K + q.c0 + c] = sh;
                    if (Lb == 0) {
                        gsc[q.c0 + c] = sc;
                        gsh[q.c0 + c] = sh;
                        if (q.save_mean) q.save_mean[c] = (float)mean;
                        if (q.save_invstd) q.save_invstd[c] = invstd;
                        if (q.running_mean) q.running_mean[c] = (1.0f - q.momentum) * q.running_mean[c] + q.momentum * (float)mean;
                        if (q.running_var) {
                            const double unb = q.count > 1.0 ? var * q.count / (q.count - 1.0) : var;
                            q.running_var[c] = (1.0f - q.momentum) * q.running_var[c] + q.momentum * (float)unb;
                        }
                    }
                }
            }
        } else {
            for (int i = tid; i < P::K; i += (int)blockDim.x) {
                sXf[i] = p.xf_scale[i];
                sXf[P::K + i] = p.xf_shift[i];
            }
        }
    }
    const bool xst = XFM != 0 && p.xa != nullptr && nt == 0;           // this workgroup also stores the activation it forms
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(XFM ? p.xa : nullptr, 0, XFM ? p.xa_bytes : 0, 0x00020000);
    // exactly PPW stores per transformed tile when this workgroup stores back (plain segments / rows beyond the tensor: dropped by
    // the out-of-range offset), so that the counted waits stay exact
    auto xf_transform = [&](int st, unsigned tile) __attribute__((always_inline)) {
        if constexpr (XFM != 0) {
            const bool tv = tile < ntiles;
            const unsigned base = tile * (unsigned)P::TP;
            const int rem = tv ? (int)((unsigned)p.Mtotal - base) : 0;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int sg = (XFM == 2 && ch0[j] >= p.xs_split) ? 1 : 0;
                const bool virt = ((p.xf_virt >> sg) & 1) != 0, act = ((p.xf_act >> sg) & 1) != 0;
                half8* q = reinterpret_cast<half8*>(sX + st * P::XT + (j * P::NW + wave) * 1024 + lane * 16);
                half8 h = *q;
                const float4v a0_ = *reinterpret_cast<const float4v*>(sXf + ch0[j]), a1_ = *reinterpret_cast<const float4v*>(sXf + ch0[j] + 4);
                const float4v b0_ = *reinterpret_cast<const float4v*>(sXf + P::K + ch0[j]), b1_ = *reinterpret_cast<const float4v*>(sXf + P::K + ch0[j] + 4);
                if (virt) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float u = __builtin_fmaf((float)h[e], e < 4 ? a0_[e & 3] : a1_[e & 3], e < 4 ? b0_[e & 3] : b1_[e & 3]);
                        if (act) u = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
                        h[e] = (half_t)u;
                    }
                    *q = h;
                }
                if (xst) {
                    const bool ok = virt & (rowj[j] < rem);
                    const unsigned off = ok ? (base + (unsigned)rowj[j]) * ((unsigned)p.ldxa * 2u) + (unsigned)ch0[j] * 2u : G_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, h), rsA, off, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // written before this wave reaches the next barrier
        }
    };

    // ---- B fragment geometry: this lane's pixel row of the stage and its swizzled chunk base; fragment kk = chunk 2 kk + hi
    const int prow = wp * 32 + (lane & 31);
    const unsigned a0 = (unsigned)prow * P::RB + (unsigned)(((lane >> 5) ^ ((prow / P::RPB) & P::SWM)) << 4);
    const int cbase = n0 + wm * 32 + 4 * (lane >> 5);

    float16v acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
    __syncthreads();                          // constants / zeroed statistics visible
    if constexpr (XFM != 0) {
        wait_vm<NP>();                        // tile 0's pieces (the older half of the two issues)
        xf_transform(0, cur_tile);
    }
    AY_PROBE(1);

    unsigned ld_tile = cur_tile + 2 * lstride;
    bool after_epi = false;
    bool done = false;
#define PW_TILE(S_)                                                                                        \
    if (!done) {                                                                                           \
        /* tile t's pieces of THIS wave landed (behind them: tile t + 1's pieces and the last epilogue's stores), then everyone's, \
           and everyone is through with the stage tile t + 2 goes to */                                   \
        AY_PROBE_STEP();                                                                                   \
        if constexpr (XFM == 0) { if (after_epi) wait_vm<NP + NSTK>(); else wait_vm<NP>(); }               \
        AY_PROBE_STEP();                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                      \
        AY_PROBE_STEP();                                                                                   \
        issue_tile(ld_tile, ((S_) + 2) % 3);                                                               \
        ld_tile += lstride;                                                                                \
        {                                                                                                  \
            const unsigned char* st = sX + (S_) * P::XT;                                                   \
            half8 bf[KC];                                                                                  \
            _Pragma("unroll") for (int kk = 0; kk < KC; ++kk) bf[kk] = *reinterpret_cast<const half8*>(st + (a0 ^ (unsigned)(kk << 5))); \
            _Pragma("unroll") for (int kk = 0; kk < KC; ++kk) mma_step(afrag[kk], bf[kk], acc[0]);         \
        }                                                                                                  \
        AY_MFMA_PAD("s_nop 11");                                                                           \
        AY_PROBE_STEP();                                                                                   \
        {                                                                                                  \
            float ssum[16], ssq[16];                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }              \
            g_epilogue<T, TM, EM, P::TP, BNR, 8>(p, cur_tile, p.coah[0], p.coaw[0], wp, lane, cbase, want_stats, rsY, acc, ssum, ssq, sStat, \
                                                 cbase - n0, bctx);                                        \
            if constexpr (EM == 0 || BNR) {                                                                \
                if (any_stats) {                                                                           \
                    int lq = lane;                                                                         \
                    asm volatile("" : "+v"(lq));                                                           \
                    double* sl = reinterpret_cast<double*>(sStat) + wm * 32;                               \
                    const int cl = g_stat_chan<BNR>(rs_index<16>(lq), lq >> 5);                            \
                    const float a = half_reduce_scatter<16>(ssum, lq);                                     \
                    const float b = half_reduce_scatter<16>(ssq, lq);                                      \
                    if (rs_reports<16>(lq)) { atomicAdd(&sl[cl], (double)a); atomicAdd(&sl[TM + cl], (double)b); } \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
        after_epi = true;                                                                                  \
        if constexpr (XFM != 0) {                                                                          \
            /* tile t + 1's pieces (behind them: the last transform's stores, tile t + 2's pieces, this epilogue's stores): \
               transformed now, published by the next barrier */                                          \
            if (xst) wait_vm<NP + NSTK + PPW>(); else wait_vm<NP + NSTK>();                                \
            xf_transform(((S_) + 1) % 3, cur_tile + lstride);                                              \
        }                                                                                                  \
        cur_tile += lstride;                                                                               \
        done = cur_tile >= ntiles;                                                                         \
    }
#ifdef AYOLO_PROBE
#define AY_PROBE_STEP() do { AY_PROBE(probe_k); ++probe_k; } while (0)
#else
#define AY_PROBE_STEP() do { } while (0)
#endif
    while (!done) {
        PW_TILE(0)
        PW_TILE(1)
        PW_TILE(2)
    }
#undef PW_TILE
#undef AY_PROBE_STEP
    wait_vm<0>();                             // trailing zero-fill DMAs must land before this LDS is released
#ifdef AYOLO_PROBE
    AY_PROBE(AY_PROBE_N - 1);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (probe_on && threadIdx.x < AY_PROBE_N) g_probe[blockIdx.x * AY_PROBE_N + threadIdx.x] = s_probe[threadIdx.x];
#endif
    if constexpr (EM == 0 || BNR) {
        if (any_stats) {
            __syncthreads();
            g_stats_to_global<TM, BNR>(p, reinterpret_cast<const double*>(sStat), tid, n0, slot);
        }
    }
}

// compute units of the CURRENT device (cached per device ordinal: one process may drive several GPUs)
static int num_cus() {
    static int cache[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cache[dev] == 0) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        cache[dev] = n > 0 ? n : 256;
    }
    return cache[dev];
}

// workgroups per CU of one k_gconv instantiation (LDS-bound; the launch bounds give the register budget to match)
template <typename T, int TM, int TPX, int EM, bool BNR, int NWX = 4>
static constexpr int gconv_bpc() {
    if constexpr (NWX == 8) return 2;          // register budget: 128 per lane = four wavefronts per SIMD = two 512-thread workgroups
    using G = GT<T, TM, TPX, NWX>;
    constexpr size_t lds_alloc = (G::lds(EM, BNR) + 1279) / 1280 * 1280;          // LDS allocation granule of gfx950
    int bpc = (int)(160 * 1024 / lds_alloc);
    const int bpc_max = sizeof(T) == 2 ? (G::lds(EM, BNR) > 56 * 1024 ? 2 : (TM == 128 ? 3 : 4)) : 1;
    return bpc > bpc_max ? bpc_max : bpc;
}

// persistent grid of one instantiation: pixel-tile slots (a multiple of the 8 XCDs) and the number of tiles the busiest
// slot walks ("waves")
struct GGrid { long long slots; long long waves; };
static GGrid gconv_grid(long long Mtotal, int tp, int ntn, int bpc) {
    const long long ntiles = (Mtotal + tp - 1) / tp;
    long long want_slots = (long long)num_cus() * bpc / ntn;
    if (want_slots < 8) want_slots = 8;
    long long slots = ntiles < want_slots ? ntiles : want_slots;
    slots = (slots + 7) / 8 * 8;
    const long long tpx = (ntiles + 7) / 8, spx = slots / 8;         // per XCD band
    return {slots, (tpx + spx - 1) / spx};
}

template <typename T, int TM, int EM, int TPX, bool BNR = false, bool XF = false, bool LIN = false, int NWX = 4>
static int launch_gconv_tp(GConvP p, hipStream_t s) {
    using G = GT<T, TM, TPX, NWX>;
    const size_t lds = G::lds(EM, BNR) + (XF ? 2 * (size_t)((p.C + BK - 1) / BK * BK) * sizeof(float) : 0);
    p.ntn = (p.Nout + TM - 1) / TM;
    constexpr int bpc_env = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if constexpr (sizeof(T) == 2 && EM != 3 && EM != 1) {
        if (p.s2f) {                         // forward 3x3 / stride 2 with the odd-column taps sharing one row run
            constexpr size_t lds2 = 3 * (size_t)(G::TP + 16) * G::ROWB + 3 * (size_t)(2 * TM * G::ROWB < 4096 ? 4096 : 2 * TM * G::ROWB) +
                                    4 * TM * sizeof(float);
            const long long slots2 = gconv_grid(p.Mtotal, G::TP, p.ntn, bpc_env > 0 ? bpc_env : 2).slots;
            p.nslots = (int)slots2;
            static bool attr2_set[16] = {false};
            if (dev < 0 || dev >= 16 || !attr2_set[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gconv_s2f<T, TM, EM, TPX>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                if (dev >= 0 && dev < 16) attr2_set[dev] = true;
            }
            hipLaunchKernelGGL((k_gconv_s2f<T, TM, EM, TPX>), dim3((unsigned)(slots2 * p.ntn)), dim3(256), lds2, s, p);
            AY_CHECK_LAUNCH("k_gconv_s2f");
            return AYOLO_OK;
        }
    }
    if constexpr (sizeof(T) == 2 && EM != 3) {
        if (p.row3) {                        // 3x3 / stride 1: x rows shared by the three taps of a kernel row (k_gconv3)
            using G3 = GT3<T, TM, TPX>;
            const long long slots3 = gconv_grid(p.Mtotal, G::TP, p.ntn, bpc_env > 0 ? bpc_env : 2).slots;
            p.nslots = (int)slots3;
            static bool attr3_set[16] = {false};
            if (dev < 0 || dev >= 16 || !attr3_set[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gconv3<T, TM, EM, TPX, BNR>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)G3::lds(EM, BNR));
                if (dev >= 0 && dev < 16) attr3_set[dev] = true;
            }
            hipLaunchKernelGGL((k_gconv3<T, TM, EM, TPX, BNR>), dim3((unsigned)(slots3 * p.ntn)), dim3(256), G3::lds(EM, BNR), s, p);
            AY_CHECK_LAUNCH("k_gconv3");
            return AYOLO_OK;
        }
    }
    int bpc = bpc_env > 0 ? bpc_env : gconv_bpc<T, TM, TPX, EM, BNR, NWX>();
    if constexpr (XF) {                      // the constants table counts against the LDS budget
        const int fit = (int)(160 * 1024 / ((lds + 1279) / 1280 * 1280));
        bpc = fit < bpc ? (fit < 1 ? 1 : fit) : bpc;
    }
    long long slots = gconv_grid(p.Mtotal, G::TP, p.ntn, bpc).slots;
    if constexpr (NWX == 8) {
        // balanced persistent grid: every workgroup walks the same number of tiles (800 tiles on 512 slots = 288 workgroups with two
        // tiles + 224 with one; 400 workgroups with two tiles each end together)
        const long long ntiles = (p.Mtotal + G::TP - 1) / G::TP, tpx = (ntiles + 7) / 8, spx = slots / 8;
        const long long waves = (tpx + spx - 1) / spx;
        slots = ((tpx + waves - 1) / waves) * 8;
    }
    p.nslots = (int)slots;
    dim3 grid((unsigned)(slots * p.ntn));
    static size_t attr_set[16] = {0};        // per device (function attributes belong to the device's context): largest size set
    if (dev < 0 || dev >= 16 || attr_set[dev] < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gconv<T, TM, EM, TPX, BNR, XF, LIN, NWX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(XF ? 160 * 1024 : lds));
        if (dev >= 0 && dev < 16) attr_set[dev] = XF ? 160 * 1024 : lds;
    }
    hipLaunchKernelGGL((k_gconv<T, TM, EM, TPX, BNR, XF, LIN, NWX>), grid, dim3(G::NT), lds, s, p);
    AY_CHECK_LAUNCH("k_gconv");
    return AYOLO_OK;
}

// k_pw: grid = 1-2 persistent workgroups per CU (and channel tile), every workgroup the same number of pixel tiles
template <int TM, int KC, int EM, bool BNR, int XFM = 0>
static int launch_pw(GConvP p, hipStream_t s) {
    using P = PW<TM, KC>;
    const size_t lds = P::lds(EM, BNR) + (XFM ? 2 * (size_t)P::K * sizeof(float) : 0);
    p.ntn = (p.Nout + TM - 1) / TM;
    const int fit = (int)(160 * 1024 / ((lds + 1279) / 1280 * 1280));
    const int bpc = KC <= 8 ? (fit < 2 ? fit : 2) : 1;                 // register budget: 128 (two workgroups per CU) / 256 per lane
    long long slots = gconv_grid(p.Mtotal, P::TP, p.ntn, bpc).slots;
    {
        const long long ntiles = (p.Mtotal + P::TP - 1) / P::TP, tpx = (ntiles + 7) / 8, spx = slots / 8;
        const long long waves = (tpx + spx - 1) / spx;
        slots = ((tpx + waves - 1) / waves) * 8;
    }
    p.nslots = (int)slots;
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw<TM, KC, EM, BNR, XFM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_pw<TM, KC, EM, BNR, XFM>), dim3((unsigned)(slots * p.ntn)), dim3(512), lds, s, p);
    AY_CHECK_LAUNCH("k_pw");
    return AYOLO_OK;
}

template <typename T, int TM, int EM, bool BNR = false, bool XF = false>
static int launch_gconv_em(const GConvP& p, hipStream_t s) {
    // the streaming 1x1 kernel (k_pw): whole reductions of 64 / 128 / 256 channels into 64- or 128-channel tiles -- forward, dgrad and
    // the transform-on-load readers.  Same-box A/B of the routes (profiles/r06_ab_pw_v1..v4.txt, r06_ab_pw_nw8.txt): TM 128 with
    // K = 128 / 256 -0.07 ms, + K = 64 and the 64-channel tiles -0.1 ... -0.2 ms, + the readers -0.23 ms in all; the 32-channel
    // tiles (K = 32 / 64 on the 160 x 160 / 320 x 320 maps) were level with k_gconv and are not instantiated.  K = 512 as two half-K
    // sub-tiles per pixel tile (128 weight registers, one workgroup per CU) was built and measured in round 6: 512 -> 256 at 40 x 40
    // 68.8 -> 53.7 us, but every 512-channel layer at 20 x 20 level or slower (three or four pixel tiles per workgroup do not amortise
    // 256 KB of weight loads: 512 -> 256 35.6 -> 44.5 us) and the step +0.08 ms (profiles/r06_ab_pw_k512.txt); not kept.
    // YOLOHead's epilogue (EM 3: fp32 logits + bias; 255 of 256 channels) on the same kernel: 128 -> 255 at 80 x 80 132 -> 119 us,
    // 256 -> 255 at 40 x 40 65 -> 51 us, step -0.04 ms (profiles/r06_ab_pw_head.txt).
    if constexpr (sizeof(T) == 2 && XF && (EM == 0 || EM == 3)) {
        // transform on load: the same tilings as the plain conv of the same shape (a layer's two routes share one kernel family, so
        // that they stay bit-identical: test_conv_transform_on_load_equals_materialised_route)
        if (p.xf && p.C % BK == 0 && p.xs_split % 8 == 0) {
            const bool two = p.x2 != nullptr;
            if constexpr (TM == 128) {
                // 256-channel tiles (eight wavefronts x 32 channels, 32-pixel tiles) where the output channels are a multiple of 256: every x
                // tile is read -- and transformed -- once instead of once per 128-channel tile (256 -> 256 at 40 x 40: 59-62 -> 48-52 us,
                // step -0.09 ms, profiles/r06_ab_pw_tm256.txt)
                if (EM == 3 ? (p.Nout > 128 && p.Nout <= 256) : p.Nout % 256 == 0) {
                    if (p.C == 128) return two ? launch_pw<256, 8, EM, false, 2>(p, s) : launch_pw<256, 8, EM, false, 1>(p, s);
                    if (p.C == 256) return two ? launch_pw<256, 16, EM, false, 2>(p, s) : launch_pw<256, 16, EM, false, 1>(p, s);
                }
                if (p.C == 64) return two ? launch_pw<TM, 4, EM, false, 2>(p, s) : launch_pw<TM, 4, EM, false, 1>(p, s);
                if (p.C == 128) return two ? launch_pw<TM, 8, EM, false, 2>(p, s) : launch_pw<TM, 8, EM, false, 1>(p, s);
                if (p.C == 256) return two ? launch_pw<TM, 16, EM, false, 2>(p, s) : launch_pw<TM, 16, EM, false, 1>(p, s);
            } else if constexpr (TM == 64) {
                if (p.C == 64) return two ? launch_pw<TM, 4, EM, false, 2>(p, s) : launch_pw<TM, 4, EM, false, 1>(p, s);
                if (p.C == 128) return two ? launch_pw<TM, 8, EM, false, 2>(p, s) : launch_pw<TM, 8, EM, false, 1>(p, s);
            }
        }
    }
    if constexpr (sizeof(T) == 2 && !XF) {
        if (p.lin) {
            if constexpr (TM == 128) {
                if (EM == 3 ? (p.Nout > 128 && p.Nout <= 256) : p.Nout % 256 == 0) {
                    if (p.C == 128) return launch_pw<256, 8, EM, BNR>(p, s);
                    if (p.C == 256) return launch_pw<256, 16, EM, BNR>(p, s);
                }
                if (p.C == 64) return launch_pw<TM, 4, EM, BNR>(p, s);
                if (p.C == 128) return launch_pw<TM, 8, EM, BNR>(p, s);
                if (p.C == 256) return launch_pw<TM, 16, EM, BNR>(p, s);
            } else if constexpr (TM == 64) {
                if (p.C == 64) return launch_pw<TM, 4, EM, BNR>(p, s);
                if (p.C == 128) return launch_pw<TM, 8, EM, BNR>(p, s);
            }
        }
    }
    // Pixel-tile size, 128 or 256 (measured per layer on the YOLOv5s shapes at batch 64, profiles/r02_conv_tile_sweep.txt):
    //  * one wave of 128-pixel tiles fits the chip: keep 128 (most workgroups in flight; the 20^2 maps);
    //  * few waves (<= 3) and 256-pixel tiles need fewer: take 256.  A workgroup's step is latency-bound there, so the
    //    number of tile "waves" is what counts: 102 400 pixels are 800 tiles of 128 on 768 slots (3 workgroups per CU)
    //    = 2 waves, the second one 4 % full, but ONE wave of 400 tiles of 256 on 512 slots (40^2 maps: -15 %);
    //  * many waves: the wider tile wins through fewer barriers and LDS fragment reads per MFMA when the reduction is
    //    deep enough -- TM 64: always; TM 128: K >= 256 (3x3); TM 32: K >= 128 (stem) -- and loses a few % on one- and
    //    two-step 1x1 layers (lower occupancy).
    static const int force = getenv("AYOLO_GCONV_TP") ? atoi(getenv("AYOLO_GCONV_TP")) : 0;
    bool wide;
    if (force) wide = force == 256;
    else {
        const int ntn = (p.Nout + TM - 1) / TM;
        const int K = p.ntaps * p.C;
        const long long w128 = gconv_grid(p.Mtotal, 128, ntn, p.row3 ? 2 : gconv_bpc<T, TM, 128, EM, BNR>()).waves;
        const long long w256 = gconv_grid(p.Mtotal, 256, ntn, p.row3 ? 2 : gconv_bpc<T, TM, 256, EM, BNR>()).waves;
        if (w128 <= 1) wide = false;
        else if (w128 <= 3 && w256 < w128) wide = true;
        else wide = TM == 64 ? true : (TM == 128 ? K >= 256 : K >= 128);
    }
    if (p.s2f && TM == 128) wide = false;          // k_gconv_s2f: the 128 x 256 tile would spill (all four fragment sets live)
    // Round 6: the 1x1 layers of the small maps on EIGHT wavefronts per 128 x 128 tile.  One 256-pixel tile per
    // workgroup at 1-1.5 workgroups per CU leaves one or two wavefronts on a SIMD, and a lone wavefront issues an instruction every
    // 8-10 cycles: prologue and epilogue (4.5 k + 5.3 k cycles of an 18 k-cycle life on 128 -> 128 at 40 x 40, tools/gconv_probe.py) run at
    // that rate.  Here every wavefront owns 32 channels x 64 pixels (32 accumulator registers), two workgroups per CU put four
    // wavefronts on every SIMD, and a workgroup walks two or more tiles so that the next tile's DMA runs under the epilogue.
    if constexpr (sizeof(T) == 2 && TM == 128 && (EM != 3 || XF)) {
        // (the layers k_pw does not take: 512-channel reductions, head levels; maps of <= 40 x 40 at batch 64.  Same-box A/B: -0.06 ms,
        // profiles/r06_ab_nw8_v1.txt)
        constexpr long long nw8_maxm = 102400;
        // (transform on load: only with whole 32-channel chunks, the 1x1 loader's condition, so that a layer's two routes -- reader
        // over z / conv over the materialised activation -- always share one tiling and stay bit-identical)
        if ((p.lin || (XF && p.C % BK == 0)) && !p.row3 && !p.s2f && p.Mtotal <= nw8_maxm) {
            if constexpr (XF) return launch_gconv_tp<T, TM, EM, 128, BNR, true, false, 8>(p, s);
            else if constexpr (EM != 3) { if (p.lin) return launch_gconv_tp<T, TM, EM, 128, BNR, false, true, 8>(p, s); }
        }
    }
    // the 1x1 loader (GConvP::lin): the Conv / dgrad / inference epilogues of the 64- and 128-channel tiles (YOLOHead keeps the
    // generic loader: three launches per step)
    if constexpr (sizeof(T) == 2 && !XF && EM != 3 && TM >= 64) {
        if (p.lin) return wide ? launch_gconv_tp<T, TM, EM, 256, BNR, false, true>(p, s) : launch_gconv_tp<T, TM, EM, 128, BNR, false, true>(p, s);
    }
    if (wide) return launch_gconv_tp<T, TM, EM, 256, BNR, XF>(p, s);
    return launch_gconv_tp<T, TM, EM, 128, BNR, XF>(p, s);
}

template <typename T, int TM>
static int launch_gconv(const GConvP& p, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (p.xf) return p.epi == AYOLO_EPI_HEAD ? launch_gconv_em<T, TM, 3, false, true>(p, s) : launch_gconv_em<T, TM, 0, false, true>(p, s);
    }
    if (p.epi == AYOLO_EPI_HEAD) return launch_gconv_em<T, TM, 3>(p, s);
    if (p.epi == AYOLO_EPI_AFFINE || p.epi == AYOLO_EPI_AFFINE_SILU) return launch_gconv_em<T, TM, 2>(p, s);
    if (p.epi == AYOLO_EPI_AFFINE_RES || p.epi == AYOLO_EPI_AFFINE_SILU_RES) return launch_gconv_em<T, TM, 4>(p, s);
    if constexpr (sizeof(T) == 2) {
        if (p.bnr) return p.accumulate ? launch_gconv_em<T, TM, 1, true>(p, s) : launch_gconv_em<T, TM, 0, true>(p, s);
    }
    if (p.accumulate) return launch_gconv_em<T, TM, 1>(p, s);
    return launch_gconv_em<T, TM, 0>(p, s);
}

template <int TM, int EM, int TPX, bool BNR = false>
static int launch_dgrad_s2(GConvP p, hipStream_t s) {
    using G = GT<half_t, TM, TPX>;
    // nine tap tiles; the one-tap step D still issues one piece per wave (4 KiB): one tile of slack behind it for TM = 32
    const size_t lds = 3 * (size_t)(G::TP + 16) * G::ROWB + (9 + (TM == 32 ? 1 : 0)) * (size_t)TM * G::ROWB + 8 * TM * sizeof(float);
    p.ntn = (p.Nout + TM - 1) / TM;
    const long long slots = gconv_grid(p.Mtotal, G::TP, p.ntn, 2).slots;
    p.nslots = (int)slots;
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dgrad_s2<half_t, TM, EM, TPX, BNR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_dgrad_s2<half_t, TM, EM, TPX, BNR>), dim3((unsigned)(slots * p.ntn)), dim3(256), lds, s, p);
    AY_CHECK_LAUNCH("k_dgrad_s2");
    return AYOLO_OK;
}

static int dispatch_gconv_one(int dtype, const GConvP& p, hipStream_t s) {
    if (p.s2d && dtype == AYOLO_F16) {       // dgrad of a 3x3 / stride-2 conv, <= 64 input channels: all four classes at once
        if (p.bnr) {
            if (p.Nout <= 32) return p.accumulate ? launch_dgrad_s2<32, 1, 256, true>(p, s) : launch_dgrad_s2<32, 0, 256, true>(p, s);
            return p.accumulate ? launch_dgrad_s2<64, 1, 128, true>(p, s) : launch_dgrad_s2<64, 0, 128, true>(p, s);
        }
        if (p.Nout <= 32) return p.accumulate ? launch_dgrad_s2<32, 1, 256>(p, s) : launch_dgrad_s2<32, 0, 256>(p, s);
        return p.accumulate ? launch_dgrad_s2<64, 1, 128>(p, s) : launch_dgrad_s2<64, 0, 128>(p, s);
    }
    // Output-channel tile: the widest that is not mostly padding.  Measured on the YOLOv5x widths (80 / 160 / 320 / 640 / 1280
    // channels, profiles/r02_conv_tm_sweep_yolov5x.txt): 128-wide tiles beat 64- and 32-wide ones by 1.3-2.5x even where
    // 37 % of the tile is padding (80 or 160 output channels) -- a narrow tile re-reads the pixel tile per channel tile and
    // halves the MFMAs per barrier -- so padding waste is NOT a reason to go narrower.  (A 256-channel x 256-pixel tile on eight
    // wavefronts existed in round 3 behind a switch; it measured neutral-to-slower -- profiles/r03_tm256_tile_sweep.txt -- and
    // was removed in round 4 together with the switch.)
    int tm = p.Nout <= 32 ? 32 : (p.Nout <= 64 ? 64 : 128);
    // ... except where the last 128-wide tile would be half empty (192 / 320 output channels) on a small map: there the
    // 64-wide tiling is exact and the extra pixel-tile reads stay in L2 (YOLOv5x, batch 8 at 1280^2: 320 -> 320 3x3 on 80x80
    // 154 vs 169 us; on the 160x160 maps the 128-wide tiles still win)
    if (p.Nout > 128 && p.Nout % 128 == 64 && p.Mtotal <= 65536) tm = 64;
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

// Fills the buffer-descriptor extents / division constants of k_gconv.  The descriptors address < 2 GiB (bit 31 of the
// offset is the out-of-range marker): a larger activation is processed as independent batch halves (NHWC, batch
// outermost; BN statistics accumulate across launches).
static int dispatch_gconv(int dtype, GConvP p, hipStream_t s) {
    const long long es = dtype == AYOLO_F16 ? 2 : 4;
    const long long yes = p.epi == AYOLO_EPI_HEAD ? 4 : es;
    const long long LIM = (1ll << 31) - 4096;
    const long long x_img = (long long)p.XH * p.XW * p.ldx * es;
    const long long y_img = (long long)p.YH * p.YW * p.ldy * yes;
    const long long w_bytes = (long long)p.Nout * p.ldw * es;
    AY_CHECK_ARG(w_bytes < (1ll << 30), "conv: weight matrix of %lld bytes unsupported", w_bytes);
    AY_CHECK_ARG(x_img < LIM && y_img < LIM, "conv: a single image of %lld / %lld bytes unsupported", x_img, y_img);
    // BNR: the z buffers of the BatchNorm segments have y's pixels (possibly a wider channel stride) and split with it
    long long z_img_max = 0;
    for (int k = 0; k < p.bnr; ++k) {
        const long long z_img = (long long)p.YH * p.YW * p.bseg[k].ldz * 2;
        z_img_max = z_img > z_img_max ? z_img : z_img_max;
    }
    AY_CHECK_ARG(z_img_max < LIM, "conv: a single image of z (%lld bytes) unsupported", z_img_max);
    const long long x2_img = p.x2 ? (long long)p.XH * p.XW * p.ldx2 * es : 0;
    AY_CHECK_ARG(x2_img < LIM, "conv: a single image of the second input segment (%lld bytes) unsupported", x2_img);
    if (x_img * p.B >= LIM || y_img * p.B >= LIM || z_img_max * p.B >= LIM || x2_img * p.B >= LIM) {
        AY_CHECK_ARG(p.B > 1, "conv: one image exceeds the 2 GiB descriptor range");
        AY_CHECK_ARG(p.xa == nullptr && p.nfin == 0, "conv_fwd_xf: store-back / in-launch finalize need tensors below 2 GiB");
        GConvP a = p, b = p;
        a.B = p.B / 2; b.B = p.B - a.B;
        a.Mtotal = (long long)a.B * p.OH * p.OW; b.Mtotal = (long long)b.B * p.OH * p.OW;
        b.x = (const char*)p.x + x_img * a.B;
        b.y = (char*)p.y + y_img * a.B;
        if (p.x2) b.x2 = (const char*)p.x2 + (long long)p.XH * p.XW * p.ldx2 * es * a.B;
        for (int k = 0; k < p.bnr; ++k)
            b.bseg[k].z = (const char*)p.bseg[k].z + (long long)p.YH * p.YW * p.bseg[k].ldz * 2 * a.B;
        int rc = dispatch_gconv(dtype, a, s);
        return rc ? rc : dispatch_gconv(dtype, b, s);
    }
    for (int k = 0; k < p.bnr; ++k) p.z_bytes[k] = (unsigned)((long long)p.YH * p.YW * p.bseg[k].ldz * 2 * p.B);
    if (p.epi != AYOLO_EPI_HEAD) {
        AY_CHECK_ARG(p.Nout % 4 == 0 && p.ldy % 4 == 0, "conv: Cout=%d / channel stride %d must be multiples of 4", p.Nout, p.ldy);
        AY_CHECK_ARG(dtype != AYOLO_F16 || (p.Nout % 8 == 0 && p.ldy % 8 == 0),
                     "conv: fp16 outputs leave in 16-byte stores: Cout=%d / channel stride %d must be multiples of 8", p.Nout, p.ldy);
    }
    p.x_bytes = (unsigned)(x_img * p.B);
    p.x2_bytes = (unsigned)(x2_img * p.B);
    p.y_bytes = (unsigned)(y_img * p.B);
    p.w_bytes = (unsigned)w_bytes;
    p.dOW = make_fastdiv((unsigned)p.OW); p.dOH = make_fastdiv((unsigned)p.OH); p.dC = make_fastdiv((unsigned)(p.C > 0 ? p.C : 1));
    p.lin = (dtype == AYOLO_F16 && p.x_linear && p.ncls <= 1 && p.ntaps == 1 && p.dh[0] == 0 && p.dw[0] == 0 && p.wt[0] == 0 &&
             p.C % BK == 0 && p.XH == p.OH && p.XW == p.OW && p.ish == 1 && p.isw == 1 && !p.xf) ? 1 : 0;
    if (p.ncls <= 0) {                       // ordinary launch: one class = all taps
        p.ncls = 1; p.ctap0[0] = 0; p.cnt[0] = p.ntaps; p.coah[0] = p.oah; p.coaw[0] = p.oaw;
    }
    // 3x3, stride 1, same-size maps (the Bottleneck 3x3 convs and their dgrad): k_gconv3
    p.row3 = 0;
    if (dtype == AYOLO_F16 && p.ncls == 1 && p.ntaps == 9 && p.ish == 1 && p.isw == 1 && p.osh == 1 && p.osw == 1 &&
        p.XH == p.OH && p.XW == p.OW && p.YH == p.OH && p.YW == p.OW && p.C >= BK && p.epi != AYOLO_EPI_HEAD &&
        // channels beyond C in the last 32-wide chunk are fetched as zeros (x and W): at most a quarter of the MFMA work
        ((p.C + BK - 1) / BK * BK - p.C) * 4 <= (p.C + BK - 1) / BK * BK) {
        int seen = 0;
        for (int t = 0; t < 9; ++t) {
            const int g = p.dh[t] + 1, j = p.dw[t] + 1;
            if (g < 0 || g > 2 || j < 0 || j > 2) { seen = -1; break; }
            seen |= 1 << (g * 3 + j);
            p.r3wt[g * 3 + j] = p.wt[t];
        }
        p.row3 = seen == 0x1ff ? 1 : 0;
    }
    // forward 3x3 / stride 2 / pad 1 on even maps -> k_gconv_s2f.  Measured on MI355X (profiles/r03_conv_layer_sweep*.txt):
    // 32 -> 64 @ 320^2 177 -> 160 us; the wider layers move by -4 .. +4 us (their step is bound by the W tiles, not by x), so
    // it is used for <= 32 input channels only
    p.s2f = 0;
    if (p.C <= 32 && dtype == AYOLO_F16 && p.ncls == 1 && p.ntaps == 9 && p.ish == 2 && p.isw == 2 && p.osh == 1 && p.osw == 1 &&
        !p.accumulate && p.XH == 2 * p.OH && p.XW == 2 * p.OW && p.YH == p.OH && p.YW == p.OW && p.C >= BK && p.epi != AYOLO_EPI_HEAD &&
        ((p.C + BK - 1) / BK * BK - p.C) * 4 <= (p.C + BK - 1) / BK * BK && p.Nout <= 128 * 1024) {
        int seen = 0;
        for (int t = 0; t < 9; ++t) {
            const int g = p.dh[t] + 1, j = p.dw[t] + 1;
            if (g < 0 || g > 2 || j < 0 || j > 2) { seen = -1; break; }
            seen |= 1 << (g * 3 + j);
            p.f2wt[g * 3 + j] = p.wt[t];
        }
        p.s2f = seen == 0x1ff ? 1 : 0;
    }
    return dispatch_gconv_one(dtype, p, s);
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

static bool is_packed_stem(const ayolo_conv_desc* d);
static int stem_fwd_dispatch(const ayolo_conv_desc* d, const void* x, const void* w, void* y, double* stats, int stat_reps, int epilogue,
                             const float* scale, const float* shift, hipStream_t s);

extern "C" int ayolo_conv_fwd(const ayolo_conv_desc* d, const void* x, const void* w, void* y, int epilogue,
                              const float* scale, const float* shift, double* stats, int stat_reps, int head_no,
                              ayolo_stream s) {
    int rc = check_desc(d, "conv_fwd");
    if (rc) return rc;
    AY_CHECK_ARG(x && w && y, "conv_fwd: null pointer");
    AY_CHECK_ARG(epilogue >= 0 && epilogue <= 5, "conv_fwd: epilogue %d", epilogue);
    AY_CHECK_ARG(epilogue != AYOLO_EPI_HEAD || (head_no > 0 && d->Cout % head_no == 0), "conv_fwd: head_no=%d", head_no);
    AY_CHECK_ARG(stats == nullptr || epilogue == AYOLO_EPI_NONE, "conv_fwd: stats need EPI_NONE");
    if ((epilogue == AYOLO_EPI_NONE || epilogue == AYOLO_EPI_AFFINE || epilogue == AYOLO_EPI_AFFINE_SILU) && is_packed_stem(d))
        return stem_fwd_dispatch(d, x, w, y, stats, stat_reps, epilogue, scale, shift, (hipStream_t)s);
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

// Forward of a 1x1 / stride-1 conv whose input is (partly) VIRTUAL: a segment's operand is act(z * xscale + xshift) formed on the
// way to the MFMAs (k_gconv<..., XF>), i.e. the consumer of a Conv-BN-act block reads the block's pre-activation z and the
// BatchNorm + activation pass that would have written the activation is not launched at all.
extern "C" int ayolo_conv_fwd_xf(const ayolo_conv_desc* d, const ayolo_xf_seg* segs, int nseg, float* xscale, float* xshift,
                                 const ayolo_xf_fin* fin, int nfin, void* xa, int ldxa, const void* w, void* y, int epilogue, const float* shift,
                                 double* stats, int stat_reps, int head_no, ayolo_stream s) {
    int rc = check_desc(d, "conv_fwd_xf");
    if (rc) return rc;
    AY_CHECK_ARG(segs && (nseg == 1 || nseg == 2) && xscale && xshift && w && y, "conv_fwd_xf: null pointer / %d segments", nseg);
    AY_CHECK_ARG(d->dtype == AYOLO_F16 && d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0,
                 "conv_fwd_xf: fp16 1x1 / stride 1 / no padding only");
    AY_CHECK_ARG(epilogue == AYOLO_EPI_NONE || epilogue == AYOLO_EPI_HEAD, "conv_fwd_xf: epilogue %d (plain + statistics, or YOLOHead)", epilogue);
    AY_CHECK_ARG(epilogue != AYOLO_EPI_HEAD || (head_no > 0 && d->Cout % head_no == 0), "conv_fwd_xf: head_no=%d", head_no);
    AY_CHECK_ARG(d->Cin <= 4096, "conv_fwd_xf: %d input channels (LDS constants table)", d->Cin);
    int c = 0;
    for (int k = 0; k < nseg; ++k) {
        AY_CHECK_ARG(segs[k].x && segs[k].C > 0 && segs[k].C % 8 == 0 && segs[k].ld % 8 == 0 && segs[k].ld >= segs[k].C,
                     "conv_fwd_xf: segment %d: C=%d ld=%d", k, segs[k].C, segs[k].ld);
        c += segs[k].C;
    }
    AY_CHECK_ARG(c == d->Cin && (nseg == 1 || segs[0].C % BK == 0), "conv_fwd_xf: segments cover %d of %d channels; the first must end on a multiple of %d",
                 c, d->Cin, BK);
    GConvP p{};
    p.x = segs[0].x; p.w = w; p.y = y;
    p.B = d->B; p.XH = d->H; p.XW = d->W; p.ldx = segs[0].ld;
    p.OH = d->Ho; p.OW = d->Wo; p.ish = 1; p.isw = 1;
    p.YH = d->Ho; p.YW = d->Wo; p.ldy = d->ldy; p.osh = 1; p.osw = 1; p.oah = 0; p.oaw = 0;
    p.C = d->Cin; p.ntaps = 1; p.K = p.C; p.ldw = p.K; p.Nout = d->Cout;
    p.epi = epilogue; p.scale = nullptr; p.shift = shift; p.stats = stats; p.head_no = head_no; p.accumulate = 0;
    p.stat_reps = stat_reps > 0 ? stat_reps : 1;
    p.y_linear = 1; p.x_linear = 1;
    p.Mtotal = (long long)d->B * d->Ho * d->Wo;
    p.dh[0] = 0; p.dw[0] = 0; p.wt[0] = 0;
    p.xf = 1; p.xf_scale = xscale; p.xf_shift = xshift;
    p.xf_act = (segs[0].act ? 1 : 0) | ((nseg > 1 && segs[1].act) ? 2 : 0);
    p.xf_virt = (segs[0].virt ? 1 : 0) | ((nseg > 1 && segs[1].virt) ? 2 : 0);
    p.xs_split = nseg > 1 ? segs[0].C : (d->Cin + BK - 1) / BK * BK;          // one segment: no step ever reaches the split
    if (nseg > 1) { p.x2 = segs[1].x; p.ldx2 = segs[1].ld; }
    AY_CHECK_ARG(nfin >= 0 && nfin <= 2 && (nfin == 0 || fin), "conv_fwd_xf: %d finalize records", nfin);
    for (int k = 0; k < nfin; ++k) {
        AY_CHECK_ARG(fin[k].stats && fin[k].C > 0 && fin[k].c0 >= 0 && fin[k].c0 + fin[k].C <= d->Cin && fin[k].sld >= fin[k].C && fin[k].reps >= 1 &&
                     fin[k].count > 0, "conv_fwd_xf: finalize record %d", k);
        p.fin[k] = fin[k];
    }
    p.nfin = nfin;
    AY_CHECK_ARG(xa == nullptr || (ldxa % 8 == 0 && ldxa >= d->Cin && (long long)d->B * d->H * d->W * ldxa * 2 < (1ll << 31) - 4096),
                 "conv_fwd_xf: store-back buffer: ld=%d (< 2 GiB)", ldxa);
    p.xa = xa; p.ldxa = ldxa; p.xa_bytes = xa ? (unsigned)((long long)d->B * d->H * d->W * ldxa * 2) : 0u;
    // the running statistics are updated by workgroup 0 of ONE launch: a conv that has to be cut into batch halves cannot carry them
    AY_CHECK_ARG(nfin == 0 || ((long long)d->B * d->H * d->W * (segs[0].ld > d->ldy ? segs[0].ld : d->ldy) * 4 < (1ll << 31) - 4096 &&
                               (nseg < 2 || (long long)d->B * d->H * d->W * segs[1].ld * 2 < (1ll << 31) - 4096)),
                 "conv_fwd_xf: tensors of 2 GiB need the separate finalize launch");
    return dispatch_gconv(d->dtype, p, (hipStream_t)s);
}

// dgrad: dx[n,h,w,ci] = sum_{kh,kw,co} dy[n,(h+ph-kh)/sh,(w+pw-kw)/sw,co] * w[co,kh,kw,ci] over exact divisions.
// Each (h mod sh, w mod sw) residue class is a stride-1 gather conv over dy with its own tap subset, so no
// MFMA work is spent on structural zeros.
static int conv_dgrad_impl(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                           const ayolo_bn_seg* segs, int nseg, int act, int sum_reps, ayolo_stream s);

extern "C" int ayolo_conv_dgrad(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                                ayolo_stream s) {
    return conv_dgrad_impl(d, dy, wt, dx, accumulate, nullptr, 0, 0, 1, s);
}

extern "C" int ayolo_conv_dgrad_bn(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                                   const ayolo_bn_seg* segs, int nseg, int act, int sum_reps, ayolo_stream s) {
    AY_CHECK_ARG(d && d->dtype == AYOLO_F16, "conv_dgrad_bn: fp16 only");
    AY_CHECK_ARG(segs && (nseg == 1 || nseg == 2) && sum_reps >= 1, "conv_dgrad_bn: nseg=%d sum_reps=%d", nseg, sum_reps);
    for (int k = 0; k < nseg; ++k) {
        const ayolo_bn_seg& g = segs[k];
        AY_CHECK_ARG(g.z && g.mean_invstd && g.sums, "conv_dgrad_bn: null pointer in segment %d", k);
        AY_CHECK_ARG(g.C > 0 && g.C % 8 == 0 && g.c0 >= 0 && g.c0 % 8 == 0 && g.ldz % 8 == 0 && g.ldz >= g.C && g.c0 + g.C <= d->Cin,
                     "conv_dgrad_bn: segment %d: c0=%d C=%d ldz=%d", k, g.c0, g.C, g.ldz);
    }
    // a 32-channel block of dx belongs to ONE segment (wave-uniform descriptor choice in the epilogue)
    AY_CHECK_ARG(nseg == 1 || (segs[1].c0 % 32 == 0 && segs[0].c0 + segs[0].C <= segs[1].c0),
                 "conv_dgrad_bn: second segment must start on a 32-channel boundary after the first");
    return conv_dgrad_impl(d, dy, wt, dx, accumulate, segs, nseg, act, sum_reps, s);
}

static int conv_dgrad_impl(const ayolo_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate,
                           const ayolo_bn_seg* segs, int nseg, int act, int sum_reps, ayolo_stream s) {
    int rc = check_desc(d, "conv_dgrad");
    if (rc) return rc;
    AY_CHECK_ARG(dy && wt && dx, "conv_dgrad: null pointer");
    const int ce = d->dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(d->Cout % ce == 0 && d->ldy % ce == 0, "conv_dgrad: Cout=%d ldy=%d must be multiples of %d", d->Cout,
                 d->ldy, ce);
    // Residue classes (a, b) = (h mod sh, w mod sw).  When every class has the same output grid (H % sh == W % sw == 0)
    // ONE launch walks all of them per dy tile: dy is read from HBM once instead of once per class (the later classes
    // hit L2) and the classes' interleaved half-line writes of dx meet in L2.  Otherwise one launch per class.
    const int ncls_all = d->sh * d->sw;
    const bool merge = ncls_all > 1 && ncls_all <= 4 && d->H % d->sh == 0 && d->W % d->sw == 0 &&
                       d->kh * d->kw <= MAX_TAPS;
    GConvP m{};
    int mt = 0;
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
            p.bnr = nseg; p.bnr_act = act; p.bnr_reps = sum_reps;
            for (int k = 0; k < nseg; ++k) p.bseg[k] = segs[k];        // z_bytes: dispatch_gconv, after its batch split
            p.y_linear = (d->sh == 1 && d->sw == 1) ? 1 : 0;
            p.x_linear = (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0) ? 1 : 0;
            p.Mtotal = (long long)d->B * p.OH * p.OW;
            if (merge && a == 0 && b == 0) m = p;
            const int cls = a * d->sw + b;
            int nt = 0;
            for (int i = 0; i < d->kh; ++i) {
                if ((a + d->ph - i) % d->sh != 0) continue;
                for (int j = 0; j < d->kw; ++j) {
                    if ((b + d->pw - j) % d->sw != 0) continue;
                    // floor division for negative numerators is not needed: exact multiples only
                    GConvP& q = merge ? m : p;
                    const int t = merge ? mt + nt : nt;
                    q.dh[t] = (signed char)((a + d->ph - i) / d->sh);
                    q.dw[t] = (signed char)((b + d->pw - j) / d->sw);
                    q.wt[t] = (signed char)(i * d->kw + j);
                    ++nt;
                }
            }
            if (merge) {
                m.ctap0[cls] = mt; m.cnt[cls] = nt; m.coah[cls] = a; m.coaw[cls] = b;
                mt += nt;
                continue;
            }
            p.ntaps = nt; p.K = nt * p.C;   // nt == 0: no tap reaches this residue class -> the kernel writes zeros
            rc = dispatch_gconv(d->dtype, p, (hipStream_t)s);
            if (rc) return rc;
        }
    if (merge) {
        m.ncls = ncls_all; m.ntaps = mt; m.K = mt * m.C;
        // 3x3 / stride 2 / pad 1 with <= 64 input channels: the nine (class, tap) products read four shifts of the dy tile
        // -> k_dgrad_s2 keeps all four classes' accumulators and loads every dy row group once
        const int cpad = (m.C + BK - 1) / BK * BK;
        // channel limit: round 3 took this kernel for <= 64 input channels only; measured in round 4 (tools/s2_sweep.py) it also wins on
        // 128 and 256 (128 -> 256 @ 80^2: 105 -> 84 us, 256 -> 512 @ 40^2: 94 -> 86 us; train step -0.09 ms) with 64-channel tiles
        // re-reading the dy rows from L2
        constexpr int dgrad_s2_maxc = 256;
        if (d->dtype == AYOLO_F16 && d->kh == 3 && d->kw == 3 && d->sh == 2 && d->sw == 2 && d->ph == 1 && d->pw == 1 &&
            d->Cin <= dgrad_s2_maxc && m.C >= BK && (cpad - m.C) * 4 <= cpad) {
            static const signed char order[9] = {4, 5, 7, 8, 3, 6, 1, 2, 0};
            for (int t = 0; t < 9; ++t) m.s2wt[t] = order[t];
            m.s2d = 1;
        }
        return dispatch_gconv(d->dtype, m, (hipStream_t)s);
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
    unsigned x_bytes, y_bytes;     // buffer descriptor extents (k_wgrad)
    unsigned gx, gy, splits;       // column tiles, channel tiles, pixel splits
    FastDiv dOW, dOH, dC;
    int linear;                    // 1x1 / stride 1 / no padding: the x row of pixel pp is row pp (no decode, no halo test)
    // split-K partials: split zz of this job stores its N x K fp32 tile sums at ws[ws_off + (zz0 + zz) * N * K ...] with plain
    // stores; k_wgrad_reduce adds the splits in a fixed order (see the header comment of k_wgrad)
    unsigned long long ws_off;     // in floats
    unsigned zz0;                  // first partial slot of this job (a logical layer cut into batch halves: the second half's
                                   // slots follow the first's)
    int dy_slot;                   // >= 0: dy is the launch's override pointer of that slot (YOLOHead levels: the loss hands a
                                   // different buffer over each step), else `dy`
    int tm;                        // output-channel tile of this job's launch class: 32 / 64 / 128
    // transform on load (1x1 / stride-1 consumers of a virtual activation, see ayolo_conv_fwd_xf): x is the producer's
    // pre-activation z and the operand is act(z * xf_scale[c] + xf_shift[c]); null = plain x
    const float* xf_scale; const float* xf_shift; int xf_act, xf_pad;
};

#define TNW 128     // dw columns per block tile

typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// (WItem -- one unit of work of a grouped launch: (job, dw tile, pixel split) -- lives in wgrad3.h, shared with k_wgrad3)
// one block of the reduction: n <= WRED_N consecutive elements of one layer's dw, S partials `stride` floats apart
#define WRED_N 2048
// (cols, ldd): the layer's dw is a column block of a wider matrix -- element e of the dense N x cols partials goes to
// dst[(e / cols) * ldd + e % cols] (the two input segments of a transform-on-load conv are two jobs over one weight matrix);
// e0 = index of the block's first element, ldd == cols: dense
// tpc (1, 2, 4 .. 64): threads that share one 16-byte column of a block and split its S partials between them (thread slice
// sl adds partials sl, sl + tpc, ..; the slices are then added in slice order through LDS: still a fixed order).  A layer with a
// tiny dw and a huge map (64 x 64 weights, 1.6 M pixels: S = 2 133 partials of 16 KB) otherwise has two blocks whose threads
// each walk all 2 133 partials one memory round trip after the other: 250 us for 35 MB.  A block covers wred_chunk(tpc) elements.
// pC > 0: the partials are k_wgrad3's TILE-MAJOR slots -- element (row n, column t * pC + c) of the dense matrix sits at
// w3_slot_offset(n, t, c) of its slot (every workgroup stores one contiguous block; dense slots made a 256 -> 256 layer's
// workgroups scatter 256-byte runs at a 1 KiB pitch: 0.9 TB/s of stores, profiles/r05_w3_probe_v5.txt); ws_off is then the
// LAYER's base, whatever e0
struct WRed { unsigned long long ws_off, stride; float* dst; unsigned n, S; float alpha; int overwrite; unsigned cols, ldd; unsigned long long e0; unsigned tpc, pC, pNB, pCB, ptc, pad; };
// element e (multiple of 4) of a dense N x (9 * C) matrix -> offset in a tile-major slot: tiles of (NB * 32 rows) x (CB * 32
// channels), inside a tile [n-block][c-block][tap][32 rows][32 channels]
__host__ __device__ static inline unsigned long long w3_perm(unsigned long long e, unsigned cols, unsigned C, unsigned NB, unsigned CB, unsigned tc) {
    const unsigned n = (unsigned)(e / cols), col = (unsigned)(e - (unsigned long long)n * cols);
    const unsigned t = col / C, c = col - t * C;
    const unsigned tni = n / (NB * 32u), nbl = (n >> 5) % NB, tci = c / (CB * 32u), cbl = (c >> 5) % CB;
    return ((unsigned long long)((tni * tc + tci) * NB * CB + nbl * CB + cbl) * 9u + t) * 1024u + (n & 31u) * 32u + (c & 31u);
}
static inline unsigned wred_tpc(unsigned S) {
    unsigned t = 1;
    while (t < 64 && S > 32 * t) t *= 2;                       // <= 32 partials per thread (16 .. 48 measured alike, 8 / 96 worse)
    return t;
}
__host__ __device__ static inline unsigned wred_chunk(unsigned tpc) { return tpc > 1 ? 1024u / tpc : (unsigned)WRED_N; }
struct WOvr { const void* q[4]; };           // dy override pointers of a launch (WGradP::dy_slot)

// A/B fragments for v_mfma_f32_32x32x16_f16 come out of the pixel-major LDS tiles t[pixel][channel] through the gfx950
// transposing LDS read (tr_frag_sw below): each 16-lane group reads a 4(pixel) x 16(channel) block, lane q supplies the
// address of row q/4, channels (q%4)*4.. and receives channel q of all 4 rows.
// ---------------------------------------------------------------------------------------------------
// k_wgrad: dw[n][tap*C + c] = sum_pixels dy[pixel][n] * x[pixel @ tap][c], split over pixel ranges.
// Same machinery as k_gconv: both operand tiles ([32 pixels][TM channels] of dy, [32 pixels][128 dw columns] of x)
// arrive by LDS-DMA two steps ahead (3 LDS stages, one raw barrier per step, counted vmcnt, branch-free loader with
// out-of-range zero fill).  Both operands are pixel-major in HBM, i.e. k-major for this GEMM: they are staged as
// they lie and transposed on the way to the MFMA by ds_read_b64_tr_b16.  The unpadded lane-linear image would put the
// 4 pixel rows of one transposing read on the same banks; the 64-byte groups of a row are therefore XOR-swizzled by the
// row (on the source address).
//
// Round 4: NO ATOMICS, FEW LAUNCHES.  Rounds 1-3 added every pixel split's tile into dw with fp32 atomics -- 30-40 % of a
// workgroup's life on the small maps (profiles/r03_wgrad_probe_*) and a summation order that changed from run to run --
// and launched one under-filled grid per layer.  Now
//   * a workgroup STORES its tile (plain 128-byte row segments) into its own slot of a split-K workspace and
//     k_wgrad_reduce adds the slots of an element in a fixed order: the weight gradient of a step is bit-reproducible;
//   * the kernel takes its problem from a JOB TABLE in device memory and its (job, tile, split) from an ITEM LIST, so ONE
//     launch covers the weight gradients of many layers (they have no mutual dependencies): the host (wgroup_plan) cuts
//     every layer of a group into items of similar length and deals them to eight per-XCD queues -- all tiles of one pixel
//     split back to back on one XCD, so the re-reads of dy (once per column tile) and x (once per tap) stay L2 hits --
//     longest items first.  A single layer (ayolo_conv_wgrad) is the same kernel with the job passed by value.
// The job struct is read through the constant address space (like the kernarg segment): uniform scalar loads the
// compiler may repeat instead of holding ~50 SGPRs.
// ---------------------------------------------------------------------------------------------------
template <typename T, int TM>
struct WT {
    static constexpr int ES = sizeof(T), CE = 16 / ES;
    static constexpr int BP = 32;                                  // pixels per step
    static constexpr int XROWB = TNW * ES, YROWB = TM * ES;        // bytes per pixel row of the two tiles
    static constexpr int XSTAGE = BP * XROWB;
    static constexpr int YSTAGE = BP * YROWB < 4096 ? 4096 : BP * YROWB;
    static constexpr int XR = XSTAGE / 4096, YR = YSTAGE / 4096, LPS = XR + YR;
    static constexpr int XRW = 1024 / XROWB, YRW = 1024 / YROWB;   // pixel rows per wave-instruction
    static constexpr int XCPR = XROWB / 16, YCPR = YROWB / 16;     // 16-byte chunks per row
    static constexpr int STAGE = XSTAGE + YSTAGE;
    static constexpr int WM = TM / 32, WN = 4 / WM, NI = TNW / (32 * WN);
    // fp16 swizzle: 64-byte group g of row r is stored at group g ^ ((r / RPB) & (G - 1)), G = min(groups per row, 4)
    static constexpr int XG = XROWB / 64 > 4 ? 4 : XROWB / 64, YG = YROWB / 64 > 4 ? 4 : YROWB / 64;
    static constexpr int XRPB = XROWB >= 256 ? 1 : 256 / XROWB, YRPB = YROWB >= 256 ? 1 : 256 / YROWB;
    static constexpr size_t LDS = (size_t)GNS * STAGE;
};

template <int ROWB, int G, int RPB>
__device__ __forceinline__ half8 tr_frag_sw(const unsigned char* tile, int k0, int c0, int lane) {
    const int q = lane & 15, rowl = q >> 2;
    const int colb = (c0 + (q & 3) * 4) * 2;
    const int sw = (rowl / RPB) & (G - 1);           // k0 is a multiple of 8: the row's swizzle only depends on rowl
    const unsigned char* p0 = tile + (k0 + rowl) * ROWB + ((((colb >> 6) ^ sw) << 6) | (colb & 63));
    fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0));
    fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0 + 4 * ROWB));
    half8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

typedef const __attribute__((address_space(4))) WGradP* wjob_cptr_t;

template <typename T, int TM, bool XFW = false>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 3 : 1)) void k_wgrad(WGradP pv, const WGradP* jobs, const WItem* items, float* ws,
                                                                            WOvr ovr) {
    using W = WT<T, TM>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % W::WM, wn = wave / W::WM;

#ifdef AYOLO_PROBE
    __shared__ unsigned long long s_probe[AY_PROBE_N];
    const bool probe_on = blockIdx.x < 512;
    int probe_k = 2;
    if (threadIdx.x < AY_PROBE_N) s_probe[threadIdx.x] = 0;
    __syncthreads();
    AY_PROBE(0);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 3] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- which (job, dw tile, pixel split): from the item list of a grouped launch, or -- single job, passed by value -- from
    // the block index: split zz on XCD zz % 8 with its gx * gy tiles consecutive there
    typedef __attribute__((address_space(4))) const char* kcptr_t;
    wjob_cptr_t pj;
    unsigned tile, zz;
    if (items != nullptr) {
        const WItem it = items[blockIdx.x];
        const unsigned job = (unsigned)__builtin_amdgcn_readfirstlane((int)it.job);
        if (job == 0xffffffffu) return;
        pj = (wjob_cptr_t)(unsigned long long)(jobs + job);
        tile = (unsigned)__builtin_amdgcn_readfirstlane((int)it.tile);
        zz = (unsigned)__builtin_amdgcn_readfirstlane((int)it.zz);
    } else {
        pj = (wjob_cptr_t)((kcptr_t)__builtin_amdgcn_kernarg_segment_ptr());     // pv is the first kernel argument
        const unsigned Lb = blockIdx.x, xcd = Lb & 7u, local = Lb >> 3;
        const unsigned ntile = pj->gx * pj->gy;
        tile = local % ntile;
        zz = (local / ntile) * 8u + xcd;
        if (zz >= pj->splits) return;
    }
#define p (*pj)
#define WFD(f_) FastDiv{pj->f_.m, pj->f_.s1, pj->f_.s2}     /* member-wise: an address-space-4 struct has no copy constructor */
    ScalarTouch<(int)sizeof(WGradP)> jt_;             // every line of the job requested at once (gfx950_dma.h)
    jt_.issue((unsigned long long)pj);
    const int j0 = (int)(tile % p.gx) * TNW;          // dw column tile (tap*C + c)
    const int n0 = (int)(tile / p.gx) * TM;           // output-channel tile
    const unsigned P = (unsigned)p.P;
    const unsigned pbeg = zz * (unsigned)p.chunk;
    const unsigned pend = pbeg + (unsigned)p.chunk < P ? pbeg + (unsigned)p.chunk : P;
    // (the host never creates an empty split: every slot of the workspace is written)
    jt_.done();

    const int slot = p.dy_slot;
    const void* dyp = slot < 0 ? p.dy : (slot == 0 ? ovr.q[0] : (slot == 1 ? ovr.q[1] : (slot == 2 ? ovr.q[2] : ovr.q[3])));
    const v4i32 rsX = make_srd(p.x, p.x_bytes), rsY = make_srd(dyp, p.y_bytes);
    const unsigned lds_tiles = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)smem_raw);

    // ---- x loader lanes: LDS position (row piece*XRW + xrowin, chunk slot xslot) <- source chunk xsrc of that row
    const int xslot = lane % W::XCPR, xrowin = lane / W::XCPR;
    int xsrc = xslot;
    if constexpr (sizeof(T) == 2) xsrc = ((((xslot >> 2) ^ ((xrowin / W::XRPB) & (W::XG - 1))) << 2) | (xslot & 3));
    const int xcol = j0 + xsrc * W::CE;
    const bool xcol_ok = xcol < p.K;
    const unsigned xtap = fdiv(xcol_ok ? (unsigned)xcol : 0u, WFD(dC));
    const int xcb = (int)((xcol_ok ? (unsigned)xcol : 0u) - xtap * (unsigned)p.C) * W::ES;
    // tap offsets of this lane's column: scalar loads of the two byte arrays + per-lane select (see k_gconv: indexed per lane
    // they would be vector loads, ~2 000 cycles before the first DMA)
    static_assert(offsetof(WGradP, dh) % 4 == 0 && offsetof(WGradP, dw_) == offsetof(WGradP, dh) + MAX_TAPS, "tap arrays");
    typedef __attribute__((address_space(4))) const int* kiptr_t;
    const kiptr_t kwords = (kiptr_t)((kcptr_t)pj + offsetof(WGradP, dh));
    int kw_[2 * (MAX_TAPS / 4)];
#pragma unroll
    for (int i = 0; i < 2 * (MAX_TAPS / 4); ++i) kw_[i] = kwords[i];
#pragma unroll
    for (int g = 0; g < 2; ++g)
        asm volatile("" : "+s"(kw_[9 * g]), "+s"(kw_[9 * g + 1]), "+s"(kw_[9 * g + 2]), "+s"(kw_[9 * g + 3]), "+s"(kw_[9 * g + 4]),
                          "+s"(kw_[9 * g + 5]), "+s"(kw_[9 * g + 6]), "+s"(kw_[9 * g + 7]), "+s"(kw_[9 * g + 8]));
    const unsigned xtq = xtap < MAX_TAPS ? xtap : 0u;
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int i = 0; i < MAX_TAPS / 4; ++i) {
        const bool me = (xtq >> 2) == (unsigned)i;
        s0 = me ? kw_[i] : s0;
        s1 = me ? kw_[MAX_TAPS / 4 + i] : s1;
    }
    const int xdh = (int)(signed char)(s0 >> (8 * (xtq & 3))), xdw = (int)(signed char)(s1 >> (8 * (xtq & 3)));
    // ---- dy loader lanes
    const int yslot = lane % W::YCPR, yrowin = lane / W::YCPR;
    int ysrc = yslot;
    if constexpr (sizeof(T) == 2 && W::YG > 1) ysrc = ((((yslot >> 2) ^ ((yrowin / W::YRPB) & (W::YG - 1))) << 2) | (yslot & 3));
    const bool ycol_ok = n0 + ysrc * W::CE < p.N;
    const unsigned ycb = (unsigned)(n0 + ysrc * W::CE) * W::ES;

    float16v acc[W::NI];
#pragma unroll
    for (int i = 0; i < W::NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // transform on load: this lane's x chunks are the same 8 dw columns (= input channels of a 1x1 conv) in every step, so its
    // 2 x 8 constants live in registers for the whole item; the chunks are transformed in place in their LDS stage by the lane
    // whose DMA wrote them (see k_gconv's xf_transform), one step before they are consumed
    // (XFW instantiations only: the constants and the branch cost 15-40 registers, i.e. a wavefront per SIMD on the narrow tiles)
    const bool xf = XFW && sizeof(T) == 2 && p.xf_scale != nullptr;
    float xfa[XFW ? 8 : 1], xfb[XFW ? 8 : 1];
#pragma unroll
    for (int e = 0; e < (XFW ? 8 : 1); ++e) { xfa[e] = 1.0f; xfb[e] = 0.0f; }
    if constexpr (XFW) if (xf && xcol_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xfa[e] = p.xf_scale[xcol + e]; xfb[e] = p.xf_shift[xcol + e]; }     // K = C, C % 8 == 0
    }
    const bool xf_act = p.xf_act != 0;
    auto xf_transform = [&](unsigned so) {
        if constexpr (sizeof(T) == 2 && XFW) {
#pragma unroll
            for (int r = 0; r < W::XR; ++r) {
                half8* q = reinterpret_cast<half8*>(smem_raw + so + (r * 4 + wave) * 1024 + lane * 16);
                half8 h = *q;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = __builtin_fmaf((float)h[e], xfa[e], xfb[e]);
                    if (xf_act) u = u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
                    h[e] = (half_t)u;
                }
                *q = h;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    const int nk = (int)((pend - pbeg + W::BP - 1) / W::BP);
    // DMA of reduction step `st` into the stage at byte offset `so`; steps >= nk lie beyond pend: all rows zero-fill
#define W_ISSUE(st, so)                                                                                             \
    {                                                                                                               \
        const unsigned pt = pbeg + (unsigned)(st) * W::BP;                                                          \
        _Pragma("unroll") for (int r = 0; r < W::XR; ++r) {                                                         \
            const unsigned pp = pt + (r * 4 + wave) * W::XRW + xrowin;                                              \
            const unsigned t = fdiv(pp, WFD(dOW));                                                                     \
            const int ow = (int)(pp - t * (unsigned)p.OW);                                                          \
            const unsigned n = fdiv(t, WFD(dOH));                                                                      \
            const int oh = (int)(t - n * (unsigned)p.OH);                                                           \
            const unsigned ih = (unsigned)(oh * p.sh + xdh), iw = (unsigned)(ow * p.sw + xdw);                      \
            const bool ok = (pp < pend) & xcol_ok & (ih < (unsigned)p.XH) & (iw < (unsigned)p.XW);                  \
            const unsigned off = ok ? ((n * (unsigned)p.XH + ih) * (unsigned)p.XW + iw) * (unsigned)p.ldx * W::ES + (unsigned)xcb : G_OOB; \
            glds16(rsX, lds_tiles + (so) + (r * 4 + wave) * 1024, off);                                             \
        }                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < W::YR; ++r) {                                                         \
            const unsigned row = (r * 4 + wave) * W::YRW + yrowin;                                                  \
            const unsigned pp = pt + row;                                                                           \
            const bool ok = (row < (unsigned)W::BP) & (pp < pend) & ycol_ok;                                        \
            const unsigned off = ok ? pp * (unsigned)p.ldy * W::ES + ycb : G_OOB;                                   \
            glds16(rsY, lds_tiles + (so) + W::XSTAGE + (r * 4 + wave) * 1024, off);                                 \
        }                                                                                                           \
    }

    unsigned so0 = 0, so1 = W::STAGE, so2 = 2 * W::STAGE;
    W_ISSUE(0, so0)
    W_ISSUE(1, so1)
    if (xf) {
        wait_vm<W::LPS>();                       // step 0's pieces
        xf_transform(so0);
    }
    AY_PROBE(1);
    for (int kt = 0; kt < nk; ++kt) {
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        wait_vm<W::LPS>();                       // step kt landed (this wave's part) ...
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        __builtin_amdgcn_s_barrier();            // ... everyone's part landed, everyone finished reading step kt-1
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        const unsigned char* cX = smem_raw + so0;
        const unsigned char* cY = smem_raw + so0 + W::XSTAGE;
        if constexpr (sizeof(T) == 4) W_ISSUE(kt + 2, so2)
        if constexpr (sizeof(T) == 2) {
            // every fragment of the step first (see g_fetch_frags: hipcc otherwise serialises ds_read -> lgkmcnt(0) ->
            // v_mfma through one register quad), then the address work of step kt+2 while the transposing reads are in
            // flight, then the MFMAs with that step's DMA pieces issued one per MFMA (see g_mma_k_issue)
            half8 fa[W::BP / 16], fb[W::BP / 16][W::NI];
            const int csub = ((lane >> 4) & 1) * 16;
#pragma unroll
            for (int kk = 0; kk < W::BP / 16; ++kk) {
                const int k0 = kk * 16 + (lane >> 5) * 8;
                fa[kk] = tr_frag_sw<W::YROWB, W::YG, W::YRPB>(cY, k0, wm * 32 + csub, lane);
#pragma unroll
                for (int ni = 0; ni < W::NI; ++ni)
                    fb[kk][ni] = tr_frag_sw<W::XROWB, W::XG, W::XRPB>(cX, k0, wn * W::NI * 32 + ni * 32 + csub, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned offs[W::LPS];
            {
                const unsigned pt = pbeg + (unsigned)(kt + 2) * W::BP;
#pragma unroll
                for (int r = 0; r < W::XR; ++r) {
                    const unsigned pp = pt + (r * 4 + wave) * W::XRW + xrowin;
                    if (p.linear) {
                        offs[r] = ((pp < pend) & xcol_ok) ? pp * (unsigned)p.ldx * W::ES + (unsigned)xcb : G_OOB;
                    } else {
                        const unsigned t = fdiv(pp, WFD(dOW));
                        const int ow = (int)(pp - t * (unsigned)p.OW);
                        const unsigned n = fdiv(t, WFD(dOH));
                        const int oh = (int)(t - n * (unsigned)p.OH);
                        const unsigned ih = (unsigned)(oh * p.sh + xdh), iw = (unsigned)(ow * p.sw + xdw);
                        const bool ok = (pp < pend) & xcol_ok & (ih < (unsigned)p.XH) & (iw < (unsigned)p.XW);
                        offs[r] = ok ? ((n * (unsigned)p.XH + ih) * (unsigned)p.XW + iw) * (unsigned)p.ldx * W::ES + (unsigned)xcb : G_OOB;
                    }
                }
#pragma unroll
                for (int r = 0; r < W::YR; ++r) {
                    const unsigned row = (r * 4 + wave) * W::YRW + yrowin;
                    const unsigned pp = pt + row;
                    const bool ok = (row < (unsigned)W::BP) & (pp < pend) & ycol_ok;
                    offs[W::XR + r] = ok ? pp * (unsigned)p.ldy * W::ES + ycb : G_OOB;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NMMA = (W::BP / 16) * W::NI;
#pragma unroll
            for (int q = 0; q < (NMMA > W::LPS ? NMMA : W::LPS); ++q) {
                if (q < NMMA) acc[q % W::NI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[q / W::NI], fb[q / W::NI][q % W::NI], acc[q % W::NI], 0, 0, 0);
                if (q < W::XR) glds16(rsX, lds_tiles + so2 + (q * 4 + wave) * 1024, offs[q]);
                else if (q < W::LPS) glds16(rsY, lds_tiles + so2 + W::XSTAGE + ((q - W::XR) * 4 + wave) * 1024, offs[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int kk = 0; kk < W::BP / 16; ++kk) {
            const int k0 = kk * 16 + (lane >> 5) * 8;
            if constexpr (sizeof(T) == 2) {
            } else {
                const float* fY = reinterpret_cast<const float*>(cY);
                const float* fX = reinterpret_cast<const float*>(cX);
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const float a = fY[(k0 + s8) * TM + wm * 32 + (lane & 31)];
#pragma unroll
                    for (int ni = 0; ni < W::NI; ++ni) {
                        const float b = fX[(k0 + s8) * TNW + wn * W::NI * 32 + ni * 32 + (lane & 31)];
                        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ni], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (sizeof(T) == 4) AY_MFMA_PAD("s_nop 15\n\ts_nop 3");   // see k_gconv: MFMA result hazard across the back edge
        if (xf) {
            wait_vm<W::LPS>();                   // step kt+1's pieces (behind them only this step's issues)
            xf_transform(so1);
        }
#ifdef AYOLO_PROBE
        AY_PROBE(probe_k); ++probe_k;
#endif
        const unsigned t = so0; so0 = so1; so1 = so2; so2 = t;
    }
#undef W_ISSUE
    wait_vm<0>();                                // trailing zero-fill DMAs must land before this LDS is released
    AY_PROBE(AY_PROBE_N - 4);
    if constexpr (sizeof(T) == 2) AY_MFMA_PAD("s_nop 11");   // the accumulators are read right below
    // acc[ni][r]: row (out channel) = n0 + wm*32 + 8*(r>>2) + 4*(lane>>5) + (r&3); col = j0 + wn*NI*32 + ni*32 + (lane&31)
    // -> this split's slot of the workspace, plain stores: lanes 0..31 of a store cover 128 contiguous bytes of one dw row
    float* slotp = ws + p.ws_off + (unsigned long long)(p.zz0 + zz) * ((unsigned long long)p.N * (unsigned long long)p.K);
    const int K = p.K, N = p.N;
#pragma unroll
    for (int ni = 0; ni < W::NI; ++ni) {
        const int col = j0 + wn * W::NI * 32 + ni * 32 + (lane & 31);
        if (col >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = n0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (row < N) slotp[(long long)row * K + col] = acc[ni][r];
        }
    }
#undef p
#undef WFD
#ifdef AYOLO_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AY_PROBE(AY_PROBE_N - 1);
    if (threadIdx.x == 0) s_probe[AY_PROBE_N - 2] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (probe_on && threadIdx.x < AY_PROBE_N) g_probe[blockIdx.x * AY_PROBE_N + threadIdx.x] = s_probe[threadIdx.x];
#endif
}

// dst[i] (+)= alpha * sum_s ws[ws_off + s * stride + i]: four interleaved running sums over s (independent loads in flight),
// combined in a fixed order -- the result does not depend on how the splits were scheduled.  One block per WRed entry of a
// grouped launch (`red`), or -- single layer -- entry `rv` advanced by the block index.
__global__ __launch_bounds__(256) void k_wgrad_reduce(WRed rv, const WRed* red, const float* ws) {
    WRed r;
    if (red != nullptr) r = red[blockIdx.x];
    else {
        r = rv;
        const unsigned chunk = wred_chunk(rv.tpc);
        const unsigned long long o = (unsigned long long)blockIdx.x * chunk;
        const unsigned long long left = (unsigned long long)rv.n > o ? (unsigned long long)rv.n - o : 0ull;     // rv.n: all elements
        r.n = (unsigned)(left < chunk ? left : chunk);
        if (!rv.pC) r.ws_off += o;
        r.e0 = o;
    }
    const bool dense = r.ldd == r.cols;
    if (r.tpc > 1) {
        __shared__ float4v part[256];
        const unsigned ncol = 256u / r.tpc;                    // 16-byte columns of this block (n <= 4 * ncol)
        const unsigned col = threadIdx.x % ncol, sl = threadIdx.x / ncol;
        const unsigned i = col * 4;
        float4v a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0, a2 = a0, a3 = a0;
        if (i < r.n) {
            const float* src = ws + r.ws_off + (r.pC ? w3_perm(r.e0 + i, r.cols, r.pC, r.pNB, r.pCB, r.ptc) : (unsigned long long)i);
            unsigned q = sl;
            for (; q + 3 * r.tpc < r.S; q += 4 * r.tpc) {
                a0 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(q) * r.stride);
                a1 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(q + r.tpc) * r.stride);
                a2 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(q + 2 * r.tpc) * r.stride);
                a3 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(q + 3 * r.tpc) * r.stride);
            }
            for (; q < r.S; q += r.tpc) a0 += *reinterpret_cast<const float4v*>(src + (unsigned long long)q * r.stride);
        }
        part[threadIdx.x] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (sl == 0 && i < r.n) {
            float4v v = part[col];
            for (unsigned t = 1; t < r.tpc; ++t) v += part[t * ncol + col];
            v *= r.alpha;
            const unsigned long long e = r.e0 + i;
            float4v* d = reinterpret_cast<float4v*>(r.dst + (dense ? e : (e / r.cols) * r.ldd + e % r.cols));
            if (!r.overwrite) v += *d;
            *d = v;
        }
        return;
    }
    for (unsigned i = threadIdx.x * 4; i < r.n; i += 1024) {
        const float* src = ws + r.ws_off + (r.pC ? w3_perm(r.e0 + i, r.cols, r.pC, r.pNB, r.pCB, r.ptc) : (unsigned long long)i);
        float4v a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0, a2 = a0, a3 = a0;
        unsigned s = 0;
        for (; s + 4 <= r.S; s += 4) {
            a0 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(s + 0) * r.stride);
            a1 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(s + 1) * r.stride);
            a2 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(s + 2) * r.stride);
            a3 += *reinterpret_cast<const float4v*>(src + (unsigned long long)(s + 3) * r.stride);
        }
        for (; s < r.S; ++s) a0 += *reinterpret_cast<const float4v*>(src + (unsigned long long)s * r.stride);
        float4v v = ((a0 + a1) + (a2 + a3)) * r.alpha;
        const unsigned long long e = r.e0 + i;                 // cols % 4 == 0: the four elements share a row
        float4v* d = reinterpret_cast<float4v*>(r.dst + (dense ? e : (e / r.cols) * r.ldd + e % r.cols));
        if (!r.overwrite) v += *d;
        *d = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_stem_wgrad: weight gradient of the packed stem conv (image as pixel pairs [B][H][W/2][8 halves]; the 6x6 / stride 2 / pad 2
// conv of kindle's first Conv row -- res/configs/model/yolov5s.yaml:21 -- is a 6 x 3 / stride (2, 1) / pad (2, 1) conv over
// pairs, see functional._Geometry), fp16, Cout <= 64.
// It is the LAST kernel of backward -- nothing is left to overlap it with -- and the generic k_wgrad gathers its x tile in
// 16-byte taps (one DMA lane per (pixel, tap)), spends half of its second 128-column tile on padding (K = 144) and pays the
// atomics of ~600 pixel splits: 320-390 us at 1.9 TB/s with the chip to itself.  Here the input patch of a 4 x 64 output tile
// is staged ONCE, as it lies (12 rows x 68 pairs, 13 KB): for output pixel (r, c) and kernel row dh the three pair taps are
// the 48 contiguous bytes at patch[2r + dh][c .. c + 2], consecutive pixels 16 bytes apart -- exactly a pixel-major operand
// with a 16-byte row stride, which the transposing LDS read turns into MFMA fragments (columns = 3 pairs x 8 halves + one
// junk pair = 32).  Per 16 pixels: one dy fragment per 32 output channels and six x fragments (one per dh), six MFMAs; every
// wavefront keeps the whole 32 x (6 x 32) gradient for its row of the tile in registers across ALL tiles of the workgroup
// and the workgroup sends ONE set of atomics at the very end.  398 -> 141 us at batch 64, 640 x 640 (tools/stem_probe.py: tile
// loop 92-126 us = the 524 MB at 4.5-5 TB/s, reduction + atomics 5 us), train step -0.2 ms (profiles/r03_stem_kernels.txt).
// ---------------------------------------------------------------------------------------------------
struct StemWP {
    const half_t* x; const half_t* dy; float* dw;
    int B, H, WP, Ho, Wo, ldy, N, K;       // WP = pairs per image row; K = 6 * 24 (row length of dw)
    float alpha;
    int tw, th;                            // tiles per row / per column of one image
    long long ntiles;
    unsigned x_bytes, y_bytes;             // buffer descriptor extents (< 2 GiB, host check)
    // BN mode (k_stem_wgrad<MB, true>): `dy` is da, the gradient of the block's OUTPUT a = act(bn(z)); the kernel forms
    // dz = bn_act_backward(da, z) on the way to LDS -- the stem has no dgrad, so nobody else reads its dz and the separate
    // BatchNorm-backward apply pass (read da, read z, write dz: 1.26 GB at batch 64) disappears
    const half_t* z; int ldz; unsigned z_bytes;
    const float* mean; const float* invstd; const float* gamma; const float* beta;
    const double* sums; int reps, act;
    int NS;                                // channel count of the sums' layout [reps][2][NS] (a launch may cover a slice of them)
    float* dgamma; float* dbeta; float grad_scale;
};
#define STEM_TR 4                          // output rows per tile = wavefronts
#define STEM_TC 64                         // output columns per tile
#define STEM_PR (2 * STEM_TR + 4)          // patch rows: 2 * TR + (6 - 2)
#define STEM_PC (STEM_TC + 4)              // patch pair columns: TC + 2 (taps) + 1 (junk pair) rounded to 68

template <int MB, bool BN = false>
__global__ __launch_bounds__(256, (MB == 1 ? 2 : 1)) void k_stem_wgrad(StemWP p) {
    constexpr int DYROW = 64 * MB;                                   // bytes per pixel row of the dy tile
    constexpr int PATCH_B = STEM_PR * STEM_PC * 16, DY_B = STEM_TR * STEM_TC * DYROW;
    constexpr int NPCH = STEM_PR * STEM_PC, NDCH = STEM_TR * STEM_TC * 4 * MB;     // 16-byte chunks per tile
    constexpr int PPT = (NPCH + 255) / 256, DPT = NDCH / 256;
    constexpr int NE = MB * 6 * 16;
    constexpr size_t TILES_B = 2 * (size_t)(PATCH_B + DY_B), RED_B = 2 * (size_t)NE * 64 * 4;
    constexpr size_t CST_OFF = TILES_B > RED_B ? TILES_B : RED_B;                 // BN constants [5][32 * MB] floats behind both
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];           // [2][patch | dy tile] (| constants)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    float16v acc[MB][6];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int g = 0; g < 6; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.0f;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.dy), 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(BN ? p.z : p.dy), 0, BN ? p.z_bytes : p.y_bytes, 0x00020000);
    float* cst = reinterpret_cast<float*>(sm + CST_OFF);
    if constexpr (BN) {
        // per-channel constants (fp16 form of k_bn_bwd_apply): u = z*A + Bc, du = da * act'(u), dz = du*P + z*R2 + Q2 with
        // P = gamma*invstd, R2 = -P*m2*invstd, Q2 = P*(m2*mean*invstd - m1), m1 = sum(du)/n, m2 = sum(du*xhat)/n
        const float invn = 1.0f / ((float)p.B * (float)p.Ho * (float)p.Wo);
        for (int c = tid; c < 32 * MB; c += 256) {
            float A = 0.0f, Bc = 0.0f, P = 0.0f, R2 = 0.0f, Q2 = 0.0f;
            if (c < p.N) {
                const float mu = p.mean[c], is = p.invstd[c], ga = opt_load(p.gamma, p.mean, c, 1.0f), be = opt_load(p.beta, p.mean, c, 0.0f);
                double d1, d2;
                rep_sum2(p.sums + c, (size_t)2 * p.NS, (size_t)p.NS, p.reps, d1, d2);
                const float s1 = (float)d1, s2 = (float)d2;
                const float m1 = s1 * invn, m2 = s2 * invn;
                A = is * ga; Bc = be - mu * A; P = ga * is;
                const float Rr = -P * m2, Q = -P * m1, nmi = -mu * is;
                R2 = is * Rr; Q2 = __builtin_fmaf(nmi, Rr, Q);
                if (blockIdx.x == 0) {
                    if (p.dbeta) p.dbeta[c] = s1 * p.grad_scale;
                    if (p.dgamma) p.dgamma[c] = s2 * p.grad_scale;
                }
            }
            cst[c] = A; cst[32 * MB + c] = Bc; cst[2 * 32 * MB + c] = P; cst[3 * 32 * MB + c] = R2; cst[4 * 32 * MB + c] = Q2;
        }
        __syncthreads();
    }
    // patch: two register sets, the loads of tile i+2 are issued while tile i is computed and written to LDS an iteration later;
    // dy (BN: da and z): one set, loaded at the top of an iteration and written (BN: transformed) at its end
    uint4 rp[2][PPT], rd[DPT], rz[BN ? DPT : 1];
    unsigned dok = 0;                                          // BN: bit j = chunk j of the fetched dy set lies inside the map / channels
    auto tile_of = [&](long long t, bool& live, int& n, int& oh0, int& ow0) {
        live = t < p.ntiles;
        const long long tt = live ? t : 0;
        const int tx = (int)(tt % p.tw);
        const long long u = tt / p.tw;
        const int ty = (int)(u % p.th);
        n = (int)(u / p.th); oh0 = ty * STEM_TR; ow0 = tx * STEM_TC;
    };
    auto fetch_patch = [&](long long t, uint4 (&fp)[PPT]) {
        bool live; int n, oh0, ow0;
        tile_of(t, live, n, oh0, ow0);
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = tid + 256 * j;
            const int row = i / STEM_PC, col = i - row * STEM_PC;
            const int ih = 2 * oh0 - 2 + row, ip = ow0 - 1 + col;
            const bool ok = live && i < NPCH && (unsigned)ih < (unsigned)p.H && (unsigned)ip < (unsigned)p.WP;
            // branch-free: out-of-image chunks take the descriptor's out-of-range offset (hardware returns 0).  Written as
            // `ok ? *ptr : 0` every load becomes a branch + s_waitcnt vmcnt(0) and the loads of a tile run one after the other
            // (see "Why LDS-DMA + counted waits" in DESIGN.md)
            const unsigned off = ok ? (unsigned)((((unsigned)n * (unsigned)p.H + (unsigned)ih) * (unsigned)p.WP + (unsigned)ip) * 16u) : G_OOB;
            fp[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
        }
    };
    auto fetch_dy = [&](long long t) {
        bool live; int n, oh0, ow0;
        tile_of(t, live, n, oh0, ow0);
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int i = tid + 256 * j;
            const int px = i / (4 * MB), part = i - px * (4 * MB);
            const int oh = oh0 + px / STEM_TC, ow = ow0 + px % STEM_TC;
            const bool ok = live && oh < p.Ho && ow < p.Wo && part * 8 < p.N;     // N % 8 == 0 (host check)
            const unsigned pix = ((unsigned)n * (unsigned)p.Ho + (unsigned)oh) * (unsigned)p.Wo + (unsigned)ow;
            if constexpr (BN) dok = ok ? (dok | (1u << j)) : (dok & ~(1u << j));
            rd[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsD, ok ? (pix * (unsigned)p.ldy + (unsigned)part * 8u) * 2u : G_OOB, 0, 0));
            if constexpr (BN)
                rz[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, ok ? (pix * (unsigned)p.ldz + (unsigned)part * 8u) * 2u : G_OOB, 0, 0));
        }
    };
    auto stash_patch = [&](int buf, const uint4 (&fp)[PPT]) {
        unsigned char* b = sm + buf * (PATCH_B + DY_B);
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = tid + 256 * j;
            if (i < NPCH) *reinterpret_cast<uint4*>(b + i * 16) = fp[j];
        }
    };
    auto stash_dy = [&](int buf) {
        unsigned char* b = sm + buf * (PATCH_B + DY_B) + PATCH_B;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            uint4 v = rd[j];
            if constexpr (BN) {
                // this thread's 8 channels are the same for every chunk it owns (256 is a multiple of 4 * MB)
                const int c8 = ((tid + 256 * j) % (4 * MB)) * 8;
                const half8 da = __builtin_bit_cast(half8, rd[j]), zz = __builtin_bit_cast(half8, rz[j]);
                half8 o;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4v kA = *reinterpret_cast<const float4v*>(cst + c8 + 4 * q);
                    const float4v kB = *reinterpret_cast<const float4v*>(cst + 32 * MB + c8 + 4 * q);
                    const float4v kP = *reinterpret_cast<const float4v*>(cst + 2 * 32 * MB + c8 + 4 * q);
                    const float4v kR = *reinterpret_cast<const float4v*>(cst + 3 * 32 * MB + c8 + 4 * q);
                    const float4v kQ = *reinterpret_cast<const float4v*>(cst + 4 * 32 * MB + c8 + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float zf = (float)zz[4 * q + e], df = (float)da[4 * q + e];
                        float du = df;
                        if (p.act) {
                            const float u = __builtin_fmaf(zf, kA[e], kB[e]);
                            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
                            du = df * (sg * __builtin_fmaf(u, 1.0f - sg, 1.0f));
                        }
                        o[4 * q + e] = (half_t)__builtin_fmaf(du, kP[e], __builtin_fmaf(zf, kR[e], kQ[e]));
                    }
                }
                // a chunk outside the map has da = z = 0 but dz = Q2 != 0 by the formula: it must contribute nothing
                v = ((dok >> j) & 1u) ? __builtin_bit_cast(uint4, o) : make_uint4(0, 0, 0, 0);
            }
            *reinterpret_cast<uint4*>(b + (tid + 256 * j) * 16) = v;
        }
    };
    const int csub = ((lane >> 4) & 1) * 16;
    auto compute = [&](int buf) {
        const unsigned char* pb = sm + buf * (PATCH_B + DY_B);
        const unsigned char* db = pb + PATCH_B;
        // this wavefront: output row `wave` of the tile, 64 pixels = four 16-deep slices
#pragma unroll
        for (int kq = 0; kq < STEM_TC / 16; ++kq) {
            const int c0 = kq * 16 + (lane >> 5) * 8;          // first of this lane's 8 pixels (k values)
            half8 fa[MB], fb[6];
#pragma unroll
            for (int m = 0; m < MB; ++m) fa[m] = tr_frag_sw<DYROW, 1, 1>(db, wave * STEM_TC + c0, m * 32 + csub, lane);
#pragma unroll
            for (int g = 0; g < 6; ++g) fb[g] = tr_frag_sw<16, 1, 1>(pb + (2 * wave + g) * (STEM_PC * 16), c0, csub, lane);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int g = 0; g < 6; ++g) acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[m], fb[g], acc[m][g], 0, 0, 0);
        }
    };

    const long long step = gridDim.x;
    long long t = blockIdx.x;
    if (t >= p.ntiles) return;
#ifdef AYOLO_PROBE
#define STEM_MARK(k_) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_probe[blockIdx.x * AY_PROBE_N + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STEM_MARK(k_) do { } while (0)
#endif
    STEM_MARK(0);
    fetch_patch(t, rp[0]);
    fetch_dy(t);
    stash_patch(0, rp[0]);
    stash_dy(0);
    fetch_patch(t + step, rp[1]);                              // tile 1's patch in flight
    __syncthreads();
    STEM_MARK(1);
    // two tiles per trip so that the register sets are named at compile time: buffer 0 holds tile t, set 1 the patch of t + step
    for (; t < p.ntiles; t += 2 * step) {
        fetch_patch(t + 2 * step, rp[0]);
        fetch_dy(t + step);
        compute(0);
        stash_patch(1, rp[1]);                                 // tile t + step (zeros beyond the last tile)
        stash_dy(1);
        __syncthreads();
        if (t + step >= p.ntiles) break;
        fetch_patch(t + 3 * step, rp[1]);
        fetch_dy(t + 2 * step);
        compute(1);
        stash_patch(0, rp[0]);                                 // tile t + 2 * step
        stash_dy(0);
        __syncthreads();
    }
    STEM_MARK(2);
    // ---- the four wavefronts' tiles summed through LDS with plain stores / loads in two rounds (waves 2, 3 -> waves 0, 1; wave 1
    // -> wave 0; ds_add_f32 for the same job took 24 us: ~190 cycles per 64-lane LDS atomic), the total back to LDS, then 1/4
    // of the global atomics from each wavefront
    float* red = reinterpret_cast<float*>(sm);                 // [2][MB * 6 * 16][64]
    auto put = [&](float* dst) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int g = 0; g < 6; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((m * 6 + g) * 16 + r) * 64 + lane] = acc[m][g][r];
    };
    auto add = [&](const float* src) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int g = 0; g < 6; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][g][r] += src[((m * 6 + g) * 16 + r) * 64 + lane];
    };
    if (wave >= 2) put(red + (wave - 2) * NE * 64);
    __syncthreads();
    if (wave < 2) add(red + wave * NE * 64);
    __syncthreads();
    if (wave == 1) put(red);
    __syncthreads();
    if (wave == 0) { add(red); }
    __syncthreads();
    if (wave == 0) put(red);
    __syncthreads();
    STEM_MARK(3);
    // acc row (output channel) = m * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); column = lane & 31 = pair tap * 8 + half
    const int col = lane & 31;
    if (col < 24) {
        for (int e = wave; e < MB * 6 * 16; e += 4) {
            const int r = e & 15, g = (e >> 4) % 6, m = e / 96;
            const int co = m * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (co < p.N) unsafeAtomicAdd(&p.dw[(long long)co * p.K + g * 24 + col], red[e * 64 + lane] * p.alpha);
        }
    }
#ifdef AYOLO_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STEM_MARK(4);
}

template <int MB, bool BN>
static int launch_stem_wgrad(StemWP p, hipStream_t s) {
    constexpr size_t lds_tiles = 2 * (size_t)(STEM_PR * STEM_PC * 16 + STEM_TR * STEM_TC * 64 * MB);
    constexpr size_t lds_red = 2 * (size_t)(MB * 6 * 16) * 64 * sizeof(float);
    constexpr size_t lds = (lds_tiles > lds_red ? lds_tiles : lds_red) + (BN ? 5 * 32 * MB * sizeof(float) : 0);
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad<MB, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    long long grid = (long long)num_cus() * (MB == 1 ? 2 : 1);
    if (grid > p.ntiles) grid = p.ntiles;
    hipLaunchKernelGGL((k_stem_wgrad<MB, BN>), dim3((unsigned)grid), dim3(256), lds, s, p);
    AY_CHECK_LAUNCH("k_stem_wgrad");
    return AYOLO_OK;
}

// More than 32 output channels (YOLOv5m / l: 48 / 64): one launch per 32-channel slice.  The 64-channel instance
// (k_stem_wgrad<2>: 192 accumulators next to three register-staged tile sets) does not fit the register file -- the compiler
// parks accumulators in scratch BETWEEN the MFMAs of the tile loop (896 bytes per lane): 2.2 ms for YOLOv5l's stem at batch
// 32, against 2 x ~0.12 ms for two slices that each re-read the (small) image.
template <bool BN>
static int launch_stem_wgrad_sliced(StemWP p, hipStream_t s) {
    const int total = p.N;
    p.NS = total;
    for (int c0 = 0; c0 < total; c0 += 32) {
        StemWP q = p;
        q.N = total - c0 < 32 ? total - c0 : 32;
        q.dy = p.dy + c0; q.y_bytes = p.y_bytes - (unsigned)c0 * 2u;
        q.dw = p.dw + (size_t)c0 * p.K;
        if constexpr (BN) {
            q.z = p.z + c0; q.z_bytes = p.z_bytes - (unsigned)c0 * 2u;
            q.mean = p.mean + c0; q.invstd = p.invstd + c0;
            q.gamma = p.gamma ? p.gamma + c0 : nullptr; q.beta = p.beta ? p.beta + c0 : nullptr;
            q.sums = p.sums + c0;
            q.dgamma = p.dgamma ? p.dgamma + c0 : nullptr; q.dbeta = p.dbeta ? p.dbeta + c0 : nullptr;
        }
        const int rc = launch_stem_wgrad<1, BN>(q, s);
        if (rc) return rc;
    }
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// k_stem_fwd: training forward of the packed stem conv (see k_stem_wgrad for the layout), fp16, Cout <= 64, plain store +
// BatchNorm statistics.  The generic k_gconv reaches this layer's 8-channel taps as 16-byte DMA gathers -- 4.5 32-deep steps
// of 5 pieces per wave for 18 MFMAs -- and ran it at 2.4 TB/s.  Here the same input patch as in k_stem_wgrad is staged
// once per 4 x 64 output tile; the 144 reduction values of an output pixel are 18 runs of 8 halves -- run (dh, tap) is the 16
// bytes at patch[2r + dh][c + tap] -- i.e. every B fragment of the 32 x 32 x 16 MFMA is ONE ds_read_b128 (nine 16-deep slices,
// no junk columns), and the weight fragments (9 x 16 bytes per lane and 32 output channels) are loaded once per kernel and stay
// in registers.  Statistics are taken from the rounded values in registers across all tiles of the workgroup.
// ---------------------------------------------------------------------------------------------------
struct StemFP {
    const half_t* x; const half_t* w; half_t* y; double* stats;
    int B, H, WP, Ho, Wo, ldy, N, ldw, stat_reps;
    int tw, th;
    long long ntiles;
    unsigned x_bytes, y_bytes;
    const float* scale; const float* shift; int act;      // AFF: y = act(conv * scale + shift) (the inference executor's fused stem)
};

// AFF: the inference epilogue of k_gconv's EM = 2 (per-channel scale / shift = folded BatchNorm + bias, optional SiLU) instead of
// the plain store + statistics of the training forward -- the same arithmetic on the fp32 accumulator, so results match the
// generic kernel bit for bit
template <int MB, bool AFF = false>
__global__ __launch_bounds__(256, (MB == 1 ? 2 : 1)) void k_stem_fwd(StemFP p) {
    constexpr int PATCH_B = STEM_PR * STEM_PC * 16;
    constexpr int NPCH = STEM_PR * STEM_PC, PPT = (NPCH + 255) / 256;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];           // [2][patch], then [2][32 * MB] doubles
    const int tid = threadIdx.x, lane = tid & 63, hsel = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // weight fragments: lane (co = lane & 31, k group = lane >> 5) of slice kk holds w[co][(2 * kk + group) * 8 .. + 8]
    half8 fw[MB][9];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int co = m * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) {
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
            if (co < p.N) v = *reinterpret_cast<const half8*>(p.w + (long long)co * p.ldw + (2 * kk + hsel) * 8);
            fw[m][kk] = v;
        }
    }
    // patch offset of this lane's run in slice kk: run = 2 * kk + group = dh * 3 + tap
    int roff[9];
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
        const int run = 2 * kk + hsel;
        roff[kk] = ((run / 3) * STEM_PC + (run % 3)) * 16;
    }
    float ssum[MB][16], ssq[MB][16];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ssum[m][r] = 0.0f; ssq[m][r] = 0.0f; }
    float* aff = reinterpret_cast<float*>(sm + 2 * PATCH_B);               // AFF: [scale | shift] of the 32 * MB channels
    if constexpr (AFF) {
        for (int c = tid; c < 32 * MB; c += 256) {
            const bool in = c < p.N;
            aff[c] = (in && p.scale) ? p.scale[c] : 1.0f;
            aff[32 * MB + c] = (in && p.shift) ? p.shift[c] : 0.0f;
        }
        __syncthreads();
    }

    uint4 rp[2][PPT];
    auto fetch = [&](long long t, uint4 (&fp)[PPT]) {
        const bool live = t < p.ntiles;
        const long long tt = live ? t : 0;
        const int tx = (int)(tt % p.tw);
        const long long u = tt / p.tw;
        const int ty = (int)(u % p.th), n = (int)(u / p.th);
        const int oh0 = ty * STEM_TR, ow0 = tx * STEM_TC;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = tid + 256 * j;
            const int row = i / STEM_PC, col = i - row * STEM_PC;
            const int ih = 2 * oh0 - 2 + row, ip = ow0 - 1 + col;
            const bool ok = live && i < NPCH && (unsigned)ih < (unsigned)p.H && (unsigned)ip < (unsigned)p.WP;
            const unsigned off = ok ? (unsigned)((((unsigned)n * (unsigned)p.H + (unsigned)ih) * (unsigned)p.WP + (unsigned)ip) * 16u) : G_OOB;
            fp[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
        }
    };
    auto stash = [&](int buf, const uint4 (&fp)[PPT]) {
        unsigned char* b = sm + buf * PATCH_B;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = tid + 256 * j;
            if (i < NPCH) *reinterpret_cast<uint4*>(b + i * 16) = fp[j];
        }
    };
    auto compute = [&](int buf, long long t) {
        const unsigned char* pb = sm + buf * PATCH_B + (2 * wave) * (STEM_PC * 16);
        const int tx = (int)(t % p.tw);
        const long long u = t / p.tw;
        const int ty = (int)(u % p.th), n = (int)(u / p.th);
        const int oh = ty * STEM_TR + wave;
#pragma unroll
        for (int blk = 0; blk < STEM_TC / 32; ++blk) {
            const int c = blk * 32 + (lane & 31);                  // this lane's output column inside the tile
            float16v acc[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
            half8 fb[9];
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) fb[kk] = *reinterpret_cast<const half8*>(pb + c * 16 + roff[kk]);
#pragma unroll
            for (int kk = 0; kk < 9; ++kk)
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[m][kk], fb[kk], acc[m], 0, 0, 0);
            // acc[m][r]: channel m * 32 + 8 * (r >> 2) + 4 * hsel + (r & 3) of pixel (oh, ow)
            const int ow = tx * STEM_TC + c;
            const bool pv = oh < p.Ho && ow < p.Wo;
            const unsigned yo = (((unsigned)n * (unsigned)p.Ho + (unsigned)oh) * (unsigned)p.Wo + (unsigned)ow) * (unsigned)p.ldy * 2u;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float v[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[g][e] = acc[m][g * 4 + e];
                        if constexpr (AFF) {
                            // channel of this register before the lane swap: m * 32 + 8 * g + 4 * hsel + e
                            const int ch0 = m * 32 + 8 * g + 4 * hsel + e;
                            const float u = v[g][e] * aff[ch0] + aff[32 * MB + ch0];
                            v[g][e] = p.act ? silu_e<half_t>(u) : u;
                        } else {
                            const float q = pv ? (float)(half_t)v[g][e] : 0.0f;
                            ssum[m][g * 4 + e] += q;
                            ssq[m][g * 4 + e] += q * q;
                        }
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const v2u32 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * j][e]), __float_as_uint(v[2 * j + 1][e]), false, false);
                        v[2 * j][e] = __uint_as_float(sw[0]);
                        v[2 * j + 1][e] = __uint_as_float(sw[1]);
                    }
                    const int ch = m * 32 + 8 * (2 * j + hsel);     // 8 channels: v[2j][0..3], v[2j+1][0..3]
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = (half_t)v[2 * j][e]; h[4 + e] = (half_t)v[2 * j + 1][e]; }
                    const unsigned off = (pv && ch < p.N) ? yo + (unsigned)ch * 2u : G_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, h), rsY, off, 0, 0);
                }
            }
        }
    };

    const long long step = gridDim.x;
    long long t = blockIdx.x;
    if (t >= p.ntiles) return;
    fetch(t, rp[0]);
    stash(0, rp[0]);
    fetch(t + step, rp[1]);
    __syncthreads();
    for (; t < p.ntiles; t += 2 * step) {
        fetch(t + 2 * step, rp[0]);
        compute(0, t);
        stash(1, rp[1]);
        __syncthreads();
        if (t + step >= p.ntiles) break;
        fetch(t + 3 * step, rp[1]);
        compute(1, t + step);
        stash(0, rp[0]);
        __syncthreads();
    }
    // ---- statistics: lane sums -> channel sums (DPP rows + one exchange), workgroup total in LDS (fp64), one replica slot
    if (!AFF && p.stats) {
        double* sst = reinterpret_cast<double*>(sm);              // [2][32 * MB]
        for (int i = tid; i < 2 * 32 * MB; i += 256) sst[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = row16_sum(ssum[m][r]), b = row16_sum(ssq[m][r]);
                a += __shfl_xor(a, 16);
                b += __shfl_xor(b, 16);
                if ((lane & 31) == 0) {
                    const int ch = m * 32 + 8 * (r >> 2) + 4 * hsel + (r & 3);
                    atomicAdd(&sst[ch], (double)a);
                    atomicAdd(&sst[32 * MB + ch], (double)b);
                }
            }
        __syncthreads();
        double* st = p.stats + (size_t)(blockIdx.x % (unsigned)p.stat_reps) * 2 * p.N;
        for (int i = tid; i < 32 * MB; i += 256) {
            if (i < p.N) {
                atomicAdd(&st[i], sst[i]);
                atomicAdd(&st[p.N + i], sst[32 * MB + i]);
            }
        }
    }
}

template <int MB, bool AFF>
static int launch_stem_fwd(StemFP p, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)(STEM_PR * STEM_PC * 16) + (AFF ? 2 * 32 * MB * sizeof(float) : 0);
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd<MB, AFF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    long long grid = (long long)num_cus() * 2;
    if (grid > p.ntiles) grid = p.ntiles;
    hipLaunchKernelGGL((k_stem_fwd<MB, AFF>), dim3((unsigned)grid), dim3(256), lds, s, p);
    AY_CHECK_LAUNCH("k_stem_fwd");
    return AYOLO_OK;
}

// the packed stem geometry of functional._Geometry (6 x 3 taps of 8 halves over pixel pairs, stride (2, 1), pad (2, 1)), fp16,
// at most 64 output channels, everything inside the 2 GiB buffer descriptors
static bool is_packed_stem(const ayolo_conv_desc* d) {
    return d->dtype == AYOLO_F16 && d->kh == 6 && d->kw == 3 && d->sh == 2 && d->sw == 1 && d->ph == 2 && d->pw == 1 && d->Cin == 8 &&
           d->ldx == 8 && d->Cout <= 64 && d->Cout % 8 == 0 && d->Ho == (d->H + 4 - 6) / 2 + 1 && d->Wo == d->W &&
           (long long)d->B * d->H * d->W * 16 < (1ll << 31) - 4096 && (long long)d->B * d->Ho * d->Wo * d->ldy * 2 < (1ll << 31) - 4096;
}

static int stem_fwd_dispatch(const ayolo_conv_desc* d, const void* x, const void* w, void* y, double* stats, int stat_reps, int epilogue,
                             const float* scale, const float* shift, hipStream_t s) {
    StemFP q{};
    q.x = (const half_t*)x; q.w = (const half_t*)w; q.y = (half_t*)y; q.stats = stats;
    q.B = d->B; q.H = d->H; q.WP = d->W; q.Ho = d->Ho; q.Wo = d->Wo; q.ldy = d->ldy; q.N = d->Cout; q.ldw = 6 * 3 * 8;
    q.stat_reps = stat_reps > 0 ? stat_reps : 1;
    q.tw = (d->Wo + STEM_TC - 1) / STEM_TC; q.th = (d->Ho + STEM_TR - 1) / STEM_TR;
    q.ntiles = (long long)d->B * q.tw * q.th;
    q.x_bytes = (unsigned)((long long)d->B * d->H * d->W * 16); q.y_bytes = (unsigned)((long long)d->B * d->Ho * d->Wo * d->ldy * 2);
    q.scale = scale; q.shift = shift; q.act = epilogue == AYOLO_EPI_AFFINE_SILU ? 1 : 0;
    if (epilogue != AYOLO_EPI_NONE) return d->Cout <= 32 ? launch_stem_fwd<1, true>(q, s) : launch_stem_fwd<2, true>(q, s);
    return d->Cout <= 32 ? launch_stem_fwd<1, false>(q, s) : launch_stem_fwd<2, false>(q, s);
}

// ---------------------------------------------------------------------------------------------------
// Weight gradients, host side: planning of a (grouped) launch, the C entries.
// ---------------------------------------------------------------------------------------------------
// geometry of one job (a layer, or one batch half of a layer whose tensors exceed the 2 GiB descriptor range)
static int wgrad_fill(const ayolo_conv_desc* d, const void* x, const void* dy, WGradP& p) {
    p = WGradP{};
    p.x = x; p.dy = dy;
    p.B = d->B; p.XH = d->H; p.XW = d->W; p.ldx = d->ldx; p.C = d->Cin;
    p.OH = d->Ho; p.OW = d->Wo; p.ldy = d->ldy; p.N = d->Cout;
    p.sh = d->sh; p.sw = d->sw; p.ntaps = d->kh * d->kw; p.K = p.ntaps * p.C; p.alpha = 1.0f;
    p.P = (long long)d->B * d->Ho * d->Wo;
    for (int i = 0; i < d->kh; ++i)
        for (int j = 0; j < d->kw; ++j) {
            p.dh[i * d->kw + j] = (signed char)(i - d->ph);
            p.dw_[i * d->kw + j] = (signed char)(j - d->pw);
        }
    p.dOW = make_fastdiv((unsigned)p.OW); p.dOH = make_fastdiv((unsigned)p.OH); p.dC = make_fastdiv((unsigned)p.C);
    p.linear = (d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0) ? 1 : 0;
    p.tm = p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128);
    p.gx = (unsigned)((p.K + TNW - 1) / TNW);
    p.gy = (unsigned)((p.N + p.tm - 1) / p.tm);
    p.dy_slot = -1;
    return AYOLO_OK;
}

// one logical layer -> 1 .. n jobs (batch halves until x and dy fit a buffer descriptor), appended to `out`
static int wgrad_halves(const ayolo_conv_desc* d, WGradP p, std::vector<WGradP>& out) {
    const long long es = d->dtype == AYOLO_F16 ? 2 : 4;
    const long long LIM = (1ll << 31) - 4096;
    const long long x_img = (long long)p.XH * p.XW * p.ldx * es, y_img = (long long)p.OH * p.OW * p.ldy * es;
    AY_CHECK_ARG(x_img < LIM && y_img < LIM, "conv_wgrad: a single image of %lld / %lld bytes unsupported", x_img, y_img);
    if (x_img * p.B >= LIM || y_img * p.B >= LIM) {
        AY_CHECK_ARG(p.dy_slot < 0, "conv_wgrad: a dy override cannot be split into batch halves");
        WGradP a = p, b = p;
        a.B = p.B / 2; b.B = p.B - a.B;
        a.P = (long long)a.B * p.OH * p.OW; b.P = (long long)b.B * p.OH * p.OW;
        b.x = (const char*)p.x + x_img * a.B;
        b.dy = (const char*)p.dy + y_img * a.B;
        int rc = wgrad_halves(d, a, out);
        return rc ? rc : wgrad_halves(d, b, out);
    }
    p.x_bytes = (unsigned)(x_img * p.B); p.y_bytes = (unsigned)(y_img * p.B);
    out.push_back(p);
    return AYOLO_OK;
}

// pixel splits of a job for a target item length of `q` 32-pixel steps (at least 4 steps per split, every split non-empty)
static void wgrad_split(WGradP& p, double q) {
    const long long steps = (p.P + 31) / 32;
    long long S = (long long)((double)steps / (q > 1.0 ? q : 1.0) + 0.5);
    const long long smax = (steps + 3) / 4;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    long long chunk = (p.P + S - 1) / S;
    chunk = (chunk + 31) / 32 * 32;
    p.chunk = chunk;
    p.splits = (unsigned)((p.P + chunk - 1) / chunk);
}

// A group table (one contiguous blob; the caller keeps a host copy and a device copy):
//   WGroupHdr | WGradP jobs[njobs] | W3P jobs3[njobs3] | WItem items[class 0] | .. | items[class 3] | WRed red[n_red]
// Classes 0 .. 2: k_wgrad with 32 / 64 / 128 output channels per tile; class 3: the 3x3 layers of k_wgrad3 (their jobs are the
// W3P array; in the introspection entries they are jobs njobs .. njobs + njobs3 - 1).
struct WGroupHdr {
    unsigned magic, dtype, njobs, n_red;
    unsigned n_items[4];               // item count per class, each a multiple of 8
    unsigned off_jobs, off_items[4], off_red;
    unsigned njobs3, off_jobs3, lds3, pad;
    unsigned long long ws_floats, table_bytes;
};
#define WGROUP_MAGIC 0x57475235u

struct WGroupPlan {
    std::vector<WGradP> jobs;
    std::vector<W3P> jobs3;
    std::vector<WItem> items[4];
    size_t lds3 = 0;
    std::vector<WRed> red;
    unsigned long long ws_floats = 0;
    int dtype = AYOLO_F16;
};

// modelled cycles of one k_wgrad3 step of job p (w3_fill's model): the unit item lengths are balanced in
static double w3_step_cost(const W3P& p) { return p.step_cost; }
// Steps per item of a k_wgrad3 job.  `q`: the launch's common item length (in steps of THIS job).  Every row range is one more
// N x K slot to store and to add: a deep layer on a small map (256 -> 256 at 20 x 20: 2.4 MB per slot for 6.5 MB of operands)
// is cut no finer than what keeps that traffic under about a third of the layer's own compute / operand time.
static void w3_split_job(W3P& p, double q) {
    const double nk = (double)p.N * p.K;
    const double px = (double)p.B * p.OH * p.OW;
    const double t_layer = 2.0 * px * nk / 1.0e15 + 2.0 * ((double)p.B * p.XH * p.XW * p.C + px * p.N) / 4.5e12;
    double smax = 0.3 * t_layer * 4.5e12 / (nk * 8.0);
    if (smax < 1.0) smax = 1.0;
    const double steps = (double)p.strips * (double)((p.NU + p.RPS - 1) / p.RPS);
    const double qmin = steps / smax;
    w3_split(p, q > qmin ? q : qmin);
}

static int wgroup_plan(const ayolo_wgrad_job* jj, int njobs, WGroupPlan& g) {
    AY_CHECK_ARG(jj && njobs > 0 && njobs < 4096, "wgrad_group: %d jobs", njobs);
    g.dtype = jj[0].conv.dtype;
    struct Layer { size_t j0, j1; float* dw; float alpha; int overwrite; unsigned ldd; bool w3; };
    std::vector<Layer> layers;
    for (int k = 0; k < njobs; ++k) {
        const ayolo_wgrad_job& a = jj[k];
        int rc = check_desc(&a.conv, "wgrad_group");
        if (rc) return rc;
        AY_CHECK_ARG(a.conv.dtype == g.dtype, "wgrad_group: job %d: mixed dtypes", k);
        AY_CHECK_ARG(a.x && (a.dy || a.dy_slot >= 0) && a.dw && a.dy_slot < 4, "wgrad_group: job %d: null pointer / dy_slot %d", k, a.dy_slot);
        const int ce = a.conv.dtype == AYOLO_F16 ? 8 : 4;
        AY_CHECK_ARG(a.conv.ldy % ce == 0, "wgrad_group: job %d: ldy=%d must be a multiple of %d", k, a.conv.ldy, ce);
        AY_CHECK_ARG(!is_packed_stem(&a.conv), "wgrad_group: job %d is the packed stem (ayolo_conv_wgrad / ayolo_stem_bn_wgrad)", k);
        AY_CHECK_ARG(((uintptr_t)a.dw % 16) == 0 && ((long long)a.conv.Cout * a.conv.kh * a.conv.kw * a.conv.Cin) % 4 == 0,
                     "wgrad_group: job %d: dw must be 16-byte aligned with a multiple of 4 elements", k);
        const int kj = a.conv.kh * a.conv.kw * a.conv.Cin;
        AY_CHECK_ARG(a.dw_ld == 0 || (a.dw_ld >= kj && a.dw_ld % 4 == 0 && kj % 4 == 0), "wgrad_group: job %d: dw_ld=%d for %d columns", k, a.dw_ld, kj);
        W3P p3;
        if (a.dy_slot < 0 && !a.xscale && !a.xshift && w3_fill(&a.conv, a.x, a.dy, p3) == 0) {      // a 3x3 layer of k_wgrad3
            g.jobs3.push_back(p3);
            layers.push_back({g.jobs3.size() - 1, g.jobs3.size(), a.dw, a.alpha, a.overwrite, (unsigned)(a.dw_ld > 0 ? a.dw_ld : kj), true});
            continue;
        }
        WGradP p;
        wgrad_fill(&a.conv, a.x, a.dy, p);
        p.dy_slot = a.dy_slot;
        if (a.xscale || a.xshift) {
            AY_CHECK_ARG(a.xscale && a.xshift && p.linear && a.conv.dtype == AYOLO_F16, "wgrad_group: job %d: transform on load needs fp16, 1x1 / stride 1", k);
            p.xf_scale = a.xscale; p.xf_shift = a.xshift; p.xf_act = a.xact ? 1 : 0;
        }
        const size_t j0 = g.jobs.size();
        rc = wgrad_halves(&a.conv, p, g.jobs);
        if (rc) return rc;
        layers.push_back({j0, g.jobs.size(), a.dw, a.alpha, a.overwrite, (unsigned)(a.dw_ld > 0 ? a.dw_ld : kj), false});
    }
    // ---- item length.  All items of the group together should fill the chip's workgroup slots a few times over (so that the
    // tail of the launch is short against its body) without cutting a layer finer than ~24 (minq) steps per item
    // (prologue + epilogue of a workgroup cost ~10 steps' worth of time; every extra split is one more N x K slot to store and add)
    const int bpc = g.dtype == AYOLO_F16 ? 3 : 1;
    const double slots = (double)num_cus() * bpc;
    const double waves = 3.0, minq = 24.0;    // re-swept on round 5's final code (profiles/r05_ab_wgrad_retune.txt); switches retired in round 6
    double total = 0.0;
    for (const WGradP& p : g.jobs) total += (double)p.gx * p.gy * (double)((p.P + 31) / 32);
    double q = total / (slots * waves);
    if (q < minq) q = minq;
    for (WGradP& p : g.jobs) wgrad_split(p, q);
    // k_wgrad3's launch: two workgroups per CU, items of equal modelled time
    if (!g.jobs3.empty()) {
        const double slots3 = (double)num_cus() * 2.0;
        // (half a round of workgroups -- one per CU: every extra row range is one more slot to store and add and one more
        // epilogue; 3 / 2 / 1 / 0.5 rounds measured +0.12 / +0.07 / 0 / -0.05 ms against one, all inside +-0.06 ms of each other on
        // a second box: profiles/r05_ab_wgrad_groups_waves.txt, r05_ab_wgrad3_waves.txt)
        const double waves3 = 0.5;
        const double minq3 = 6.0;
        double total3 = 0.0;
        for (const W3P& p : g.jobs3) total3 += (double)w3_tiles(p) * p.strips * (double)((p.NU + p.RPS - 1) / p.RPS) * w3_step_cost(p);
        const double t_item = total3 / (slots3 * waves3);
        for (W3P& p : g.jobs3) {
            double q3 = t_item / w3_step_cost(p);
            if (q3 < minq3) q3 = minq3;
            w3_split_job(p, q3);
            const size_t l = w3_lds_bytes(p);
            g.lds3 = l > g.lds3 ? l : g.lds3;
        }
    }
    // ---- workspace slots + reduction blocks, layer by layer
    unsigned long long off = 0;
    for (const Layer& L : layers) {
        unsigned S = 0;
        unsigned long long nk, slotf;
        unsigned cols;
        WRed proto{};
        if (L.w3) {
            W3P& p = g.jobs3[L.j0];
            nk = (unsigned long long)p.N * (unsigned long long)p.K; cols = (unsigned)p.K;
            slotf = w3_slot_floats(p);
            p.ws_off = off; p.zz0 = 0; S = w3_splits(p);
            proto.pC = (unsigned)p.C; proto.pNB = (unsigned)p.NB; proto.pCB = (unsigned)p.CB; proto.ptc = (unsigned)p.tc;
        } else {
            nk = (unsigned long long)g.jobs[L.j0].N * (unsigned long long)g.jobs[L.j0].K; cols = (unsigned)g.jobs[L.j0].K;
            slotf = nk;
            for (size_t j = L.j0; j < L.j1; ++j) { g.jobs[j].ws_off = off; g.jobs[j].zz0 = S; S += g.jobs[j].splits; }
        }
        const unsigned tpc = wred_tpc(S), chunk = wred_chunk(tpc);
        for (unsigned long long e = 0; e < nk; e += chunk) {
            WRed r = proto;
            r.tpc = tpc;
            r.ws_off = proto.pC ? off : off + e; r.stride = slotf; r.dst = L.dw; r.e0 = e; r.n = (unsigned)(nk - e < chunk ? nk - e : chunk); r.S = S;
            r.alpha = L.alpha; r.overwrite = L.overwrite; r.cols = cols; r.ldd = L.ldd;
            g.red.push_back(r);
        }
        off += (unsigned long long)S * slotf;
        off = (off + 63) / 64 * 64;
    }
    g.ws_floats = off;
    // ---- items: per class, (job, split) groups -- the tiles of one pixel split, back to back on one XCD -- dealt longest first
    // to the XCD queue with the least work so far; queues padded to equal length, interleaved block % 8 = XCD
    for (int c = 0; c < 4; ++c) {
        const int tm = 32 << c;
        struct Grp { unsigned job, zz, nt; double cost; };
        std::vector<Grp> grps;
        if (c < 3) {
            for (size_t j = 0; j < g.jobs.size(); ++j) {
                const WGradP& p = g.jobs[j];
                if (p.tm != tm) continue;
                for (unsigned z = 0; z < p.splits; ++z) {
                    const long long pb = (long long)z * p.chunk, pe = pb + p.chunk < p.P ? pb + p.chunk : p.P;
                    grps.push_back({(unsigned)j, z, p.gx * p.gy, (double)((pe - pb + 31) / 32)});
                }
            }
        } else {
            for (size_t j = 0; j < g.jobs3.size(); ++j) {
                const W3P& p = g.jobs3[j];
                const double sc = w3_step_cost(p);
                for (unsigned z = 0; z < w3_splits(p); ++z)
                    grps.push_back({(unsigned)j, z, w3_tiles(p), (double)w3_item_steps(p, z) * sc});
            }
        }
        if (grps.empty()) continue;
        std::stable_sort(grps.begin(), grps.end(), [](const Grp& a, const Grp& b) { return a.cost > b.cost; });
        std::vector<WItem> qs[8];
        double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (const Grp& gr : grps) {
            int best = 0;
            for (int x = 1; x < 8; ++x)
                if (load[x] < load[best]) best = x;
            for (unsigned t = 0; t < gr.nt; ++t) qs[best].push_back({gr.job, t, gr.zz, 0u});
            load[best] += gr.cost * gr.nt;
        }
        size_t len = 0;
        for (int x = 0; x < 8; ++x) len = qs[x].size() > len ? qs[x].size() : len;
        g.items[c].resize(len * 8);
        for (size_t i = 0; i < len; ++i)
            for (int x = 0; x < 8; ++x) g.items[c][i * 8 + x] = i < qs[x].size() ? qs[x][i] : WItem{0xffffffffu, 0u, 0u, 0u};
    }
    return AYOLO_OK;
}

static size_t wgroup_bytes(const WGroupPlan& g, WGroupHdr* h) {
    WGroupHdr t{};
    size_t o = (sizeof(WGroupHdr) + 63) / 64 * 64;
    t.off_jobs = (unsigned)o; o += (g.jobs.size() * sizeof(WGradP) + 63) / 64 * 64;
    t.off_jobs3 = (unsigned)o; o += (g.jobs3.size() * sizeof(W3P) + 63) / 64 * 64;
    for (int c = 0; c < 4; ++c) { t.off_items[c] = (unsigned)o; t.n_items[c] = (unsigned)g.items[c].size(); o += (g.items[c].size() * sizeof(WItem) + 63) / 64 * 64; }
    t.off_red = (unsigned)o; o += (g.red.size() * sizeof(WRed) + 63) / 64 * 64;
    t.magic = WGROUP_MAGIC; t.dtype = (unsigned)g.dtype; t.njobs = (unsigned)g.jobs.size(); t.n_red = (unsigned)g.red.size();
    t.njobs3 = (unsigned)g.jobs3.size(); t.lds3 = (unsigned)g.lds3;
    t.ws_floats = g.ws_floats; t.table_bytes = o;
    if (h) *h = t;
    return o;
}

extern "C" int ayolo_wgrad_group_size(const ayolo_wgrad_job* jobs, int njobs, size_t* table_bytes, size_t* ws_bytes) {
    AY_CHECK_ARG(table_bytes && ws_bytes, "wgrad_group_size: null pointer");
    WGroupPlan g;
    int rc = wgroup_plan(jobs, njobs, g);
    if (rc) return rc;
    *table_bytes = wgroup_bytes(g, nullptr);
    *ws_bytes = (size_t)g.ws_floats * sizeof(float);
    return AYOLO_OK;
}

extern "C" int ayolo_wgrad_group_build(const ayolo_wgrad_job* jobs, int njobs, void* table, size_t table_bytes) {
    AY_CHECK_ARG(table, "wgrad_group_build: null table");
    WGroupPlan g;
    int rc = wgroup_plan(jobs, njobs, g);
    if (rc) return rc;
    WGroupHdr h;
    AY_CHECK_ARG(wgroup_bytes(g, &h) <= table_bytes, "wgrad_group_build: table of %zu bytes, %zu needed", table_bytes, (size_t)h.table_bytes);
    unsigned char* t = (unsigned char*)table;
    memset(t, 0, (size_t)h.table_bytes);
    memcpy(t, &h, sizeof(h));
    if (!g.jobs.empty()) memcpy(t + h.off_jobs, g.jobs.data(), g.jobs.size() * sizeof(WGradP));     // (a group of k_wgrad3 jobs only: no k_wgrad jobs)
    if (!g.jobs3.empty()) memcpy(t + h.off_jobs3, g.jobs3.data(), g.jobs3.size() * sizeof(W3P));
    for (int c = 0; c < 4; ++c)
        if (!g.items[c].empty()) memcpy(t + h.off_items[c], g.items[c].data(), g.items[c].size() * sizeof(WItem));
    memcpy(t + h.off_red, g.red.data(), g.red.size() * sizeof(WRed));
    return AYOLO_OK;
}

// introspection of a group table (tests, tools): header counts and the split geometry of one job
extern "C" int ayolo_wgrad_group_info(const void* table_host, int job, long long* out, int nout) {
    AY_CHECK_ARG(table_host && out && nout >= 12, "wgrad_group_info: out[12]");
    const WGroupHdr& h = *(const WGroupHdr*)table_host;
    AY_CHECK_ARG(h.magic == WGROUP_MAGIC, "wgrad_group_info: not a group table");
    out[0] = h.njobs; out[1] = h.n_items[0]; out[2] = h.n_items[1]; out[3] = h.n_items[2]; out[4] = h.n_red; out[5] = (long long)h.ws_floats;
    for (int k = 6; k < nout; ++k) out[k] = 0;
    if (nout >= 14) { out[12] = h.n_items[3]; out[13] = h.njobs3; }
    if (job >= 0) {
        AY_CHECK_ARG((unsigned)job < h.njobs + h.njobs3, "wgrad_group_info: job %d of %u", job, h.njobs + h.njobs3);
        if ((unsigned)job < h.njobs) {
            const WGradP& p = ((const WGradP*)((const unsigned char*)table_host + h.off_jobs))[job];
            out[6] = p.tm; out[7] = p.gx; out[8] = p.gy; out[9] = p.splits; out[10] = p.chunk; out[11] = p.zz0;
        } else {
            // a k_wgrad3 job: "tile class" 0 (none of 32 / 64 / 128), tiles along C / along N, (strip, row range) splits, virtual
            // rows per item
            const W3P& p = ((const W3P*)((const unsigned char*)table_host + h.off_jobs3))[(unsigned)job - h.njobs];
            out[6] = 0; out[7] = p.tc; out[8] = p.tn; out[9] = w3_splits(p); out[10] = p.uch; out[11] = p.zz0;
            if (nout >= 20) { out[14] = p.TC; out[15] = p.RPS; out[16] = p.strips; out[17] = p.NB; out[18] = p.CB; out[19] = p.stage; }
        }
    }
    return AYOLO_OK;
}
/* item `i` of class `cls` (0: 32, 1: 64, 2: 128 output channels per tile of k_wgrad; 3: k_wgrad3, job numbers continue behind
 * k_wgrad's): out = {job or -1, tile, split} */
extern "C" int ayolo_wgrad_group_item(const void* table_host, int cls, long long i, long long* out) {
    AY_CHECK_ARG(table_host && out && cls >= 0 && cls < 4, "wgrad_group_item: bad args");
    const WGroupHdr& h = *(const WGroupHdr*)table_host;
    AY_CHECK_ARG(h.magic == WGROUP_MAGIC && i >= 0 && i < (long long)h.n_items[cls], "wgrad_group_item: index %lld", i);
    const WItem& it = ((const WItem*)((const unsigned char*)table_host + h.off_items[cls]))[i];
    out[0] = it.job == 0xffffffffu ? -1 : (long long)it.job + (cls == 3 ? (long long)h.njobs : 0); out[1] = it.tile; out[2] = it.zz;
    return AYOLO_OK;
}

template <typename T, int TM, bool XFW>
static int launch_wgrad_k(const WGradP& pv, const WGradP* jobs, const WItem* items, unsigned blocks, float* ws, const WOvr& ovr, hipStream_t s) {
    using W = WT<T, TM>;
    static bool attr_set[16] = {false};
    // (a cap on a group's resident workgroups -- asking for more LDS than the kernel uses, so that the main stream's kernels always
    // find a free slot -- was measured in round 6 and lost: 55 KB +0.05 ms, 82 KB +0.35 ms, profiles/r06_ab_fork_placement_1.txt)
    const size_t lds = W::LDS;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad<T, TM, XFW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)W::LDS);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_wgrad<T, TM, XFW>), dim3(blocks), dim3(256), lds, s, pv, jobs, items, ws, ovr);
    AY_CHECK_LAUNCH("k_wgrad");
    return AYOLO_OK;
}

static int launch_wgrad_any(int dtype, int tm, bool xfw, const WGradP& pv, const WGradP* jobs, const WItem* items, unsigned blocks, float* ws,
                            const WOvr& ovr, hipStream_t s) {
    if (dtype == AYOLO_F16) {
        if (xfw) {                           // some job of the launch transforms x on load
            if (tm == 32) return launch_wgrad_k<half_t, 32, true>(pv, jobs, items, blocks, ws, ovr, s);
            if (tm == 64) return launch_wgrad_k<half_t, 64, true>(pv, jobs, items, blocks, ws, ovr, s);
            return launch_wgrad_k<half_t, 128, true>(pv, jobs, items, blocks, ws, ovr, s);
        }
        if (tm == 32) return launch_wgrad_k<half_t, 32, false>(pv, jobs, items, blocks, ws, ovr, s);
        if (tm == 64) return launch_wgrad_k<half_t, 64, false>(pv, jobs, items, blocks, ws, ovr, s);
        return launch_wgrad_k<half_t, 128, false>(pv, jobs, items, blocks, ws, ovr, s);
    }
    if (tm == 32) return launch_wgrad_k<float, 32, false>(pv, jobs, items, blocks, ws, ovr, s);
    if (tm == 64) return launch_wgrad_k<float, 64, false>(pv, jobs, items, blocks, ws, ovr, s);
    return launch_wgrad_k<float, 128, false>(pv, jobs, items, blocks, ws, ovr, s);
}

extern "C" int ayolo_wgrad_group_run(const void* table_host, const void* table_dev, void* ws, size_t ws_bytes,
                                     const void* const* dy_override, int n_override, ayolo_stream s) {
    AY_CHECK_ARG(table_host && table_dev && ws, "wgrad_group_run: null pointer");
    const WGroupHdr& h = *(const WGroupHdr*)table_host;
    AY_CHECK_ARG(h.magic == WGROUP_MAGIC, "wgrad_group_run: not a group table");
    AY_CHECK_ARG((size_t)h.ws_floats * 4 <= ws_bytes && ((uintptr_t)ws % 16) == 0, "wgrad_group_run: workspace of %zu bytes, %llu needed (16-byte aligned)",
                 ws_bytes, (unsigned long long)h.ws_floats * 4);
    AY_CHECK_ARG(n_override >= 0 && n_override <= 4 && (n_override == 0 || dy_override), "wgrad_group_run: %d overrides", n_override);
    WOvr ovr{};
    for (int k = 0; k < n_override; ++k) ovr.q[k] = dy_override[k];
    const WGradP* hjobs = (const WGradP*)((const unsigned char*)table_host + h.off_jobs);
    for (unsigned j = 0; j < h.njobs; ++j)
        AY_CHECK_ARG(hjobs[j].dy_slot < n_override && (hjobs[j].dy_slot < 0 || ovr.q[hjobs[j].dy_slot]), "wgrad_group_run: job %u needs dy override %d", j,
                     hjobs[j].dy_slot);
    const unsigned char* td = (const unsigned char*)table_dev;
    const WGradP* djobs = (const WGradP*)(td + h.off_jobs);
    const WGradP none{};
    // (A cap on the group's resident workgroups per CU -- workgroups walking the item list with a grid stride -- was measured in
    // round 4 and removed: 13.58 ms uncapped, 14.45 / 16.3 ms with two / one workgroup per CU, and the loop cost 20+ registers.)
    for (int c = 0; c < 3; ++c) {
        if (!h.n_items[c]) continue;
        bool xfw = false;
        for (unsigned j = 0; j < h.njobs; ++j) xfw = xfw || (hjobs[j].tm == (32 << c) && hjobs[j].xf_scale != nullptr);
        int rc = launch_wgrad_any((int)h.dtype, 32 << c, xfw, none, djobs, (const WItem*)(td + h.off_items[c]), h.n_items[c], (float*)ws, ovr, (hipStream_t)s);
        if (rc) return rc;
    }
    if (h.n_items[3]) {
        const W3P none3{};
        // the row-pitch instantiation all k_wgrad3 jobs of the group share (YOLOv5's stride-1 3x3 layers do), else the generic one
        const W3P* hj3 = (const W3P*)((const unsigned char*)table_host + h.off_jobs3);
        int rp = w3_rp_class(hj3[0]);
        for (unsigned j = 1; j < h.njobs3; ++j) rp = w3_rp_class(hj3[j]) == rp ? rp : 0;
        int rc = w3_launch(none3, (const W3P*)(td + h.off_jobs3), (const WItem*)(td + h.off_items[3]), h.n_items[3], h.lds3, (float*)ws, rp, (hipStream_t)s);
        if (rc) return rc;
    }
    if (h.n_red) {
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(h.n_red), dim3(256), 0, (hipStream_t)s, WRed{}, (const WRed*)(td + h.off_red), (const float*)ws);
        AY_CHECK_LAUNCH("k_wgrad_reduce");
    }
    return AYOLO_OK;
}

static bool is_packed_stem(const ayolo_conv_desc* d);

// single layer: the job travels by value (no table, no copy); split for the chip to itself
static int wgrad_single_plan(const ayolo_conv_desc* d, const void* x, const void* dy, std::vector<WGradP>& jobs, unsigned long long* ws_floats) {
    WGradP p;
    wgrad_fill(d, x, dy, p);
    int rc = wgrad_halves(d, p, jobs);
    if (rc) return rc;
    const int bpc = d->dtype == AYOLO_F16 ? 3 : 1;
    double total = 0.0;
    for (const WGradP& q : jobs) total += (double)q.gx * q.gy * (double)((q.P + 31) / 32);
    double q = total / ((double)num_cus() * bpc * 2.0);
    const double minq = 24.0;
    if (q < minq) q = minq;
    unsigned S = 0;
    for (WGradP& j : jobs) { wgrad_split(j, q); j.ws_off = 0; j.zz0 = S; S += j.splits; }
    *ws_floats = (unsigned long long)S * (unsigned long long)p.N * (unsigned long long)p.K;
    return AYOLO_OK;
}

// single 3x3 layer on k_wgrad3: items for the chip to itself (two workgroups per CU, about two rounds of them)
static bool w3_single_plan(const ayolo_conv_desc* d, const void* x, const void* dy, W3P& p) {
    if (w3_fill(d, x, dy, p) != 0) return false;
    const double steps = (double)p.strips * (double)((p.NU + p.RPS - 1) / p.RPS);
    const double want = (double)num_cus() * 2.0 * 2.0 / (double)w3_tiles(p);                    // row ranges per tile
    double q = steps / (want < 1.0 ? 1.0 : want);
    const double minq = 6.0;
    w3_split_job(p, q < minq ? minq : q);
    p.ws_off = 0; p.zz0 = 0;
    return true;
}

extern "C" size_t ayolo_conv_wgrad_workspace(const ayolo_conv_desc* d) {
    if (!d || check_desc(d, "conv_wgrad_workspace") != AYOLO_OK || is_packed_stem(d)) return 0;
    {
        W3P p3;
        if (w3_single_plan(d, d, d, p3)) return (size_t)w3_splits(p3) * (size_t)w3_slot_floats(p3) * sizeof(float);
    }
    std::vector<WGradP> jobs;
    unsigned long long wf = 0;
    if (wgrad_single_plan(d, d, d, jobs, &wf) != AYOLO_OK) return 0;      // pointers are not looked at by the sizing
    return (size_t)wf * sizeof(float);
}

extern "C" int ayolo_conv_wgrad(const ayolo_conv_desc* d, const void* x, const void* dy, float* dw, float alpha, void* ws, size_t ws_bytes,
                                ayolo_stream s) {
    int rc = check_desc(d, "conv_wgrad");
    if (rc) return rc;
    AY_CHECK_ARG(x && dy && dw, "conv_wgrad: null pointer");
    const int ce = d->dtype == AYOLO_F16 ? 8 : 4;
    AY_CHECK_ARG(d->ldy % ce == 0, "conv_wgrad: ldy=%d must be a multiple of %d", d->ldy, ce);
    if (is_packed_stem(d)) {         // the packed stem (see k_stem_wgrad): one set of atomics per workgroup, no workspace
        StemWP q{};
        q.x = (const half_t*)x; q.dy = (const half_t*)dy; q.dw = dw;
        q.B = d->B; q.H = d->H; q.WP = d->W; q.Ho = d->Ho; q.Wo = d->Wo; q.ldy = d->ldy; q.N = d->Cout; q.K = d->kh * d->kw * d->Cin; q.alpha = alpha;
        q.tw = (d->Wo + STEM_TC - 1) / STEM_TC; q.th = (d->Ho + STEM_TR - 1) / STEM_TR;
        q.ntiles = (long long)d->B * q.tw * q.th;
        q.x_bytes = (unsigned)((long long)d->B * d->H * d->W * 16); q.y_bytes = (unsigned)((long long)d->B * d->Ho * d->Wo * d->ldy * 2);
        return launch_stem_wgrad_sliced<false>(q, (hipStream_t)s);
    }
    std::vector<WGradP> jobs;
    unsigned long long wf = 0;
    W3P p3;
    const bool use3 = w3_single_plan(d, x, dy, p3);
    if (use3) wf = (unsigned long long)w3_splits(p3) * w3_slot_floats(p3);
    else {
        rc = wgrad_single_plan(d, x, dy, jobs, &wf);
        if (rc) return rc;
    }
    AY_CHECK_ARG(ws && ((uintptr_t)ws % 16) == 0 && (size_t)wf * 4 <= ws_bytes,
                 "conv_wgrad: split-K workspace of %zu bytes, %llu needed (ayolo_conv_wgrad_workspace; 16-byte aligned)", ws_bytes, wf * 4);
    const unsigned long long nk = (unsigned long long)d->Cout * (unsigned long long)(d->kh * d->kw * d->Cin);
    AY_CHECK_ARG(((uintptr_t)dw % 16) == 0 && nk % 4 == 0, "conv_wgrad: dw must be 16-byte aligned with a multiple of 4 elements");
    unsigned S = 0;
    const WOvr ovr{};
    if (use3) {
        S = w3_splits(p3);
        const long long blocks = (long long)w3_tiles(p3) * ((S + 7) / 8 * 8);
        AY_CHECK_ARG(blocks < (1ll << 31), "conv_wgrad: grid of %lld workgroups", blocks);
        rc = w3_launch(p3, nullptr, nullptr, (unsigned)blocks, w3_lds_bytes(p3), (float*)ws, w3_rp_class(p3), (hipStream_t)s);
        if (rc) return rc;
    }
    for (const WGradP& j : jobs) {
        const long long blocks = (long long)j.gx * j.gy * ((j.splits + 7) / 8 * 8);
        AY_CHECK_ARG(blocks < (1ll << 31), "conv_wgrad: grid of %lld workgroups", blocks);
        rc = launch_wgrad_any(d->dtype, j.tm, false, j, nullptr, nullptr, (unsigned)blocks, (float*)ws, ovr, (hipStream_t)s);
        if (rc) return rc;
        S += j.splits;
    }
    WRed r{};
    r.ws_off = 0; r.stride = nk; r.dst = dw; r.S = S; r.alpha = alpha; r.overwrite = 0; r.cols = r.ldd = (unsigned)(d->kh * d->kw * d->Cin); r.e0 = 0;
    if (use3) { r.stride = w3_slot_floats(p3); r.pC = (unsigned)p3.C; r.pNB = (unsigned)p3.NB; r.pCB = (unsigned)p3.CB; r.ptc = (unsigned)p3.tc; }
    AY_CHECK_ARG(nk < (1ull << 32), "conv_wgrad: dw of %llu elements", nk);
    r.n = (unsigned)nk;
    r.tpc = wred_tpc(S);
    const unsigned rchunk = wred_chunk(r.tpc);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((nk + rchunk - 1) / rchunk)), dim3(256), 0, (hipStream_t)s, r, (const WRed*)nullptr, (const float*)ws);
    AY_CHECK_LAUNCH("k_wgrad_reduce");
    return AYOLO_OK;
}

// Stem block backward in ONE launch (see k_stem_wgrad<MB, true>): BatchNorm + activation backward of the stem's output
// gradient and the weight gradient of its conv.  `d` is the packed-stem descriptor with ldy = row stride of da.
extern "C" int ayolo_stem_bn_wgrad(const ayolo_conv_desc* d, const void* x, const void* z, int ldz, const void* da,
                                   const float* save_mean, const float* save_invstd, const float* gamma, const float* beta,
                                   int act, const double* sums, int sum_reps, float* dw, float* dgamma, float* dbeta,
                                   float alpha, float grad_scale, ayolo_stream s) {
    int rc = check_desc(d, "stem_bn_wgrad");
    if (rc) return rc;
    AY_CHECK_ARG(x && z && da && dw && save_mean && save_invstd && sums && sum_reps >= 1, "stem_bn_wgrad: null pointer");
    AY_CHECK_ARG(is_packed_stem(d) && ldz % 8 == 0 && ldz >= d->Cout && d->ldy % 8 == 0 &&
                 (long long)d->B * d->Ho * d->Wo * ldz * 2 < (1ll << 31) - 4096,
                 "stem_bn_wgrad: needs the packed stem geometry (fp16, 6x3 pair taps, <= 64 output channels, < 2 GiB tensors)");
    StemWP q{};
    q.x = (const half_t*)x; q.dy = (const half_t*)da; q.dw = dw;
    q.B = d->B; q.H = d->H; q.WP = d->W; q.Ho = d->Ho; q.Wo = d->Wo; q.ldy = d->ldy; q.N = d->Cout; q.K = 6 * 3 * 8; q.alpha = alpha;
    q.tw = (d->Wo + STEM_TC - 1) / STEM_TC; q.th = (d->Ho + STEM_TR - 1) / STEM_TR;
    q.ntiles = (long long)d->B * q.tw * q.th;
    q.x_bytes = (unsigned)((long long)d->B * d->H * d->W * 16); q.y_bytes = (unsigned)((long long)d->B * d->Ho * d->Wo * d->ldy * 2);
    q.z = (const half_t*)z; q.ldz = ldz; q.z_bytes = (unsigned)((long long)d->B * d->Ho * d->Wo * ldz * 2);
    q.mean = save_mean; q.invstd = save_invstd; q.gamma = gamma; q.beta = beta; q.sums = sums; q.reps = sum_reps; q.act = act ? 1 : 0;
    q.dgamma = dgamma; q.dbeta = dbeta; q.grad_scale = grad_scale;
    return launch_stem_wgrad_sliced<true>(q, (hipStream_t)s);
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

// all layers of a model in ONE launch: blockIdx.y = job (table in device memory, written once when the plan is built)
template <typename T>
__global__ __launch_bounds__(256) void k_cast_weights(const ayolo_cast_job* jobs) {
    const ayolo_cast_job J = jobs[blockIdx.y];
    const unsigned tot = (unsigned)J.Cout_pad * (unsigned)J.taps * (unsigned)J.Cin_pad;
    T* w = reinterpret_cast<T*>(J.w);
    T* wt = reinterpret_cast<T*>(J.wt);
    for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < tot; t += gridDim.x * 256) {
        const unsigned c = t % (unsigned)J.Cin_pad, r = t / (unsigned)J.Cin_pad;
        const unsigned tap = r % (unsigned)J.taps, co = r / (unsigned)J.taps;
        const float v = (c < (unsigned)J.Cin && co < (unsigned)J.Cout) ? J.w32[((size_t)co * J.taps + tap) * J.Cin + c] : 0.0f;
        if (w) w[t] = (T)v;
        if (wt) wt[((size_t)c * J.taps + tap) * (J.wt_ld > 0 ? J.wt_ld : J.Cout_pad) + co] = (T)v;
    }
}

// The same through 64 x 64 (output channel x input channel) tiles of one tap: rows of w32 are read and rows of w written along the
// input channels, and the TRANSPOSED copy leaves through an LDS transpose so that its rows (along the output channels) are written
// contiguously too.  The per-element kernel above writes wt with a lane stride of taps * Cout elements (one 2-byte store per
// cache line: 84 us for YOLOv5s's 7.2 M parameters, 0.69 TB/s) and pays two integer divisions per element.
template <typename T>
__global__ __launch_bounds__(256) void k_cast_weights_t(const ayolo_cast_job* jobs) {
    const ayolo_cast_job J = jobs[blockIdx.y];
    constexpr int TS = 64;
    __shared__ T tile[TS][TS + 2];
    const int nco = (J.Cout_pad + TS - 1) / TS, nc = (J.Cin_pad + TS - 1) / TS;
    const int ntile = nco * nc * J.taps;
    T* w = reinterpret_cast<T*>(J.w);
    T* wt = reinterpret_cast<T*>(J.wt);
    const int wt_ld = J.wt_ld > 0 ? J.wt_ld : J.Cout_pad;
    const int tr = threadIdx.x >> 4, tc4 = (threadIdx.x & 15) * 4;
    for (int tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
        const int tap = tl % J.taps;
        const int rest = tl / J.taps;
        const int c0 = (rest % nc) * TS, co0 = (rest / nc) * TS;
        // interior tiles of layers whose rows are 16-byte aligned in all three arrays: one 16-byte load and one 4-element store
        // per thread and row instead of four scalar loads and four 2-byte stores (YOLOv5l: 372 MB in 249 us = 1.5 TB/s before)
        const bool vec = sizeof(T) == 2 && (J.Cin & 3) == 0 && (J.Cin_pad & 3) == 0 && (wt_ld & 3) == 0 && c0 + TS <= J.Cin &&
                         co0 + TS <= J.Cout && (((uintptr_t)J.w32 | (uintptr_t)w | (uintptr_t)wt) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + tr + 16 * i;
                const float4 v = *reinterpret_cast<const float4*>(J.w32 + ((size_t)co * J.taps + tap) * J.Cin + c0 + tc4);
                T h[4] = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[tr + 16 * i][tc4 + j] = h[j];
                if (w) *reinterpret_cast<uint2*>(w + ((size_t)co * J.taps + tap) * J.Cin_pad + c0 + tc4) = *reinterpret_cast<const uint2*>(h);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + tr + 16 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + tc4 + j;
                    const float v = (c < J.Cin && co < J.Cout) ? J.w32[((size_t)co * J.taps + tap) * J.Cin + c] : 0.0f;
                    tile[tr + 16 * i][tc4 + j] = (T)v;
                    if (w && co < J.Cout_pad && c < J.Cin_pad) w[((size_t)co * J.taps + tap) * J.Cin_pad + c] = (T)v;
                }
            }
        }
        __syncthreads();
        if (wt) {
            if (vec) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + tr + 16 * i;
                    T h[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = tile[tc4 + j][tr + 16 * i];
                    *reinterpret_cast<uint2*>(wt + ((size_t)c * J.taps + tap) * wt_ld + co0 + tc4) = *reinterpret_cast<const uint2*>(h);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + tr + 16 * i;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = co0 + tc4 + j;
                        if (c < J.Cin_pad && co < J.Cout_pad) wt[((size_t)c * J.taps + tap) * wt_ld + co] = tile[tc4 + j][tr + 16 * i];
                    }
                }
            }
        }
        __syncthreads();
    }
}

extern "C" int ayolo_cast_weights(const ayolo_cast_job* jobs_dev, int njobs, int dtype, ayolo_stream s) {
    AY_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535, "cast_weights: njobs=%d", njobs);
    dim3 grid(48, (unsigned)njobs);              // up to 48 workgroups walk a layer's tiles (the 512 x 512 x 9 weight has 576)
    if (dtype == AYOLO_F16) hipLaunchKernelGGL(k_cast_weights_t<half_t>, grid, dim3(256), 0, (hipStream_t)s, jobs_dev);
    else hipLaunchKernelGGL(k_cast_weights_t<float>, grid, dim3(256), 0, (hipStream_t)s, jobs_dev);
    AY_CHECK_LAUNCH("k_cast_weights");
    return AYOLO_OK;
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
