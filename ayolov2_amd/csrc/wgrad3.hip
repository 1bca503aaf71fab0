// k_wgrad3: weight gradient of the 3x3 / pad 1 convs (stride 1 and 2), fp16 operands, on a once-staged input patch.
//
//   dw[n][(dh, dw) * C + c] = sum over output pixels of dy[pixel][n] * x[pixel * s + (dh, dw) - 1][c]
//
// Replaces, for these layers, the autograd weight-gradient launches behind scripts/train/yolo_trainer.py:329 (the 3x3 Conv rows of
// res/configs/model/yolov5s.yaml:22,25,28,31,46,50 and the Bottleneck 3x3 convs inside every C3).
//
// The generic k_wgrad (conv.hip) treats the nine taps as nine gathers: a 32-pixel step DMAs the x tile once per tap and the dy
// tile once per 128-column tile of dw, 3-6 LDS-DMA pieces next to 4-8 MFMAs, and a 3x3 layer runs at 250-350 TFLOP/s whatever
// its size (profiles/r03_conv_layer_sweep*.txt).  Here a step stages the input PATCH of RPS output rows x TC columns once -- the
// padded input rows as they lie, 64 bytes (one 32-channel block) per pixel -- and every tap is the SAME patch read at a shifted
// LDS address by the transposing fragment read (ds_read_b64_tr_b16: each lane supplies its own pixel-row address, so a stride-2
// pixel walk and a two-row step cost nothing).  A wavefront owns one (32 output channels) x (32 input channels) block of dw for
// ALL nine taps -- 9 accumulator blocks, 144 registers -- so one dy fragment and nine x fragments feed nine MFMAs (2.2 LDS reads
// per MFMA instead of 2.5-3), a step of 80 pixels is 45 MFMAs per wavefront behind ONE barrier, and the DMA traffic per MFMA
// drops from 0.4-0.75 pieces to 0.1-0.25.  Planes of [pixel][32 channels] are 64-byte rows: any four consecutive pixels are 256
// contiguous bytes = every LDS bank once, shifted or not, so the fragment reads are conflict-free without a swizzle (stride 2
// keeps odd and even input columns in separate planes for the same reason).
// Partial sums leave through the split-K workspace of k_wgrad (plain stores, fixed-order k_wgrad_reduce): bit-reproducible.
#include "wgrad3.h"
#include <stdlib.h>
#include <string.h>

typedef __fp16 w3_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef const __attribute__((address_space(4))) W3P* w3job_cptr_t;

#define W3_XOOB 0x40000000u     // a lane offset that is out of range on its own AND on top of any row base (tensors < 1 GiB)
#define W3_MAXI 6               // sub-steps per wavefront (nsub <= 6)
#define W3_STAGE_MAX 64512      // a stage: < 64 KiB (16-bit fragment offsets); <= 40 KiB: two workgroups of two stages per CU
#define W3_TAB_BYTES 9216       // loader tables of the current strip ([ppr + NB * nsub <= 36][64 lanes]), behind the two stages
#define W3_RED_BYTES 28672      // cross-wavefront reduction: up to 7 wavefronts x 16 registers x 64 lanes x 4 bytes per round

// 16 pixels x this lane's channel out of a [pixel][32 channel] plane: lane supplies the row address of pixel (q >> 2) (+ 4 for the
// second read), channels (q & 3) * 4 .. of its 16-channel half, and receives channel q of the 4 + 4 rows (see tr_frag_sw in conv.hip)
__device__ __forceinline__ half8 w3_frag(const unsigned char* lo_, const unsigned char* hi_) {
    const w3_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) w3_fp16x4*)(lo_));
    const w3_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) w3_fp16x4*)(hi_));
    half8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Timeline probe (tools/w3_probe.py; -DAYOLO_PROBE builds only, never the product library): wave 0 of the first 512 workgroups
// records s_memtime at the marks of its step loop
#ifdef AYOLO_PROBE
#define W3_PROBE_N 128
__device__ unsigned long long g_probe3[512 * W3_PROBE_N];
extern "C" int ayolo_probe3_read(void* dst, unsigned long long bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_probe3), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#define W3_MARK()                                                                                      \
    do {                                                                                               \
        if (blockIdx.x < 512 && probe_k < W3_PROBE_N) {                                                \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                \
            if (threadIdx.x == 0) s_probe[probe_k] = t_;                                               \
        }                                                                                              \
        ++probe_k;                                                                                     \
    } while (0)
#else
#define W3_MARK() do { } while (0)
#endif

// RP > 0: the window's row pitch in bytes is this compile-time constant AND the stride is 1 -- the nine tap offsets then are
// immediates of the fragment reads (3 instructions per MFMA instead of 5: the MFMA phase of a wavefront alone on its SIMD is
// issue-bound); RP == 0: any geometry, tap offsets from registers.
template <int RP>
__global__ __launch_bounds__(512, 2) void k_wgrad3(W3P pv, const W3P* jobs, const WItem* items, float* ws) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AYOLO_PROBE
    __shared__ unsigned long long s_probe[W3_PROBE_N];
    int probe_k = 0;
    if (threadIdx.x < W3_PROBE_N) s_probe[threadIdx.x] = 0;
    __syncthreads();
    W3_MARK();                                   // 0: start
#endif

    typedef __attribute__((address_space(4))) const char* kcptr_t;
    w3job_cptr_t pj;
    unsigned tile, zz;
    if (items != nullptr) {
        const WItem it = items[blockIdx.x];
        const unsigned job = (unsigned)__builtin_amdgcn_readfirstlane((int)it.job);
        if (job == 0xffffffffu) return;
        pj = (w3job_cptr_t)(unsigned long long)(jobs + job);
        tile = (unsigned)__builtin_amdgcn_readfirstlane((int)it.tile);
        zz = (unsigned)__builtin_amdgcn_readfirstlane((int)it.zz);
    } else {
        // single job, passed by value: split zz on XCD zz % 8 with its tiles consecutive there (they re-read the same x / dy)
        pj = (w3job_cptr_t)((kcptr_t)__builtin_amdgcn_kernarg_segment_ptr());
        const unsigned Lb = blockIdx.x, xcd = Lb & 7u, local = Lb >> 3;
        const unsigned ntile = (unsigned)(pj->tn * pj->tc);
        tile = local % ntile;
        zz = (local / ntile) * 8u + xcd;
        if (zz >= pj->uranges) return;
    }
#define p (*pj)
#define W3FD(f_) FastDiv{pj->f_.m, pj->f_.s1, pj->f_.s2}
    // Everything the step loop needs lives in registers from here on.  Left to itself hipcc RE-LOADS fields of the job (constant
    // address space) inside the loader's row loop -- an s_load + s_waitcnt lgkmcnt(0) round trip per window row: the first probe
    // of this kernel showed 3 200 cycles of "issue" per step for seven DMA pieces per wavefront (profiles/r05_w3_probe_v1.txt).
    // The asm statements pin the values (no rematerialisation from memory); what does not fit the SGPR file is spilled to VGPR lanes.
    int s = p.s, TC = p.TC, RPS = p.RPS, PX = p.PX, nsub = p.nsub;
    int NB = p.NB, CB = p.CB, NP = p.NP, SL = p.SL;
    int nrows = p.nrows, ppr = p.ppr, rowpitch = p.rowpitch, plo = p.plo, ple = p.ple, xstage = p.xstage;
    unsigned XP = (unsigned)p.XP, UP = (unsigned)p.UP, XH = (unsigned)p.XH, OH = (unsigned)p.OH, Bn = (unsigned)p.B;
    unsigned xrowb = (unsigned)p.XW * (unsigned)p.ldx * 2u, yrowb = (unsigned)p.OW * (unsigned)p.ldy * 2u, ypixb = (unsigned)p.ldy * 2u;
    unsigned stage = (unsigned)p.stage;
    int nstrips = p.strips;
    const FastDiv fXP = W3FD(dXP), fUP = W3FD(dUP);
    unsigned fxm = fXP.m, fxs1 = fXP.s1, fxs2 = fXP.s2, fum = fUP.m, fus1 = fUP.s1, fus2 = fUP.s2;
    asm volatile("" : "+s"(s), "+s"(RPS), "+s"(nsub), "+s"(nrows), "+s"(ppr), "+s"(rowpitch), "+s"(xstage), "+s"(stage));
    asm volatile("" : "+s"(XP), "+s"(UP), "+s"(XH), "+s"(OH), "+s"(Bn), "+s"(xrowb), "+s"(yrowb), "+s"(ypixb), "+s"(nstrips));
    asm volatile("" : "+s"(fxm), "+s"(fxs1), "+s"(fxs2), "+s"(fum), "+s"(fus1), "+s"(fus2));
    const int tni = (int)(tile / (unsigned)p.tc), tci = (int)(tile - (unsigned)tni * (unsigned)p.tc);
    const int nb0 = tni * NB, cb0 = tci * CB;
    // item = (tile, row range zz): the virtual output rows [u0, u1) of EVERY column strip, strip after strip
    unsigned u0 = zz * p.uch;
    unsigned u1 = u0 + p.uch < p.NU ? u0 + p.uch : p.NU;
    const int nsteps = (int)((u1 - u0 + (unsigned)RPS - 1) / (unsigned)RPS);
    asm volatile("" : "+s"(u0), "+s"(u1));
    // this wavefront: block `pair` of the tile, slice `slice` of the sub-steps (EIGHT wavefronts: SL = 8 / NP of them share a block)
    const int pair = wave & (NP - 1);
    const int slice = NP == 1 ? wave : (NP == 2 ? wave >> 1 : wave >> 2);
    const int nb = CB == 2 ? pair >> 1 : pair, cb = CB == 2 ? pair & 1 : 0;

    const v4i32 rsX = make_srd(p.x, p.x_bytes), rsY = make_srd(p.dy, p.y_bytes);
    const unsigned lds_tiles = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)smem_raw);
    const int XW = p.XW, OW = p.OW, Cc = p.C, Nn = p.N, ldx = p.ldx, ldy = p.ldy;
    const FastDiv fTC = W3FD(dTC);

    // ---- loader tables of column strip c0, shared by the eight wavefronts, in LDS behind the two stages.
    // XT[j][l] (j < ppr): byte offset (inside an input row) of the 16-byte chunk ci = j * 64 + l of the window row image
    // [c-block][odd / all columns | even columns][pixel][4 chunks] -- what lane l of piece j of ANY row fetches;
    // DT[e][l] (e < NB * nsub): offset of chunk (l & 3) of pixel sub * 16 + (l >> 2) of the step inside its dy row, the pixel's row
    // (pp / TC) in the low bits -- what lane l of dy piece e = [n-block][sub-step] fetches.
    unsigned* ltab = reinterpret_cast<unsigned*>(smem_raw + 2 * (unsigned)p.stage);
    const int ndy = NB * nsub;
    auto setup_strip = [&](int c0) __attribute__((always_inline)) {
        const int cbsz = (plo + ple) >> 4;               // chunks per c-block
        for (int idx = tid; idx < (ppr + ndy) * 64; idx += 512) {
            const int e = idx >> 6, l = idx & 63;
            unsigned v;
            if (e < ppr) {
                const int ci = e * 64 + l;
                const int cbi = ci >= cbsz ? 1 : 0;
                const int rem = ci - cbi * cbsz;
                const bool even = rem >= (plo >> 4);             // stride 2 only (ple == 0 otherwise: rem < plo / 16 for every live chunk)
                const int rem2 = even ? rem - (plo >> 4) : rem;
                const int q = rem2 >> 2, ch = rem2 & 3;
                const int ic = s == 1 ? c0 - 1 + q : (even ? 2 * (c0 + q) : 2 * (c0 + q) - 1);
                const int chan = (cb0 + cbi) * 32 + ch * 8;
                const bool ok = (ci < CB * cbsz) & (ic >= 0) & (ic < XW) & (chan < Cc);
                v = ok ? (unsigned)((ic * ldx + chan) * 2) : W3_XOOB;
            } else {
                const int ed = e - ppr;
                int nbk = 0, sub = ed;
                while (sub >= nsub) { sub -= nsub; ++nbk; }
                const unsigned pp = (unsigned)(sub * 16 + (l >> 2));
                const unsigned row = fdiv(pp, fTC);
                const int col = (int)(pp - row * (unsigned)TC);
                const int chan = (nb0 + nbk) * 32 + (l & 3) * 8;
                const bool ok = (pp < (unsigned)PX) & (c0 + col < OW) & (chan < Nn);
                v = ok ? ((unsigned)(((c0 + col) * ldy + chan) * 2) | row) : W3_XOOB;
            }
            ltab[idx] = v;
        }
    };
    // ---- fragment geometry.  Sub-step `sub`, half h of its 16 pixels: this lane's pixel is 4 * (sub * 4 + (lane >> 5) * 2 + h) + rowl
    // (TC % 4 == 0: the four pixels of a read lie in one output row); pixels beyond the step are clamped (their dy is zero, the
    // x they meet only has to be finite).  XO[i]: byte offsets of the two halves' pixels in the window for the wavefront's i-th
    // sub-step, tap (0, 0), this wavefront's c-block (packed: a stage is < 64 KiB).
    const int q16 = lane & 15, rowl = q16 >> 2;
    const unsigned chanb = (unsigned)(((q16 & 3) * 4 + ((lane >> 4) & 1) * 16) * 2);
    unsigned XO[W3_MAXI];
#pragma unroll
    for (int i = 0; i < W3_MAXI; ++i) {
        unsigned o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sub = slice + SL * i;
            unsigned pp = (unsigned)(4 * (sub * 4 + (lane >> 5) * 2 + h) + rowl);
            pp = pp < (unsigned)PX ? pp : (unsigned)(PX - 1);
            const unsigned row = fdiv(pp, fTC);
            o[h] = row * (unsigned)(s * rowpitch) + (pp - row * (unsigned)TC) * 64u + chanb + (unsigned)(cb * (plo + ple));
        }
        XO[i] = o[0] | (o[1] << 16);
    }
    const unsigned DYL = (unsigned)(((lane >> 5) * 8 + rowl) * 64) + chanb + (unsigned)(nb * nsub * 1024);
    // tap (dh, dw) -> byte offset inside the window: immediates when RP > 0
    unsigned tapo[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dh = t / 3, dw = t % 3;
        const int colo = s == 1 ? dw * 64 : (dw == 1 ? plo : (dw == 2 ? 64 : 0));
        tapo[t] = RP > 0 ? (unsigned)(dh * RP + dw * 64) : (unsigned)(dh * rowpitch + colo);
    }

    float16v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // ---- loader cursor: the step whose DMA is issued next.  (xn, xvi): image and padded row of the window's first row
    // V = s * u; (yn, yoh, yu): image, row and virtual row of the step's first output row.  Both advance by a step without a division.
    // (Macros over plain locals, not lambdas: with the mutable cursor captured by reference in nested lambdas hipcc kept the
    // closure in memory -- every uniform counter became a scratch load in divergent control flow.)
    unsigned xn = 0, xvi = 0, yn = 0, yoh = 0, yu = 0;
#define W3_CURSOR_RESET()                                                    \
    {                                                                        \
        const unsigned V0_ = (unsigned)s * u0;                               \
        const unsigned tq_ = __umulhi(fxm, V0_);                             \
        xn = (tq_ + ((V0_ - tq_) >> fxs1)) >> fxs2;                          \
        xvi = V0_ - xn * XP;                                                 \
        const unsigned tu_ = __umulhi(fum, u0);                              \
        yn = (tu_ + ((u0 - tu_) >> fus1)) >> fus2;                           \
        yoh = u0 - yn * UP;                                                  \
        yu = u0;                                                             \
    }
#define W3_CURSOR_STEP()                                                     \
    {                                                                        \
        xvi += (unsigned)(s * RPS);                                          \
        if (xvi >= XP) { xvi -= XP; ++xn; }                                  \
        yoh += (unsigned)RPS; yu += (unsigned)RPS;                           \
        if (yoh >= UP) { yoh -= UP; ++yn; }                                  \
    }
    // ---- DMA of the cursor's step.  What the probes of the first four versions showed (profiles/r05_w3_probe_v*.txt,
    // r05_ldsdma_microbench.txt): an LDS-DMA piece costs ~45 cycles to issue while fewer than ~40 are in flight on the CU; what
    // cost 300-400 cycles per piece was the CODE AROUND IT -- with or without the DMA instruction (v3 experiments), branchy or
    // straight-line and predicated (v4: 33 slots of ~25 instructions = 8 000 cycles) -- because a wavefront alone on its SIMD
    // issues an instruction only every ~8-10 cycles.  So (1) eight wavefronts, two per SIMD: the two slices of a block split its
    // sub-steps AND every wavefront issues an eighth of the pieces, one wavefront's loader code runs under the other's MFMAs;
    // (2) as few instructions per piece as the hardware allows: the per-lane offsets come from the shared LDS tables, the row
    // bases of the whole window are computed ONCE per step, one row per lane (rbv: lane r = window row r; dyv: lane r = dy row r),
    // and a piece picks its own with v_readlane / ds_bpermute.
    // x pieces q = wave, wave + 8, .. of the nrows * ppr pieces [row][j]; dy pieces e = 7 - wave, 15 - wave, .. (dealt from the
    // other end).
    const int nxq = nrows * ppr;
    const int wrev = 7 - wave;
    const FastDiv fPPR = W3FD(dPPR);
    unsigned rbv = 0, dyv = 0, p_x = 0, p_y = 0;
#define W3_ISSUE_BEGIN(sb_)                                                                          \
    {                                                                                                \
        p_x = lds_tiles + (sb_);                                                                     \
        p_y = lds_tiles + (sb_) + (unsigned)xstage;                                                  \
        unsigned vi_ = xvi + (unsigned)lane, n_ = xn;                                                \
        n_ = vi_ >= XP ? n_ + 1u : n_; vi_ = vi_ >= XP ? vi_ - XP : vi_;                             \
        const unsigned ih_ = vi_ - 1u;              /* vi == 0: the top padding row wraps to out of range */   \
        rbv = ((ih_ < XH) & (n_ < Bn) & (lane < nrows)) ? (n_ * XH + ih_) * xrowb : G_OOB;           \
        unsigned oh_ = yoh + (unsigned)lane, n2_ = yn;                                               \
        n2_ = oh_ >= UP ? n2_ + 1u : n2_; oh_ = oh_ >= UP ? oh_ - UP : oh_;                          \
        dyv = ((lane < RPS) & (oh_ < OH) & (yu + (unsigned)lane < u1)) ? (n2_ * OH + oh_) * yrowb : G_OOB;   \
    }
    // This wavefront's pieces of a step: at most four x pieces and three dy pieces (w3_fill keeps nrows * ppr <= 32 and
    // NB * nsub <= 24).  Which window row / piece-in-row each of them is never changes (pr_ / plds_: row and LDS offset, uniform),
    // their table entries change with the strip (pt_ / pd_: W3_PIECES_OF_STRIP) -- all of it out of the step loop: a step's loader
    // is the row bases (one row per lane) + per piece a v_readlane, an add and the DMA (probe v5c: 1 280 cycles of issue per step
    // with this bookkeeping inside the loop, 9-10 cycles per instruction)
    unsigned pr_[4], plds_[4], pt_[4], pd_[3];
#pragma unroll
    for (int k_ = 0; k_ < 4; ++k_) {
        const unsigned q_ = (unsigned)(wave + 8 * k_) < (unsigned)nxq ? (unsigned)(wave + 8 * k_) : 0u;
        pr_[k_] = fdiv(q_, fPPR);
        plds_[k_] = pr_[k_] * (unsigned)rowpitch + (q_ - pr_[k_] * (unsigned)ppr) * 1024u;
        pt_[k_] = 0;
    }
#pragma unroll
    for (int k_ = 0; k_ < 3; ++k_) pd_[k_] = 0;
#define W3_PIECES_OF_STRIP()                                                                         \
    {                                                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                           \
            const unsigned q_ = (unsigned)(wave + 8 * k_) < (unsigned)nxq ? (unsigned)(wave + 8 * k_) : 0u;   \
            pt_[k_] = ltab[(q_ - pr_[k_] * (unsigned)ppr) * 64u + (unsigned)lane];                   \
        }                                                                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                             \
            pd_[k_] = ltab[(unsigned)(ppr + (wrev + 8 * k_ < ndy ? wrev + 8 * k_ : 0)) * 64u + (unsigned)lane];   \
    }
#define W3_ISSUE_ALL()                                                                               \
    {                                                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_)                                             \
            if (wave + 8 * k_ < nxq) {                                                               \
                const unsigned rb_ = (unsigned)__builtin_amdgcn_readlane((int)rbv, (int)pr_[k_]);    \
                glds16(rsX, p_x + plds_[k_], rb_ + pt_[k_]);                                         \
            }                                                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_)                                             \
            if (wrev + 8 * k_ < ndy) {                                                               \
                const unsigned bb_ = (unsigned)__builtin_amdgcn_ds_bpermute((int)((pd_[k_] & 3u) << 2), (int)dyv);   \
                glds16_b(rsY, p_y + (unsigned)(wrev + 8 * k_) * 1024u, bb_ + (pd_[k_] & ~3u));       \
            }                                                                                        \
    }

    W3_MARK();                                   // 1: prologue done
    const int G = nstrips * nsteps;              // steps of the item, strip after strip
    int strip_ld = 0, st_ld = 0;
    setup_strip(0);
    __syncthreads();                             // tables visible
    W3_PIECES_OF_STRIP()
    W3_CURSOR_RESET()
    W3_ISSUE_BEGIN(0u)
    W3_ISSUE_ALL()
    W3_MARK();                                   // 2: first issue done
    const unsigned char* sy0 = smem_raw + xstage + DYL;
    for (int g = 0; g < G; ++g) {
        const unsigned sb = (g & 1) ? stage : 0u;
        wait_vm<0>();                            // step g landed (this wavefront's pieces) ...
        W3_MARK();                               // 3 + 4 g: DMA waited for
        __builtin_amdgcn_s_barrier();            // ... everyone's pieces landed, everyone finished reading step g - 1
        W3_MARK();                               // 4 + 4 g: barrier passed
        const bool more = g + 1 < G;
        if (more) {
            if (++st_ld == nsteps) {             // next strip: every wavefront has issued its last pieces of this one (barrier above)
                st_ld = 0; ++strip_ld;
                setup_strip(strip_ld * TC);
                __syncthreads();
                W3_PIECES_OF_STRIP()
                W3_CURSOR_RESET()
            } else W3_CURSOR_STEP()
            W3_ISSUE_BEGIN(stage - sb)
            W3_ISSUE_ALL()
        }
        W3_MARK();                               // 5 + 4 g: next step issued
        // The wavefront's sub-steps of the stage at `sb`: one dy fragment and nine shifted x fragments feed nine MFMAs.  Software
        // pipeline: the fragments of the NEXT sub-step are fetched behind the MFMAs of this one that read the registers they
        // replace (x fragment t right after MFMA t), so the LDS latency runs in the matrix pipe's shadow.
        if (slice < nsub) {
            const unsigned char* sx = smem_raw + sb;
            const unsigned char* sy = sy0 + sb;
            half8 a = w3_frag(sy + slice * 1024, sy + slice * 1024 + 256);
            half8 b[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) b[t] = w3_frag(sx + (XO[0] & 0xffffu) + tapo[t], sx + (XO[0] >> 16) + tapo[t]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < W3_MAXI; ++i) {
                const int sub = slice + SL * i;
                if (sub < nsub) {
                    const int in = i + 1 < W3_MAXI ? i + 1 : i;
                    const int subn = sub + SL < nsub ? sub + SL : sub;                    // (past the last one: a harmless re-read)
                    const unsigned char* x0 = sx + (XO[in] & 0xffffu);
                    const unsigned char* x1 = sx + (XO[in] >> 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], acc[t], 0, 0, 0);
                        b[t] = w3_frag(x0 + tapo[t], x1 + tapo[t]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    a = w3_frag(sy + subn * 1024, sy + subn * 1024 + 256);
                }
            }
        }
        W3_MARK();                               // 6 + 4 g: MFMAs issued
    }
#undef W3_CURSOR_RESET
#undef W3_CURSOR_STEP
#undef W3_ISSUE_BEGIN
#undef W3_ISSUE_ALL
#undef W3_PIECES_OF_STRIP
    // the accumulators are read below: MFMA result hazard (see AY_MFMA_PAD in conv.hip)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 11" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                             // the stages are free
#ifdef AYOLO_PROBE
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) s_probe[W3_PROBE_N - 6] = t_; }
#endif
    // ---- wavefronts that share a block: slices 1 .. SL - 1 hand their sums to slice 0 through LDS, one tap per round, added in
    // slice order (a fixed order: the partial is bit-reproducible)
    if (SL > 1) {
        float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (slice > 0) {
                float* dst = red + (size_t)((slice - 1) * NP + pair) * (16 * 64) + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r * 64] = acc[t][r];
            }
            __syncthreads();
            if (slice == 0) {
                for (int sl = 1; sl < SL; ++sl) {
                    const float* src = red + (size_t)((sl - 1) * NP + pair) * (16 * 64) + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += src[r * 64];
                }
            }
            __syncthreads();
        }
    }
#ifdef AYOLO_PROBE
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) s_probe[W3_PROBE_N - 5] = t_; }
#endif
    // ---- acc[t][r]: output channel 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), input channel lane & 31 of block (nb, cb), tap t
    // -> this split's TILE-MAJOR slot of the workspace ([tile][n-block][c-block][tap][32 rows][32 channels], w3_perm in
    // conv.hip): the workgroup's 9 * NB * CB blocks are one contiguous run, a store covers two 128-byte rows
    if (slice == 0) {
        float* slotp = ws + p.ws_off + (unsigned long long)(p.zz0 + zz) * ((unsigned long long)p.tn * p.tc * NB * CB * 9ull * 1024ull)
                       + ((unsigned long long)(tile * (unsigned)(NB * CB) + (unsigned)(nb * CB + cb)) * 9ull) * 1024ull
                       + (unsigned)((4 * (lane >> 5)) * 32 + (lane & 31));
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) slotp[t * 1024 + (8 * (r >> 2) + (r & 3)) * 32] = acc[t][r];
    }
#undef p
#undef W3FD
#ifdef AYOLO_PROBE
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) s_probe[W3_PROBE_N - 4] = t_; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) { s_probe[W3_PROBE_N - 1] = t_; s_probe[W3_PROBE_N - 2] = (unsigned long long)probe_k; }
    }
    __syncthreads();
    if (blockIdx.x < 512 && threadIdx.x < W3_PROBE_N) g_probe3[blockIdx.x * W3_PROBE_N + threadIdx.x] = s_probe[threadIdx.x];
#endif
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int w3_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

size_t w3_lds_bytes(const W3P& p) {
    const size_t a = 2 * (size_t)p.stage + W3_TAB_BYTES;
    return a < W3_RED_BYTES ? (size_t)W3_RED_BYTES : a;
}

int w3_fill(const ayolo_conv_desc* d, const void* x, const void* dy, W3P& p, bool any_route) {
    // Routing.  Alone on the chip the kernel beats the generic k_wgrad on the stride-1 3x3 layers of YOLOv5s at batch 64 except the
    // smallest map (122 vs 154 us at 160 x 160, 62 vs 82 at 80 x 80, 67 vs 73 at 40 x 40, 82 vs 73 at 20 x 20:
    // profiles/r05_wgrad3_layer_sweep.txt).  Inside the train step -- one more launch and tail per weight-gradient group, an
    // 8-wavefront workgroup owning a CU while the main stream's kernels look for slots -- it never paid for itself: round 5's best
    // routing (maps of >= 80 rows) measured -0.06 ... -0.11 ms on one box and level on the next, and on round 6's code on/off is
    // 11.973 / 11.972 ms (profiles/r06_ab_sppf_v2_wgrad3.txt).  RETIRED from the default route in round 6:
    // nothing is routed here unless AYOLO_WGRAD3 is set (1: every stride-1 3x3 layer; 2: the stride-2 layers too -- slower on every
    // stride-2 layer of YOLOv5s, r05_wgrad3_layer_sweep.txt); the variable is read at every planning call so that the tests that
    // keep the kernel honest can switch it.  The former tuning switches (_MINHW, _S2, _WAVES, _MINQ) are gone.
    const int on = w3_env("AYOLO_WGRAD3", 0);
    if (!on && !any_route) return 1;
    if (d->dtype != AYOLO_F16 || d->kh != 3 || d->kw != 3 || d->ph != 1 || d->pw != 1 || d->sh != d->sw || (d->sh != 1 && d->sh != 2)) return 1;
    if (d->Cin % 8 || d->Cout % 8 || d->ldx % 8 || d->ldy % 8 || d->Cin < 16 || d->Cout < 16) return 1;
    const int s = d->sh;
    // stride 2: the window is four input pixels per output pixel -- more DMA bytes per MFMA than the generic kernel's widest tiles
    if (s == 2 && on < 2 && !any_route) return 1;
    if (d->Ho != (d->H + 2 - 3) / s + 1 || d->Wo != (d->W + 2 - 3) / s + 1) return 1;
    const long long xb = (long long)d->B * d->H * d->W * d->ldx * 2, yb = (long long)d->B * d->Ho * d->Wo * d->ldy * 2;
    if (xb >= (1ll << 30) || yb >= (1ll << 30)) return 1;
    p = W3P{};
    p.x = x; p.dy = dy;
    p.B = d->B; p.XH = d->H; p.XW = d->W; p.ldx = d->ldx; p.C = d->Cin;
    p.OH = d->Ho; p.OW = d->Wo; p.ldy = d->ldy; p.N = d->Cout;
    p.s = s; p.K = 9 * p.C;
    p.UP = p.OH + (s == 1 ? 2 : 1); p.XP = s * p.UP;
    if ((long long)d->B * p.XP >= (1ll << 30)) return 1;
    p.NU = (unsigned)(d->B * p.UP);
    p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
    const int NBt = (p.N + 31) / 32, CBt = (p.C + 31) / 32;
    if (CBt >= 2 && NBt >= 2) { p.NB = 2; p.CB = 2; }
    else if (CBt == 1) { p.CB = 1; p.NB = NBt >= 3 ? 4 : NBt; }
    else { p.NB = 1; p.CB = 2; }
    p.NP = p.NB * p.CB; p.SL = 8 / p.NP;
    p.tn = (NBt + p.NB - 1) / p.NB; p.tc = (CBt + p.CB - 1) / p.CB;
    // step geometry: the (TC, RPS) with the least modelled time per output pixel (tools/w3_probe.py prints the marks these numbers
    // come from).  A wavefront's instruction stream per step: ~300 cycles of waits / barrier / row bases, ~150 per DMA piece it
    // issues (an eighth of the step's), ~330 per 16-pixel sub-step of its own (9 MFMAs with the fragment fetch pipelined behind
    // them, live pixels or not); two wavefronts share a SIMD, so the matrix pipe needs 2 x 288 cycles per sub-step of a slice.
    double best = 1e30;
    for (int TC = 4; TC <= 96; TC += 4)
        for (int RPS = 1; RPS <= 4; ++RPS) {
            const int PX = TC * RPS;
            if (PX > 96 || (TC > p.OW + 3 && TC > 4)) continue;
            const int nsub = (PX + 15) / 16;
            const int pw = s == 1 ? TC + 2 : 2 * TC + 1;
            const int ppr = (p.CB * pw * 4 + 63) / 64;
            const int nrows = s * (RPS - 1) + 3;
            if (nrows > p.XP || nrows * ppr > 32 || p.NB * nsub > 24 || (ppr + p.NB * nsub) * 256 > W3_TAB_BYTES) continue;
            const int stage = nrows * ppr * 1024 + p.NB * nsub * 1024;
            if (stage > W3_STAGE_MAX) continue;
            const int strips = (p.OW + TC - 1) / TC;
            const int wsub = (nsub + p.SL - 1) / p.SL;
            const int npw = (nrows * ppr + 7) / 8 + (p.NB * nsub + 7) / 8;
            const double issue = 300.0 + 150.0 * npw + 330.0 * wsub, pipe = 200.0 + 576.0 * wsub;
            const double tstep = issue > pipe ? issue : pipe;
            const double t = tstep * strips / ((double)RPS * p.OW);
            if (t < best - 1e-9) {
                best = t;
                p.TC = TC; p.RPS = RPS; p.PX = PX; p.nsub = nsub; p.strips = strips;
                p.nrows = nrows; p.ppr = ppr; p.rowpitch = ppr * 1024;
                p.plo = (s == 1 ? TC + 2 : TC + 1) * 64; p.ple = s == 1 ? 0 : TC * 64;
                p.xstage = nrows * ppr * 1024; p.stage = stage;
                p.step_cost = tstep;
            }
        }
    if (best > 1e29) return 1;
    p.dXP = make_fastdiv((unsigned)p.XP); p.dUP = make_fastdiv((unsigned)p.UP); p.dTC = make_fastdiv((unsigned)p.TC);
    p.dPPR = make_fastdiv((unsigned)p.ppr);
    p.uch = (unsigned)p.RPS; p.uranges = (p.NU + p.uch - 1) / p.uch;
    return 0;
}

void w3_split(W3P& p, double steps) {
    const unsigned total = (p.NU + (unsigned)p.RPS - 1) / (unsigned)p.RPS;        // steps of one strip over all rows
    double q = steps / p.strips < 2.0 ? 2.0 : steps / p.strips;                   // an item walks its rows once per strip
    unsigned n = (unsigned)((double)total / q + 0.5);
    if (n < 1) n = 1;
    unsigned per = (total + n - 1) / n;
    p.uch = per * (unsigned)p.RPS;
    p.uranges = (p.NU + p.uch - 1) / p.uch;
}

template <int RP>
static int w3_launch_rp(const W3P& pv, const W3P* jobs, const WItem* items, unsigned blocks, size_t lds, float* ws, hipStream_t s) {
    static bool attr_set[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
#ifdef AYOLO_PROBE
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<RP>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W3_STAGE_MAX + W3_TAB_BYTES);
#else
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<RP>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W3_STAGE_MAX + W3_TAB_BYTES);
#endif
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_wgrad3<RP>, dim3(blocks), dim3(512), lds, s, pv, jobs, items, ws);
    AY_CHECK_LAUNCH("k_wgrad3");
    return AYOLO_OK;
}

/* rp: the launch's compile-time row pitch (w3_rp_class of every job of the launch), 0 = the generic instantiation */
int w3_launch(const W3P& pv, const W3P* jobs, const WItem* items, unsigned blocks, size_t lds, float* ws, int rp, hipStream_t s) {
    switch (rp) {
        case 2048: return w3_launch_rp<2048>(pv, jobs, items, blocks, lds, ws, s);
        case 3072: return w3_launch_rp<3072>(pv, jobs, items, blocks, lds, ws, s);
        case 4096: return w3_launch_rp<4096>(pv, jobs, items, blocks, lds, ws, s);
        case 6144: return w3_launch_rp<6144>(pv, jobs, items, blocks, lds, ws, s);
        default: return w3_launch_rp<0>(pv, jobs, items, blocks, lds, ws, s);
    }
}

/* introspection (tests/test_kernel_math.py restates the kernel's index algebra on the CPU from these numbers): the step
 * geometry w3_fill chooses for `d`; returns AYOLO_EINVAL when the layer is not k_wgrad3's */
extern "C" int ayolo_wgrad3_geometry(const ayolo_conv_desc* d, long long* out, int nout) {
    AY_CHECK_ARG(d && out && nout >= 24, "wgrad3_geometry: out[24]");
    W3P p;
    AY_CHECK_ARG(w3_fill(d, d, d, p, true) == 0, "wgrad3_geometry: not a fp16 3x3 / pad 1 / stride 1 or 2 layer of k_wgrad3");
    const long long v[24] = {p.TC, p.RPS, p.PX, p.nsub, p.strips, p.NB, p.CB, p.NP, p.SL, p.tn, p.tc, p.nrows, p.ppr, p.rowpitch,
                             p.plo, p.ple, p.xstage, p.stage, p.UP, p.XP, (long long)p.NU, (long long)p.x_bytes, (long long)p.y_bytes,
                             (long long)w3_lds_bytes(p)};
    for (int i = 0; i < 24; ++i) out[i] = v[i];
    return AYOLO_OK;
}
