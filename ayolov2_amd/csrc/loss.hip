// Fused YOLO loss (forward value + analytic backward) -- ComputeLoss.__call__ of the reference
// (scripts/loss/losses.py:227-300) without the ~600 small torch kernels and the autograd graph it records:
//
//   forward : k_loss_rows (one wavefront per matched target row: gather the 5+nc logits of its cell, CIoU box term,
//             class BCE, objectness target of the row, "last row wins" ownership of the cell)
//             k_loss_obj  (every (b, a, y, x) cell: objectness BCE against the owner row's detached IoU)
//             k_loss_finalize (means, balance, hyper-parameter gains -> [loss*bs, lbox, lobj, lcls, loss])
//   backward: k_loss_grad_dense (d loss / d logits for every element: zero except the objectness channel)
//             k_loss_grad_rows  (box + class gradients of the matched rows, atomically added: rows that share a
//             cell accumulate exactly like the index_put backward of `pi[b, a, gj, gi]`)
//
// Arithmetic follows scripts/utils/metrics.py:60-135 (bbox_iou, CIoU with eps 1e-7, alpha under no_grad) and
// torch.nn.BCEWithLogitsLoss(pos_weight) in fp32; the partial sums are accumulated in fp64 atomics.
#include "common.h"

#define LOSS_MAX_LEVELS 8
#define LOSS_SLOTS 64        // hashed fp64 accumulators per (level, term): spreads the atomics of a launch

struct LossP {
    ayolo_loss_level lv[LOSS_MAX_LEVELS];
    int nl;
    float cp, cn, cls_pw, obj_pw, gr;
    float h_box, h_obj, h_cls;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// BCEWithLogits(x, t, pos_weight): (1 - t) * x + lw * (log1p(exp(-|x|)) + max(-x, 0)),  lw = 1 + (pw - 1) * t
__device__ __forceinline__ float bce_logits(float x, float t, float pw) {
    const float lw = 1.0f + (pw - 1.0f) * t;
    return (1.0f - t) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f));
}
__device__ __forceinline__ float bce_logits_grad(float x, float t, float pw) {
    const float lw = 1.0f + (pw - 1.0f) * t;
    return (1.0f - t) + lw * (sigmoid_f(x) - 1.0f);
}

struct CIoU {
    float ciou;
    float d[4];    // d ciou / d (p0, p1, p2, p3): the four box logits
};

// pbox = (sig(p0)*2-0.5, sig(p1)*2-0.5, (sig(p2)*2)^2*aw, (sig(p3)*2)^2*ah) vs target (tx, ty, tw, th), both cxcywh
template <bool GRAD>
__device__ __forceinline__ CIoU ciou_eval(const float p[4], const float t[4], float aw, float ah) {
    const float eps = 1e-7f;
    const float s0 = sigmoid_f(p[0]), s1 = sigmoid_f(p[1]), s2 = sigmoid_f(p[2]), s3 = sigmoid_f(p[3]);
    const float x = s0 * 2.0f - 0.5f, y = s1 * 2.0f - 0.5f;
    const float w = (s2 * 2.0f) * (s2 * 2.0f) * aw, h = (s3 * 2.0f) * (s3 * 2.0f) * ah;
    const float b1x1 = x - w / 2, b1x2 = x + w / 2, b1y1 = y - h / 2, b1y2 = y + h / 2;
    const float b2x1 = t[0] - t[2] / 2, b2x2 = t[0] + t[2] / 2, b2y1 = t[1] - t[3] / 2, b2y2 = t[1] + t[3] / 2;
    const float dxi = fminf(b1x2, b2x2) - fmaxf(b1x1, b2x1), dyi = fminf(b1y2, b2y2) - fmaxf(b1y1, b2y1);
    const float iw = fmaxf(dxi, 0.0f), ih = fmaxf(dyi, 0.0f);
    const float inter = iw * ih;
    const float w1 = b1x2 - b1x1, h1 = b1y2 - b1y1 + eps;
    const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + eps;
    const float uni = w1 * h1 + w2 * h2 - inter + eps;
    const float iou = inter / uni;
    const float cw = fmaxf(b1x2, b2x2) - fminf(b1x1, b2x1), ch = fmaxf(b1y2, b2y2) - fminf(b1y1, b2y1);
    const float c2 = cw * cw + ch * ch + eps;
    const float sx = b2x1 + b2x2 - b1x1 - b1x2, sy = b2y1 + b2y2 - b1y1 - b1y2;
    const float rho2 = (sx * sx + sy * sy) / 4;
    const float kpi = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
    const float da = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = kpi * (da * da);
    const float alpha = v / (v - iou + (1.0f + eps));
    CIoU r;
    r.ciou = iou - (rho2 / c2 + v * alpha);
    if constexpr (GRAD) {
        // reverse mode with d ciou = 1; alpha is a constant (torch.no_grad)
        const float g_rho2 = -1.0f / c2, g_c2 = rho2 / (c2 * c2), g_v = -alpha;
        float g_inter = 1.0f / uni;
        const float g_uni = -inter / (uni * uni);
        float g_w1 = g_uni * h1, g_h1 = g_uni * w1;
        g_inter -= g_uni;
        const float g_a1 = -2.0f * kpi * da * g_v;                 // d v / d atan(w1/h1) = -2 k da
        const float q = w1 * w1 + h1 * h1;
        g_w1 += g_a1 * (h1 / q);
        g_h1 += g_a1 * (-w1 / q);
        const float g_iw = g_inter * ih, g_ih = g_inter * iw;
        float gx1 = 0.0f, gx2 = 0.0f, gy1 = 0.0f, gy2 = 0.0f;   // wrt b1x1, b1x2, b1y1, b1y2
        // clamp(0) passes the gradient where its input is >= 0; min/max split ties evenly (torch semantics)
        if (dxi >= 0.0f) {
            gx2 += g_iw * (b1x2 < b2x2 ? 1.0f : (b1x2 == b2x2 ? 0.5f : 0.0f));
            gx1 -= g_iw * (b1x1 > b2x1 ? 1.0f : (b1x1 == b2x1 ? 0.5f : 0.0f));
        }
        if (dyi >= 0.0f) {
            gy2 += g_ih * (b1y2 < b2y2 ? 1.0f : (b1y2 == b2y2 ? 0.5f : 0.0f));
            gy1 -= g_ih * (b1y1 > b2y1 ? 1.0f : (b1y1 == b2y1 ? 0.5f : 0.0f));
        }
        const float g_cw = g_c2 * 2.0f * cw, g_ch = g_c2 * 2.0f * ch;
        gx2 += g_cw * (b1x2 > b2x2 ? 1.0f : (b1x2 == b2x2 ? 0.5f : 0.0f));
        gx1 -= g_cw * (b1x1 < b2x1 ? 1.0f : (b1x1 == b2x1 ? 0.5f : 0.0f));
        gy2 += g_ch * (b1y2 > b2y2 ? 1.0f : (b1y2 == b2y2 ? 0.5f : 0.0f));
        gy1 -= g_ch * (b1y1 < b2y1 ? 1.0f : (b1y1 == b2y1 ? 0.5f : 0.0f));
        const float g_sx = g_rho2 * sx / 2, g_sy = g_rho2 * sy / 2;
        gx1 -= g_sx; gx2 -= g_sx; gy1 -= g_sy; gy2 -= g_sy;
        gx2 += g_w1; gx1 -= g_w1; gy2 += g_h1; gy1 -= g_h1;
        const float g_x = gx1 + gx2, g_y = gy1 + gy2, g_w = (gx2 - gx1) / 2, g_h = (gy2 - gy1) / 2;
        r.d[0] = g_x * 2.0f * s0 * (1.0f - s0);
        r.d[1] = g_y * 2.0f * s1 * (1.0f - s1);
        r.d[2] = g_w * aw * 8.0f * s2 * s2 * (1.0f - s2);
        r.d[3] = g_h * ah * 8.0f * s3 * s3 * (1.0f - s3);
    }
    return r;
}

// acc layout: double acc[nl][3][LOSS_SLOTS] = {sum (1 - ciou), sum objectness BCE, sum class BCE}, slot = workgroup % LOSS_SLOTS
template <bool BWD>
__global__ __launch_bounds__(256) void k_loss_rows(LossP P, double* acc, const float* grad_out) {
    const int l = blockIdx.y;
    const ayolo_loss_level& L = P.lv[l];
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= L.n) return;
    const long long b = L.b[row], a = L.a[row], gj = L.gj[row], gi = L.gi[row];
    const float* ps = L.pred + b * L.sb + a * L.sa + gj * L.sy + gi * L.sx;
    const long long cell = ((b * L.na + a) * L.ny + gj) * L.nx + gi;
    float p[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = ps[i]; t[i] = L.tbox[(long long)row * 4 + i]; }
    const float aw = L.anch[(long long)row * 2], ah = L.anch[(long long)row * 2 + 1];
    const int nc = L.no - 5;
    const int tc = (int)L.tcls[row];
    if constexpr (!BWD) {
        const CIoU r = ciou_eval<false>(p, t, aw, ah);
        float cls = 0.0f;
        if (nc > 1)
            for (int c = lane; c < nc; c += 64) cls += bce_logits(ps[5 + c], c == tc ? P.cp : P.cn, P.cls_pw);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cls += __shfl_xor(cls, off);
        if (lane == 0) {
            const int slot = blockIdx.x % LOSS_SLOTS;
            atomicAdd(&acc[(l * 3 + 0) * LOSS_SLOTS + slot], (double)(1.0f - r.ciou));
            if (nc > 1) atomicAdd(&acc[(l * 3 + 2) * LOSS_SLOTS + slot], (double)cls);
            L.score[row] = (1.0f - P.gr) + P.gr * fmaxf(r.ciou, 0.0f);
            atomicMax(&L.own[cell], row + 1);        // duplicate cells: the LAST row's objectness target wins
            if (L.head) L.next[row] = atomicExch(&L.head[cell], row + 1);   // all rows of the cell (packed backward)
        }
    } else {
        const float g = grad_out[0] * (float)L.B;    // d (loss * bs)
        float* gp = L.grad + cell * L.no;
        if (lane == 0) {
            const CIoU r = ciou_eval<true>(p, t, aw, ah);
            const float k = -g * P.h_box / (float)L.n;           // lbox = mean(1 - ciou) * h_box
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(&gp[i], k * r.d[i]);
        }
        if (nc > 1) {
            const float k = g * P.h_cls / ((float)L.n * (float)nc);
            for (int c = lane; c < nc; c += 64)
                atomicAdd(&gp[5 + c], k * bce_logits_grad(ps[5 + c], c == tc ? P.cp : P.cn, P.cls_pw));
        }
    }
}

__global__ __launch_bounds__(256) void k_loss_obj(LossP P, double* acc) {
    const int l = blockIdx.y;
    const ayolo_loss_level& L = P.lv[l];
    const long long cells = (long long)L.B * L.na * L.ny * L.nx;
    float s = 0.0f;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < cells; c += (long long)gridDim.x * 256) {
        const unsigned cu = (unsigned)c;
        const unsigned x = cu % (unsigned)L.nx, r1 = cu / (unsigned)L.nx;
        const unsigned y = r1 % (unsigned)L.ny, r2 = r1 / (unsigned)L.ny;
        const unsigned a = r2 % (unsigned)L.na, b = r2 / (unsigned)L.na;
        const float v = L.pred[(long long)b * L.sb + (long long)a * L.sa + (long long)y * L.sy + (long long)x * L.sx + 4];
        const int o = L.own[c];
        s += bce_logits(v, o ? L.score[o - 1] : 0.0f, P.obj_pw);
    }
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(&acc[(l * 3 + 1) * LOSS_SLOTS + blockIdx.x % LOSS_SLOTS], (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
}

__global__ void k_loss_finalize(LossP P, const double* acc, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lbox = 0.0f, lobj = 0.0f, lcls = 0.0f;
    for (int l = 0; l < P.nl; ++l) {
        const ayolo_loss_level& L = P.lv[l];
        const double cells = (double)L.B * L.na * L.ny * L.nx;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < LOSS_SLOTS; ++k) {
            s0 += acc[(l * 3 + 0) * LOSS_SLOTS + k]; s1 += acc[(l * 3 + 1) * LOSS_SLOTS + k]; s2 += acc[(l * 3 + 2) * LOSS_SLOTS + k];
        }
        if (L.n > 0) {
            lbox += (float)(s0 / (double)L.n);
            if (L.no - 5 > 1) lcls += (float)(s2 / ((double)L.n * (double)(L.no - 5)));
        }
        lobj += (float)(s1 / cells) * L.balance;
    }
    lbox *= P.h_box; lobj *= P.h_obj; lcls *= P.h_cls;
    const float loss = lbox + lobj + lcls;
    out[0] = loss * (float)P.lv[0].B;
    out[1] = lbox; out[2] = lobj; out[3] = lcls; out[4] = loss;
}

// grad[(b, a, y, x, o)] contiguous: objectness channel gets its BCE gradient, everything else 0
__device__ __forceinline__ float loss_obj_grad(const LossP& P, const ayolo_loss_level& L, long long c, float k) {
    const unsigned cu = (unsigned)c;
    const unsigned x = cu % (unsigned)L.nx, r1 = cu / (unsigned)L.nx;
    const unsigned y = r1 % (unsigned)L.ny, r2 = r1 / (unsigned)L.ny;
    const unsigned a = r2 % (unsigned)L.na, b = r2 / (unsigned)L.na;
    const float v = L.pred[(long long)b * L.sb + (long long)a * L.sa + (long long)y * L.sy + (long long)x * L.sx + 4];
    const int ow = L.own[c];
    return k * bce_logits_grad(v, ow ? L.score[ow - 1] : 0.0f, P.obj_pw);
}

__global__ __launch_bounds__(256) void k_loss_grad_dense(LossP P, const float* grad_out) {
    const int l = blockIdx.y;
    const ayolo_loss_level& L = P.lv[l];
    const long long cells = (long long)L.B * L.na * L.ny * L.nx;
    const long long total = cells * L.no, total4 = total / 4;
    const float k = grad_out[0] * (float)L.B * P.h_obj * L.balance / (float)cells;
    const unsigned no = (unsigned)L.no;
    const bool small = total < (1ll << 32);
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long long)gridDim.x * 256) {
        float4v o4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const long long e0 = q * 4;
        const long long c0 = small ? (long long)((unsigned)e0 / no) : e0 / no;      // 32-bit division when it fits
        const unsigned r0 = (unsigned)(e0 - c0 * no);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned r = r0 + i;
            long long c = c0;
            if (r >= no) { r -= no; c += 1; }
            if (r == 4) o4[i] = loss_obj_grad(P, L, c, k);
        }
        *reinterpret_cast<float4v*>(L.grad + e0) = o4;
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(total - total4 * 4)) {       // < 4 trailing elements
        const long long e = total4 * 4 + threadIdx.x;
        const long long c = e / no;
        L.grad[e] = (unsigned)(e - c * no) == 4 ? loss_obj_grad(P, L, c, k) : 0.0f;
    }
}

// ---- packed backward: gradient written as the head conv's dz (NHWC, compute dtype) + dbias ------------------------------
__global__ __launch_bounds__(256) void k_loss_rowbox(LossP P, const float* grad_out) {
    const int l = blockIdx.y;
    const ayolo_loss_level& L = P.lv[l];
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= L.n) return;
    const long long b = L.b[row], a = L.a[row], gj = L.gj[row], gi = L.gi[row];
    const float* ps = L.pred + b * L.sb + a * L.sa + gj * L.sy + gi * L.sx;
    float p[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = ps[i]; t[i] = L.tbox[(long long)row * 4 + i]; }
    const CIoU r = ciou_eval<true>(p, t, L.anch[(long long)row * 2], L.anch[(long long)row * 2 + 1]);
    const float k = -grad_out[0] * (float)L.B * P.h_box / (float)L.n;
#pragma unroll
    for (int i = 0; i < 4; ++i) L.rowbox[(long long)row * 4 + i] = k * r.d[i];
}

// Packed gradient of one level, sparse form.  dz (B*ny*nx rows of ldz channels) is almost all zeros: per pixel only the na
// objectness channels are dense, the box / class channels are non-zero only in cells that own matched rows.  Three stages:
//   1. zero fill of dz (k_fill_zero, HBM speed)
//   2. MODE 1: the 8-channel group holding the objectness channel of every (pixel, anchor)
//   3. MODE 2: the other groups of every cell that owns rows (visited once, by the head row of the cell's list)
// (the dense form walked all 32 groups of every pixel with two dependent list-head loads each: 0.28 ms for 275 MB, 1 TB/s).
// A group's value is computed by loss_group() from the cells of BOTH anchors it may span, so every group is written whole
// and exactly once; the bias gradient is accumulated from the same fp32 values.
template <typename T>
__device__ __forceinline__ void loss_group_values(const LossP& P, const ayolo_loss_level& L, long long pix, int cg, float k_obj, float k_cls,
                                                  float* sb, T (&out)[8]) {
    const unsigned hw = (unsigned)(L.ny * L.nx), nx = (unsigned)L.nx;
    const int no = L.no, nc = no - 5, Cc = L.na * no, ldz = L.ldz;
    const unsigned pu = (unsigned)pix;
    const unsigned b = pu / hw, pp = pu - b * hw;
    const unsigned y = pp / nx, x = pp - y * nx;
    const int c0 = cg * 8;
    const int a_lo = c0 / no;
    const int a_hi = (c0 + 7 < Cc ? c0 + 7 : Cc - 1) / no;
    const long long cell_lo = ((long long)b * L.na + a_lo) * hw + pp;
    const int h_lo = L.head[cell_lo];
    const int h_hi = a_hi != a_lo ? L.head[cell_lo + hw] : h_lo;
    const float* ps0 = L.pred + (long long)b * L.sb + (long long)y * L.sy + (long long)x * L.sx;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c0 + i;
        if (c >= Cc) continue;
        const int a = c >= (a_lo + 1) * no ? a_hi : a_lo;
        const int o = c - a * no;
        if (o == 4) {
            const long long cell = cell_lo + (long long)(a - a_lo) * hw;
            const int ow = L.own[cell];
            v[i] = k_obj * bce_logits_grad(ps0[(long long)a * L.sa + 4], ow ? L.score[ow - 1] : 0.0f, P.obj_pw);
        } else {
            const float* ps = ps0 + (long long)a * L.sa;
            for (int h = a == a_lo ? h_lo : h_hi; h; h = L.next[h - 1]) {        // rows matched to this cell (usually none)
                const int row = h - 1;
                if (o < 4) v[i] += L.rowbox[(long long)row * 4 + o];
                else if (nc > 1) v[i] += k_cls * bce_logits_grad(ps[o], (o - 5) == (int)L.tcls[row] ? P.cp : P.cn, P.cls_pw);
            }
        }
        if (sb && v[i] != 0.0f) atomicAdd(&sb[c], v[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (T)v[i];
}

template <typename T>
__device__ __forceinline__ void loss_group(const LossP& P, const ayolo_loss_level& L, long long pix, int cg, float k_obj, float k_cls,
                                           float* sb) {
    T out[8];
    loss_group_values<T>(P, L, pix, cg, k_obj, k_cls, sb, out);
    T* dst = reinterpret_cast<T*>(L.dz) + pix * L.ldz + cg * 8;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(out);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst + 4) = *reinterpret_cast<const uint4*>(out + 4);
}

#define LOSS_PB 64                        // pixels per pass of the full-row writer (MODE 3): 64 * na <= 256 threads for na <= 4
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_loss_grad_packed(LossP P, const float* grad_out) {
    const int l = blockIdx.y;
    const ayolo_loss_level& L = P.lv[l];
    const unsigned hw = (unsigned)(L.ny * L.nx);
    const long long npix = (long long)L.B * hw;
    const long long cells = npix * L.na;
    const int no = L.no, nc = no - 5, Cc = L.na * no;
    const float g = grad_out[0] * (float)L.B;
    const float k_obj = g * P.h_obj * L.balance / (float)cells;
    const float k_cls = (L.n > 0 && nc > 1) ? g * P.h_cls / ((float)L.n * (float)nc) : 0.0f;
    __shared__ float sb[2048];                       // bias-gradient partials of this workgroup (ldz <= 2048)
    if (L.dbias) {
        for (int c = threadIdx.x; c < Cc; c += 256) sb[c] = 0.0f;
        __syncthreads();
    }
    float* acc = L.dbias ? sb : nullptr;
    if (MODE == 3) {
        // stages 1 + 2 in one pass of FULL rows: a workgroup takes LOSS_PB pixels at a time, computes their na objectness groups
        // (one thread per (pixel, anchor)) into LDS and then writes the pixels' whole dz rows -- those groups in place, zeros
        // everywhere else -- in consecutive 16-byte pieces.  The zero fill (275 MB at HBM speed) plus the scattered 16-byte
        // writes of stage 2 into it (one per 512-byte row and anchor) were 65 + 127 us of the step's serial head.
        constexpr int PB = LOSS_PB;
        __shared__ T sg[PB * 4][8];                                    // [pixel][anchor (na <= 4)][8 channels]
        const int gpr = L.ldz / 8;                                     // groups per row
        const long long nblk = (npix + PB - 1) / PB;
        for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
            const long long p0 = blk * PB;
            const int t = threadIdx.x;
            if (t < PB * L.na) {
                const int pl = t / L.na, a = t - pl * L.na;
                if (p0 + pl < npix) loss_group_values<T>(P, L, p0 + pl, (a * no + 4) / 8, k_obj, k_cls, acc, sg[pl * 4 + a]);
            }
            __syncthreads();
            const int cpg = (int)(8 * sizeof(T) / 16);                  // 16-byte pieces per group: 1 (fp16) or 2 (fp32)
            const int total = PB * gpr * cpg;
            for (int q = t; q < total; q += 256) {
                const int pl = q / (gpr * cpg), r = q - pl * gpr * cpg;
                if (p0 + pl >= npix) break;
                const int cg = r / cpg, half = r - cg * cpg;
                uint4 v = make_uint4(0, 0, 0, 0);
                // is cg the objectness group of an anchor?  (a * no + 4) / 8 == cg  <=>  a in [ceil((8 cg - 4) / no), (8 cg + 3) / no]
                const int a = (8 * cg + 3) / no;
                if (a < L.na && (a * no + 4) / 8 == cg) v = reinterpret_cast<const uint4*>(sg[pl * 4 + a])[half];
                reinterpret_cast<uint4*>(reinterpret_cast<T*>(L.dz) + (p0 + pl) * L.ldz)[r] = v;
            }
            __syncthreads();
        }
    } else if (MODE == 1) {
        // (pixel, anchor) -> the group of that anchor's objectness channel; two anchors never share one (no >= 8)
        const long long total = npix * L.na;
        for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < total; w += (long long)gridDim.x * 256) {
            const long long pix = w / L.na;
            const int a = (int)(w - pix * L.na);
            loss_group<T>(P, L, pix, (a * no + 4) / 8, k_obj, k_cls, acc);
        }
    } else {
        // (row, k-th group of the row's anchor): only the head row of a cell's list acts, so a cell is visited once; a group
        // shared with the neighbouring anchor belongs to the lower anchor's cell if that cell owns rows, else to the upper one
        const int G = (no + 7) / 8 + 1;
        const long long total = (long long)L.n * G;
        for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < total; w += (long long)gridDim.x * 256) {
            const int row = (int)(w / G), k = (int)(w - (long long)row * G);
            const long long b = L.b[row], a = L.a[row], gj = L.gj[row], gi = L.gi[row];
            const long long pp = gj * L.nx + gi;
            const long long cell = (b * L.na + a) * hw + pp;
            if (L.head[cell] - 1 != row) continue;
            const int cg = (int)(a * no) / 8 + k;
            const int c0 = cg * 8;
            if (c0 >= (a + 1) * no || c0 >= Cc) continue;                          // beyond this anchor's channels
            const int a_lo = c0 / no, a_hi = (c0 + 7 < Cc ? c0 + 7 : Cc - 1) / no;
            bool mine = true;
            if (a_lo != a_hi && a == a_hi) mine = L.head[cell - hw] == 0;          // the lower anchor's cell owns a shared group
            const int obj_lo = a_lo * no + 4, obj_hi = a_hi * no + 4;
            if ((obj_lo >= c0 && obj_lo < c0 + 8) || (obj_hi >= c0 && obj_hi < c0 + 8)) mine = false;   // stage 2 wrote it
            if (mine) loss_group<T>(P, L, b * hw + pp, cg, k_obj, k_cls, acc);
        }
    }
    if (L.dbias) {
        __syncthreads();
        for (int c = threadIdx.x; c < Cc; c += 256)
            if (sb[c] != 0.0f) atomicAdd(&L.dbias[c], sb[c]);
    }
}

static int loss_pack(LossP* P, const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw, float gr,
                     float h_box, float h_obj, float h_cls, int need_grad, int* max_n) {
    AY_CHECK_ARG(lv && nl > 0 && nl <= LOSS_MAX_LEVELS, "yolo_loss: nl=%d", nl);
    *max_n = 0;
    for (int l = 0; l < nl; ++l) {
        const ayolo_loss_level& L = lv[l];
        AY_CHECK_ARG(L.pred && L.own && L.B > 0 && L.na > 0 && L.ny > 0 && L.nx > 0 && L.no >= 6, "yolo_loss: level %d", l);
        AY_CHECK_ARG((long long)L.B * L.na * L.ny * L.nx < (1ll << 31), "yolo_loss: level %d has too many cells", l);
        AY_CHECK_ARG(L.n == 0 || (L.b && L.a && L.gj && L.gi && L.tcls && L.tbox && L.anch && L.score), "yolo_loss: level %d rows", l);
        AY_CHECK_ARG(!need_grad || L.grad, "yolo_loss: level %d grad", l);
        P->lv[l] = L;
        if (L.n > *max_n) *max_n = L.n;
    }
    P->nl = nl; P->cp = cp; P->cn = cn; P->cls_pw = cls_pw; P->obj_pw = obj_pw; P->gr = gr;
    P->h_box = h_box; P->h_obj = h_obj; P->h_cls = h_cls;
    return AYOLO_OK;
}

extern "C" int ayolo_yolo_loss_fwd(const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw,
                                   float gr, float h_box, float h_obj, float h_cls, double* acc, float* out,
                                   ayolo_stream s) {
    LossP P{};
    int max_n = 0;
    int rc = loss_pack(&P, lv, nl, cp, cn, cls_pw, obj_pw, gr, h_box, h_obj, h_cls, 0, &max_n);
    if (rc) return rc;
    AY_CHECK_ARG(acc && out, "yolo_loss_fwd: null pointer");
    hipStream_t st = (hipStream_t)s;
    rc = ayolo_fill_zero(acc, (size_t)nl * 3 * LOSS_SLOTS * sizeof(double), s);
    if (rc) return rc;
    if (max_n > 0) {
        hipLaunchKernelGGL(k_loss_rows<false>, dim3((unsigned)((max_n + 3) / 4), (unsigned)nl), dim3(256), 0, st, P, acc, (const float*)nullptr);
        AY_CHECK_LAUNCH("k_loss_rows");
    }
    hipLaunchKernelGGL(k_loss_obj, dim3(512, (unsigned)nl), dim3(256), 0, st, P, acc);
    AY_CHECK_LAUNCH("k_loss_obj");
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, P, acc, out);
    AY_CHECK_LAUNCH("k_loss_finalize");
    return AYOLO_OK;
}

extern "C" int ayolo_yolo_loss_bwd(const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw,
                                   float gr, float h_box, float h_obj, float h_cls, const float* grad_out,
                                   ayolo_stream s) {
    LossP P{};
    int max_n = 0;
    int rc = loss_pack(&P, lv, nl, cp, cn, cls_pw, obj_pw, gr, h_box, h_obj, h_cls, 1, &max_n);
    if (rc) return rc;
    AY_CHECK_ARG(grad_out, "yolo_loss_bwd: null grad_out");
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(k_loss_grad_dense, dim3(2048, (unsigned)nl), dim3(256), 0, st, P, grad_out);
    AY_CHECK_LAUNCH("k_loss_grad_dense");
    if (max_n > 0) {
        hipLaunchKernelGGL(k_loss_rows<true>, dim3((unsigned)((max_n + 3) / 4), (unsigned)nl), dim3(256), 0, st, P, (double*)nullptr, grad_out);
        AY_CHECK_LAUNCH("k_loss_grad_rows");
    }
    return AYOLO_OK;
}

extern "C" int ayolo_yolo_loss_bwd_packed(const ayolo_loss_level* lv, int nl, float cp, float cn, float cls_pw, float obj_pw,
                                          float gr, float h_box, float h_obj, float h_cls, const float* grad_out,
                                          ayolo_stream s) {
    LossP P{};
    int max_n = 0;
    int rc = loss_pack(&P, lv, nl, cp, cn, cls_pw, obj_pw, gr, h_box, h_obj, h_cls, 0, &max_n);
    if (rc) return rc;
    AY_CHECK_ARG(grad_out, "yolo_loss_bwd_packed: null grad_out");
    int dt = lv[0].dz_dtype;
    for (int l = 0; l < nl; ++l) {
        const ayolo_loss_level& L = lv[l];
        AY_CHECK_ARG(L.dz && L.head && L.ldz >= L.na * L.no && L.ldz % 8 == 0 && L.ldz <= 2048 && L.dz_dtype == dt,
                     "yolo_loss_bwd_packed: level %d (ldz=%d)", l, L.ldz);
        AY_CHECK_ARG(L.n == 0 || (L.next && L.rowbox), "yolo_loss_bwd_packed: level %d rows", l);
    }
    hipStream_t st = (hipStream_t)s;
    if (max_n > 0) {
        hipLaunchKernelGGL(k_loss_rowbox, dim3((unsigned)((max_n + 255) / 256), (unsigned)nl), dim3(256), 0, st, P, grad_out);
        AY_CHECK_LAUNCH("k_loss_rowbox");
    }
    const size_t es = dt == AYOLO_F16 ? 2 : 4;
    bool rows = true;                            // (more than four anchors per cell: the fill + scatter route below)
    for (int l = 0; l < nl; ++l) rows = rows && lv[l].na <= 4;
    if (rows) {                                  // zero fill + objectness groups as one pass of whole rows (MODE 3)
        if (dt == AYOLO_F16) hipLaunchKernelGGL((k_loss_grad_packed<half_t, 3>), dim3(2048, (unsigned)nl), dim3(256), 0, st, P, grad_out);
        else hipLaunchKernelGGL((k_loss_grad_packed<float, 3>), dim3(2048, (unsigned)nl), dim3(256), 0, st, P, grad_out);
    } else {
        for (int l = 0; l < nl; ++l) {
            const ayolo_loss_level& L = lv[l];
            rc = ayolo_fill_zero(L.dz, (size_t)L.B * L.ny * L.nx * L.ldz * es, s);
            if (rc) return rc;
        }
        if (dt == AYOLO_F16) hipLaunchKernelGGL((k_loss_grad_packed<half_t, 1>), dim3(1024, (unsigned)nl), dim3(256), 0, st, P, grad_out);
        else hipLaunchKernelGGL((k_loss_grad_packed<float, 1>), dim3(1024, (unsigned)nl), dim3(256), 0, st, P, grad_out);
    }
    if (max_n > 0) {
        const unsigned gr_ = (unsigned)(((long long)max_n * 13 + 255) / 256);
        if (dt == AYOLO_F16) hipLaunchKernelGGL((k_loss_grad_packed<half_t, 2>), dim3(gr_ < 2048 ? gr_ : 2048, (unsigned)nl), dim3(256), 0, st, P, grad_out);
        else hipLaunchKernelGGL((k_loss_grad_packed<float, 2>), dim3(gr_ < 2048 ? gr_ : 2048, (unsigned)nl), dim3(256), 0, st, P, grad_out);
    }
    AY_CHECK_LAUNCH("k_loss_grad_packed");
    return AYOLO_OK;
}
