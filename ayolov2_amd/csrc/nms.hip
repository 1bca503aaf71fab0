// NMS path for gfx950: candidate filter (HBM-streaming), key sort (rocPRIM), wave64 IoU bit-matrix,
// single-wavefront greedy scan, dense IoU and the fast/matrix-NMS column reductions.
//
// Replaces what the reference delegates to torch boolean-mask gathers + torchvision.ops.nms
// (scripts/utils/metrics.py:285-443, scripts/utils/nms.py:15-116).  All float arithmetic that decides an
// index is done as the plain IEEE single-precision sequence of the CPU path (no FMA contraction, true
// division) so kept indices are bit-identical to the oracle.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <string.h>
#include <math.h>

#define CAND_ROWS 64

// ---------------------------------------------------------------------------------------------------
// Stage A: candidates
// ---------------------------------------------------------------------------------------------------
struct CandParams {
    const float* pred;
    int B, N, no;
    float ct;
    int multi_label, require_obj;
    const uint64_t* class_mask;
    const int32_t* rows;
    int rows_per_img;
    float* det;
    uint64_t* keys;
    uint32_t* counters;
    uint32_t capacity;
    int seq_bits;
    int order_by_seq;
    int cpb;                 // chunks of CAND_ROWS rows per workgroup
    int64_t nchunks;
    int key_mode;            // 0: image | ~conf | seq (or image | seq);  2: image | class | ~conf | row (fixed-shape NMS)
    int cls_bits;            // key_mode 2: width of the class field
    int box_xyxy;            // rows already carry x1 y1 x2 y2 (head exported with out_xyxy)
};

__device__ __forceinline__ bool class_ok(const uint64_t* m, int c) {
    return m == nullptr || ((m[c >> 6] >> (c & 63)) & 1ull);
}

// One workgroup walks `cpb` consecutive chunks of CAND_ROWS rows.  Hits are staged in LDS (CAND_STAGE slots) and
// output slots are reserved with ONE returning global atomic per flush: with a reservation per 64-row chunk the
// 12 600 same-address atomics of the 8 x 100 800 workload serialised in L2 (~12 ns each) and were the whole
// 0.30 ms of this kernel.  The next chunk is prefetched into registers while the current one is scanned.
#define CAND_STAGE 512
#define CAND_SPAN 8
#define CAND_NPRE 6          // 16-byte prefetch registers per lane: covers no <= 96

__global__ __launch_bounds__(256) void k_candidates(CandParams p) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t* stage_key = reinterpret_cast<uint64_t*>(lds_raw);
    float* stage_det = reinterpret_cast<float*>(lds_raw + (size_t)CAND_STAGE * 8);
    float* lds = reinterpret_cast<float*>(lds_raw + (size_t)CAND_STAGE * 32);
    __shared__ unsigned s_total, s_local, s_fill, s_base;
    __shared__ unsigned s_img[CAND_SPAN];
    const int nc = p.no - 5;
    const int tid = threadIdx.x;
    const int64_t total_rows = (int64_t)p.B * p.rows_per_img;
    const int64_t c0 = (int64_t)blockIdx.x * p.cpb;
    const int64_t c1 = min(c0 + (int64_t)p.cpb, p.nchunks);
    const int img_lo = (int)((c0 * CAND_ROWS) / p.rows_per_img);
    const int img_hi = (int)((min(c1 * CAND_ROWS, total_rows) - 1) / p.rows_per_img);
    const bool img_in_lds = (img_hi - img_lo) < CAND_SPAN;
    if (tid == 0) s_fill = 0;
    if (tid < CAND_SPAN) s_img[tid] = 0;

    const int n4 = (CAND_ROWS * p.no) >> 2;       // CAND_ROWS is a multiple of 4: a full chunk is whole 16-byte words
    const bool vec_ok = p.rows == nullptr && p.no <= 16 * CAND_NPRE &&
                        ((reinterpret_cast<uintptr_t>(p.pred) & 15) == 0);
    // two chunks in flight per workgroup (register prefetch, depth 2): with one, the four resident workgroups of a
    // CU keep ~87 KB outstanding, about half of what HBM latency x bandwidth asks for
    f4v preA[CAND_NPRE], preB[CAND_NPRE];
    auto prefetch = [&](f4v (&pre)[CAND_NPRE], int64_t ch) {
        const f4v* s4 = reinterpret_cast<const f4v*>(p.pred + ch * CAND_ROWS * p.no);
#pragma unroll
        for (int k = 0; k < CAND_NPRE; ++k) pre[k] = __builtin_nontemporal_load(s4 + min(tid + 256 * k, n4 - 1));
    };
    auto full_chunk = [&](int64_t ch) { return vec_ok && ch < c1 && (ch + 1) * CAND_ROWS <= total_rows; };
    bool haveA = full_chunk(c0), haveB = full_chunk(c0 + 1);
    if (haveA) prefetch(preA, c0);
    if (haveB) prefetch(preB, c0 + 1);

    constexpr int TPR = 256 / CAND_ROWS;         // threads per row
    const int pr = tid / TPR, sub = tid % TPR;
    const float* L = lds + pr * p.no;
    const int iters = p.multi_label ? (nc + TPR - 1) / TPR : 1;
    const bool mask_path = iters <= 64;          // nc <= 256
    uint64_t allow = ~0ull;                      // bit it: class sub + it*TPR passes the `classes` filter
    if (p.multi_label && mask_path && p.class_mask != nullptr) {
        allow = 0;
        for (int it = 0; it < iters; ++it) {
            const int c = sub + it * TPR;
            if (c < nc && class_ok(p.class_mask, c)) allow |= 1ull << it;
        }
    }

    auto do_chunk = [&](int64_t ch, f4v (&pre)[CAND_NPRE], bool& have_pre) {
        const int64_t r0 = ch * CAND_ROWS;
        const int nrows = (int)min((int64_t)CAND_ROWS, total_rows - r0);
        // ---- chunk -> LDS (the previous chunk ended on a barrier)
        if (have_pre) {
            f4v* l4 = reinterpret_cast<f4v*>(lds);
#pragma unroll
            for (int k = 0; k < CAND_NPRE; ++k)
                if (tid + 256 * k < n4) l4[tid + 256 * k] = pre[k];
        } else {
            const int nelem = nrows * p.no;
            if (p.rows == nullptr) {
                const float* src = p.pred + r0 * p.no;   // rows_per_img == N: flattened rows are contiguous
                for (int e = tid; e < nelem; e += 256) lds[e] = src[e];
            } else {
                for (int e = tid; e < nelem; e += 256) {
                    int rr = e / p.no, cc = e - rr * p.no;
                    int64_t r = r0 + rr;
                    int img = (int)(r / p.rows_per_img);
                    int prop = p.rows[r];
                    lds[e] = p.pred[((int64_t)img * p.N + prop) * p.no + cc];
                }
            }
        }
        have_pre = full_chunk(ch + 2);
        if (have_pre) prefetch(pre, ch + 2);
        if (tid == 0) { s_total = 0; s_local = 0; }
        __syncthreads();

        const bool row_ok = pr < nrows;
        const int64_t r = r0 + pr;
        // rows_per_img >= CAND_ROWS (or B == 1): a chunk touches at most two images -- one wave-uniform division
        const int img_c = (int)(r0 / p.rows_per_img);
        const int64_t img_c_end = (int64_t)(img_c + 1) * p.rows_per_img;
        const int img = row_ok ? img_c + (r >= img_c_end ? 1 : 0) : img_lo;
        const uint32_t rowpos = row_ok ? (uint32_t)(r - (r >= img_c_end ? img_c_end : img_c_end - p.rows_per_img)) : 0u;
        const float obj = row_ok ? L[4] : 0.0f;
        const bool pass = row_ok && (!p.require_obj || obj > p.ct);

        float best = -INFINITY;
        int besti = 0;
        if (!p.multi_label) {
            // best class: first maximal index (torch.max semantics), conf = cls*obj computed before the max
            for (int c = sub; c < nc; c += TPR) {
                float v = row_ok ? L[5 + c] * obj : -INFINITY;
                if (v > best) { best = v; besti = c; }
            }
            for (int off = 1; off < TPR; off <<= 1) {
                float ov = __shfl_xor(best, off);
                int oi = __shfl_xor(besti, off);
                if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
            }
        }

        // ---- pass 1: count.  Up to 64 classes per lane the hits are kept as a bit mask (branch-free loop, LDS reads
        // pipelined by the unroll) and pass 2 only visits the set bits.
        unsigned mine = 0;
        uint64_t hm = 0;
        if (!p.multi_label) {
            hm = (pass && sub == 0 && best > p.ct && class_ok(p.class_mask, besti)) ? 1ull : 0ull;
            mine = (unsigned)hm;
        } else if (mask_path) {
#pragma unroll 4
            for (int it = 0; it < iters; ++it) {
                const int c = sub + it * TPR;
                const float conf = L[5 + min(c, nc - 1)] * obj;
                const bool hit = pass & (c < nc) & (conf > p.ct);
                hm |= (uint64_t)(hit ? 1u : 0u) << it;
            }
            hm &= allow;
            mine = (unsigned)__popcll(hm);
        } else {
            for (int it = 0; it < iters; ++it) {
                int c = sub + it * TPR;
                float conf = (c < nc && row_ok) ? L[5 + c] * obj : 0.0f;
                mine += (pass && c < nc && conf > p.ct && class_ok(p.class_mask, c)) ? 1u : 0u;
            }
        }
        if (mine) {
            atomicAdd(&s_total, mine);
            if (img_in_lds) atomicAdd(&s_img[img - img_lo], mine);
            else atomicAdd(&p.counters[1 + img], mine);
        }
        __syncthreads();
        const unsigned tot = s_total;
        if (tot == 0) { __syncthreads(); return; }
        unsigned fill = s_fill;
        const bool direct = tot > CAND_STAGE;    // a dense chunk bypasses the staging buffer
        if (direct || fill + tot > CAND_STAGE) {
            // one reservation covers the staged hits and, for a dense chunk, this chunk's own
            if (tid == 0) s_base = atomicAdd(&p.counters[0], fill + (direct ? tot : 0u));
            __syncthreads();
            const unsigned fb = s_base;
            for (unsigned e = tid; e < fill * 6; e += 256)
                if (fb + e / 6 < p.capacity) p.det[(size_t)fb * 6 + e] = stage_det[e];
            for (unsigned e = tid; e < fill; e += 256)
                if (fb + e < p.capacity) p.keys[fb + e] = stage_key[e];
            __syncthreads();
            if (direct && tid == 0) s_base = fb + fill;   // read below, after the barrier that follows pass 2's setup
            fill = 0;
            if (tid == 0) s_fill = 0;
            __syncthreads();
        }
        const unsigned base = s_base;

        // box (general.py:297-321 with ratio = wh = 1, pad = 0): c -/+ size/2
        float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        if (row_ok && p.box_xyxy) {
            x1 = L[0]; y1 = L[1]; x2 = L[2]; y2 = L[3];
        } else if (row_ok) {
            float hw = L[2] / 2.0f, hh = L[3] / 2.0f;
            x1 = 1.0f * (L[0] - hw) + 0.0f;
            y1 = 1.0f * (L[1] - hh) + 0.0f;
            x2 = 1.0f * (L[0] + hw) + 0.0f;
            y2 = 1.0f * (L[1] + hh) + 0.0f;
        }
        // ---- pass 2: emit (order inside the workgroup is irrelevant: keys are unique and sorted afterwards)
        unsigned my_off = mine ? atomicAdd(&s_local, mine) : 0u;
        auto emit = [&](int c, float conf) {
            uint64_t seq = p.multi_label ? (uint64_t)rowpos * (uint64_t)nc + (uint64_t)c : (uint64_t)rowpos;
            uint64_t key;
            if (p.key_mode == 2) {
                uint32_t cb = ~__float_as_uint(conf);
                key = ((((uint64_t)img << p.cls_bits) | (uint64_t)c) << (32 + p.seq_bits)) | ((uint64_t)cb << p.seq_bits) |
                      (uint64_t)rowpos;
            } else if (p.order_by_seq) {
                key = ((uint64_t)img << p.seq_bits) | seq;
            } else {
                uint32_t cb = ~__float_as_uint(conf);   // conf > ct >= 0: bits monotonic
                key = ((uint64_t)img << (32 + p.seq_bits)) | ((uint64_t)cb << p.seq_bits) | seq;
            }
            if (direct) {
                uint32_t slot = base + my_off++;
                if (slot < p.capacity) {
                    float* d = p.det + (size_t)slot * 6;
                    d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = conf; d[5] = (float)c;
                    p.keys[slot] = key;
                }
            } else {
                unsigned idx = fill + my_off++;
                float* d = stage_det + idx * 6;
                d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = conf; d[5] = (float)c;
                stage_key[idx] = key;
            }
        };
        if (!p.multi_label) {
            if (hm) emit(besti, best);
        } else if (mask_path) {
            while (hm) {
                const int it = __ffsll((unsigned long long)hm) - 1;
                hm &= hm - 1;
                const int c = sub + it * TPR;
                emit(c, L[5 + c] * obj);
            }
        } else if (mine) {
            for (int it = 0; it < iters; ++it) {
                int c = sub + it * TPR;
                float conf = (c < nc && row_ok) ? L[5 + c] * obj : 0.0f;
                if (pass && c < nc && conf > p.ct && class_ok(p.class_mask, c)) emit(c, conf);
            }
        }
        if (!direct && tid == 0) s_fill = fill + tot;
        __syncthreads();
    };
    for (int64_t ch = c0; ch < c1; ch += 2) {
        do_chunk(ch, preA, haveA);
        if (ch + 1 < c1) do_chunk(ch + 1, preB, haveB);
    }

    // ---- final flush
    __syncthreads();
    const unsigned fill = s_fill;
    if (fill) {
        if (tid == 0) s_base = atomicAdd(&p.counters[0], fill);
        __syncthreads();
        const unsigned fb = s_base;
        for (unsigned e = tid; e < fill * 6; e += 256)
            if (fb + e / 6 < p.capacity) p.det[(size_t)fb * 6 + e] = stage_det[e];
        for (unsigned e = tid; e < fill; e += 256)
            if (fb + e < p.capacity) p.keys[fb + e] = stage_key[e];
    }
    if (img_in_lds && tid < CAND_SPAN && s_img[tid]) atomicAdd(&p.counters[1 + img_lo + tid], s_img[tid]);
}

static int bits_for(uint64_t n) {   // bits needed to represent values in [0, n)
    int b = 0;
    while (b < 63 && (1ull << b) < n) ++b;
    return b;
}

extern "C" int ayolo_nms_key_bits(int B, int rows_per_img, int nc_eff, int order_by_seq, int* seq_bits,
                                  int* total_bits) {
    int sb = bits_for((uint64_t)rows_per_img * (uint64_t)(nc_eff > 0 ? nc_eff : 1));
    int ib = bits_for((uint64_t)B);
    int tb = order_by_seq ? sb + ib : sb + 32 + ib;
    if (seq_bits) *seq_bits = sb;
    if (total_bits) *total_bits = tb;
    return tb <= 64 ? AYOLO_OK : AYOLO_EINVAL;
}

static int run_candidates(CandParams p, int no, ayolo_stream s) {
    AY_CHECK_ARG(p.rows_per_img >= CAND_ROWS || p.B == 1, "nms_candidates: fewer than %d rows per image with B > 1", CAND_ROWS);
    int64_t total_rows = (int64_t)p.B * p.rows_per_img;
    p.nchunks = cdiv64(total_rows, CAND_ROWS);
    // one resident wave of workgroups (4 per CU at ~38 KB LDS each); more chunks per workgroup = fewer reservations
    const char* env = getenv("AYOLO_CAND_CPB");   // test hook: force several chunks per workgroup at small sizes
    const int env_cpb = env ? atoi(env) : 0;
    p.cpb = env_cpb > 0 ? env_cpb : (int)std::max<int64_t>(1, cdiv64(p.nchunks, (int64_t)256 * 4));   /* MI355X: 256 CUs */
    int64_t nblk = cdiv64(p.nchunks, p.cpb);
    size_t lds = (size_t)CAND_STAGE * 32 + (size_t)CAND_ROWS * no * sizeof(float);
    AY_CHECK_ARG(lds <= 64 * 1024, "nms_candidates: no=%d too large", no);
    hipLaunchKernelGGL(k_candidates, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)s, p);
    AY_CHECK_LAUNCH("k_candidates");
    return AYOLO_OK;
}

extern "C" int ayolo_nms_candidates(const float* pred, int B, int N, int no, float conf_thres, int multi_label,
                                    int require_obj, const uint64_t* class_mask, const int32_t* rows,
                                    int rows_per_img, float* det, uint64_t* keys, uint32_t* counters,
                                    uint32_t capacity, int order_by_seq, ayolo_stream s) {
    AY_CHECK_ARG(pred && det && keys && counters, "nms_candidates: null pointer");
    AY_CHECK_ARG(B > 0 && N > 0 && no > 5, "nms_candidates: bad dims B=%d N=%d no=%d", B, N, no);
    if (rows == nullptr) rows_per_img = N;
    AY_CHECK_ARG(rows_per_img > 0, "nms_candidates: rows_per_img=%d", rows_per_img);
    const int nc = no - 5;
    int seq_bits, total_bits;
    if (ayolo_nms_key_bits(B, rows_per_img, multi_label ? nc : 1, order_by_seq, &seq_bits, &total_bits) != AYOLO_OK) {
        ayolo_set_error("nms_candidates: key needs %d bits (B*rows*nc too large)", total_bits);
        return AYOLO_EINVAL;
    }
    CandParams p{pred, B, N, no, conf_thres, multi_label, require_obj, class_mask, rows, rows_per_img,
                 det, keys, counters, capacity, seq_bits, order_by_seq, 0, 0, 0, 0, 0};
    return run_candidates(p, no, s);
}

// ---------------------------------------------------------------------------------------------------
// sort + gather
// ---------------------------------------------------------------------------------------------------
extern "C" int ayolo_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                                    uint32_t* vals_out, uint32_t n, int begin_bit, int end_bit, void* ws,
                                    size_t* ws_bytes, ayolo_stream s) {
    AY_CHECK_ARG(ws_bytes, "sort: ws_bytes null");
    size_t need = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (int)n,
                                                      begin_bit, end_bit, (hipStream_t)s);
    if (e != hipSuccess) { ayolo_set_error("sort size query: %s", hipGetErrorString(e)); return AYOLO_ELAUNCH; }
    if (ws == nullptr) { *ws_bytes = need; return AYOLO_OK; }
    if (*ws_bytes < need) { ayolo_set_error("sort: workspace %zu < %zu", *ws_bytes, need); return AYOLO_ENOSPC; }
    if (n == 0) return AYOLO_OK;
    e = hipcub::DeviceRadixSort::SortPairs(ws, need, keys_in, keys_out, vals_in, vals_out, (int)n, begin_bit,
                                           end_bit, (hipStream_t)s);
    if (e != hipSuccess) { ayolo_set_error("sort: %s", hipGetErrorString(e)); return AYOLO_ELAUNCH; }
    return AYOLO_OK;
}

// fills run as kernels on the caller's stream (see csrc/plan.hip: hipMemsetAsync ordering on ROCm 7.2)
__global__ void k_fill_u32(uint32_t* v, uint32_t n, uint32_t value) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = value;
}

__global__ void k_iota(uint32_t* v, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

extern "C" int ayolo_iota_u32(uint32_t* v, uint32_t n, ayolo_stream s) {
    if (n == 0) return AYOLO_OK;
    hipLaunchKernelGGL(k_iota, dim3(cdiv((int)n, 256)), dim3(256), 0, (hipStream_t)s, v, n);
    AY_CHECK_LAUNCH("k_iota");
    return AYOLO_OK;
}

__global__ void k_obj_keys(const float* pred, int B, int N, int no, int idx_bits, uint64_t* keys, uint32_t* vals) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * N) return;
    int b = (int)(t / N);
    uint32_t i = (uint32_t)(t - (int64_t)b * N);
    float obj = pred[t * no + 4];
    // descending by value, NaN-free input assumed; negative values ordered correctly via sign fix-up
    uint32_t u = __float_as_uint(obj);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-order key of the float
    u = ~u;                                            // descending
    keys[t] = ((uint64_t)b << (32 + idx_bits)) | ((uint64_t)u << idx_bits) | i;
    vals[t] = i;
}

extern "C" int ayolo_nms_obj_keys(const float* pred, int B, int N, int no, uint64_t* keys, uint32_t* vals,
                                  ayolo_stream s) {
    int idx_bits = bits_for((uint64_t)N), ib = bits_for((uint64_t)B);
    AY_CHECK_ARG(idx_bits + 32 + ib <= 64, "obj_keys: B*N too large for a 64-bit key");
    int64_t n = (int64_t)B * N;
    hipLaunchKernelGGL(k_obj_keys, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)s, pred, B, N, no,
                       idx_bits, keys, vals);
    AY_CHECK_LAUNCH("k_obj_keys");
    return AYOLO_OK;
}

__global__ void k_gather_rows(const float* src, const uint32_t* order, float* dst, uint32_t n, int width) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t row = t / width, c = t - row * width;
    if (row < n) dst[(size_t)row * width + c] = src[(size_t)order[row] * width + c];
}

extern "C" int ayolo_gather_rows(const float* src, const uint32_t* order, float* dst, uint32_t n, int width,
                                 ayolo_stream s) {
    if (n == 0) return AYOLO_OK;
    uint64_t tot = (uint64_t)n * width;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)s, src, order,
                       dst, n, width);
    AY_CHECK_LAUNCH("k_gather_rows");
    return AYOLO_OK;
}

// per-image max coordinate + 1 (torchvision batched_nms coordinate trick): out[b] = max(boxes) + 1
__global__ void k_seg_max_coord(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n, float* out) {
    int b = blockIdx.x;
    uint32_t off = seg_off[b], n = seg_n[b];
    float m = -INFINITY;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float* d = sdet + (size_t)(off + i) * 6;
        m = fmaxf(fmaxf(fmaxf(m, d[0]), fmaxf(d[1], d[2])), d[3]);
    }
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = red[0] + 1.0f;
}

extern "C" int ayolo_seg_max_coord(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n, int B,
                                   float* out, ayolo_stream s) {
    hipLaunchKernelGGL(k_seg_max_coord, dim3(B), dim3(256), 0, (hipStream_t)s, sdet, seg_off, seg_n, out);
    AY_CHECK_LAUNCH("k_seg_max_coord");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// Stage C: wave64 ballot IoU bit-matrix.  One wavefront per 64x64 tile; lane = column box; the 64 row
// boxes are broadcast through SGPRs (v_readlane), v_cmp writes the 64-bit row word directly.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// IOU 0: torchvision (area = w*h, pairs without overlap skipped).  IOU 1: TensorRT batchedNMSPlugin with
// isNormalized = 0 (allClassNMS jaccardOverlap: +1 on every extent, disjoint boxes intersect in the unit box).
template <int IOU>
__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ sdet, const uint32_t* seg_off,
                                                 const uint32_t* seg_n, const uint64_t* mask_off, float thr,
                                                 float offset_scale, const float* per_img_offset,
                                                 int class_aware, int ge_mode, uint64_t* mask) {
    const int b = blockIdx.z;
    const uint32_t n = seg_n[b];
    const uint32_t rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
    const uint32_t words = (n + 63) >> 6;
    const int lane = threadIdx.x;
    const float* base = sdet + (size_t)seg_off[b] * 6;
    const float scale = per_img_offset ? per_img_offset[b] : offset_scale;

    const uint32_t ri = rb * 64 + lane, cj = cb * 64 + lane;
    float rx1 = 0, ry1 = 0, rx2 = 0, ry2 = 0, rc = -1.f, cx1 = 0, cy1 = 0, cx2 = 0, cy2 = 0, cc = -2.f;
    if (ri < n) {
        const float* d = base + (size_t)ri * 6;
        rc = d[5];
        float o = rc * scale;
        rx1 = d[0] + o; ry1 = d[1] + o; rx2 = d[2] + o; ry2 = d[3] + o;
    }
    if (cj < n) {
        const float* d = base + (size_t)cj * 6;
        cc = d[5];
        float o = cc * scale;
        cx1 = d[0] + o; cy1 = d[1] + o; cx2 = d[2] + o; cy2 = d[3] + o;
    }
    float rarea, carea;
    if (IOU == 0) {
        rarea = (rx2 - rx1) * (ry2 - ry1);
        carea = (cx2 - cx1) * (cy2 - cy1);
    } else {   // bboxSize: 0 for an inverted box, else (w + 1) * (h + 1)
        rarea = (rx2 < rx1 || ry2 < ry1) ? 0.0f : ((rx2 - rx1) + 1.0f) * ((ry2 - ry1) + 1.0f);
        carea = (cx2 < cx1 || cy2 < cy1) ? 0.0f : ((cx2 - cx1) + 1.0f) * ((cy2 - cy1) + 1.0f);
    }
    const bool cvalid = cj < n;

    uint64_t myword = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const float ix1 = rl(rx1, i), iy1 = rl(ry1, i), ix2 = rl(rx2, i), iy2 = rl(ry2, i);
        const float ia = rl(rarea, i), ic = rl(rc, i);
        float xx1 = fmaxf(ix1, cx1), yy1 = fmaxf(iy1, cy1);
        float xx2 = fminf(ix2, cx2), yy2 = fminf(iy2, cy2);
        float w, h, inter;
        bool cand = cvalid && (rb * 64 + i) < n && cj > (rb * 64 + i);
        if (IOU == 0) {
            w = fmaxf(0.0f, xx2 - xx1); h = fmaxf(0.0f, yy2 - yy1);
            inter = w * h;
            cand = cand && inter > 0.0f;
        } else {
            const bool disjoint = (cx1 > ix2) | (cx2 < ix1) | (cy1 > iy2) | (cy2 < iy1);   // intersectBbox -> (0,0,0,0)
            w = disjoint ? 1.0f : (xx2 - xx1) + 1.0f;
            h = disjoint ? 1.0f : (yy2 - yy1) + 1.0f;
            inter = w * h;
            cand = cand && w > 0.0f && h > 0.0f;
        }
        if (class_aware) cand = cand && (ic == cc);
        uint64_t any = __ballot(cand);
        uint64_t word = 0;
        if (any) {   // wave-uniform: only divide where some pair intersects
            float ovr = inter / (ia + carea - inter);
            bool sup = cand && (ge_mode ? (ovr >= thr) : (ovr > thr));
            word = __ballot(sup);
        }
        if (lane == i) myword = word;
    }
    if (ri < n) mask[mask_off[b] + (uint64_t)ri * words + cb] = myword;
}

extern "C" int ayolo_nms_mask(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n,
                              const uint64_t* mask_off, int B, uint32_t max_n, float thr_f, float offset_scale,
                              const float* per_img_offset, int class_aware, uint64_t* mask, ayolo_stream s) {
    if (max_n == 0 || B == 0) return AYOLO_OK;
    unsigned nb = (max_n + 63) / 64;
    AY_CHECK_ARG(nb <= 65535 && B <= 65535, "nms_mask: grid too large");
    hipLaunchKernelGGL(k_nms_mask<0>, dim3(nb, nb, B), dim3(64), 0, (hipStream_t)s, sdet, seg_off, seg_n, mask_off,
                       thr_f, offset_scale, per_img_offset, class_aware, 0, mask);
    AY_CHECK_LAUNCH("k_nms_mask");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// Stage D: greedy scan.  One wavefront per image, removed-set in LDS, 64-box blocks resolved in SGPRs.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rl64(uint64_t v, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void k_nms_reduce(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n,
                                                   const uint64_t* mask_off, const uint64_t* mask, uint32_t max_out,
                                                   float* out, int32_t* out_idx, uint32_t* out_count) {
    extern __shared__ __attribute__((aligned(16))) uint64_t remv[];
    const int b = blockIdx.x;
    const uint32_t n = seg_n[b];
    const uint32_t words = (n + 63) >> 6;
    const int lane = threadIdx.x;
    const uint64_t* M = mask + mask_off[b];
    for (uint32_t w = lane; w < words; w += 64) remv[w] = 0;
    __syncthreads();
    uint32_t kept = 0;
    int32_t* oidx = out_idx + (size_t)b * max_out;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t blk = 0; blk < words && kept < max_out; ++blk) {
        const uint32_t i = blk * 64 + lane;
        uint64_t d = (i < n) ? M[(uint64_t)i * words + blk] : 0ull;
        uint64_t valid = (blk * 64 + 64 <= n) ? ~0ull : ((1ull << (n - blk * 64)) - 1ull);
        uint64_t alive = rfl64(~remv[blk] & valid);
        uint64_t keepmask = 0;
        uint32_t k2 = kept;
        while (alive != 0 && k2 < max_out) {
            int t = __ffsll((unsigned long long)alive) - 1;
            keepmask |= 1ull << t;
            ++k2;
            uint64_t dt = rl64(d, t);
            alive &= ~(dt | (1ull << t));
        }
        if ((keepmask >> lane) & 1ull) oidx[kept + (uint32_t)__popcll(keepmask & lt_mask)] = (int32_t)i;
        kept = k2;
        if (kept >= max_out) break;
        // OR the rows of this block's kept boxes into the removed-set for later blocks
        for (uint32_t w0 = blk + 1; w0 < words; w0 += 64) {
            uint32_t w = w0 + lane;
            uint64_t acc = 0;
            uint64_t km = keepmask;
            while (km != 0) {
                int t = __ffsll((unsigned long long)km) - 1;
                km &= km - 1;
                if (w < words) acc |= M[(uint64_t)(blk * 64 + t) * words + w];
            }
            if (w < words) remv[w] |= acc;
        }
        __syncthreads();
    }
    if (lane == 0) out_count[b] = kept;
    __syncthreads();
    // gather output rows
    const float* base = sdet + (size_t)seg_off[b] * 6;
    for (uint32_t t = lane; t < kept * 6; t += 64) {
        uint32_t k = t / 6, c = t - k * 6;
        out[((size_t)b * max_out + k) * 6 + c] = base[(size_t)oidx[k] * 6 + c];
    }
}

extern "C" int ayolo_nms_reduce(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n,
                                const uint64_t* mask_off, const uint64_t* mask, int B, uint32_t max_out, float* out,
                                int32_t* out_idx, uint32_t* out_count, uint32_t max_n, ayolo_stream s) {
    if (B == 0) return AYOLO_OK;
    size_t lds = (size_t)((max_n + 63) / 64) * 8;
    if (lds < 8) lds = 8;
    AY_CHECK_ARG(lds <= 64 * 1024, "nms_reduce: %u boxes per image exceeds the 524288 limit", max_n);
    hipLaunchKernelGGL(k_nms_reduce, dim3(B), dim3(64), lds, (hipStream_t)s, sdet, seg_off, seg_n, mask_off, mask,
                       max_out, out, out_idx, out_count);
    AY_CHECK_LAUNCH("k_nms_reduce");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// Per-class route of the `nms` branch (metrics.py:383-388, boxes offset by cls * 4096): regroup the conf-sorted
// image segments into (image, class) segments, and merge the kept rows back into per-image confidence order.
// These four entry points replace ~45 torch launches (sort / bincount / index / cumsum chains) with 8.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int seg_of(const uint32_t* off, int n, uint32_t i) {
    // largest b in [0, n) with off[b] <= i (off ascending, off[0] == 0, duplicates = empty segments)
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t float_order(float f) {   // monotonic map float -> uint32 (NaN above +inf)
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#define CK_RPT 4      // rows per thread: 1024-row tiles, one pair of span atomics per workgroup
__global__ __launch_bounds__(256) void k_class_keys(const float* sdet, const uint32_t* seg_off, const uint32_t* sel_off,
                                                    int B, int nc, uint32_t tot, float* rows1, uint64_t* keys,
                                                    uint32_t* vals, uint32_t* span) {
    __shared__ uint32_t s_hi[4], s_lo[4];
    uint32_t hi = 0, lo = 0;      // order codes of max(coord) and max(-coord); 0 is below every float
#pragma unroll
    for (int k = 0; k < CK_RPT; ++k) {
        const uint32_t i = (blockIdx.x * CK_RPT + k) * 256 + threadIdx.x;
        if (i >= tot) continue;
        const int b = seg_of(sel_off, B, i);
        const size_t g = (size_t)seg_off[b] + (i - sel_off[b]);
        const float2* src = reinterpret_cast<const float2*>(sdet + g * 6);
        float2 a = src[0], c = src[1], d = src[2];
        float2* dst = reinterpret_cast<float2*>(rows1 + (size_t)i * 6);
        dst[0] = a; dst[1] = c; dst[2] = d;
        keys[i] = (uint64_t)((int64_t)b * nc + (int64_t)d.y);
        vals[i] = i;
        hi = max(max(hi, float_order(a.x)), max(float_order(a.y), max(float_order(c.x), float_order(c.y))));
        lo = max(max(lo, float_order(-a.x)), max(float_order(-a.y), max(float_order(-c.x), float_order(-c.y))));
    }
    for (int off = 32; off > 0; off >>= 1) {
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
        lo = max(lo, (uint32_t)__shfl_xor((int)lo, off));
    }
    if ((threadIdx.x & 63) == 0) { s_hi[threadIdx.x >> 6] = hi; s_lo[threadIdx.x >> 6] = lo; }
    __syncthreads();
    // same-address atomics serialise in L2 (~12 ns each): one pair per workgroup, and only for a new maximum
    if (threadIdx.x == 0) {
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        lo = max(max(s_lo[0], s_lo[1]), max(s_lo[2], s_lo[3]));
        if (hi > __hip_atomic_load(&span[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&span[0], hi);
        if (lo > __hip_atomic_load(&span[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&span[1], lo);
    }
}

extern "C" int ayolo_nms_class_keys(const float* sdet, const uint32_t* seg_off, const uint32_t* sel_off, int B, int nc,
                                    uint32_t tot, float* rows1, uint64_t* keys, uint32_t* vals, uint32_t* span,
                                    ayolo_stream s) {
    AY_CHECK_ARG(sdet && seg_off && sel_off && rows1 && keys && vals && span, "nms_class_keys: null pointer");
    AY_CHECK_ARG(B > 0 && nc > 0, "nms_class_keys: B=%d nc=%d", B, nc);
    if (tot == 0) return AYOLO_OK;
    hipLaunchKernelGGL(k_class_keys, dim3(cdiv((int)tot, 256 * CK_RPT)), dim3(256), 0, (hipStream_t)s, sdet, seg_off, sel_off, B,
                       nc, tot, rows1, keys, vals, span);
    AY_CHECK_LAUNCH("k_class_keys");
    return AYOLO_OK;
}

// one workgroup: segment s = run of key == s in the sorted keys; exclusive scans give the row and mask offsets
__global__ __launch_bounds__(1024) void k_class_layout(const uint64_t* keys_sorted, uint32_t tot, int nseg,
                                                       uint32_t* seg_off2, uint32_t* seg_n2, uint64_t* mask_off,
                                                       int64_t* summary) {
    __shared__ unsigned long long s_scan[1024];
    __shared__ unsigned long long s_carry;
    __shared__ unsigned s_max;
    const int tid = threadIdx.x;
    if (tid == 0) { s_carry = 0; s_max = 0; }
    __syncthreads();
    for (int base = 0; base < nseg; base += 1024) {
        const int sg = base + tid;
        uint32_t start = 0, n = 0;
        if (sg < nseg) {
            // lower_bound(keys, sg) and lower_bound(keys, sg + 1)
            uint32_t lo = 0, hi = tot;
            while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (keys_sorted[m] < (uint64_t)sg) lo = m + 1; else hi = m; }
            start = lo;
            hi = tot;
            while (lo < hi) { uint32_t m = (lo + hi) >> 1; if (keys_sorted[m] < (uint64_t)sg + 1) lo = m + 1; else hi = m; }
            n = lo - start;
            seg_off2[sg] = start;
            seg_n2[sg] = n;
            atomicMax(&s_max, n);
        }
        unsigned long long w = (unsigned long long)n * ((n + 63) / 64);
        s_scan[tid] = w;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {       // Hillis-Steele inclusive scan
            unsigned long long add = tid >= off ? s_scan[tid - off] : 0ull;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        const unsigned long long carry = s_carry;
        if (sg < nseg) mask_off[sg] = carry + s_scan[tid] - w;
        __syncthreads();
        if (tid == 1023) s_carry = carry + s_scan[1023];
        __syncthreads();
    }
    if (tid == 0) { summary[0] = (int64_t)s_max; summary[1] = (int64_t)s_carry; }
}

extern "C" int ayolo_nms_class_layout(const uint64_t* keys_sorted, uint32_t tot, int nseg, uint32_t* seg_off2,
                                      uint32_t* seg_n2, uint64_t* mask_off, int64_t* summary, ayolo_stream s) {
    AY_CHECK_ARG(keys_sorted && seg_off2 && seg_n2 && mask_off && summary, "nms_class_layout: null pointer");
    AY_CHECK_ARG(nseg > 0, "nms_class_layout: nseg=%d", nseg);
    hipLaunchKernelGGL(k_class_layout, dim3(1), dim3(1024), 0, (hipStream_t)s, keys_sorted, tot, nseg, seg_off2, seg_n2,
                       mask_off, summary);
    AY_CHECK_LAUNCH("k_class_layout");
    return AYOLO_OK;
}

__global__ void k_class_mark(const int32_t* out_idx, const uint32_t* out_count, const uint32_t* seg_off2,
                             const uint32_t* perm, int nseg, uint32_t max_out, uint32_t* flags) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sg = t / max_out, k = t - sg * max_out;
    if (sg < (uint32_t)nseg && k < out_count[sg]) flags[perm[seg_off2[sg] + (uint32_t)out_idx[(size_t)sg * max_out + k]]] = 1u;
}

__global__ __launch_bounds__(256) void k_class_emit(const float* rows1, const uint32_t* flags, const uint32_t* scan,
                                                    const uint32_t* sel_off, int B, uint32_t tot, uint32_t max_det,
                                                    float* out, uint32_t* kept) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < (uint32_t)B) {
        const uint32_t end_all = scan[tot - 1] + flags[tot - 1];
        const uint32_t a = sel_off[i], b = sel_off[i + 1];
        const uint32_t va = a < tot ? scan[a] : end_all, vb = b < tot ? scan[b] : end_all;
        kept[i] = min(max_det, vb - va);
    }
    if (i >= tot || !flags[i]) return;
    const int b = seg_of(sel_off, B, i);
    const uint32_t within = scan[i] - scan[sel_off[b]];
    if (within >= max_det) return;
    const float2* src = reinterpret_cast<const float2*>(rows1 + (size_t)i * 6);
    float2* dst = reinterpret_cast<float2*>(out + ((size_t)b * max_det + within) * 6);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

extern "C" int ayolo_nms_class_merge(const float* rows1, const int32_t* out_idx, const uint32_t* out_count,
                                     const uint32_t* seg_off2, const uint32_t* perm, int nseg, uint32_t max_out,
                                     const uint32_t* sel_off, int B, uint32_t tot, uint32_t max_det, uint32_t* flags,
                                     uint32_t* scan, float* out, uint32_t* kept, void* ws, size_t* ws_bytes,
                                     ayolo_stream s) {
    AY_CHECK_ARG(ws_bytes, "nms_class_merge: ws_bytes null");
    size_t need = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, flags, scan, (int)tot, (hipStream_t)s);
    if (e != hipSuccess) { ayolo_set_error("nms_class_merge: scan size query: %s", hipGetErrorString(e)); return AYOLO_ELAUNCH; }
    if (ws == nullptr) { *ws_bytes = need; return AYOLO_OK; }
    if (*ws_bytes < need) { ayolo_set_error("nms_class_merge: workspace %zu < %zu", *ws_bytes, need); return AYOLO_ENOSPC; }
    AY_CHECK_ARG(rows1 && out_idx && out_count && seg_off2 && perm && sel_off && flags && scan && out && kept,
                 "nms_class_merge: null pointer");
    AY_CHECK_ARG(tot > 0 && B > 0 && nseg > 0 && max_out > 0 && max_det > 0,
                 "nms_class_merge: bad sizes");
    hipStream_t st = (hipStream_t)s;
    const uint64_t nthreads = (uint64_t)nseg * max_out;
    hipLaunchKernelGGL(k_class_mark, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, out_idx, out_count,
                       seg_off2, perm, nseg, max_out, flags);
    AY_CHECK_LAUNCH("k_class_mark");
    e = hipcub::DeviceScan::ExclusiveSum(ws, need, flags, scan, (int)tot, st);
    if (e != hipSuccess) { ayolo_set_error("nms_class_merge: scan: %s", hipGetErrorString(e)); return AYOLO_ELAUNCH; }
    const uint32_t nthr = tot > (uint32_t)B ? tot : (uint32_t)B;
    hipLaunchKernelGGL(k_class_emit, dim3(cdiv((int)nthr, 256)), dim3(256), 0, st, rows1, flags, scan, sel_off, B, tot,
                       max_det, out, kept);
    AY_CHECK_LAUNCH("k_class_emit");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fixed-shape batched NMS with the TensorRT BatchedNMS_TRT contract the reference builds into its engines
// (scripts/model_converter/model_converter.py:268-388: shareLocation 1, backgroundLabelId -1, isNormalized 0,
// clipBoxes 0; outputs consumed by train_utils.py:262-283).  Every buffer has a size fixed by (B, N, nc, topK,
// keepTopK, capacity); no host read-back anywhere, so the sequence can be captured in a hipGraph.
//   candidates (key = image | class | ~score | row) -> radix sort -> per (image, class) top-K layout ->
//   IoU bit matrix (plugin jaccard) -> greedy scan -> keys (image | ~score | class | rank) -> sort -> emit
// ---------------------------------------------------------------------------------------------------
extern "C" int ayolo_trt_nms_key_bits(int B, int N, int nc, int* row_bits, int* cls_bits, int* img_bits) {
    int rb = bits_for((uint64_t)N), cb = bits_for((uint64_t)nc + 1), ib = bits_for((uint64_t)B + 1);
    if (row_bits) *row_bits = rb;
    if (cls_bits) *cls_bits = cb;
    if (img_bits) *img_bits = ib;
    return rb + 32 + cb + ib <= 64 ? AYOLO_OK : AYOLO_EINVAL;
}

__global__ void k_fill_u64(uint64_t* v, uint64_t n, uint64_t value) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = value;
}

extern "C" int ayolo_trt_nms_candidates(const float* pred, int B, int N, int no, float score_thres, int box_xyxy,
                                        float* det, uint64_t* keys, uint32_t* counters, uint32_t capacity,
                                        ayolo_stream s) {
    AY_CHECK_ARG(pred && det && keys && counters, "trt_nms_candidates: null pointer");
    AY_CHECK_ARG(B > 0 && N > 0 && no > 5 && capacity > 0, "trt_nms_candidates: bad dims B=%d N=%d no=%d", B, N, no);
    int rb, cb, ib;
    if (ayolo_trt_nms_key_bits(B, N, no - 5, &rb, &cb, &ib) != AYOLO_OK) {
        ayolo_set_error("trt_nms_candidates: key needs %d bits", rb + 32 + cb + ib);
        return AYOLO_EINVAL;
    }
    hipStream_t st = (hipStream_t)s;
    // unused slots sort behind every candidate: all-ones class field >= nc by construction of cls_bits
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)cdiv64(capacity, 256)), dim3(256), 0, st, keys, (uint64_t)capacity, ~0ull);
    AY_CHECK_LAUNCH("k_fill_u64");
    hipLaunchKernelGGL(k_fill_u32, dim3(cdiv(1 + B, 256)), dim3(256), 0, st, counters, (uint32_t)(1 + B), 0u);
    AY_CHECK_LAUNCH("k_fill_u32");
    CandParams p{pred, B, N, no, score_thres, 1, 0, nullptr, nullptr, N, det, keys, counters, capacity, rb, 0, 0, 0, 2, cb,
                 box_xyxy};
    return run_candidates(p, no, s);
}

// segment sg = (image, class): the first min(count, top_k) sorted keys whose high field equals sg
__global__ void k_trt_layout(const uint64_t* keys_sorted, uint32_t cap, int B, int nc, int cls_bits, int shift,
                             uint32_t top_k, uint32_t* seg_off2, uint32_t* seg_n2, uint64_t* mask_off) {
    const int sg = blockIdx.x * blockDim.x + threadIdx.x;
    if (sg >= B * nc) return;
    const uint64_t pre = (((uint64_t)(sg / nc)) << cls_bits) | (uint64_t)(sg % nc);
    uint32_t lo = 0, hi = cap;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if ((keys_sorted[m] >> shift) < pre) lo = m + 1; else hi = m; }
    const uint32_t start = lo;
    hi = cap;
    while (lo < hi) { uint32_t m = (lo + hi) >> 1; if ((keys_sorted[m] >> shift) <= pre) lo = m + 1; else hi = m; }
    seg_off2[sg] = start;
    seg_n2[sg] = min(lo - start, top_k);
    mask_off[sg] = (uint64_t)sg * top_k * ((top_k + 63) / 64);
}

extern "C" int ayolo_trt_nms_layout(const uint64_t* keys_sorted, uint32_t capacity, int B, int N, int nc,
                                    uint32_t top_k, uint32_t* seg_off2, uint32_t* seg_n2, uint64_t* mask_off,
                                    ayolo_stream s) {
    AY_CHECK_ARG(keys_sorted && seg_off2 && seg_n2 && mask_off && top_k > 0, "trt_nms_layout: bad args");
    int rb, cb, ib;
    AY_CHECK_ARG(ayolo_trt_nms_key_bits(B, N, nc, &rb, &cb, &ib) == AYOLO_OK, "trt_nms_layout: key too wide");
    hipLaunchKernelGGL(k_trt_layout, dim3(cdiv(B * nc, 256)), dim3(256), 0, (hipStream_t)s, keys_sorted, capacity, B, nc,
                       cb, 32 + rb, top_k, seg_off2, seg_n2, mask_off);
    AY_CHECK_LAUNCH("k_trt_layout");
    return AYOLO_OK;
}

extern "C" int ayolo_trt_nms_mask(const float* sdet, const uint32_t* seg_off, const uint32_t* seg_n,
                                  const uint64_t* mask_off, int nseg, uint32_t max_n, float iou_thres, uint64_t* mask,
                                  ayolo_stream s) {
    if (max_n == 0 || nseg == 0) return AYOLO_OK;
    unsigned nb = (max_n + 63) / 64;
    AY_CHECK_ARG(nb <= 65535 && nseg <= 65535, "trt_nms_mask: grid too large");
    hipLaunchKernelGGL(k_nms_mask<1>, dim3(nb, nb, nseg), dim3(64), 0, (hipStream_t)s, sdet, seg_off, seg_n, mask_off,
                       iou_thres, 0.0f, (const float*)nullptr, 0, 0, mask);
    AY_CHECK_LAUNCH("k_nms_mask<trt>");
    return AYOLO_OK;
}

// kept rows of every (image, class) -> key image | ~score | class | rank, value = row of `out`; the rest ~0
__global__ void k_trt_final_keys(const float* out, const uint32_t* out_count, int nseg, uint32_t max_out, int nc,
                                 int cls_bits, int rank_bits, uint64_t* fkeys, uint32_t* fvals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)nseg * max_out) return;
    const uint32_t sg = t / max_out, k = t - sg * max_out;
    uint64_t key = ~0ull;
    if (k < out_count[sg]) {
        const uint32_t sb = ~__float_as_uint(out[(size_t)t * 6 + 4]);
        key = ((((uint64_t)(sg / nc) << 32) | sb) << (cls_bits + rank_bits)) | ((uint64_t)(sg % nc) << rank_bits) | k;
    }
    fkeys[t] = key;
    fvals[t] = t;
}

__global__ void k_trt_emit(const uint64_t* fkeys_sorted, const uint32_t* fvals_sorted, uint32_t E, const float* out,
                           int shift, uint32_t keep, int32_t* num_det, float* boxes, float* scores, float* classes) {
    __shared__ uint32_t s_start, s_cnt;
    const uint64_t b = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = E;
        while (lo < hi) { uint32_t m = (lo + hi) >> 1; if ((fkeys_sorted[m] >> shift) < b) lo = m + 1; else hi = m; }
        s_start = lo;
        hi = E;
        while (lo < hi) { uint32_t m = (lo + hi) >> 1; if ((fkeys_sorted[m] >> shift) <= b) lo = m + 1; else hi = m; }
        s_cnt = min(lo - s_start, keep);
        num_det[b] = (int32_t)s_cnt;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < keep; j += blockDim.x) {
        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, -1.f};          // gatherNMSOutputs padding: box 0, score 0, class -1
        if (j < s_cnt) {
            const float* r = out + (size_t)fvals_sorted[s_start + j] * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = r[c];
        }
        float* bo = boxes + (b * keep + j) * 4;
        bo[0] = v[0]; bo[1] = v[1]; bo[2] = v[2]; bo[3] = v[3];
        scores[b * keep + j] = v[4];
        classes[b * keep + j] = v[5];
    }
}

extern "C" int ayolo_trt_nms_final_keys(const float* out, const uint32_t* out_count, int B, int nc, uint32_t max_out,
                                        uint64_t* fkeys, uint32_t* fvals, int* total_bits, ayolo_stream s) {
    const int cb = bits_for((uint64_t)nc + 1), kb = bits_for((uint64_t)max_out), ib = bits_for((uint64_t)B + 1);
    AY_CHECK_ARG(ib + 32 + cb + kb <= 64, "trt_nms_final_keys: key too wide");
    if (total_bits) *total_bits = ib + 32 + cb + kb;
    if (out == nullptr) return AYOLO_OK;                          // size query
    AY_CHECK_ARG(out_count && fkeys && fvals && max_out > 0, "trt_nms_final_keys: bad args");
    const uint64_t E = (uint64_t)B * nc * max_out;
    hipLaunchKernelGGL(k_trt_final_keys, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, (hipStream_t)s, out, out_count,
                       B * nc, max_out, nc, cb, kb, fkeys, fvals);
    AY_CHECK_LAUNCH("k_trt_final_keys");
    return AYOLO_OK;
}

extern "C" int ayolo_trt_nms_emit(const uint64_t* fkeys_sorted, const uint32_t* fvals_sorted, const float* out, int B,
                                  int nc, uint32_t max_out, uint32_t keep_top_k, int32_t* num_det, float* boxes,
                                  float* scores, float* classes, ayolo_stream s) {
    AY_CHECK_ARG(fkeys_sorted && fvals_sorted && out && num_det && boxes && scores && classes && keep_top_k > 0,
                 "trt_nms_emit: bad args");
    const int cb = bits_for((uint64_t)nc + 1), kb = bits_for((uint64_t)max_out);
    hipLaunchKernelGGL(k_trt_emit, dim3(B), dim3(128), 0, (hipStream_t)s, fkeys_sorted, fvals_sorted,
                       (uint32_t)((uint64_t)B * nc * max_out), out, 32 + cb + kb, keep_top_k, num_det, boxes, scores,
                       classes);
    AY_CHECK_LAUNCH("k_trt_emit");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// Dense IoU (metrics.py:138-164) and the fast / matrix NMS column reductions
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_pair(float ax1, float ay1, float ax2, float ay2, float aa, float bx1, float by1,
                                          float bx2, float by2, float ba) {
    float rx = fminf(ax2, bx2) - fmaxf(ax1, bx1);
    float ry = fminf(ay2, by2) - fmaxf(ay1, by1);
    rx = rx > 0.0f ? rx : 0.0f;
    ry = ry > 0.0f ? ry : 0.0f;
    float inter = rx * ry;
    return inter / (aa + ba - inter);
}

__global__ void k_box_iou(const float* a, int64_t N, const float* b, int64_t M, float* out) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t i = blockIdx.y;
    if (j >= M) return;
    const float* p = a + i * 4;
    const float* q = b + j * 4;
    float aa = (p[2] - p[0]) * (p[3] - p[1]);
    float ba = (q[2] - q[0]) * (q[3] - q[1]);
    out[i * M + j] = iou_pair(p[0], p[1], p[2], p[3], aa, q[0], q[1], q[2], q[3], ba);
}

extern "C" int ayolo_box_iou(const float* a, int64_t N, const float* b, int64_t M, float* out, ayolo_stream s) {
    if (N == 0 || M == 0) return AYOLO_OK;
    AY_CHECK_ARG(N <= 65535 * 32768LL, "box_iou: N too large");
    // rows in grid.y (<= 65535): loop in chunks
    for (int64_t i0 = 0; i0 < N; i0 += 65535) {
        int64_t rows = N - i0 < 65535 ? N - i0 : 65535;
        hipLaunchKernelGGL(k_box_iou, dim3((unsigned)cdiv64(M, 256), (unsigned)rows), dim3(256), 0, (hipStream_t)s,
                           a + i0 * 4, rows, b, M, out + i0 * M);
    }
    AY_CHECK_LAUNCH("k_box_iou");
    return AYOLO_OK;
}

// colmax[j] = max_{i<j} iou(i, j) over boxes offset by cls*scale; zero-initialised by the caller (the
// upper-triangular matrix's zeros take part in the max, so colmax >= 0; NaN propagates as in torch).
#define CM_ROWS 256
__global__ __launch_bounds__(64) void k_iou_colmax(const float* boxes, const float* cls, float scale, uint32_t n,
                                                   int* colmax_bits) {
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    const uint32_t i0 = blockIdx.y * CM_ROWS;
    if (i0 >= (blockIdx.x + 1) * 64) return;   // rows all >= every column of this block
    float bx1 = 0, by1 = 0, bx2 = 0, by2 = 0, ba = 0;
    if (j < n) {
        float o = cls ? cls[j] * scale : 0.0f;
        bx1 = boxes[j * 4 + 0] + o; by1 = boxes[j * 4 + 1] + o; bx2 = boxes[j * 4 + 2] + o; by2 = boxes[j * 4 + 3] + o;
        ba = (bx2 - bx1) * (by2 - by1);
    }
    float m = 0.0f;
    bool isnan_ = false;
    uint32_t i1 = min(i0 + CM_ROWS, n);
    for (uint32_t i = i0; i < i1; ++i) {
        float o = cls ? cls[i] * scale : 0.0f;   // uniform loads
        float ax1 = boxes[i * 4 + 0] + o, ay1 = boxes[i * 4 + 1] + o, ax2 = boxes[i * 4 + 2] + o, ay2 = boxes[i * 4 + 3] + o;
        float aa = (ax2 - ax1) * (ay2 - ay1);
        if (i < j && j < n) {
            float v = iou_pair(ax1, ay1, ax2, ay2, aa, bx1, by1, bx2, by2, ba);
            if (v != v) isnan_ = true;
            else if (v > m) m = v;
        }
    }
    if (j < n) {
        if (isnan_) atomicMax(&colmax_bits[j], 0x7fc00000);
        else atomicMax(&colmax_bits[j], __float_as_int(m));
    }
}

extern "C" int ayolo_iou_colmax(const float* boxes, const float* cls, float offset_scale, uint32_t n, float* colmax,
                                ayolo_stream s) {
    if (n == 0) return AYOLO_OK;
    hipLaunchKernelGGL(k_fill_u32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, (uint32_t*)colmax, n, 0u);
    hipLaunchKernelGGL(k_iou_colmax, dim3((n + 63) / 64, (n + CM_ROWS - 1) / CM_ROWS), dim3(64), 0, (hipStream_t)s,
                       boxes, cls, offset_scale, n, (int*)colmax);
    AY_CHECK_LAUNCH("k_iou_colmax");
    return AYOLO_OK;
}

// decay[j] = min_i exp(-(iou(i,j)^2 - colmax[i]^2) / 0.5), iou(i,j) = 0 for i >= j (metrics.py:408-417).
__global__ __launch_bounds__(64) void k_matrix_decay(const float* boxes, const float* cls, float scale, uint32_t n,
                                                     const float* colmax, int* decay_bits) {
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    const uint32_t i0 = blockIdx.y * CM_ROWS;
    float bx1 = 0, by1 = 0, bx2 = 0, by2 = 0, ba = 0;
    if (j < n) {
        float o = cls ? cls[j] * scale : 0.0f;
        bx1 = boxes[j * 4 + 0] + o; by1 = boxes[j * 4 + 1] + o; bx2 = boxes[j * 4 + 2] + o; by2 = boxes[j * 4 + 3] + o;
        ba = (bx2 - bx1) * (by2 - by1);
    }
    float mn = INFINITY;
    uint32_t i1 = min(i0 + CM_ROWS, n);
    for (uint32_t i = i0; i < i1; ++i) {
        float o = cls ? cls[i] * scale : 0.0f;
        float ax1 = boxes[i * 4 + 0] + o, ay1 = boxes[i * 4 + 1] + o, ax2 = boxes[i * 4 + 2] + o, ay2 = boxes[i * 4 + 3] + o;
        float aa = (ax2 - ax1) * (ay2 - ay1);
        float v = 0.0f;
        if (i < j && j < n) v = iou_pair(ax1, ay1, ax2, ay2, aa, bx1, by1, bx2, by2, ba);
        float mi = colmax[i];
        float e = expf(-(v * v - mi * mi) / 0.5f);
        mn = fminf(mn, e);
    }
    if (j < n) atomicMin(&decay_bits[j], __float_as_int(mn));   // decay > 0: int order == float order
}

extern "C" int ayolo_matrix_nms_decay(const float* boxes, const float* cls, float offset_scale, uint32_t n,
                                      const float* colmax, float* decay, ayolo_stream s) {
    if (n == 0) return AYOLO_OK;
    hipLaunchKernelGGL(k_fill_u32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, (uint32_t*)decay, n, 0x7f7f7f7fu);   // large positive float
    hipLaunchKernelGGL(k_matrix_decay, dim3((n + 63) / 64, (n + CM_ROWS - 1) / CM_ROWS), dim3(64), 0, (hipStream_t)s,
                       boxes, cls, offset_scale, n, colmax, (int*)decay);
    AY_CHECK_LAUNCH("k_matrix_decay");
    return AYOLO_OK;
}

// merge_nms (metrics.py:418-435): for each kept row k: weights = (iou(box_off[k], box_off[:]) > thr) * scores;
// merged[k] = (weights @ boxes) / sum(weights); redundant[k] = count(iou > thr) > 1.
__global__ __launch_bounds__(256) void k_merge_boxes(const float* det, uint32_t n, float scale, const int32_t* kept,
                                                     uint32_t nk, float thr, float* merged, int32_t* redundant) {
    const uint32_t k = blockIdx.x;
    if (k >= nk) return;
    const float* dk = det + (size_t)kept[k] * 6;
    float ok = dk[5] * scale;
    float ax1 = dk[0] + ok, ay1 = dk[1] + ok, ax2 = dk[2] + ok, ay2 = dk[3] + ok;
    float aa = (ax2 - ax1) * (ay2 - ay1);
    float s[5] = {0, 0, 0, 0, 0};
    int cnt = 0;
    for (uint32_t j = threadIdx.x; j < n; j += 256) {
        const float* d = det + (size_t)j * 6;
        float o = d[5] * scale;
        float bx1 = d[0] + o, by1 = d[1] + o, bx2 = d[2] + o, by2 = d[3] + o;
        float ba = (bx2 - bx1) * (by2 - by1);
        float v = iou_pair(ax1, ay1, ax2, ay2, aa, bx1, by1, bx2, by2, ba);
        if (v > thr) {
            float w = d[4];
            s[0] += w * d[0]; s[1] += w * d[1]; s[2] += w * d[2]; s[3] += w * d[3]; s[4] += w;
            ++cnt;
        }
    }
    __shared__ float red[5][256];
    __shared__ int redc[256];
    for (int q = 0; q < 5; ++q) red[q][threadIdx.x] = s[q];
    redc[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + st];
            redc[threadIdx.x] += redc[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) merged[k * 4 + threadIdx.x] = red[threadIdx.x][0] / red[4][0];
    if (threadIdx.x == 0) redundant[k] = redc[0] > 1;
}

extern "C" int ayolo_merge_boxes(const float* det, uint32_t n, float offset_scale, const int32_t* kept, uint32_t nk,
                                 float thr_f32, float* merged, int32_t* redundant, ayolo_stream s) {
    if (nk == 0) return AYOLO_OK;
    hipLaunchKernelGGL(k_merge_boxes, dim3(nk), dim3(256), 0, (hipStream_t)s, det, n, offset_scale, kept, nk, thr_f32,
                       merged, redundant);
    AY_CHECK_LAUNCH("k_merge_boxes");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// Validator matching (scripts/utils/train_utils.py:294-333 `YoloValidator.process_batch`), every image of a batch
// in one go and without the per-image D2H + numpy argsort / unique of the reference:
//   candidate pairs (label l, detection d): same class and IoU(l, d) >= iouv[0]
//   sort by IoU descending, unique by detection  ->  each detection keeps its best label
//   unique by label (in detection order)         ->  each label keeps its LOWEST-index detection (detections arrive
//                                                    sorted by confidence, so: its most confident one)
//   correct[d][j] = IoU >= iouv[j] for the surviving pairs, false elsewhere.
// IoU ties between the labels of one detection resolve to the higher label index.
// ---------------------------------------------------------------------------------------------------
__global__ void k_match_best(const float* det, const int* det_img, int64_t N, const float* lab, const int* lab_off, const float* iouv,
                             int* best_l, float* best_iou, int* owner) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= N) return;
    const float* q = det + d * 6;
    const float bx1 = q[0], by1 = q[1], bx2 = q[2], by2 = q[3], cls = q[5];
    const float ba = (bx2 - bx1) * (by2 - by1);
    const int img = det_img[d];
    const float thr0 = iouv[0];
    int bl = -1;
    float bi = -1.0f;
    for (int l = lab_off[img]; l < lab_off[img + 1]; ++l) {
        const float* p = lab + (int64_t)l * 5;
        if (p[0] != cls) continue;
        const float aa = (p[3] - p[1]) * (p[4] - p[2]);
        const float v = iou_pair(p[1], p[2], p[3], p[4], aa, bx1, by1, bx2, by2, ba);
        if (v >= thr0 && v >= bi) { bi = v; bl = l; }
    }
    best_l[d] = bl;
    best_iou[d] = bi;
    if (bl >= 0) atomicMin(&owner[bl], (int)d);
}

__global__ void k_match_correct(const int* best_l, const float* best_iou, const int* owner, int64_t N, const float* iouv, int niou,
                                unsigned char* correct) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * niou) return;
    const int64_t d = t / niou;
    const int j = (int)(t - d * niou);
    const int bl = best_l[d];
    correct[t] = (bl >= 0 && owner[bl] == (int)d && best_iou[d] >= iouv[j]) ? 1 : 0;
}

extern "C" int ayolo_match_detections(const float* det, const int* det_img, int64_t N, const float* lab, const int* lab_off,
                                      int64_t M, const float* iouv_dev, int niou, int* best_l, float* best_iou, int* owner,
                                      unsigned char* correct, ayolo_stream s) {
    if (N == 0) return AYOLO_OK;
    AY_CHECK_ARG(det && det_img && lab_off && iouv_dev && best_l && best_iou && correct && niou > 0 && N < (1ll << 31) && M < (1ll << 31),
                 "match_detections: bad args");
    AY_CHECK_ARG(M == 0 || (lab && owner), "match_detections: labels");
    hipStream_t st = (hipStream_t)s;
    if (M > 0) {
        hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)cdiv64(M, 256)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(owner), (uint32_t)M,
                           0x7fffffffu);
        AY_CHECK_LAUNCH("k_fill_u32");
    }
    hipLaunchKernelGGL(k_match_best, dim3((unsigned)cdiv64(N, 128)), dim3(128), 0, st, det, det_img, N, lab, lab_off, iouv_dev, best_l,
                       best_iou, owner);
    AY_CHECK_LAUNCH("k_match_best");
    hipLaunchKernelGGL(k_match_correct, dim3((unsigned)cdiv64(N * niou, 256)), dim3(256), 0, st, best_l, best_iou, owner, N, iouv_dev,
                       niou, correct);
    AY_CHECK_LAUNCH("k_match_correct");
    return AYOLO_OK;
}

// ---------------------------------------------------------------------------------------------------
// One-call class-aware NMS of the `nms` branch (metrics.py:313-388 with the boxes offset by cls * 4096), no library sorts
// and no host read-back until the result:
//   k_candidates (conf filter, 340 B per raw proposal)            -> candidate rows + keys (image | ~conf | seq)
//   k_seg_hist     per (image, class) counts, coordinate span      -> nseg counters
//   k_seg_scan     exclusive scan, limits checked on the device    -> segment offsets, fallback flags
//   k_seg_scatter  candidates grouped by segment (any order)       -> 24 B per candidate
//   k_seg_nms      ONE workgroup per segment: bitonic sort by key in LDS, greedy NMS on the offset boxes in LDS
//                  (64-box blocks resolved by one wavefront, later boxes checked against the block's kept boxes by all),
//                  kept rows appended to the image's list
//   k_img_topk     ONE workgroup per image: bitonic sort of the kept keys, first max_det rows emitted
// The offset makes boxes of different classes disjoint whenever the candidates' coordinates span less than 4096 (checked
// on the device), so the global greedy scan in confidence order equals independent per-class scans merged in confidence
// order -- ~nc times fewer box pairs, and segments small enough (<= SEG_CAP) to live in LDS.  Arithmetic that decides
// an index is the CPU sequence (offset add, area, IoU with a true division, strict >).  Whenever a limit does not hold
// (span, a segment above SEG_CAP, more than IMG_CAP kept boxes or more than max_nms candidates in an image, candidate
// buffer too small) a status flag is set and the caller takes the general path instead.
// ---------------------------------------------------------------------------------------------------
#define SEG_CAP 2048
#define IMG_CAP 8192
#define NMSF_OVERFLOW 1u      // candidate buffer too small (status[0] > capacity)
#define NMSF_SPAN 2u          // coordinates span >= 4096: classes not separable by the offset
#define NMSF_SEGCAP 4u        // a (image, class) segment has more than SEG_CAP candidates
#define NMSF_IMGCAP 8u        // an image kept more than IMG_CAP boxes before the max_det cut
#define NMSF_MAXNMS 16u       // an image has more than max_nms candidates (the reference truncates before NMS)

struct SegNmsP {
    const float* det; const uint64_t* keys; const uint32_t* counters;   // k_candidates outputs (counters[0] = total)
    uint32_t capacity; int B, nc, img_shift;                            // image = key >> img_shift
    uint32_t* seg_cnt; uint32_t* seg_off; uint32_t* seg_fill;           // [nseg], [nseg + 1], [nseg]
    uint32_t* span; uint32_t* flags;                                    // [2], [1]
    uint64_t* gkey; float* gbox;                                        // [capacity], [capacity][4]
    uint64_t* kkey; float* kbox; uint32_t* kcls; uint32_t* kcnt;        // per image kept lists [B][IMG_CAP] (+ [B])
    // max_nms truncation (metrics.py:378-379: an image with more candidates keeps its max_nms most confident ones):
    uint32_t* thist; uint32_t* tpick; uint64_t* tlist; uint32_t* tfill; uint64_t* tkey; uint32_t* preflags;   // [B][TR_BINS], [B][2], [B][IMG_CAP], [B], [B], [1]
    float thr; double thr_mid; int thr_odd; uint32_t max_det, max_nms;
    float* out; uint32_t* out_cnt;                                      // [B][max_det][6], [B]
};

// ---- max_nms truncation: the max_nms smallest keys (= most confident candidates) of an over-full image, exactly, as a
// radix select: two histogram levels of TR_BITS bits over the confidence field (workgroup-private in LDS) narrow the
// max_nms-th key down to one bucket of both levels, the (few) keys of that bucket are sorted by one workgroup, and the
// resulting threshold key selects the candidates.  Every kernel returns at once when no image is over-full.
#define TR_BITS 11
#define TR_BINS (1 << TR_BITS)
#define TR_LDS_IMGS 8          // images whose histograms fit one workgroup's LDS (64 KiB)
// Bucket of a key at a select level.  conf = obj * cls <= 1 puts float(conf) below 0x40000000, so the two leading bits of the
// ~conf field are 11 for every candidate: the buckets start BELOW them (level 0: exponent low 6 bits + 5 mantissa bits = 32
// buckets per octave; with the leading bits included all confidences of a typical batch fell into ~30 buckets and the LDS
// histogram atomics serialised).  A key whose leading bits are not 11 (conf >= 2: not a probability) is smaller than all
// others and goes to bucket 0 of both levels, which keeps the map monotone.
__device__ __forceinline__ uint32_t tr_bin(uint64_t key, int img_shift, int level = 0) {
    const uint32_t f = (uint32_t)(key >> (img_shift - 32));            // the ~conf field
    return (f >> 30) != 3u ? 0u : (f >> (30 - TR_BITS * (level + 1))) & (TR_BINS - 1);
}

__device__ __forceinline__ bool any_image_over(const SegNmsP& p) {
    bool over = false;
    for (int b = 0; b < p.B; ++b) over |= p.counters[1 + b] > p.max_nms;     // uniform, B scalar loads
    return over;
}

// tpick[img]: {bucket level 0, keys of it still wanted, bucket level 1, keys of it still wanted}
template <int LEVEL>
__global__ __launch_bounds__(256) void k_trunc_hist(SegNmsP p) {
    extern __shared__ uint32_t s_th[];                     // [min(B, TR_LDS_IMGS)][TR_BINS]
    if (!any_image_over(p)) return;
    const bool in_lds = p.B <= TR_LDS_IMGS;
    uint32_t* gh = p.thist + (size_t)LEVEL * p.B * TR_BINS;
    if (in_lds) {
        for (int i = threadIdx.x; i < p.B * TR_BINS; i += 256) s_th[i] = 0;
        __syncthreads();
    }
    const uint32_t total = min(p.counters[0], p.capacity);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const uint64_t key = p.keys[i];
        const int img = (int)(key >> p.img_shift);
        bool in = p.counters[1 + img] > p.max_nms;
        if (LEVEL == 1) in = in && p.tpick[4 * img] == tr_bin(key, p.img_shift, 0);
        if (in) {
            const uint32_t bin = (uint32_t)img * TR_BINS + tr_bin(key, p.img_shift, LEVEL);
            if (in_lds) atomicAdd(&s_th[bin], 1u); else atomicAdd(&gh[bin], 1u);
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < p.B * TR_BINS; i += 256)
            if (s_th[i]) atomicAdd(&gh[i], s_th[i]);
    }
}

// one workgroup per image: the bucket of this level in which the cumulative count crosses the wanted number
template <int LEVEL>
__global__ __launch_bounds__(1024) void k_trunc_pick(SegNmsP p) {
    __shared__ uint32_t s_scan[1024];
    const int img = blockIdx.x, tid = threadIdx.x;
    if (p.counters[1 + img] <= p.max_nms) {
        if (tid == 0 && LEVEL == 0) { p.tkey[img] = ~0ull; p.tpick[4 * img] = 0xffffffffu; p.tpick[4 * img + 1] = 0; p.tpick[4 * img + 2] = 0xffffffffu; p.tpick[4 * img + 3] = 0; }
        return;
    }
    const uint32_t want = LEVEL == 0 ? p.max_nms : p.tpick[4 * img + 1];
    const uint32_t* gh = p.thist + ((size_t)LEVEL * p.B + img) * TR_BINS;
    constexpr int PER = TR_BINS / 1024;
    static_assert(PER >= 1, "one or more bins per thread");
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { c[k] = gh[tid * PER + k]; sum += c[k]; }
    s_scan[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        s_scan[tid] += add;
        __syncthreads();
    }
    uint32_t below = s_scan[tid] - sum;                    // keys in the buckets before this thread's first one
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (below < want && below + c[k] >= want) {        // exactly one (thread, k) satisfies this
            p.tpick[4 * img + 2 * LEVEL] = (uint32_t)(tid * PER + k);
            p.tpick[4 * img + 2 * LEVEL + 1] = want - below;           // keys of this bucket that are still wanted (>= 1)
        }
        below += c[k];
    }
}

// keys of the deciding bucket (both levels) -> the image's list.  Workgroup-level aggregation: all hits of an image share
// ONE global counter, and same-address atomics serialise in L2 (~12 ns each)
__global__ __launch_bounds__(256) void k_trunc_collect(SegNmsP p) {
    __shared__ uint32_t s_cnt[TR_LDS_IMGS], s_base[TR_LDS_IMGS];
    if (!any_image_over(p)) return;
    const bool agg = p.B <= TR_LDS_IMGS;
    const uint32_t total = min(p.counters[0], p.capacity);
    const uint32_t per = (total + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(total, lo + per);
    if (threadIdx.x < TR_LDS_IMGS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // pass 1: count this workgroup's hits per image; pass 2: write them behind ONE reservation per image
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
            const uint64_t key = p.keys[i];
            const int img = (int)(key >> p.img_shift);
            if (p.tpick[4 * img] == tr_bin(key, p.img_shift, 0) && p.tpick[4 * img + 2] == tr_bin(key, p.img_shift, 1)) {
                if (!agg) {
                    if (pass == 0) continue;
                    const uint32_t slot = atomicAdd(&p.tfill[img], 1u);
                    if (slot < IMG_CAP) p.tlist[(size_t)img * IMG_CAP + slot] = key;
                } else if (pass == 0) atomicAdd(&s_cnt[img], 1u);
                else {
                    const uint32_t slot = s_base[img] + atomicAdd(&s_cnt[img], 1u);
                    if (slot < IMG_CAP) p.tlist[(size_t)img * IMG_CAP + slot] = key;
                }
            }
        }
        __syncthreads();
        if (pass == 0 && agg && threadIdx.x < p.B) {
            s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&p.tfill[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
            s_cnt[threadIdx.x] = 0;
        }
        __syncthreads();
    }
}

template <int NT> __device__ __forceinline__ void lds_bitonic(uint64_t* key, uint32_t* val, int N2, int tid);

__global__ __launch_bounds__(1024) void k_trunc_thr(SegNmsP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char thr_lds[];
    uint64_t* skey = reinterpret_cast<uint64_t*>(thr_lds);
    uint32_t* sidx = reinterpret_cast<uint32_t*>(thr_lds + (size_t)IMG_CAP * 8);
    const int img = blockIdx.x, tid = threadIdx.x;
    if (p.tpick[4 * img] == 0xffffffffu) return;           // image not truncated: tkey = ~0 (k_trunc_pick)
    const uint32_t n = p.tfill[img];
    if (n > IMG_CAP) { if (tid == 0) { atomicOr(p.preflags, NMSF_SEGCAP); p.tkey[img] = ~0ull; } return; }
    int N2 = 64;
    while (N2 < (int)n) N2 <<= 1;
    for (int i = tid; i < N2; i += 1024) { skey[i] = i < (int)n ? p.tlist[(size_t)img * IMG_CAP + i] : ~0ull; sidx[i] = 0; }
    __syncthreads();
    lds_bitonic<1024>(skey, sidx, N2, tid);
    if (tid == 0) p.tkey[img] = skey[p.tpick[4 * img + 3] - 1];
}

__global__ __launch_bounds__(256) void k_seg_hist(SegNmsP p) {
    extern __shared__ uint32_t s_hist[];        // [nseg]
    __shared__ uint32_t s_hi[4], s_lo[4];
    const int nseg = p.B * p.nc;
    for (int i = threadIdx.x; i < nseg; i += 256) s_hist[i] = 0;
    __syncthreads();
    const uint32_t total = min(p.counters[0], p.capacity);
    uint32_t hi = 0, lo = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const float2* src = reinterpret_cast<const float2*>(p.det + (size_t)i * 6);
        const float2 a = src[0], c = src[1], d = src[2];
        const uint64_t key = p.keys[i];
        const int img = (int)(key >> p.img_shift);
        if (key > p.tkey[img]) continue;                   // beyond the image's max_nms most confident candidates
        atomicAdd(&s_hist[img * p.nc + (int)d.y], 1u);
        hi = max(max(hi, float_order(a.x)), max(float_order(a.y), max(float_order(c.x), float_order(c.y))));
        lo = max(max(lo, float_order(-a.x)), max(float_order(-a.y), max(float_order(-c.x), float_order(-c.y))));
    }
    for (int off = 32; off > 0; off >>= 1) {
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
        lo = max(lo, (uint32_t)__shfl_xor((int)lo, off));
    }
    if ((threadIdx.x & 63) == 0) { s_hi[threadIdx.x >> 6] = hi; s_lo[threadIdx.x >> 6] = lo; }
    __syncthreads();
    for (int i = threadIdx.x; i < nseg; i += 256)
        if (s_hist[i]) atomicAdd(&p.seg_cnt[i], s_hist[i]);
    if (threadIdx.x == 0) {
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        lo = max(max(s_lo[0], s_lo[1]), max(s_lo[2], s_lo[3]));
        if (hi > __hip_atomic_load(&p.span[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.span[0], hi);
        if (lo > __hip_atomic_load(&p.span[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&p.span[1], lo);
    }
}

__device__ __forceinline__ float order_float(uint32_t code) {      // inverse of float_order
    return __uint_as_float((code & 0x80000000u) ? (code & 0x7fffffffu) : ~code);
}

// one workgroup: exclusive scan of the segment counts; every limit of the fast path is checked here, on the device
__global__ __launch_bounds__(1024) void k_seg_scan(SegNmsP p) {
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_carry, s_flags;
    const int tid = threadIdx.x, nseg = p.B * p.nc;
    if (tid == 0) { s_carry = 0; s_flags = 0; }
    __syncthreads();
    for (int base = 0; base < nseg; base += 1024) {
        const int sg = base + tid;
        const uint32_t n = sg < nseg ? p.seg_cnt[sg] : 0u;
        if (n > SEG_CAP) atomicOr(&s_flags, NMSF_SEGCAP);
        s_scan[tid] = n;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        const uint32_t carry = s_carry;
        if (sg < nseg) p.seg_off[sg] = carry + s_scan[tid] - n;
        __syncthreads();
        if (tid == 1023) s_carry = carry + s_scan[1023];
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        p.seg_off[nseg] = s_carry;
        uint32_t f = s_flags | p.preflags[0];              // raised by the truncation kernels before this one
        if (p.counters[0] > p.capacity) f |= NMSF_OVERFLOW;
        if (p.counters[0] > 0) {
            const float hi = order_float(p.span[0]), lo = order_float(p.span[1]);     // max(coord), max(-coord)
            if (!(hi + lo < 4096.0f)) f |= NMSF_SPAN;
        }
        p.flags[0] = f;
    }
}

// each workgroup owns a contiguous range of candidates: local rank per segment by LDS atomics, ONE global reservation per
// (workgroup, segment), then the rows are written to seg_off[segment] + reservation + local rank
#define SCAT_PER 8             // candidates per thread
__global__ __launch_bounds__(256) void k_seg_scatter(SegNmsP p) {
    extern __shared__ uint32_t s_sc[];                     // [nseg] counts, then [nseg] global bases
    if (p.flags[0]) return;
    const int nseg = p.B * p.nc;
    uint32_t* s_cnt = s_sc;
    uint32_t* s_base = s_sc + nseg;
    for (int i = threadIdx.x; i < nseg; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const uint32_t total = min(p.counters[0], p.capacity);
    const uint32_t lo = blockIdx.x * (256 * SCAT_PER);
    uint32_t rank[SCAT_PER];
    int sgs[SCAT_PER];
#pragma unroll
    for (int k = 0; k < SCAT_PER; ++k) {
        const uint32_t i = lo + k * 256 + threadIdx.x;
        sgs[k] = -1;
        if (i < total) {
            const uint64_t key = p.keys[i];
            const int img = (int)(key >> p.img_shift);
            if (key <= p.tkey[img]) {
                sgs[k] = img * p.nc + (int)p.det[(size_t)i * 6 + 5];
                rank[k] = atomicAdd(&s_cnt[sgs[k]], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nseg; i += 256)
        if (s_cnt[i]) s_base[i] = atomicAdd(&p.seg_fill[i], s_cnt[i]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAT_PER; ++k) {
        if (sgs[k] < 0) continue;
        const uint32_t i = lo + k * 256 + threadIdx.x;
        const float2* src = reinterpret_cast<const float2*>(p.det + (size_t)i * 6);
        const float2 a = src[0], c = src[1];
        const uint32_t slot = p.seg_off[sgs[k]] + s_base[sgs[k]] + rank[k];
        p.gkey[slot] = p.keys[i];
        reinterpret_cast<float4*>(p.gbox)[slot] = make_float4(a.x, a.y, c.x, c.y);
    }
}

// ascending bitonic sort of n (key, payload) pairs in LDS; N2 = power of two >= n, slots [n, N2) hold ~0 keys
template <int NT>
__device__ __forceinline__ void lds_bitonic(uint64_t* key, uint32_t* val, int N2, int tid) {
    for (int k = 2; k <= N2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < N2 / 2; t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // lower index of pair t at distance j
                const int l = i | j;
                const bool up = (i & k) == 0;
                const uint64_t a = key[i], b = key[l];
                if ((a > b) == up) {
                    key[i] = b; key[l] = a;
                    const uint32_t va = val[i]; val[i] = val[l]; val[l] = va;
                }
            }
            __syncthreads();
        }
}

// Segments of at most MAT_CAP candidates (the common case: ~nc segments share an image's candidates): rank sort instead of
// a bitonic network (no barrier per stage), the whole suppression bit matrix of the segment in LDS -- every 64 x 64 tile by
// one wavefront, the row word straight from a ballot, all tiles in parallel -- and then ONE wavefront walks the matrix
// (k_nms_reduce's scan).  The blockwise k_seg_nms below serialises (resolve block, test later boxes) per 64-box block.
// `inter / u > thr` for the correctly rounded float quotient WITHOUT the division (an IEEE division is ~40 VALU operations,
// more than the rest of a box-pair test): RN(x) > thr  <=>  x > m or (x == m and the float above thr has an even mantissa),
// m = thr + ulp(thr) / 2 being the round-to-nearest boundary.  m has 25 significant bits and u 24, so u * m is EXACT in
// double and the comparison decides the real inequality exactly.  u <= 0 (degenerate boxes) follows the quotient's sign /
// infinity: u < 0 gives a negative quotient (never above thr >= 0), u == 0 gives +inf.
__device__ __forceinline__ bool iou_above(float inter, float u, double thr_mid, int thr_odd) {
    const double di = (double)inter, prod = (double)u * thr_mid;
    const bool pos = (di > prod) | ((di == prod) & (thr_odd != 0));
    return u > 0.0f ? pos : (u == 0.0f);
}

#define MAT_CAP 512
// SMALL_NT threads: the three parallel phases (rank sort, overlap bits, exact-IoU pass) are the segment's critical path -- one
// workgroup per segment, the largest segment bounds the launch -- so eight wavefronts instead of four halve them; three
// workgroups of ~50 KB LDS still fit a CU, i.e. the 640 segments of an 8 x 80-class batch stay one round
#define SMALL_NT 512
__global__ __launch_bounds__(SMALL_NT) void k_seg_nms_small(SegNmsP p) {
    __shared__ uint64_t skin[MAT_CAP], skey[MAT_CAP];          // keys as loaded / sorted
    __shared__ float4 sbox[MAT_CAP];                           // sorted boxes + class offset
    __shared__ uint16_t sidx[MAT_CAP], skl[MAT_CAP];           // sorted position -> position in the segment; kept list
    __shared__ uint64_t smat[MAT_CAP * (MAT_CAP / 64)];        // row i, word w: later boxes 64w .. 64w+63 that box i suppresses
    __shared__ uint64_t srem[MAT_CAP / 64];
    __shared__ uint32_t s_kept, s_base;
    if (p.flags[0]) return;
    const int sg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (int)p.seg_cnt[sg];
    if (n == 0 || n > MAT_CAP) return;
    const uint32_t off = p.seg_off[sg];
    const int img = sg / p.nc, cls = sg - img * p.nc;
    for (int i = tid; i < n; i += SMALL_NT) skin[i] = p.gkey[off + i];
    __syncthreads();
    // rank sort: keys are unique, so the ranks are a permutation
    {
        static_assert(SMALL_NT >= MAT_CAP, "one key per thread");
        const uint64_t k0 = tid < n ? skin[tid] : 0ull;
        int r0 = 0;
        const int n8 = n & ~7;
        for (int j = 0; j < n8; j += 8) {                      // eight independent broadcast reads in flight per step
            uint64_t kj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kj[u] = skin[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) r0 += kj[u] < k0;
        }
        for (int j = n8; j < n; ++j) r0 += skin[j] < k0;
        if (tid < n) { skey[r0] = k0; sidx[r0] = (uint16_t)tid; }
    }
    __syncthreads();
    const float o = (float)cls * 4096.0f;                      // metrics.py:383: boxes + cls * max_wh, in float32
    for (int i = tid; i < n; i += SMALL_NT) {
        const float4 b = reinterpret_cast<const float4*>(p.gbox)[off + sidx[i]];
        sbox[i] = make_float4(b.x + o, b.y + o, b.z + o, b.w + o);
    }
    __syncthreads();
    const int nb = (n + 63) >> 6;
    // ---- suppression matrix in two passes.  Only a few % of the box pairs of a class overlap at all, and the IoU test with
    // its exact threshold comparison is ~4x the instructions of an overlap test, so:
    //   pass 1: OVERLAP bits of every pair of the upper triangle -- 64 x 64 tiles, lane = column box, row boxes broadcast
    //           from LDS, the row word straight from a ballot (4 compares per pair; a superset of `inter > 0`);
    //   pass 2: every thread walks the set bits of its share of the words and keeps those whose IoU is above the threshold.
    for (int t = wave; t < nb * nb; t += SMALL_NT / 64) {
        const int rb = t / nb, cb = t - rb * nb;
        if (cb < rb) continue;
        const int j = cb * 64 + lane;
        const bool cvalid = j < n;
        const float4 bj = sbox[cvalid ? j : 0];
        const int rows = min(64, n - rb * 64);
        for (int i0 = 0; i0 < rows; i0 += 4) {                 // four row boxes fetched per step (independent broadcast reads)
            float4 bi4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bi4[u] = sbox[min(rb * 64 + i0 + u, n - 1)];
            uint64_t myword = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ri = rb * 64 + i0 + u;
                const float4 bi = bi4[u];
                const bool ov = cvalid && j > ri && i0 + u < rows && bi.x < bj.z && bj.x < bi.z && bi.y < bj.w && bj.y < bi.w;
                const uint64_t word = __ballot(ov);
                if (lane == u) myword = word;
            }
            if (lane < 4 && i0 + lane < rows) smat[(rb * 64 + i0 + lane) * (MAT_CAP / 64) + cb] = myword;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n * nb; idx += SMALL_NT) {
        const int ri = idx / nb, cb = idx - ri * nb;
        if (cb < (ri >> 6)) continue;
        uint64_t word = smat[ri * (MAT_CAP / 64) + cb], out = 0;
        if (word == 0) continue;
        const float4 bi = sbox[ri];
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
        while (word != 0) {
            const int t = __ffsll((unsigned long long)word) - 1;
            word &= word - 1;
            const float4 bj = sbox[cb * 64 + t];
            const float aj = (bj.z - bj.x) * (bj.w - bj.y);
            const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
            const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;                         // torchvision nms: pairs without overlap skipped, strict >
            if (inter > 0.0f && iou_above(inter, ai + aj - inter, p.thr_mid, p.thr_odd)) out |= 1ull << t;
        }
        smat[ri * (MAT_CAP / 64) + cb] = out;
    }
    if (tid < MAT_CAP / 64) srem[tid] = 0;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    // ---- greedy scan by wavefront 0
    if (wave == 0) {
        uint32_t kept = 0;
        for (int blk = 0; blk < nb && kept < p.max_det; ++blk) {
            const int i = blk * 64 + lane;
            const uint64_t d = i < n ? smat[i * (MAT_CAP / 64) + blk] : 0ull;
            const uint64_t valid = (blk * 64 + 64 <= n) ? ~0ull : ((1ull << (n - blk * 64)) - 1ull);
            uint64_t alive = ~srem[blk] & valid;               // uniform (LDS broadcast)
            uint64_t keep = 0;
            uint32_t k2 = kept;
            while (alive != 0 && k2 < p.max_det) {
                const int t = __ffsll((unsigned long long)alive) - 1;
                keep |= 1ull << t;
                ++k2;
                alive &= ~(rl64(d, t) | (1ull << t));
            }
            if ((keep >> lane) & 1ull) skl[kept + __popcll(keep & ((1ull << lane) - 1ull))] = (uint16_t)i;
            kept = k2;
            // OR the rows of this block's kept boxes into the removed set of the later blocks: lane = kept box, one wave
            // reduction per later word (a loop over the kept boxes by the few "word" lanes was 64 dependent LDS reads)
            const bool mine = ((keep >> lane) & 1ull) != 0;
            for (int w = blk + 1; w < nb; ++w) {
                uint64_t v = mine ? smat[i * (MAT_CAP / 64) + w] : 0ull;
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) {
                    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o2), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o2);
                    v |= ((uint64_t)hi << 32) | lo;
                }
                if (lane == 0) srem[w] |= v;
            }
        }
        if (lane == 0) s_kept = kept;
    }
    __syncthreads();
    const uint32_t kept = s_kept;
    if (tid == 0) s_base = atomicAdd(&p.kcnt[img], kept);
    __syncthreads();
    const uint32_t base = s_base;
    for (uint32_t r = tid; r < kept; r += SMALL_NT) {
        const uint32_t j = skl[r];
        const size_t dst = (size_t)img * ((size_t)p.nc * p.max_det) + base + r;
        p.kkey[dst] = skey[j];
        reinterpret_cast<float4*>(p.kbox)[dst] = reinterpret_cast<const float4*>(p.gbox)[off + sidx[j]];
        p.kcls[dst] = (uint32_t)cls;
    }
}

__global__ __launch_bounds__(256) void k_seg_nms(SegNmsP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char seg_lds[];
    uint64_t* skey = reinterpret_cast<uint64_t*>(seg_lds);                       // [SEG_CAP]
    float4* sbox = reinterpret_cast<float4*>(seg_lds + (size_t)SEG_CAP * 8);     // [SEG_CAP] boxes + class offset, SORTED order
    uint32_t* sidx = reinterpret_cast<uint32_t*>(seg_lds + (size_t)SEG_CAP * 24);// [SEG_CAP] sort payload: position in the segment
    uint16_t* skl = reinterpret_cast<uint16_t*>(seg_lds + (size_t)SEG_CAP * 28); // [SEG_CAP] kept boxes (sorted positions)
    __shared__ uint64_t s_keep;
    __shared__ uint32_t s_kept, s_base;
    if (p.flags[0]) return;
    const int sg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = p.seg_cnt[sg];
    if (n <= MAT_CAP) return;                              // k_seg_nms_small's
    const uint32_t off = p.seg_off[sg];
    const int img = sg / p.nc, cls = sg - img * p.nc;
    int N2 = 64;
    while (N2 < (int)n) N2 <<= 1;
    for (int i = tid; i < N2; i += 256) {
        skey[i] = i < (int)n ? p.gkey[off + i] : ~0ull;
        sidx[i] = (uint32_t)i;
    }
    __syncthreads();
    lds_bitonic<256>(skey, sidx, N2, tid);
    // boxes in sorted order, with the reference's class offset applied (metrics.py:383: boxes + cls * max_wh, in float32)
    const float o = (float)cls * 4096.0f;
    for (int i = tid; i < (int)n; i += 256) {
        const float4 b = reinterpret_cast<const float4*>(p.gbox)[off + sidx[i]];
        sbox[i] = make_float4(b.x + o, b.y + o, b.z + o, b.w + o);
    }
    if (tid == 0) s_kept = 0;
    __syncthreads();
    // greedy scan over blocks of 64 sorted boxes: wavefront 0 resolves a block (intra-block suppression words, then the
    // sequential scan in SGPRs), then all four wavefronts test every later, still alive box against the block's kept boxes
    unsigned char* salive = reinterpret_cast<unsigned char*>(skl + SEG_CAP);       // [SEG_CAP]
    for (int i = tid; i < (int)n; i += 256) salive[i] = 1;
    __syncthreads();
    // suppressed(i -> j): torchvision nms on the offset boxes (pairs without overlap skipped, strict >)
    // `on`: this lane has a pair to test.  The IEEE division runs only where some lane of the wavefront sees an overlap
    // (wave-uniform branch): most box pairs are disjoint
    auto sup = [&](bool on, const float4 bi, float ai, const float4 bj, float aj) -> bool {
        const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
        const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
        const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
        const float inter = w * h;
        return on && inter > 0.0f && iou_above(inter, ai + aj - inter, p.thr_mid, p.thr_odd);
    };
    constexpr int SLOTS = SEG_CAP / 256;
    const int nblk = ((int)n + 63) >> 6;
    for (int blk = 0; blk < nblk; ++blk) {
        if (wave == 0) {
            const int j = blk * 64 + lane;
            const bool valid = j < (int)n;
            const float4 bj = sbox[valid ? j : 0];
            const float aj = (bj.z - bj.x) * (bj.w - bj.y);
            uint64_t m = 0;                               // bit i: box blk*64 + i (i < lane) would suppress this lane's box
            const int lim = min(63, (int)n - blk * 64);
            for (int i = 0; i < lim; ++i) {
                const float4 bi = sbox[blk * 64 + i];
                const float ai = (bi.z - bi.x) * (bi.w - bi.y);
                if (sup(i < lane && valid, bi, ai, bj, aj)) m |= 1ull << i;
            }
            uint64_t rem = __ballot(valid && salive[valid ? j : 0] != 0);
            uint64_t keep = 0;
            const uint32_t kept0 = s_kept;
            uint32_t kept = kept0;
            while (rem != 0 && kept < p.max_det) {
                const int t = __ffsll((unsigned long long)rem) - 1;
                keep |= 1ull << t;
                ++kept;
                rem &= ~(__ballot((m >> t) & 1ull) | (1ull << t));
            }
            if ((keep >> lane) & 1ull) skl[kept0 + __popcll(keep & ((1ull << lane) - 1ull))] = (uint16_t)j;
            if (lane == 0) { s_keep = keep; s_kept = kept; }
        }
        __syncthreads();
        const uint64_t keep = s_keep;
        if (s_kept >= p.max_det || blk + 1 == nblk) break;   // uniform
        const int nslots = ((int)n - (blk + 1) * 64 + 255) >> 8;      // slots of 256 later boxes that exist (uniform)
        float4 mine[SLOTS];
        float marea[SLOTS];
        bool live[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            live[q] = false;
            if (q < nslots) {
                const int j = (blk + 1) * 64 + q * 256 + tid;
                live[q] = j < (int)n && salive[j < (int)n ? j : 0] != 0;
                mine[q] = sbox[j < (int)n ? j : 0];
                marea[q] = (mine[q].z - mine[q].x) * (mine[q].w - mine[q].y);
            }
        }
        uint64_t km = keep;
        while (km != 0) {
            const int t = __ffsll((unsigned long long)km) - 1;
            km &= km - 1;
            const float4 bi = sbox[blk * 64 + t];
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
#pragma unroll
            for (int q = 0; q < SLOTS; ++q)
                if (q < nslots) { if (sup(live[q], bi, ai, mine[q], marea[q])) live[q] = false; }
        }
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            if (q < nslots) {
                const int j = (blk + 1) * 64 + q * 256 + tid;
                if (j < (int)n && !live[q]) salive[j] = 0;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // kept boxes (sorted positions skl[0 .. kept)) -> the image's list, with their ORIGINAL coordinates
    const uint32_t kept = s_kept;
    if (tid == 0) s_base = atomicAdd(&p.kcnt[img], kept);
    __syncthreads();
    const uint32_t base = s_base;                          // < nc * max_det: every segment keeps at most max_det boxes
    for (uint32_t r = tid; r < kept; r += 256) {
        const uint32_t j = skl[r];
        const size_t dst = (size_t)img * ((size_t)p.nc * p.max_det) + base + r;
        p.kkey[dst] = skey[j];
        reinterpret_cast<float4*>(p.kbox)[dst] = reinterpret_cast<const float4*>(p.gbox)[off + sidx[j]];
        p.kcls[dst] = (uint32_t)cls;
    }
}

// one workgroup per image: the max_det most confident kept boxes of all its classes, emitted in confidence order as
// [x1, y1, x2, y2, conf, cls] rows (conf recovered from the key: ~conf bits above the seq field).  Up to nc * max_det boxes
// arrive (random boxes hardly overlap: every class keeps its first max_det), so the cut is a SELECTION, not a sort of all of
// them: an LDS histogram over the leading confidence bits finds the bucket that holds the max_det-th key, the keys below it
// are taken, the bucket's own keys are sorted and contribute the rest; only the <= max_det winners are sorted for output.
#define TOPK_OUT 1024          // max_det supported by this path
#define TOPK_BND 4096          // keys in the deciding bucket
#define TOPK_KPT 32            // kept keys per thread (registers): nc * max_det <= 32 768
__global__ __launch_bounds__(1024) void k_img_topk(SegNmsP p, int seq_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char img_lds[];
    uint64_t* okey = reinterpret_cast<uint64_t*>(img_lds);                                   // [TOPK_OUT]
    uint64_t* bkey = okey + TOPK_OUT;                                                          // [TOPK_BND]
    uint32_t* oidx = reinterpret_cast<uint32_t*>(bkey + TOPK_BND);                             // [TOPK_OUT]
    uint32_t* bidx = oidx + TOPK_OUT;                                                          // [TOPK_BND]
    uint32_t* hist = bidx + TOPK_BND;                                                          // [TR_BINS]
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_no, s_nb, s_bstar, s_r;
    const int img = blockIdx.x, tid = threadIdx.x;
    if (p.flags[0]) { if (tid == 0) p.out_cnt[img] = 0; return; }
    const uint32_t kcap = (uint32_t)p.nc * p.max_det;
    const uint32_t K = min(p.kcnt[img], kcap);
    const uint32_t want = min(K, p.max_det);
    const uint64_t* kk = p.kkey + (size_t)img * kcap;
    if (tid == 0) { s_no = 0; s_nb = 0; s_bstar = 0xffffffffu; s_r = 0; }
    // this thread's keys, fetched ONCE with independent loads (a load per loop iteration in front of an LDS atomic was a
    // chain of ~1 us round trips: 24 per pass)
    uint64_t kreg[TOPK_KPT];
#pragma unroll
    for (int k = 0; k < TOPK_KPT; ++k) {
        const uint32_t i = (uint32_t)k * 1024u + tid;
        kreg[k] = i < K ? kk[i] : ~0ull;
    }
    if (K > TOPK_OUT) {
        for (int i = tid; i < TR_BINS; i += 1024) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TOPK_KPT; ++k)
            if ((uint32_t)k * 1024u + tid < K) atomicAdd(&hist[tr_bin(kreg[k], p.img_shift)], 1u);
        __syncthreads();
        constexpr int PER = TR_BINS / 1024;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { c[k] = hist[tid * PER + k]; sum += c[k]; }
        s_scan[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        uint32_t below = s_scan[tid] - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (below < want && below + c[k] >= want) { s_bstar = (uint32_t)(tid * PER + k); s_r = want - below; }
            below += c[k];
        }
    }
    __syncthreads();
    const uint32_t bstar = s_bstar;
#pragma unroll
    for (int k = 0; k < TOPK_KPT; ++k) {
        const uint32_t i = (uint32_t)k * 1024u + tid;
        if (i < K) {
            const uint64_t key = kreg[k];
            const uint32_t bin = K > TOPK_OUT ? tr_bin(key, p.img_shift) : 0u;
            if (K <= TOPK_OUT || bin < bstar) {
                const uint32_t slot = atomicAdd(&s_no, 1u);
                okey[slot] = key; oidx[slot] = i;
            } else if (bin == bstar) {
                const uint32_t slot = atomicAdd(&s_nb, 1u);
                if (slot < TOPK_BND) { bkey[slot] = key; bidx[slot] = i; }
            }
        }
    }
    __syncthreads();
    if (K > TOPK_OUT) {
        const uint32_t nb = s_nb;
        if (nb > TOPK_BND) { if (tid == 0) { atomicOr(&p.flags[0], NMSF_IMGCAP); p.out_cnt[img] = 0; } return; }
        // the r smallest keys of the deciding bucket join the winners: rank of every bucket key among the bucket's keys
        // (unique keys: ranks are a permutation; batched broadcast reads, no barrier per stage as a sorting network has)
        const uint32_t no = s_no, r = s_r;
        uint64_t mk[TOPK_BND / 1024];
        uint32_t rk[TOPK_BND / 1024];
#pragma unroll
        for (int q = 0; q < TOPK_BND / 1024; ++q) { mk[q] = (uint32_t)q * 1024u + tid < nb ? bkey[q * 1024 + tid] : ~0ull; rk[q] = 0; }
        const int nb8 = (int)nb & ~7;
        for (int j = 0; j < nb8; j += 8) {
            uint64_t kj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kj[u] = bkey[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < TOPK_BND / 1024; ++q) rk[q] += kj[u] < mk[q];
        }
        for (int j = nb8; j < (int)nb; ++j) {
            const uint64_t kj = bkey[j];
#pragma unroll
            for (int q = 0; q < TOPK_BND / 1024; ++q) rk[q] += kj < mk[q];
        }
#pragma unroll
        for (int q = 0; q < TOPK_BND / 1024; ++q)
            if ((uint32_t)q * 1024u + tid < nb && rk[q] < r) { okey[no + rk[q]] = mk[q]; oidx[no + rk[q]] = bidx[q * 1024 + tid]; }
        __syncthreads();
    }
    // okey / oidx hold the winners in any order (K <= TOPK_OUT: all K kept boxes): rank them, emit the first `want`
    const int cnt = K > TOPK_OUT ? (int)want : (int)K;
    {
        const uint64_t mkey = tid < cnt ? okey[tid] : ~0ull;
        const uint32_t midx = tid < cnt ? oidx[tid] : 0u;
        uint32_t rank = 0;
        const int c8 = cnt & ~7;
        for (int j = 0; j < c8; j += 8) {
            uint64_t kj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kj[u] = okey[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += kj[u] < mkey;
        }
        for (int j = c8; j < cnt; ++j) rank += okey[j] < mkey;
        if (tid < cnt) { bkey[rank] = mkey; bidx[rank] = midx; }       // sorted winners (the bucket buffers are free now)
    }
    __syncthreads();
    for (uint32_t r = tid; r < want; r += 1024) {
        const size_t src = (size_t)img * kcap + bidx[r];
        const float4 b = reinterpret_cast<const float4*>(p.kbox)[src];
        const float conf = __uint_as_float(~(uint32_t)(bkey[r] >> seq_bits));
        float* d = p.out + ((size_t)img * p.max_det + r) * 6;
        d[0] = b.x; d[1] = b.y; d[2] = b.z; d[3] = b.w; d[4] = conf; d[5] = (float)p.kcls[src];
    }
    if (tid == 0) p.out_cnt[img] = want;
}

// status (device, uint32): [0] flags (NMSF_*: non-zero = take the general path), [1] candidates, [2 .. 2 + B) rows per image
// in `out`, [2 + B .. 2 + 2B) candidates per image.  ws == NULL: *ws_bytes receives the workspace size for `capacity`.
extern "C" int ayolo_nms_class_fast(const float* pred, int B, int N, int no, float conf_thres, int multi_label,
                                    const uint64_t* class_mask, float iou_thres_f, uint32_t max_det, uint32_t max_nms,
                                    uint32_t capacity, void* ws, size_t* ws_bytes, float* out, uint32_t* status,
                                    ayolo_stream s) {
    AY_CHECK_ARG(ws_bytes, "nms_class_fast: ws_bytes null");
    AY_CHECK_ARG(B > 0 && N > 0 && no > 6 && max_det > 0 && max_det <= TOPK_OUT && capacity > 0, "nms_class_fast: bad sizes");
    AY_CHECK_ARG((uint64_t)(no - 5) * max_det <= (uint64_t)TOPK_KPT * 1024, "nms_class_fast: nc * max_det = %llu kept boxes per image unsupported", (unsigned long long)(no - 5) * max_det);
    const int nc = no - 5, nseg = B * nc;
    AY_CHECK_ARG(B <= 1024 && (size_t)nseg * 4 <= 48 * 1024, "nms_class_fast: B * nc = %d segments unsupported", nseg);
    int seq_bits, total_bits;
    if (ayolo_nms_key_bits(B, N, multi_label ? nc : 1, 0, &seq_bits, &total_bits) != AYOLO_OK) {
        ayolo_set_error("nms_class_fast: key needs %d bits", total_bits);
        return AYOLO_EINVAL;
    }
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t o = 0;
    const size_t o_det = o; o += al((size_t)capacity * 24);
    const size_t o_keys = o; o += al((size_t)capacity * 8);
    const size_t o_gkey = o; o += al((size_t)capacity * 8);
    const size_t o_gbox = o; o += al((size_t)capacity * 16);
    const size_t kimg = (size_t)nc * max_det;               // kept boxes of an image before the max_det cut: exact bound
    const size_t o_kkey = o; o += al((size_t)B * kimg * 8);
    const size_t o_kbox = o; o += al((size_t)B * kimg * 16);
    const size_t o_kcls = o; o += al((size_t)B * kimg * 4);
    const size_t o_tlist = o; o += al((size_t)B * IMG_CAP * 8);
    const size_t o_tkey = o; o += al((size_t)B * 8);
    const size_t o_tpick = o; o += al((size_t)B * 16);
    const size_t o_zero = o;       // zeroed every call: counters | seg_cnt | seg_fill | span | kcnt | tfill | preflags | thist
    const size_t n_zero = (size_t)(1 + B) + 2 * (size_t)nseg + 2 + 2 * (size_t)B + 1 + 2 * (size_t)B * TR_BINS;
    o += al(n_zero * 4);
    const size_t o_segoff = o; o += al((size_t)(nseg + 1) * 4);
    if (ws == nullptr) { *ws_bytes = o; return AYOLO_OK; }
    if (*ws_bytes < o) { ayolo_set_error("nms_class_fast: workspace %zu < %zu", *ws_bytes, o); return AYOLO_ENOSPC; }
    AY_CHECK_ARG(pred && out && status && ((uintptr_t)ws % 256) == 0, "nms_class_fast: null / misaligned pointer");
    unsigned char* w = (unsigned char*)ws;
    uint32_t* z = (uint32_t*)(w + o_zero);
    int rc = ayolo_fill_zero(z, n_zero * 4, s);
    if (rc) return rc;
    SegNmsP p{};
    p.det = (float*)(w + o_det); p.keys = (uint64_t*)(w + o_keys); p.counters = z;
    p.capacity = capacity; p.B = B; p.nc = nc; p.img_shift = 32 + seq_bits;
    p.seg_cnt = z + (1 + B); p.seg_fill = p.seg_cnt + nseg; p.span = p.seg_fill + nseg; p.kcnt = p.span + 2;
    p.tfill = p.kcnt + B; p.preflags = p.tfill + B; p.thist = p.preflags + 1;
    p.tlist = (uint64_t*)(w + o_tlist); p.tkey = (uint64_t*)(w + o_tkey); p.tpick = (uint32_t*)(w + o_tpick);
    p.seg_off = (uint32_t*)(w + o_segoff);
    p.flags = status;
    p.gkey = (uint64_t*)(w + o_gkey); p.gbox = (float*)(w + o_gbox);
    p.kkey = (uint64_t*)(w + o_kkey); p.kbox = (float*)(w + o_kbox); p.kcls = (uint32_t*)(w + o_kcls);
    p.thr = iou_thres_f; p.max_det = max_det; p.max_nms = max_nms;
    {   // round-to-nearest boundary above thr: the quotient test of the segment kernels (iou_above)
        AY_CHECK_ARG(iou_thres_f >= 0.0f && iou_thres_f < 3.0e38f, "nms_class_fast: iou threshold %g", (double)iou_thres_f);
        const float next = nextafterf(iou_thres_f, INFINITY);
        p.thr_mid = ((double)iou_thres_f + (double)next) * 0.5;
        uint32_t bits;
        memcpy(&bits, &iou_thres_f, 4);
        p.thr_odd = (int)(bits & 1u);
    }
    p.out = out; p.out_cnt = status + 2;
    CandParams cp{pred, B, N, no, conf_thres, multi_label, 1, class_mask, nullptr, N,
                  (float*)(w + o_det), (uint64_t*)(w + o_keys), z, capacity, seq_bits, 0, 0, 0, 0, 0, 0};
    rc = run_candidates(cp, no, s);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)s;
    const unsigned gridc = (unsigned)std::min<uint64_t>(((uint64_t)capacity + 255) / 256, 1024);
    static bool attr_t[16] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t img_lds = (size_t)IMG_CAP * 12;
    if (dev < 0 || dev >= 16 || !attr_t[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunc_thr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunc_hist<0>), hipFuncAttributeMaxDynamicSharedMemorySize, TR_LDS_IMGS * TR_BINS * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trunc_hist<1>), hipFuncAttributeMaxDynamicSharedMemorySize, TR_LDS_IMGS * TR_BINS * 4);
        if (dev >= 0 && dev < 16) attr_t[dev] = true;
    }
    const size_t th_lds = (size_t)(B <= TR_LDS_IMGS ? B : 0) * TR_BINS * 4;
    hipLaunchKernelGGL(k_trunc_hist<0>, dim3(256), dim3(256), th_lds, st, p);
    AY_CHECK_LAUNCH("k_trunc_hist");
    hipLaunchKernelGGL(k_trunc_pick<0>, dim3((unsigned)B), dim3(1024), 0, st, p);
    AY_CHECK_LAUNCH("k_trunc_pick");
    hipLaunchKernelGGL(k_trunc_hist<1>, dim3(256), dim3(256), th_lds, st, p);
    AY_CHECK_LAUNCH("k_trunc_hist");
    hipLaunchKernelGGL(k_trunc_pick<1>, dim3((unsigned)B), dim3(1024), 0, st, p);
    AY_CHECK_LAUNCH("k_trunc_pick");
    hipLaunchKernelGGL(k_trunc_collect, dim3(256), dim3(256), 0, st, p);
    AY_CHECK_LAUNCH("k_trunc_collect");
    hipLaunchKernelGGL(k_trunc_thr, dim3((unsigned)B), dim3(1024), img_lds, st, p);
    AY_CHECK_LAUNCH("k_trunc_thr");
    hipLaunchKernelGGL(k_seg_hist, dim3(256), dim3(256), (size_t)nseg * 4, st, p);
    AY_CHECK_LAUNCH("k_seg_hist");
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1024), 0, st, p);
    AY_CHECK_LAUNCH("k_seg_scan");
    hipLaunchKernelGGL(k_seg_scatter, dim3((unsigned)(((uint64_t)capacity + 256 * SCAT_PER - 1) / (256 * SCAT_PER))), dim3(256), (size_t)nseg * 8, st, p);
    AY_CHECK_LAUNCH("k_seg_scatter");
    static bool attr_set[16] = {false};
    const size_t seg_lds = (size_t)SEG_CAP * 31, topk_lds = (size_t)(TOPK_OUT + TOPK_BND) * 12 + (size_t)TR_BINS * 4;
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_nms), hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_img_topk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)topk_lds);
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_seg_nms_small, dim3((unsigned)nseg), dim3(SMALL_NT), 0, st, p);
    AY_CHECK_LAUNCH("k_seg_nms_small");
    hipLaunchKernelGGL(k_seg_nms, dim3((unsigned)nseg), dim3(256), seg_lds, st, p);
    AY_CHECK_LAUNCH("k_seg_nms");
    hipLaunchKernelGGL(k_img_topk, dim3((unsigned)B), dim3(1024), topk_lds, st, p, seq_bits);
    AY_CHECK_LAUNCH("k_img_topk");
    // status[1] = total, status[2 + B ..] = candidates per image: copied from the counters (device to device, 4 * (1 + B) bytes)
    AY_CHECK_HIP(hipMemcpyAsync(status + 1, z, 4, hipMemcpyDeviceToDevice, st));
    AY_CHECK_HIP(hipMemcpyAsync(status + 2 + B, z + 1, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    return AYOLO_OK;
}
