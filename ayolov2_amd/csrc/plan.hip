// Batched launcher: one C call enqueues a whole pre-compiled list of kernel launches (a model forward or
// backward) on one stream.  The op list is built once per (model, shape, dtype) by ayolov2_amd/plan.py over
// static buffers, so a training step costs two host calls instead of ~1600 Python-level launches, and the
// list is a straight-line HIP stream program (graph-capturable).
#include "common.h"
#include <time.h>
#include <stdlib.h>

static inline double now_us() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// Zero fill as an ordinary kernel on the caller's stream.  hipMemsetAsync was observed (ROCm 7.2, null stream) to
// complete AFTER kernels enqueued behind it when the GPU was idle, wiping partially accumulated BN sums.
__global__ void k_fill_zero(uint4* p16, size_t n16, unsigned char* tail, size_t ntail) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) p16[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

extern "C" int ayolo_fill_zero(void* ptr, size_t bytes, ayolo_stream s) {
    if (bytes == 0) return AYOLO_OK;
    AY_CHECK_ARG(ptr && ((uintptr_t)ptr % 16) == 0, "fill_zero: pointer must be 16-byte aligned");
    size_t n16 = bytes / 16, ntail = bytes % 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (uint4*)ptr, n16,
                       (unsigned char*)ptr + n16 * 16, ntail);
    AY_CHECK_LAUNCH("k_fill_zero");
    return AYOLO_OK;
}

// side stream of the executor (one per process = per GPU): created on first use
static hipStream_t g_side = nullptr;
static hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;
static int side_fork(hipStream_t main) {
    if (!g_side) {
        AY_CHECK_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
        AY_CHECK_HIP(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming));
        AY_CHECK_HIP(hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming));
    }
    AY_CHECK_HIP(hipEventRecord(g_ev_fork, main));
    AY_CHECK_HIP(hipStreamWaitEvent(g_side, g_ev_fork, 0));
    return AYOLO_OK;
}

extern "C" int ayolo_run_ops(const ayolo_op* ops, int n, ayolo_stream s) {
    AY_CHECK_ARG(ops || n == 0, "run_ops: null op list");
    static const bool debug_stall = getenv("AYOLO_DEBUG_STALL") != nullptr;
    double t_prev = debug_stall ? now_us() : 0.0;
    bool used_side = false;
    for (int k = 0; k < n; ++k) {
        const ayolo_op& o = ops[k];
        int rc = AYOLO_OK;
        // AYOLO_OP_SIDE: this op depends on everything enqueued so far but nothing later depends on it before the end of
        // the list (weight gradients): it runs on the executor's side stream, concurrently with the ops that follow
        ayolo_stream cs = s;
        if (o.kind & AYOLO_OP_SIDE) {
            rc = side_fork((hipStream_t)s);
            if (rc != AYOLO_OK) return rc;
            cs = (ayolo_stream)g_side;
            used_side = true;
        }
        if (debug_stall) {
            double t = now_us();
            if (t - t_prev > 500.0) fprintf(stderr, "[ayolo] run_ops: op %d/%d (kind %d) was blocked %.1f us in the HIP runtime\n", k - 1, n, k ? ops[k - 1].kind : 0, t - t_prev);
            t_prev = t;
        }
        switch (o.kind & 0xff) {
        case AYOLO_OP_NOP:
            break;
        case AYOLO_OP_CONV_FWD:
            rc = ayolo_conv_fwd(&o.conv, o.p[0], o.p[1], o.p[2], o.i[0], (const float*)o.p[3], (const float*)o.p[4],
                                (float*)o.p[5], o.i[1], o.i[2], cs);
            break;
        case AYOLO_OP_CONV_DGRAD:
            rc = ayolo_conv_dgrad(&o.conv, o.p[0], o.p[1], o.p[2], o.i[0], cs);
            break;
        case AYOLO_OP_CONV_WGRAD:
            rc = ayolo_conv_wgrad(&o.conv, o.p[0], o.p[1], (float*)o.p[2], o.f[0], cs);
            break;
        case AYOLO_OP_CAST_WEIGHT:
            rc = ayolo_cast_weight((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.p[1], o.p[2], cs);
            break;
        case AYOLO_OP_CAST_WEIGHTS:
            rc = ayolo_cast_weights((const ayolo_cast_job*)o.p[0], o.i[0], o.i[1], cs);
            break;
        case AYOLO_OP_BN_FINALIZE:
            rc = ayolo_bn_finalize((const float*)o.p[0], o.i[0], o.i[1], o.d[0], (const float*)o.p[1], (const float*)o.p[2], o.f[0],
                                   o.f[1], (float*)o.p[3], (float*)o.p[4], (float*)o.p[5], (float*)o.p[6], (float*)o.p[7],
                                   (float*)o.p[8], cs);
            break;
        case AYOLO_OP_AFFINE_ACT:
            rc = ayolo_affine_act_res(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], (const float*)o.p[2],
                                      (const float*)o.p[3], o.i[4], o.p[4], o.i[5], cs);
            break;
        case AYOLO_OP_BN_TRAIN_ACT:
            rc = ayolo_bn_train_act(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], (const float*)o.p[2], o.i[4], o.i[7], o.d[0],
                                    (const float*)o.p[3], (const float*)o.p[4], o.f[0], o.f[1], (float*)o.p[5], (float*)o.p[6],
                                    (float*)o.p[7], (float*)o.p[8], o.i[5], o.p[9], o.i[6], cs);
            break;
        case AYOLO_OP_BN_BWD_REDUCE:
            rc = ayolo_bn_act_bwd_reduce(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], (const float*)o.p[2],
                                         (const float*)o.p[3], (const float*)o.p[4], (const float*)o.p[5], o.i[4], (float*)o.p[6],
                                         o.i[5], cs);
            break;
        case AYOLO_OP_BN_BWD_APPLY:
            rc = ayolo_bn_act_bwd_apply(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.p[2], o.i[3], o.l[0], o.i[4],
                                        (const float*)o.p[3], (const float*)o.p[4], (const float*)o.p[5], (const float*)o.p[6],
                                        o.i[5], (const float*)o.p[7], o.i[6], (float*)o.p[8], (float*)o.p[9], o.f[0], cs);
            break;
        case AYOLO_OP_MAXPOOL_FWD:
            rc = ayolo_maxpool_fwd(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], (unsigned char*)o.p[2], o.i[3], o.i[4], o.i[5], o.i[6],
                                   o.i[7], cs);
            break;
        case AYOLO_OP_MAXPOOL_BWD:
            rc = ayolo_maxpool_bwd(o.i[0], (const unsigned char*)o.p[0], o.p[1], o.i[1], o.p[2], o.i[2], o.i[3], o.i[4], o.i[5],
                                   o.i[6], o.i[7], o.i[8], cs);
            break;
        case AYOLO_OP_UPSAMPLE_FWD:
            rc = ayolo_upsample2x_fwd(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], cs);
            break;
        case AYOLO_OP_UPSAMPLE_BWD:
            rc = ayolo_upsample2x_bwd(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.i[7], cs);
            break;
        case AYOLO_OP_PACK_INPUT:
            rc = ayolo_pack_input((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.p[1], o.i[5], cs);
            break;
        case AYOLO_OP_HEAD_GRAD_PACK:
            rc = ayolo_head_grad_pack((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.p[1], o.i[6],
                                      (float*)o.p[2], cs);
            break;
        case AYOLO_OP_COPY2D:
            rc = ayolo_copy2d(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], o.i[4], cs);
            break;
        case AYOLO_OP_MEMSET:
            rc = ayolo_fill_zero(o.p[0], (size_t)o.l[0], cs);
            break;
        case AYOLO_OP_BN_EVAL_AFFINE:
            rc = ayolo_bn_eval_affine((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2], (const float*)o.p[3],
                                      (const float*)o.p[4], o.f[0], o.i[0], (float*)o.p[5], (float*)o.p[6], cs);
            break;
        default:
            ayolo_set_error("run_ops: unknown op kind %d at index %d", o.kind, k);
            return AYOLO_EINVAL;
        }
        if (rc != AYOLO_OK) return rc;
    }
    if (used_side) {                       // the caller's stream continues only after the side stream has drained
        AY_CHECK_HIP(hipEventRecord(g_ev_join, g_side));
        AY_CHECK_HIP(hipStreamWaitEvent((hipStream_t)s, g_ev_join, 0));
    }
    return AYOLO_OK;
}
