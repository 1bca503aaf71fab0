// Batched launcher: one C call enqueues a whole pre-compiled list of kernel launches (a model forward or
// backward) on one stream.  The op list is built once per (model, shape, dtype) by ayolov2_amd/plan.py over
// static buffers, so a training step costs two host calls instead of ~1600 Python-level launches, and the
// list is a straight-line HIP stream program (graph-capturable).
#include "common.h"
#include <stdlib.h>

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

// Side stream of the executor.  No process-global state: every (host thread, device) pair gets its own side stream and
// fork / join events on first use, so one process may drive several devices (the nn.DataParallel branch of
// scripts/train/train_model_builder.py:132-133 runs one thread per device) and two threads may run op lists on the
// same device without sharing events.  ayolo_release_thread_state() frees the calling thread's objects.
#define AY_MAX_DEVICES 16
// ONE side stream (several round-robin streams measured no better, profiles/r03_scheduling_ab.txt)
struct SideCtx { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static thread_local SideCtx t_side[AY_MAX_DEVICES];

static int side_ctx(SideCtx** out) {
    int dev = 0;
    AY_CHECK_HIP(hipGetDevice(&dev));
    AY_CHECK_ARG(dev >= 0 && dev < AY_MAX_DEVICES, "run_ops: device ordinal %d unsupported", dev);
    SideCtx& c = t_side[dev];
    if (!c.side) {
        // lowest priority: the side stream carries the weight gradients, which only have to finish by the end of the
        // list -- the dependent chain on the caller's stream (BN passes, dgrads) should win the CUs when both have work.
        // Measured alternatives, none kept: highest / default priority (+0.8 % / +0.3 %), several side streams, a CU mask
        // (hipExtStreamCreateWithCUMask, 64 / 128 CUs for the side stream: 24.0 / 21.4 ms per step against 15.1 --
        // profiles/r03_scheduling_ab.txt: the weight gradients are latency-bound per workgroup, so confining them to a
        // quarter of the chip stretches them far beyond the backward window)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        AY_CHECK_HIP(hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, least));
        AY_CHECK_HIP(hipEventCreateWithFlags(&c.fork, hipEventDisableTiming));
        AY_CHECK_HIP(hipEventCreateWithFlags(&c.join, hipEventDisableTiming));
    }
    *out = &c;
    return AYOLO_OK;
}

extern "C" int ayolo_release_thread_state(void) {
    for (int d = 0; d < AY_MAX_DEVICES; ++d) {
        SideCtx& c = t_side[d];
        if (!c.side) continue;
        (void)hipStreamSynchronize(c.side);
        (void)hipStreamDestroy(c.side);
        (void)hipEventDestroy(c.fork);
        (void)hipEventDestroy(c.join);
        c = SideCtx();
    }
    return AYOLO_OK;
}

// fork: the side stream waits for everything enqueued so far on the caller's stream; returns it
static int side_fork(SideCtx* c, hipStream_t main, hipStream_t* out) {
    AY_CHECK_HIP(hipEventRecord(c->fork, main));
    AY_CHECK_HIP(hipStreamWaitEvent(c->side, c->fork, 0));
    *out = c->side;
    return AYOLO_OK;
}

// the caller's stream continues only after the side stream has drained (also on the error path: work already forked
// must not outlive the call unordered)
static int side_join(SideCtx* c, hipStream_t main) {
    AY_CHECK_HIP(hipEventRecord(c->join, c->side));
    AY_CHECK_HIP(hipStreamWaitEvent(main, c->join, 0));
    return AYOLO_OK;
}

// Make stream `waiter` wait for everything enqueued so far on the calling thread's executor side stream (a no-op when
// this thread never used one on the current device).  With AYOLO_RUN_NO_JOIN this lets a communication stream pick up
// finished weight-gradient buckets without stalling the compute stream (trainer.FlatGradDDP).
extern "C" int ayolo_side_stream_join(ayolo_stream waiter) {
    int dev = 0;
    AY_CHECK_HIP(hipGetDevice(&dev));
    AY_CHECK_ARG(dev >= 0 && dev < AY_MAX_DEVICES, "side_stream_join: device ordinal %d unsupported", dev);
    SideCtx& c = t_side[dev];
    if (!c.side) return AYOLO_OK;
    return side_join(&c, (hipStream_t)waiter);
}

static int run_ops_impl(const ayolo_op* ops, int n, ayolo_stream s, int flags, hipEvent_t* ev);

extern "C" int ayolo_run_ops(const ayolo_op* ops, int n, ayolo_stream s) { return run_ops_impl(ops, n, s, 0, nullptr); }
extern "C" int ayolo_run_ops_ex(const ayolo_op* ops, int n, ayolo_stream s, int flags) { return run_ops_impl(ops, n, s, flags, nullptr); }

// Measurement mode of the executor (bench.py's in-situ roofline): the same list on the same streams, with a HIP event
// recorded before and after every op ON THE STREAM THE OP RUNS ON (side-stream weight gradients included), so ms[k] is
// the duration of op k inside the real step -- cold caches, concurrent side-stream work and all.  Blocks until the list
// has finished.
extern "C" int ayolo_run_ops_timed(const ayolo_op* ops, int n, ayolo_stream s, float* ms) {
    AY_CHECK_ARG((ops && ms) || n == 0, "run_ops_timed: null pointer");
    if (n <= 0) return AYOLO_OK;
    hipEvent_t* ev = (hipEvent_t*)calloc((size_t)2 * n, sizeof(hipEvent_t));
    AY_CHECK_ARG(ev, "run_ops_timed: out of memory");
    int rc = AYOLO_OK;
    for (int k = 0; k < 2 * n && rc == AYOLO_OK; ++k)
        if (hipEventCreate(&ev[k]) != hipSuccess) { ayolo_set_error("run_ops_timed: hipEventCreate failed"); rc = AYOLO_ELAUNCH; }
    if (rc == AYOLO_OK) rc = run_ops_impl(ops, n, s, 0, ev);
    if (rc == AYOLO_OK && hipStreamSynchronize((hipStream_t)s) != hipSuccess) { ayolo_set_error("run_ops_timed: sync failed"); rc = AYOLO_ELAUNCH; }
    for (int k = 0; k < n && rc == AYOLO_OK; ++k) {
        ms[k] = 0.0f;
        if ((ops[k].kind & 0xff) == AYOLO_OP_NOP) continue;
        if (hipEventElapsedTime(&ms[k], ev[2 * k], ev[2 * k + 1]) != hipSuccess) ms[k] = -1.0f;
    }
    for (int k = 0; k < 2 * n; ++k)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    free(ev);
    return rc;
}

static int run_ops_impl(const ayolo_op* ops, int n, ayolo_stream s, int flags, hipEvent_t* ev) {
    AY_CHECK_ARG(ops || n == 0, "run_ops: null op list");
    bool used_side = false;
    SideCtx* sc = nullptr;
    for (int k = 0; k < n; ++k) {
        const ayolo_op& o = ops[k];
        int rc = AYOLO_OK;
        // AYOLO_OP_SIDE: this op depends on everything enqueued so far but nothing later depends on it before the end of
        // the list (weight gradients): it runs on the executor's side stream, concurrently with the ops that follow
        ayolo_stream cs = s;
        if (o.kind & AYOLO_OP_SIDE) {
            if (!sc) rc = side_ctx(&sc);
            hipStream_t forked = nullptr;
            if (rc == AYOLO_OK) rc = side_fork(sc, (hipStream_t)s, &forked);
            if (rc != AYOLO_OK) {
                if (used_side) (void)side_join(sc, (hipStream_t)s);
                return rc;
            }
            cs = (ayolo_stream)forked;
            used_side = true;
        }
        const bool timed = ev && (o.kind & 0xff) != AYOLO_OP_NOP;
        if (timed) AY_CHECK_HIP(hipEventRecord(ev[2 * k], (hipStream_t)cs));
        switch (o.kind & 0xff) {
        case AYOLO_OP_NOP:
            break;
        case AYOLO_OP_CONV_FWD:
            // p[6] / p[7]: transform on load (ayolo_conv_fwd_xf) -- p[0] is segment 0 (channel stride conv.ldx, i[5] channels, or all
            // of them when there is no second segment), p[8] / i[4] segment 1 and its channel stride; i[3] bits: 0 act of segment 0,
            // 1 segment 0 is virtual (the producer's z), 2 / 3 the same for segment 1
            if (o.p[6]) {
                ayolo_xf_seg sg[2];
                const int two = o.p[8] != nullptr;
                sg[0].x = o.p[0]; sg[0].ld = o.conv.ldx; sg[0].C = two ? o.i[5] : o.conv.Cin; sg[0].act = o.i[3] & 1; sg[0].virt = (o.i[3] >> 1) & 1;
                sg[1].x = o.p[8]; sg[1].ld = o.i[4]; sg[1].C = o.conv.Cin - sg[0].C; sg[1].act = (o.i[3] >> 2) & 1; sg[1].virt = (o.i[3] >> 3) & 1;
                // p[9] / i[6]: host array of ayolo_xf_fin -- the producers' BatchNorm finalize inside this launch
                // p[10] / i[7]: store-back buffer (the materialised activation) and its channel stride
                rc = ayolo_conv_fwd_xf(&o.conv, sg, two ? 2 : 1, (float*)o.p[6], (float*)o.p[7], (const ayolo_xf_fin*)o.p[9], o.i[6], o.p[10], o.i[7],
                                       o.p[1], o.p[2], o.i[0], (const float*)o.p[4], (double*)o.p[5], o.i[1], o.i[2], cs);
            }
            else rc = ayolo_conv_fwd(&o.conv, o.p[0], o.p[1], o.p[2], o.i[0], (const float*)o.p[3], (const float*)o.p[4],
                                     (double*)o.p[5], o.i[1], o.i[2], cs);
            break;
        case AYOLO_OP_CONV_DGRAD:
            // i[1] > 0: the BatchNorm-backward sums of the i[1] block(s) that produced dx ride in the epilogue
            // (p[3]: host array of ayolo_bn_seg, i[2]: activation, i[3]: accumulator replicas)
            if (o.i[1] > 0) rc = ayolo_conv_dgrad_bn(&o.conv, o.p[0], o.p[1], o.p[2], o.i[0], (const ayolo_bn_seg*)o.p[3], o.i[1], o.i[2], o.i[3], cs);
            else rc = ayolo_conv_dgrad(&o.conv, o.p[0], o.p[1], o.p[2], o.i[0], cs);
            break;
        case AYOLO_OP_CONV_WGRAD:
            // p[3] / l[0]: split-K workspace and its bytes
            rc = ayolo_conv_wgrad(&o.conv, o.p[0], o.p[1], (float*)o.p[2], o.f[0], o.p[3], (size_t)o.l[0], cs);
            break;
        case AYOLO_OP_WGRAD_GROUP:
            rc = ayolo_wgrad_group_run(o.p[0], o.p[1], o.p[2], (size_t)o.l[0], (const void* const*)&o.p[3], o.i[0], cs);
            break;
        case AYOLO_OP_CAST_WEIGHT:
            rc = ayolo_cast_weight((const float*)o.p[0], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], o.i[6], o.p[1], o.p[2], cs);
            break;
        case AYOLO_OP_CAST_WEIGHTS:
            rc = ayolo_cast_weights((const ayolo_cast_job*)o.p[0], o.i[0], o.i[1], cs);
            break;
        case AYOLO_OP_BN_FINALIZE:
            // i[2]: channel stride of the accumulators (0: = C)
            rc = ayolo_bn_finalize_ld((const double*)o.p[0], o.i[0], o.i[2] > 0 ? o.i[2] : o.i[1], o.i[1], o.d[0], (const float*)o.p[1],
                                      (const float*)o.p[2], o.f[0], o.f[1], (float*)o.p[3], (float*)o.p[4], (float*)o.p[5], (float*)o.p[6],
                                      (float*)o.p[7], (float*)o.p[8], cs);
            break;
        case AYOLO_OP_AFFINE_ACT:
            rc = ayolo_affine_act_res(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], (const float*)o.p[2],
                                      (const float*)o.p[3], o.i[4], o.p[4], o.i[5], cs);
            break;
        case AYOLO_OP_BN_TRAIN_ACT:
            rc = ayolo_bn_train_act(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], (const double*)o.p[2], o.i[4], o.i[7], o.d[0],
                                    (const float*)o.p[3], (const float*)o.p[4], o.f[0], o.f[1], (float*)o.p[5], (float*)o.p[6],
                                    (float*)o.p[7], (float*)o.p[8], o.i[5], o.p[9], o.i[6], cs);
            break;
        case AYOLO_OP_BN_BWD_REDUCE:
            rc = ayolo_bn_act_bwd_reduce(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], o.i[3], (const float*)o.p[2],
                                         (const float*)o.p[3], (const float*)o.p[4], (const float*)o.p[5], o.i[4], (double*)o.p[6],
                                         o.i[5], cs);
            break;
        case AYOLO_OP_BN_BWD_APPLY:
            // p[10] / i[7] / i[8]: gradient buffer of a shortcut fed by the same output (NULL: none), its row stride, accumulate
            rc = ayolo_bn_act_bwd_apply_res(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.p[2], o.i[3], o.l[0], o.i[4],
                                            (const float*)o.p[3], (const float*)o.p[4], (const float*)o.p[5], (const float*)o.p[6],
                                            o.i[5], (const double*)o.p[7], o.i[6], (float*)o.p[8], (float*)o.p[9], o.f[0],
                                            o.p[10], o.i[7], o.i[8], cs);
            break;
        case AYOLO_OP_BN_BWD_APPLY2: {
            ayolo_bn_apply_seg g[2];
            for (int k = 0; k < 2; ++k) {
                void* const* q = o.p + 2 + 7 * k;
                g[k].da = q[0];
                g[k].save_mean = (const float*)q[1];
                g[k].save_invstd = (const float*)q[1] + o.i[5 + k];
                g[k].gamma = (const float*)q[2]; g[k].beta = (const float*)q[3];
                g[k].sums = (const double*)q[4];
                g[k].dgamma = (float*)q[5]; g[k].dbeta = (float*)q[6];
                g[k].C = o.i[5 + k]; g[k].ldda = o.i[7 + k];
            }
            rc = ayolo_bn_act_bwd_apply2(o.i[0], o.p[0], o.i[1], o.p[1], o.i[2], o.l[0], &g[0], &g[1], o.i[3], o.i[4], o.f[0], cs);
            break;
        }
        case AYOLO_OP_SPPF_FWD:
            rc = ayolo_sppf_pool_fwd(o.i[0], o.p[0], o.i[1], (unsigned char*)o.p[1], o.i[2], o.i[3], o.i[4], o.i[5], cs);
            break;
        case AYOLO_OP_SPPF_BWD:
            rc = ayolo_sppf_pool_bwd(o.i[0], (const unsigned char*)o.p[0], o.p[1], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], cs);
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
        case AYOLO_OP_JOIN_SIDE:
            if (used_side) rc = side_join(sc, (hipStream_t)s);
            break;
        case AYOLO_OP_STEM_BN_WGRAD:
            // p: x, z, da, mean, invstd, gamma, beta, sums, dw, dgamma, dbeta; i: ldz, act, sum_reps; f: alpha, grad_scale
            rc = ayolo_stem_bn_wgrad(&o.conv, o.p[0], o.p[1], o.i[0], o.p[2], (const float*)o.p[3], (const float*)o.p[4],
                                     (const float*)o.p[5], (const float*)o.p[6], o.i[1], (const double*)o.p[7], o.i[2],
                                     (float*)o.p[8], (float*)o.p[9], (float*)o.p[10], o.f[0], o.f[1], cs);
            break;
        case AYOLO_OP_HEAD_DECODE: {
            const int64_t st[4] = {o.i[5], o.i[6], o.i[7], o.i[8]};            // element strides of (b, a, y, x)
            rc = ayolo_head_decode((const float*)o.p[0], st, o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], (const float*)o.p[1], o.f[0],
                                   (float*)o.p[2], (int64_t)o.l[0], (int64_t)o.i[9], cs);
            break;
        }
        case AYOLO_OP_BN_EVAL_AFFINE:
            rc = ayolo_bn_eval_affine((const float*)o.p[0], (const float*)o.p[1], (const float*)o.p[2], (const float*)o.p[3],
                                      (const float*)o.p[4], o.f[0], o.i[0], (float*)o.p[5], (float*)o.p[6], cs);
            break;
        default:
            ayolo_set_error("run_ops: unknown op kind %d at index %d", o.kind, k);
            rc = AYOLO_EINVAL;
        }
        if (rc != AYOLO_OK) {
            if (used_side) (void)side_join(sc, (hipStream_t)s);
            return rc;
        }
        if (timed) AY_CHECK_HIP(hipEventRecord(ev[2 * k + 1], (hipStream_t)cs));
    }
    // AYOLO_RUN_NO_JOIN: a segment of a longer list -- side-stream work stays in flight, the caller joins later
    if (used_side && !(flags & AYOLO_RUN_NO_JOIN)) return side_join(sc, (hipStream_t)s);
    return AYOLO_OK;
}
