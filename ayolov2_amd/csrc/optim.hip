// Multi-tensor SGD (momentum / nesterov / weight decay) with the GradScaler hand-shake, every parameter tensor of a
// model in ONE launch.  Replaces the per-parameter-group torch kernels behind `scaler.step(optimizer)` in the
// reference's training step (scripts/train/yolo_trainer.py:332-338; optimiser built at :149-168 as
// SGD(pg_bn, nesterov=True) + {pg_w, weight_decay} + {pg_b}).
//
// Arithmetic per element, in torch.optim.SGD's order (torch/optim/sgd.py _single_tensor_sgd):
//   g = grad / grad_scale (when a scale is given);  g += weight_decay * p;
//   buf = (first || isnan(buf)) ? g : momentum * buf + (1 - dampening) * g;   g = nesterov ? g + momentum * buf : buf;   p -= lr * g
// and nothing at all when *found_inf != 0 (the step GradScaler would skip).  A NaN momentum entry means "not yet
// initialised": the host creates buffers NaN-filled, so a FIRST step that the scaler skips on the device leaves them
// uninitialised exactly as torch leaves momentum_buffer = None (the host cannot know without a sync).
#include "common.h"

__global__ __launch_bounds__(256) void k_sgd_step(const ayolo_sgd_job* jobs, ayolo_sgd_groups G, const float* grad_scale,
                                                  const float* found_inf) {
    if (found_inf && *found_inf != 0.0f) return;
    const ayolo_sgd_job J = jobs[blockIdx.y];
    const ayolo_sgd_group g = G.g[J.group];
    const float inv_one = 1.0f;
    const float scale = grad_scale ? *grad_scale : inv_one;
    const bool scaled = grad_scale != nullptr;
    const long long n4 = (J.n % 4 == 0 && ((uintptr_t)J.p % 16 == 0) && ((uintptr_t)J.g % 16 == 0) &&
                          (!J.buf || (uintptr_t)J.buf % 16 == 0)) ? J.n / 4 : 0;
    // vector body (all tensors of the YOLOv5 models qualify), scalar tail / fallback
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 p = reinterpret_cast<float4*>(J.p)[i];
        const float4 gr = reinterpret_cast<const float4*>(J.g)[i];
        float4 b = J.buf ? reinterpret_cast<float4*>(J.buf)[i] : make_float4(0, 0, 0, 0);
        float* pv = reinterpret_cast<float*>(&p);
        const float* gv = reinterpret_cast<const float*>(&gr);
        float* bv = reinterpret_cast<float*>(&b);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = scaled ? gv[e] / scale : gv[e];
            if (g.weight_decay != 0.0f) d = d + g.weight_decay * pv[e];
            if (g.momentum != 0.0f) {
                bv[e] = (J.first || bv[e] != bv[e]) ? d : g.momentum * bv[e] + (1.0f - g.dampening) * d;
                d = g.nesterov ? d + g.momentum * bv[e] : bv[e];
            }
            pv[e] = pv[e] - g.lr * d;
        }
        reinterpret_cast<float4*>(J.p)[i] = p;
        if (J.buf && g.momentum != 0.0f) reinterpret_cast<float4*>(J.buf)[i] = b;
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (long long)gridDim.x * 256) {
        float d = scaled ? J.g[i] / scale : J.g[i];
        float pv = J.p[i];
        if (g.weight_decay != 0.0f) d = d + g.weight_decay * pv;
        if (g.momentum != 0.0f) {
            const float b0 = J.buf[i];
            float b = (J.first || b0 != b0) ? d : g.momentum * b0 + (1.0f - g.dampening) * d;
            J.buf[i] = b;
            d = g.nesterov ? d + g.momentum * b : b;
        }
        J.p[i] = pv - g.lr * d;
    }
}

extern "C" int ayolo_sgd_step(const ayolo_sgd_job* jobs_dev, int njobs, const ayolo_sgd_groups* groups,
                              const float* grad_scale, const float* found_inf, ayolo_stream s) {
    AY_CHECK_ARG(jobs_dev && groups && njobs > 0 && njobs <= 65535, "sgd_step: njobs=%d", njobs);
    // blockIdx.y = job; callers split large tensors into jobs of a few 10^4 elements (8 workgroups x 256 lanes x float4
    // cover 8192 elements per sweep)
    hipLaunchKernelGGL(k_sgd_step, dim3(8, (unsigned)njobs), dim3(256), 0, (hipStream_t)s, jobs_dev, *groups, grad_scale,
                       found_inf);
    AY_CHECK_LAUNCH("k_sgd_step");
    return AYOLO_OK;
}
