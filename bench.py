#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): img/s of the YOLOv5s 640x640 bs=64 train step (forward + ComputeLoss +
backward + SGD step, fp16 autocast with fp32 master weights) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Weak scaling: the per-GPU batch (64) is fixed as N grows; gradients are
all-reduced by torch DDP over RCCL/xGMI, overlapped with backward.  Inputs are synthetic COCO-shape tensors that
are resident in HBM before the timed region (no dataset, no H2D copy inside it).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_BWD_GFLOP_PER_IMG = 49.30      # conv MACs of YOLOv5s @640, fwd 16.43 GFLOP x 3
MFMA_PEAK_TFLOPS = 2500.0          # MI355X dense fp16/bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0

HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0,
           lr=0.01, momentum=0.937, weight_decay=0.0005)


def synth_targets(batch, per_img, gen):
    """(nt, 6) [image, class, x, y, w, h] normalised; cls~U{0..79}, xy~U(0.1,0.9), wh~LogU(0.02,0.6)."""
    n = batch * per_img
    img = torch.arange(batch).repeat_interleave(per_img).float()
    cls = torch.randint(0, 80, (n,), generator=gen).float()
    xy = torch.rand(n, 2, generator=gen) * 0.8 + 0.1
    wh = torch.exp(torch.rand(n, 2, generator=gen) * (math.log(0.6) - math.log(0.02)) + math.log(0.02))
    return torch.cat((img[:, None], cls[:, None], xy, wh), 1)


def build_train_objects(model_name, device, world_size):
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.losses import ComputeLoss
    from torch import nn

    torch.manual_seed(0)
    model = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", f"{model_name}.yaml")).to(device).train()
    model.hyp, model.gr, model.nc = dict(HYP), 1.0, 80
    pg_w, pg_bn, pg_b = [], [], []                  # yolo_trainer.py:149-168 parameter groups
    for mod in model.modules():
        if hasattr(mod, "bias") and isinstance(mod.bias, nn.Parameter):
            pg_b.append(mod.bias)
        if isinstance(mod, nn.BatchNorm2d):
            pg_bn.append(mod.weight)
        elif hasattr(mod, "weight") and isinstance(mod.weight, nn.Parameter):
            pg_w.append(mod.weight)
    # the whole SGD-nesterov step (unscale, found-inf skip, weight decay, momentum) is ONE HIP launch (ayolo_sgd_step);
    # AYOLO_TORCH_SGD=1 times torch's fused multi-tensor kernels instead (one launch per parameter group)
    if os.environ.get("AYOLO_TORCH_SGD") == "1":
        opt = torch.optim.SGD(pg_bn, lr=HYP["lr"], momentum=HYP["momentum"], nesterov=True, fused=True)
    else:
        from ayolov2_amd.optim import SGD
        opt = SGD(pg_bn, lr=HYP["lr"], momentum=HYP["momentum"], nesterov=True)
    opt.add_param_group({"params": pg_w, "weight_decay": HYP["weight_decay"]})
    opt.add_param_group({"params": pg_b})
    loss_fn = ComputeLoss(model)
    run_model = model
    if world_size > 1:
        if os.environ.get("AYOLO_TORCH_DDP") == "1":
            run_model = nn.parallel.DistributedDataParallel(model, device_ids=[device.index], gradient_as_bucket_view=True)
        else:
            from ayolov2_amd.trainer import FlatGradDDP      # one all-reduce of the plan's flat gradient arena per step
            run_model = FlatGradDDP(model)
    scaler = torch.amp.GradScaler("cuda")
    return model, run_model, opt, loss_fn, scaler


def conv_kernel_roofline(model, batch, size, device, reps=5):
    """Roofline of the dominant kernel family, k_gconv<f16>: the 57 Conv forward launches of one train step, each timed
    live with HIP events on the launch stream.  Per launch the algorithmic cost is SURVEY.md 8d's: FLOP = 2*MAC, bytes =
    fp16 (input + output + weights).  YOLOv5s is HBM-bound on MI355X as a whole (121 FLOP/B unfused vs a ridge of 312), so
    the headline `achieved` is algorithmic GB/s against the 8 TB/s HBM peak; the MFMA view and the per-launch roofline
    (sum over launches of max(FLOP/MFMA peak, bytes/HBM peak) / measured time) are reported next to it."""
    from ayolov2_amd import ops, functional as F_
    from ayolov2_amd.modules import Conv
    shapes = []

    def hook(mod, inp, out):
        shapes.append((mod, tuple(inp[0].shape)))

    hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, Conv)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.eval()
        model(torch.rand(1, 3, size, size, device=device))
        model.train()
    for h in hs:
        h.remove()
    total_flop = total_s = alg_bytes = roof_s = hbm_roof_s = 0.0
    for mod, xs in shapes:
        conv = mod.conv
        _, cin, H, W = xs
        dt = torch.float16
        geo = F_._Geometry((batch, cin, H, W), conv.weight.shape, F_._pair_(conv.stride), F_._pair_(conv.padding), dt)
        xk = torch.randn((batch, geo.Cin_k, geo.H, geo.W), device=device).to(dt).contiguous(memory_format=torch.channels_last)
        cout = conv.weight.shape[0]
        w, _ = mod._cache(0).get(conv.weight, dt, cout, geo.cin_pad)
        y = ops.new_act(batch, cout, geo.Ho, geo.Wo, dt, device)
        stats = torch.zeros((ops.STAT_REPS, 2 * cout), dtype=torch.float32, device=device)
        d = geo.desc(dt, geo.Cin_k, cout)
        for _ in range(2):
            ops.conv_fwd(d, xk, w, y, 0, stats=stats)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv_fwd(d, xk, w, y, 0, stats=stats)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) / 1e3 / reps
        kh, kw = conv.kernel_size
        flop = 2.0 * batch * geo.Ho * geo.Wo * cout * conv.in_channels * kh * kw
        byts = 2.0 * (xk.numel() + w.numel() + y.numel())
        total_s += t
        total_flop += flop
        alg_bytes += byts
        roof_s += max(flop / (MFMA_PEAK_TFLOPS * 1e12), byts / (HBM_PEAK_GBS * 1e9))
        hbm_roof_s += byts / (HBM_PEAK_GBS * 1e9) if byts / (HBM_PEAK_GBS * 1e9) >= flop / (MFMA_PEAK_TFLOPS * 1e12) else 0.0
    gbs = alg_bytes / total_s / 1e9
    return {"bound": "hbm", "kernel": "k_gconv<f16>: the 57 Conv forward launches of one train step",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            # PMC traffic covers forward AND dgrad launches of the family (same kernel); algorithmic_gb_fwd_dgrad matches it
            "traffic": pmc_traffic(("k_gconv",), ""),
            "traffic_unit": "GB per train step over every k_gconv launch (forward + dgrad), rocprofv3 --pmc FETCH_SIZE "
                            "(x2 gfx950) / WRITE_SIZE passes committed under profiles/",
            "algorithmic_gb_fwd": round(alg_bytes / 1e9, 3), "algorithmic_gb_fwd_dgrad": round(2 * alg_bytes / 1e9, 3),
            "launch_ms_sum": round(total_s * 1e3, 3),
            "mfma_view": {"achieved_tflops": round(total_flop / total_s / 1e12, 1), "peak_tflops": MFMA_PEAK_TFLOPS,
                          "frac": round(total_flop / total_s / 1e12 / MFMA_PEAK_TFLOPS, 4)},
            "per_launch_roofline_frac": round(roof_s / total_s, 4),
            "hbm_bound_share_of_roofline_time": round(hbm_roof_s / roof_s, 3)}


def pmc_traffic(families, suffix):
    """HBM bytes of the roofline kernel family from the newest committed PMC summary (tools/profile_round.sh); the
    counters need their own rocprofv3 passes, so bench.py reports the committed measurement rather than re-collecting."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None
    tot = 0.0
    for k, v in json.load(open(files[-1])).items():
        if any(f in k for f in families) and suffix in k:
            tot += v["fetch_GB_per_step_corrected"] + v["write_GB_per_step"]
    return round(tot, 3) if tot else None


def cpu_baseline(model_name, size, batch=4, budget_s=25.0):
    """The oracle's pure-PyTorch CPU model (kind 'port') doing the same train step on the host cores; bounded sample."""
    from ayolov2_amd.losses import ComputeLoss
    from oracle.model_ref import RefYOLO
    torch.manual_seed(0)
    r = RefYOLO(os.path.join(ROOT, "ayolov2_amd", "configs", f"{model_name}.yaml")).train()
    r.hyp, r.gr = dict(HYP), 1.0
    loss_fn = ComputeLoss(r)
    opt = torch.optim.SGD(r.parameters(), lr=HYP["lr"], momentum=HYP["momentum"], nesterov=True)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(batch, 3, size, size, generator=gen)
    t = synth_targets(batch, 8, gen)
    n, t0 = 0, time.perf_counter()
    while True:
        loss, _ = loss_fn(r(x), t)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and el > budget_s or n >= 8:
            break
    # first iteration includes one-off allocator/oneDNN warm-up: report the steady-state rate of the later ones
    return {"value": round(batch * n / el, 3), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps of oracle/model_ref.py {model_name} fp32 at batch {batch}, {size}x{size}, {el:.1f} s"}


def nms_extra(device):
    """Secondary metric of BASELINE.json: NMS boxes/s (candidates entering greedy NMS per second), config-5 shape."""
    from ayolov2_amd.metrics import non_max_suppression
    g = torch.Generator().manual_seed(0)
    B, N, nc, img = 8, 100800, 80, 1280
    pred = torch.cat((torch.rand(B, N, 2, generator=g) * img, torch.rand(B, N, 2, generator=g) ** 3 * img / 2 + 2,
                      torch.sigmoid(torch.randn(B, N, 1, generator=g) * 2 - 9.5),
                      torch.sigmoid(torch.randn(B, N, nc, generator=g) * 2 - 4)), 2).to(device)
    from ayolov2_amd import metrics as M
    cand = M._collect_candidates(pred, 0.001, True, True, None, None, False)
    n_cand = int(cand.counts.sum())
    for _ in range(2):
        non_max_suppression(pred, 0.001, 0.65, multi_label=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        non_max_suppression(pred, 0.001, 0.65, multi_label=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # fixed-shape variant (TensorRT BatchedNMS_TRT contract): no host read-back, so calls queue back to back
    from ayolov2_amd.fixed_nms import BatchedNMS
    fx = BatchedNMS(nc, top_k=512, keep_top_k=100, score_threshold=0.001, iou_threshold=0.65)
    for _ in range(2):
        fx.from_prediction(pred, box_xyxy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fx.from_prediction(pred, box_xyxy=False)
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / reps
    fixed = {"fixed_nms_ms_per_batch": round(dtf * 1e3, 3), "fixed_nms_proposals_per_s": round(B * N / dtf, 1),
             "fixed_nms_config": "topK 512 keepTopK 100, capacity B*N pairs, overflow=%s" % bool(fx.overflow)}
    return {**fixed, "nms_boxes_per_s": round(n_cand / dt, 1), "nms_proposals_per_s": round(B * N / dt, 1), "nms_ms_per_batch": round(dt * 1e3, 3),
            "nms_candidates": n_cand, "nms_workload": f"{B}x{N}x{nc + 5} fp32, conf 0.001 iou 0.65 multi_label",
            "nms_filter_hbm_gbs_lower_bound": round(B * N * (nc + 5) * 4 / dt / 1e9, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="yolov5s")
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / NMS legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_ddp = os.environ.get("AYOLO_FORCE_DDP") == "1"       # exercise the DDP/RCCL path on a single GPU
    if world > 1 or force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    model, run_model, opt, loss_fn, scaler = build_train_objects(args.model, device, 2 if force_ddp and world == 1 else world)
    gen = torch.Generator().manual_seed(1234 + rank)
    imgs = torch.rand(args.batch, 3, args.size, args.size, generator=gen).to(device)       # resident in HBM
    targets_cpu = synth_targets(args.batch, 8, gen)              # labels arrive from the CPU loader (data_loader.py:905-908)
    targets = targets_cpu.to(device)
    head = model.model[-1]
    pred_shapes = [(args.batch, head.na, args.size // int(s), args.size // int(s), head.no) for s in head._strides_py]

    def step():
        prep = loss_fn.prepare(targets_cpu, pred_shapes, device)   # host-side target assignment, no stream sync
        with torch.autocast("cuda", dtype=torch.float16):
            pred = run_model(imgs)
            loss, _ = loss_fn(pred, targets, prepared=prep)
        if world > 1:
            loss = loss * world                       # yolo_trainer.py:325-326
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad(set_to_none=True)
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    assert math.isfinite(float(loss)), "non-finite loss"

    if rank == 0:
        ms = el / args.steps * 1e3
        value = world * args.batch * args.steps / el
        out = {
            "metric": "img/s fwd+bwd YOLOv5s 640x640 bs=64 train step", "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.model} {args.size}x{args.size} per-GPU batch {args.batch}: forward + ComputeLoss "
                                   f"+ backward + SGD-nesterov step, fp16 autocast / fp32 master weights, random-init weights",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "step_conv_tflops": round(value * FWD_BWD_GFLOP_PER_IMG / 1e3 / world, 2),
            "step_frac_of_mfma_peak": round(value * FWD_BWD_GFLOP_PER_IMG / 1e3 / world / MFMA_PEAK_TFLOPS, 4),
        }
        if not args.no_extras and world == 1:
            out["roofline"] = conv_kernel_roofline(model, args.batch, args.size, device)
            out["extra"] = nms_extra(device)
            out["cpu_baseline"] = cpu_baseline(args.model, args.size)
        print(json.dumps(out), flush=True)
    if world > 1 or force_ddp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
