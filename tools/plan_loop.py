import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from ayolov2_amd import plan as P
dev = torch.device("cuda", 0)
model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
imgs = torch.rand(64, 3, 640, 640, device=dev)
with torch.autocast("cuda", dtype=torch.float16):
    raws = model(imgs)
plan = [v for v in model._plans.values()][0]
draws = [torch.randn(r.shape, device=dev) * 1e-3 for r in raws]
def loop(fn, n=30, sync_each=False):
    ts = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(); evs[0].record()
    for i in range(n):
        fn()
        evs[i + 1].record()
        if sync_each: evs[i + 1].synchronize()
        elif i >= 1: evs[i - 1].synchronize()       # keep ~1 iteration of run-ahead
    torch.cuda.synchronize()
    return [round(evs[i].elapsed_time(evs[i + 1]), 1) for i in range(n)]
print("fwd only :", loop(lambda: plan.run_forward(imgs)))
print("bwd only :", loop(lambda: plan.run_backward(draws)))
print("fwd+bwd  :", loop(lambda: (plan.run_forward(imgs), plan.run_backward(draws))))
print("fwd+bwd sync each:", loop(lambda: (plan.run_forward(imgs), plan.run_backward(draws)), sync_each=True))

gen = torch.Generator().manual_seed(0)
tc = bench.synth_targets(64, 8, gen); targets = tc.to(dev)
head = model.model[-1]
shapes = [(64, 3, 640 // int(s), 640 // int(s), 85) for s in head._strides_py]
prep0 = loss_fn.prepare(tc, shapes, dev)
def A(prep=None):
    with torch.autocast("cuda", dtype=torch.float16):
        pred = model(imgs)
        loss, _ = loss_fn(pred, targets, prepared=prep or prep0)
    loss.backward()
    model.zero_grad(set_to_none=True)
def A2():
    with torch.autocast("cuda", dtype=torch.float16):
        pred = model(imgs)
        loss = sum((p.float() ** 2).mean() for p in pred)
    loss.backward()
    model.zero_grad(set_to_none=True)
def B(prep=None):
    with torch.autocast("cuda", dtype=torch.float16):
        pred = model(imgs)
        loss, _ = loss_fn(pred, targets, prepared=prep or prep0)
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
def C():
    B(loss_fn.prepare(tc, shapes, dev))
print("A2 model+trivial loss+bwd:", loop(A2, 20))
print("A  model+loss+bwd        :", loop(A, 20))
print("B  A+scaler+opt          :", loop(B, 20))
print("C  B+prepare             :", loop(C, 20))

for mode in ("side", "pageable", "same"):
    loss_fn.h2d_mode = mode
    print("C mode", mode, ":", loop(C, 16))
def D():
    with torch.autocast("cuda", dtype=torch.float16):
        pred = model(imgs)
        loss, _ = loss_fn(pred, targets)          # GPU-side build_targets (syncs)
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
print("D gpu build_targets:", loop(D, 16))
loss_fn.h2d_mode = "same"
def C_nocopy():
    t_cpu = tc
    fake = [torch.empty(tuple(s), device="meta") for s in shapes]
    loss_fn.build_targets(fake, t_cpu, anchors=loss_fn._anchors_cpu)     # CPU work only, result dropped
    B(prep0)
print("C cpu-work only, no copy:", loop(C_nocopy, 16))
torch.set_num_threads(1)
print("C 1 thread (same)        :", loop(C, 16))
print("C_nocopy 1 thread        :", loop(C_nocopy, 16))
