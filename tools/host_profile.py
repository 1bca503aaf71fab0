import cProfile, pstats, io, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
gen = torch.Generator().manual_seed(0)
imgs = torch.rand(64, 3, 640, 640, generator=gen).to(dev)
tc = bench.synth_targets(64, 8, gen); targets = tc.to(dev)
head = model.model[-1]
shapes = [(64, 3, 640 // int(s), 640 // int(s), 85) for s in head._strides_py]
T = {}
def step(timed=False):
    t0 = time.perf_counter()
    prep = loss_fn.prepare(tc, shapes, dev)
    t1 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.float16):
        pred = run_model(imgs)
        t2 = time.perf_counter()
        loss, _ = loss_fn(pred, targets, prepared=prep)
    t3 = time.perf_counter()
    scaler.scale(loss).backward()
    t4 = time.perf_counter()
    scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
    t5 = time.perf_counter()
    if timed:
        for k, v in zip(("prepare", "forward", "loss", "backward", "optim"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            T[k] = T.get(k, 0) + v
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(5): step(True)
torch.cuda.synchronize()
print({k: round(v / 5 * 1e3, 2) for k, v in T.items()})
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
