import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
gen = torch.Generator().manual_seed(0)
imgs = torch.rand(64, 3, 640, 640, generator=gen).to(dev)
targets = bench.synth_targets(64, 8, gen).to(dev)
def step(m):
    with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
        pred = m(imgs)
        loss, _ = loss_fn(pred, targets)
    scaler.scale(loss).backward(); scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
def timeit(m, n=5):
    for _ in range(2): step(m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(m)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    return t_issue / n * 1e3, t / n * 1e3
print("eager  issue/wall ms:", timeit(model))
# forward-only and backward split timings
with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
    for _ in range(2): pred = model(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): pred = model(imgs)
    ti = time.perf_counter() - t0; torch.cuda.synchronize(); print("fwd issue/wall ms", ti / 5 * 1e3, (time.perf_counter() - t0) / 5 * 1e3)
try:
    with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
        g = torch.cuda.make_graphed_callables(model, (imgs,), num_warmup_iters=2)
    print("graphed issue/wall ms:", timeit(g))
except Exception as e:
    import traceback; traceback.print_exc()
