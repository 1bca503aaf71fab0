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

# ---- GPU-side section times (events in stream order; include any idle gaps inside a section)
names = ("prepare", "forward", "loss", "backward", "optim")
acc = {k: 0.0 for k in names}
def step_ev():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    prep = loss_fn.prepare(tc, shapes, dev); ev[1].record()
    with torch.autocast("cuda", dtype=torch.float16):
        pred = run_model(imgs); ev[2].record()
        loss, _ = loss_fn(pred, targets, prepared=prep)
    ev[3].record()
    scaler.scale(loss).backward(); ev[4].record()
    scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True); ev[5].record()
    return ev
for _ in range(2): step_ev()
torch.cuda.synchronize()
t0 = time.perf_counter()
evs = [step_ev() for _ in range(5)]
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
for ev in evs:
    for k in range(5): acc[names[k]] += ev[k].elapsed_time(ev[k + 1])
print("GPU-side ms per section:", {k: round(v / 5, 2) for k, v in acc.items()}, "wall/step", round(wall, 2))
print("step-to-step GPU ms:", [round(evs[i][0].elapsed_time(evs[i + 1][0]), 2) for i in range(4)])

# ---- isolated sections (sync before/after each)
def sync_time(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, r
iso = {k: 0.0 for k in names}
for _ in range(4):
    d, prep = sync_time(lambda: loss_fn.prepare(tc, shapes, dev)); iso["prepare"] += d
    with torch.autocast("cuda", dtype=torch.float16):
        d, pred = sync_time(lambda: run_model(imgs)); iso["forward"] += d
        d, lo = sync_time(lambda: loss_fn(pred, targets, prepared=prep)); iso["loss"] += d
    d, _ = sync_time(lambda: scaler.scale(lo[0]).backward()); iso["backward"] += d
    d, _ = sync_time(lambda: (scaler.step(opt), scaler.update(), opt.zero_grad(set_to_none=True))); iso["optim"] += d
print("isolated (synced) ms per section:", {k: round(v / 4, 2) for k, v in iso.items()}, "sum", round(sum(iso.values()) / 4, 2))
print("mem allocated GB", torch.cuda.memory_allocated() / 1e9, "reserved", torch.cuda.memory_reserved() / 1e9, "alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"), "cudaMalloc calls", torch.cuda.memory_stats().get("num_device_alloc"))
