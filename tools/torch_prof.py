import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5s", dev, 1)
gen = torch.Generator().manual_seed(0)
imgs = torch.rand(64, 3, 640, 640, generator=gen).to(dev)
tc = bench.synth_targets(64, 8, gen); targets = tc.to(dev)
head = model.model[-1]
shapes = [(64, 3, 640 // int(s), 640 // int(s), 85) for s in head._strides_py]
def step():
    prep = loss_fn.prepare(tc, shapes, dev)
    with torch.autocast("cuda", dtype=torch.float16):
        pred = run_model(imgs)
        loss, _ = loss_fn(pred, targets, prepared=prep)
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad(set_to_none=True)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
evs=[e for e in prof.events() if e.name=="aten::copy_"]
evs.sort(key=lambda e:-e.cpu_time_total)
for e in evs[:12]:
    print(round(e.cpu_time_total),"us", e.input_shapes, [str(f)[-70:] for f in (e.stack or [])[:4]])
import collections
c=collections.Counter(str(e.input_shapes) for e in evs)
print(c.most_common(12))
