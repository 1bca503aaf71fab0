import os, sys, time, torch, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
model, run_model, opt, loss_fn, scaler = bench.build_train_objects("yolov5n", dev, 1)
imgs = torch.rand(8, 3, 256, 256, device=dev)
print("eager fwd"); 
with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
    for _ in range(3): out = model(imgs)
torch.cuda.synchronize()
print("capture fwd only")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
    model(imgs)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
    with torch.cuda.graph(g):
        out_g = model(imgs)
print("captured; replay")
g.replay(); torch.cuda.synchronize()
print("replay ok", float(out_g[0].float().abs().mean()))
print("make_graphed_callables")
with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
    gm = torch.cuda.make_graphed_callables(model, (imgs,), num_warmup_iters=2)
print("graphed ok")
with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
    o = gm(imgs)
sum(t.float().square().mean() for t in o).backward()
torch.cuda.synchronize()
print("graphed fwd+bwd ok", float(model.model[0].conv.weight.grad.abs().mean()))
