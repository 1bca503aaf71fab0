"""N > 1 path on CPU: two gloo ranks run the package's TrainModelBuilder + training_step around a CPU module
(the oracle network -- the HIP modules refuse CPU tensors by design) and must end with identical parameters that
equal a single-process run on the concatenated batch (gradient averaging x world_size loss convention)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    from oracle.model_ref import RefYOLO
    torch.manual_seed(0)
    m = RefYOLO(os.path.join(ROOT, "ayolov2_amd", "configs", "yolov5n.yaml"))
    for mod in m.modules():                      # BN in eval mode: the check is about gradient sync, not batch stats
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    m.hyp, m.gr = dict(HYP), 1.0
    return m


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    imgs = torch.rand(2, 3, 64, 64, generator=g)
    t = torch.tensor([[0, 3, 0.5, 0.5, 0.3, 0.4], [1, 7, 0.3, 0.6, 0.2, 0.2]])
    t[:, 2:] += 0.05 * rank
    return imgs, t


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from ayolov2_amd.losses import ComputeLoss
    from ayolov2_amd.trainer import TrainModelBuilder, training_step
    model = _make_model()
    b = TrainModelBuilder(model, {"train": {"batch_size": 4}})
    b.cuda = False
    b.device = torch.device("cpu")
    b.ddp_init()
    ddp, _, _ = b.prepare()
    loss_fn = ComputeLoss(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    imgs, t = _data(rank)
    for mod in ddp.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    training_step(ddp, loss_fn, opt, None, imgs, t, world_size=world, amp=False)
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_ddp_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "rank0.pt"))
    b = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in a:
        assert torch.equal(a[k], b[k]), f"ranks diverged at {k}"
    # single process reference: sum of the per-rank losses (mean-over-ranks gradient x world_size)
    sys.path.insert(0, ROOT)
    from ayolov2_amd.losses import ComputeLoss
    model = _make_model()
    loss_fn = ComputeLoss(model)
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    total = 0
    for r in range(2):
        imgs, t = _data(r)
        l, _ = loss_fn(model(imgs), t)
        total = total + l
    total.backward()
    opt.step()
    ref = model.state_dict()
    for k in a:
        if a[k].dtype.is_floating_point:
            assert torch.allclose(a[k], ref[k], rtol=1e-4, atol=1e-6), k


def _flat_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from ayolov2_amd.trainer import FlatGradDDP
    dist.init_process_group("gloo")
    torch.manual_seed(10 + rank)                      # ranks start from DIFFERENT parameters
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    w = FlatGradDDP(net)
    sync = net._ayolo_grad_sync
    flat = torch.arange(8, dtype=torch.float32) * (rank + 1)      # what the plan's gradient arena would hold
    sync.reduce_flat(flat)
    # the plan's overlapped exchange: the same arena reduced bucket by bucket (reverse-layer order) must equal the single
    # all-reduce; sync_bn's in-stream average of a statistics slice uses the same primitive
    flat_b = torch.arange(8, dtype=torch.float32) * (rank + 1)
    for lo, hi in ((5, 8), (2, 5), (0, 2)):
        sync.launch_bucket(flat_b[lo:hi])
    sync.wait_all()
    assert torch.equal(flat_b, flat), (flat_b, flat)
    stats = torch.tensor([2.0, 4.0]) * (rank + 1)
    sync.average_now(stats)
    assert torch.equal(stats, torch.tensor([3.0, 6.0]))
    import pickle
    restored = pickle.loads(pickle.dumps(sync))                    # checkpoints pickle the model: no process group inside
    assert restored.world == 1 and restored.group is None
    assert len(list(w.modules())) == len(list(net.modules())) + 1  # the sync object is not a submodule (no cycle)
    torch.save({"state": {k: v.clone() for k, v in net.state_dict().items()}, "flat": flat}, os.path.join(out, f"flat{rank}.pt"))
    dist.destroy_process_group()


def test_flat_grad_ddp_two_ranks_gloo(tmp_path):
    """FlatGradDDP (the plan executor's data-parallel wrapper): rank-0 broadcast of parameters and buffers at
    construction, one averaged all-reduce of the flat gradient arena."""
    port = _free_port()
    mp.spawn(_flat_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "flat0.pt"))
    b = torch.load(os.path.join(tmp_path, "flat1.pt"))
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), f"not broadcast: {k}"
    want = torch.arange(8, dtype=torch.float32) * 1.5               # mean of x1 and x2
    assert torch.equal(a["flat"], want) and torch.equal(b["flat"], want)


def _plan_bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from ayolov2_amd import YOLOModel
    from ayolov2_amd.plan import TrainPlan
    from ayolov2_amd.trainer import _FlatSync
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    model = YOLOModel(os.path.join(ROOT, "ayolov2_amd", "configs", "yolov5s.yaml")).train()
    # the REAL plan of the benchmark model (op lists and arenas are built without launching anything: the CPU will do)
    plan = TrainPlan(model, (2, 3, 64, 64), torch.float16, torch.device("cpu"))
    total = plan.gradarena.total
    buckets = plan.buckets
    # structure: the buckets tile the arena from the top down, ready indices never decrease, and a bucket is only handed
    # over after the last backward op that writes into its range has been enqueued
    assert buckets[0][2] == total and buckets[-1][1] == 0
    for (r0, lo0, hi0), (r1, lo1, hi1) in zip(buckets, buckets[1:]):
        assert lo0 == hi1 and r1 >= r0
    for ready, lo, hi in buckets:
        last_writer = max(idx for idx, off, n in plan.grad_done if off < hi and off + n > lo)
        assert ready >= last_writer, (ready, last_writer, lo, hi)
    assert (buckets[-1][2] - buckets[-1][1]) * 4 < 256 << 10, "the bucket that can only leave after backward must be small"
    g = torch.Generator().manual_seed(1000 + rank)
    arena = torch.randn(total, generator=g) * (rank + 1)                   # what this rank's backward would leave in the arena
    sync = _FlatSync(None, world)
    flat = arena.clone()
    sync.reduce_flat(flat)                                                 # reference: ONE averaged all-reduce
    bucketed = arena.clone()
    for ready, lo, hi in buckets:                                          # the overlapped exchange, in hand-over order
        sync.launch_bucket(bucketed[lo:hi])
    sync.wait_all()
    assert torch.equal(bucketed, flat)
    # 16-bit compression of the exchange: the mean of the ROUNDED per-rank buckets, written back into the fp32 arena
    for kind, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        csync = _FlatSync(None, world, compress=kind)
        small = (arena * 1e-2).clone()                                     # inside fp16's range
        comp = small.clone()
        for ready, lo, hi in buckets:
            csync.launch_bucket(comp[lo:hi])
        csync.wait_all()
        gathered = [torch.empty_like(small) for _ in range(world)]
        dist.all_gather(gathered, small)
        want = sum(t.to(dt) for t in gathered)                             # gloo: sum in the 16-bit type, then scale
        want = (want * (1.0 / world)).float()
        assert torch.allclose(comp, want, rtol=2 ** -7, atol=1e-6), kind
        exact = sum(gathered) / world
        assert float((comp - exact).abs().max()) <= 2 ** (-7 if kind == "bf16" else -10) * float(exact.abs().max()) + 1e-6
    torch.save({"flat": flat[:64].clone(), "n": len(buckets)}, os.path.join(out, f"pb{rank}.pt"))
    dist.destroy_process_group()


def test_real_plan_buckets_two_ranks_gloo(tmp_path):
    """The bucket table of a compiled YOLOv5s TrainPlan (plan.buckets: ranges + hand-over points of the real backward op
    list) driven through _FlatSync on two gloo ranks: bucket by bucket == one flat all-reduce, with and without 16-bit
    compression of the exchange."""
    port = _free_port()
    mp.spawn(_plan_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "pb0.pt"))
    b = torch.load(os.path.join(tmp_path, "pb1.pt"))
    assert torch.equal(a["flat"], b["flat"]) and a["n"] == b["n"] >= 5
