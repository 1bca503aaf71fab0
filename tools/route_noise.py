#!/usr/bin/env python3
"""How far apart are the plan executor and the per-module path in an fp16 train step, seed by seed?  (tests/test_gpu_infer.py
::test_fp16_train_step_is_reproducible_and_routes_agree holds ONE seed to a threshold: this prints the band the number lives in.)
usage (GPU box): [AYOLO_LIB=...] python tools/route_noise.py [seeds...]"""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_infer as T  # noqa: E402

seeds = [int(a) for a in sys.argv[1:]] or [37, 38, 39, 40, 41, 42]
for seed in seeds:
    m, _ = T._pair("s", seed=seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.fill_(0.3)
    m.hyp, m.gr, m.nc = dict(T.HYP), 1.0, 80
    m = m.cuda().train()
    x, t = torch.rand(4, 3, 320, 320).cuda(), T._targets(4, seed + 1).cuda()
    sd = copy.deepcopy(m.state_dict())
    out = []
    for use_plan in (True, False, True):
        m.load_state_dict(sd)
        m.__dict__.pop("_plans", None)
        m.use_plan = use_plan
        loss, _, g = T._train_step(m, x, t, amp=True)
        out.append((loss, T._flat(g)))
    m.use_plan = True
    print("seed %d: plan vs module cos %.6f (loss %.3e rel)   plan twice cos %.9f" % (
        seed, T._cos(out[0][1], out[1][1]), abs(out[0][0] - out[1][0]) / abs(out[0][0]), T._cos(out[0][1], out[2][1])))
