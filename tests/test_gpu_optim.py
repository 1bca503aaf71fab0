"""ayolov2_amd.optim.SGD (one HIP launch per step) vs torch.optim.SGD on the CPU: the optimiser the reference builds at
scripts/train/yolo_trainer.py:149-168 (BN group, weight-decay group, bias group; nesterov) and steps through GradScaler
(yolo_trainer.py:332-338).  Tolerance: 1 ulp-level (rtol 2e-6) -- torch's own CPU / CUDA kernels differ among
themselves in whether `g + alpha*p` is contracted to an fma."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, device):
    g = torch.Generator().manual_seed(seed)
    shapes = [(32,), (64, 32, 3, 3), (255,), (16, 8, 1, 1), (7,), (128, 64, 1, 1), (3, 5)]
    ps = []
    for i, sh in enumerate(shapes):
        t = torch.randn(sh, generator=g)
        if len(sh) == 4 and i % 2 == 1:
            t = t.contiguous(memory_format=torch.channels_last)       # conv weights live in KRSC memory
        ps.append(torch.nn.Parameter(t.to(device)))
    return ps


def _groups(ps, lr, wd):
    return [{"params": [ps[0], ps[4]]}, {"params": [ps[1], ps[3], ps[5], ps[6]], "weight_decay": wd}, {"params": [ps[2]], "lr": lr * 2}]


@pytest.mark.parametrize("nesterov,momentum,dampening,wd", [(True, 0.937, 0.0, 5e-4), (False, 0.9, 0.1, 0.0), (False, 0.0, 0.0, 1e-3)])
def test_sgd_matches_torch(nesterov, momentum, dampening, wd):
    from ayolov2_amd.optim import SGD
    pc, pg = _params(0, "cpu"), _params(0, "cuda")
    oc = torch.optim.SGD(_groups(pc, 0.01, wd), lr=0.01, momentum=momentum, dampening=dampening, nesterov=nesterov)
    og = SGD(_groups(pg, 0.01, wd), lr=0.01, momentum=momentum, dampening=dampening, nesterov=nesterov)
    gen = torch.Generator().manual_seed(1)
    for step in range(4):
        for a, b in zip(pc, pg):
            gr = torch.randn(a.shape, generator=gen)
            a.grad = gr.clone()
            # a gradient in a different memory order than the parameter must still be applied element by element
            b.grad = gr.to("cuda") if step % 2 == 0 else gr.to("cuda").contiguous()
        if step == 2:                        # LR schedulers rewrite param_groups[i]["lr"] every step
            for o in (oc, og):
                for g in o.param_groups:
                    g["lr"] *= 0.5
        oc.step()
        og.step()
        for a, b in zip(pc, pg):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-6)
    if momentum:
        for a, b in zip(pc, pg):
            np.testing.assert_allclose(og.state[b]["momentum_buffer"].cpu().numpy(), oc.state[a]["momentum_buffer"].numpy(),
                                       rtol=2e-6, atol=1e-6)
    # the state dict has torch.optim.SGD's layout (checkpoints of one load into the other)
    oc2 = torch.optim.SGD(_groups(_params(0, "cuda"), 0.01, wd), lr=0.01, momentum=momentum, dampening=dampening, nesterov=nesterov)
    oc2.load_state_dict(og.state_dict())


def test_sgd_with_gradscaler_skips_and_unscales():
    """GradScaler hands grad_scale / found_inf over as device tensors: gradients are divided on the device, an inf step
    is skipped without a host sync -- including a skipped FIRST step, after which the momentum buffer must still start
    from the first applied gradient (torch: momentum_buffer stays None)."""
    from ayolov2_amd.optim import SGD
    pc, pg = _params(3, "cpu"), _params(3, "cuda")
    oc = torch.optim.SGD(_groups(pc, 0.01, 5e-4), lr=0.01, momentum=0.937, nesterov=True)
    og = SGD(_groups(pg, 0.01, 5e-4), lr=0.01, momentum=0.937, nesterov=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=1000)
    scaler.scale(torch.zeros(1, device="cuda"))          # creates the device-side scale tensor
    gen = torch.Generator().manual_seed(2)
    for step in range(4):
        bad = step in (0, 2)
        scale = float(scaler.get_scale())
        for a, b in zip(pc, pg):
            gr = torch.randn(a.shape, generator=gen)
            a.grad = gr.clone()
            gs = gr * scale
            if bad and a.dim() == 1:
                gs[0] = float("inf")
            b.grad = gs.to("cuda")
        before = [b.detach().clone() for b in pg]
        scaler.step(og)                      # _step_supports_amp_scaling: inf check, then our kernel with grad_scale / found_inf
        scaler.update()
        if bad:
            for b0, b in zip(before, pg):
                assert torch.equal(b0, b.detach()), "an inf step must leave the parameters untouched"
            assert float(scaler.get_scale()) == scale * 0.5
        else:
            oc.step()
            for a, b in zip(pc, pg):
                np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-6)
