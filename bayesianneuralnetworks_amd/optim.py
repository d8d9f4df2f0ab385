"""Optimizer step of the reference's training loop on the device (examples/MNIST/train.py:41,65):
torch.optim.Adam semantics (amsgrad = False, maximize = False, L2 weight_decay), every parameter
tensor of the model updated by ONE HIP launch (csrc/bnn_train.hip) -- graph-capturable, the step
counter lives on the device.

Deviations from torch.optim.Adam (pinned by tests/test_hip_parity.py::test_fused_adam_matches_torch_adam at
rtol 1e-5 over the first steps, where they are largest): ONE step counter per parameter group (torch: one per
parameter -- the same value whenever every parameter of the group has a gradient, as in the reference's loop)
and the bias corrections 1 - beta^t evaluated in fp32 on the device (torch: Python doubles), about 6e-5 relative
on the very first updates.  The state_dict therefore holds `step` in the group, not per parameter: it is not
interchangeable with torch.optim.Adam's."""
import ctypes

import torch

from . import _lib
from ._lib import AdamTensor, BnnHipError, check, ptr, stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, advance=None):
        """advance (optional, a device epoch cell of _rng.EpsGenerator.epoch_dev): bumped by one in the step's last launch
        -- the fresh-noise step of a captured training step without a launch of its own (last parameter group only)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        groups = [g for g in self.param_groups if any(p.grad is not None for p in g["params"])]
        if advance is not None and not groups:
            check(lib.bnn_rng_advance(ptr(advance), 1, stream_ptr(advance.device)), "bnn_rng_advance")
        for gi, group in enumerate(groups):
            ps = [p for p in group["params"] if p.grad is not None]
            dev = ps[0].device
            arr = (AdamTensor * len(ps))()
            for i, p in enumerate(ps):
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                    raise BnnHipError("optim.Adam: parameters must be contiguous fp32 tensors on one GPU")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise BnnHipError("optim.Adam: gradients must be contiguous fp32")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                arr[i].p, arr[i].g = p.data_ptr(), g.data_ptr()
                arr[i].m, arr[i].v, arr[i].n = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            if "step" not in group:
                group["step"] = torch.zeros(1, dtype=torch.float32, device=dev)
            adv = advance if (advance is not None and gi == len(groups) - 1) else None
            check(lib.bnn_adam_step_advance(arr, len(ps), float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                            float(group["eps"]), float(group["weight_decay"]), ptr(group["step"]),
                                            ptr(adv) if adv is not None else None, 1, stream_ptr(dev)), "bnn_adam_step")
        return loss
