"""Compute-mode switch of the contraction kernels.

'f32'  : fp32 operands and accumulation -- the parity path (1e-5 vs the reference); wide forward layers run as three-way
        bf16 splits on the bf16 MFMA (fp32-accurate, csrc/bnn_linear.hip kComputeBf16x3), the rest on v_mfma_f32_16x16x4_f32;
'bf16' : operands rounded to bf16, fp32 accumulate (v_mfma_f32_16x16x32_bf16) -- the
         throughput configuration BASELINE.json names (tolerance stated in tests/DESIGN.md).
"""
_default = "f32"


def set_compute(mode):
    global _default
    if mode not in ("f32", "bf16"):
        raise ValueError("compute mode must be 'f32' or 'bf16'")
    _default = mode


def get_compute():
    return _default


def fuse_kl_gradient(enable=True):
    """Opt-in for training loops of the form `(likelihood + KLDivergence(...)(model)).backward()` (the reference's
    train.py:57-64): KLDivergence's backward then leaves its gradient to the layers' weight-gradient launches,
    which add the closed form in their final store (ops._tls.kl_pending) -- no separate KL pass, no accumulation add
    per parameter.  Leave it off when the KL gradient is taken with torch.autograd.grad(): the parked gradient
    reaches `.grad`, not the returned tuple."""
    from .. import ops
    ops.FUSE_KL_GRADIENT = bool(enable)


def fuse_activations(module, bf16_activations=False, fuse_head=False):
    """Opt-in graph rewrite inside every torch.nn.Sequential:

    * a `NormalLinear` directly followed by `torch.nn.ReLU` gets the ReLU folded into its kernel
      epilogue (layer.activation = 'relu'); the ReLU module becomes `torch.nn.Identity`.
      Numerics unchanged (max(y, 0) of the same fp32 accumulator); one launch and one round trip
      of the activation tensor fewer per layer.
    * bf16_activations=True additionally lets a fused NormalLinear whose output feeds the next
      NormalLinear (through the Identity) emit that hidden activation in bf16 -- used only while
      the compute mode is 'bf16', where the consumer would round it to bf16 anyway, so results
      are identical to fp32 hidden activations; it halves the consumer's activation stream.  In the
      'f32' mode the same pairs (wide consumer, inference) pass the activation as the three bf16
      planes of its fp32 value (ops.X3Activation, exact to 2^-24), the operand format of the dense
      kernel's parity mode.
    * fuse_head=True (with bf16_activations): a hidden layer whose consumer is the classifier head (<= 16 outputs) at the END
      of the Sequential additionally learns that (`_fuse_head`): `BayesianNetworkModule.predictive_mean` then runs the pair as
      ONE launch (bnn_dense_forward_head) and the head's logits exist only as partial sums that the MC reduction adds up.
      Opt-in because it is only right when `_forward` returns that Sequential's output AS IT IS (anything applied to the
      logits afterwards -- a softmax -- would meet partial sums, not a tensor).

    Returns the number of fused pairs."""
    import torch
    from .dense import NormalLinear
    fused = 0
    for m in module.modules():
        if isinstance(m, torch.nn.Sequential):
            names = list(m._modules.keys())
            for a, b in zip(names[:-1], names[1:]):
                la, lb = m._modules[a], m._modules[b]
                if type(la) is NormalLinear and type(lb) is torch.nn.ReLU and la.activation is None:
                    la.activation = 'relu'
                    m._modules[b] = torch.nn.Identity()
                    fused += 1
            if bf16_activations:
                mods = [m._modules[n] for n in names]
                for i, la in enumerate(mods):
                    if type(la) is NormalLinear and la.activation == 'relu':
                        j = i + 1
                        while j < len(mods) and type(mods[j]) is torch.nn.Identity:
                            j += 1
                        if j < len(mods) and type(mods[j]) is NormalLinear and mods[j].in_channels % 8 == 0:
                            # a narrow consumer that ENDS the Sequential: under predictive_mean (bf16 mode, inference) the pair
                            # runs as one launch (ops._dense_head_raw) -- the layer learns who its head is
                            if fuse_head and mods[j].weight.mean.shape[0] <= 16 and j == len(mods) - 1:
                                la.__dict__["_fuse_head"] = mods[j]      # (not a submodule registration)
                            la.out_dtype = torch.bfloat16
                            # ... and in the fp32 parity mode, when the consumer runs on the dense kernels too, as the three
                            # bf16 planes they read (ops.X3Activation): same values, no split pass
                            la.out_x3 = mods[j].weight.mean.shape[0] > 16 or mods[j].in_channels <= 2048
    return fused
