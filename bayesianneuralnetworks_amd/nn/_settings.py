"""Compute-mode switch of the contraction kernels.

'f32'  : v_mfma_f32_16x16x4_f32, exact fp32 products -- the parity path (1e-5 vs the reference);
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
