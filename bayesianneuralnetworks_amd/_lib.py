"""ctypes binding of libbnn_hip.so (C-ABI in include/bnn_hip.h).

The library is the product's only compute path for CUDA (HIP) tensors.  There is
no fallback: if it is missing or a call fails, `BnnHipError` is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BNN_HIP_LIB: another build of the same C-ABI (A/B measurements); default = the in-tree library
LIB_PATH = os.environ.get("BNN_HIP_LIB") or os.path.join(_HERE, "libbnn_hip.so")

F32, BF16, BF16X3 = 0, 1, 2
COMPUTE_F32, COMPUTE_BF16 = 0, 1
FLAG_RELU = 1
FLAG_X_BF16 = 2
E_ALIGN, E_UNSUPPORTED, E_DEVICE = -4, -6, -7     # BNN_E_ALIGN, BNN_E_UNSUPPORTED, BNN_E_DEVICE (include/bnn_hip.h)
FLAG_Y_BF16 = 4


class BnnHipError(RuntimeError):
    pass


ABI_VERSION = 2            # include/bnn_hip.h BNN_ABI_VERSION


class Rng(ctypes.Structure):
    """bnn_rng_t"""
    _fields_ = [("seed", ctypes.c_uint64),
                ("stream", ctypes.c_uint32),
                ("sample0", ctypes.c_uint32),
                ("epoch_host", ctypes.c_uint32),
                ("epoch_dev_delta", ctypes.c_int32),
                ("epoch_dev", ctypes.c_void_p),
                ("generator", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


class KlTensor(ctypes.Structure):
    """bnn_kl_tensor_t"""
    _fields_ = [("mu", ctypes.c_void_p),
                ("rho", ctypes.c_void_p),
                ("n", ctypes.c_int64),
                ("prior_mu", ctypes.c_float),
                ("prior_sigma", ctypes.c_float)]


class AdamTensor(ctypes.Structure):
    """bnn_adam_tensor_t"""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_int64)]


class KlFuse(ctypes.Structure):
    """bnn_kl_fuse_t"""
    _fields_ = [("upstream", ctypes.c_void_p), ("mu_w", ctypes.c_void_p), ("mu_b", ctypes.c_void_p),
                ("scale_w", ctypes.c_float), ("prior_mu_w", ctypes.c_float), ("prior_sigma_w", ctypes.c_float),
                ("scale_b", ctypes.c_float), ("prior_mu_b", ctypes.c_float), ("prior_sigma_b", ctypes.c_float)]


class DrawTensor(ctypes.Structure):
    """bnn_draw_tensor_t"""
    _fields_ = [("mu", ctypes.c_void_p), ("rho", ctypes.c_void_p), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64),
                ("out", ctypes.c_void_p), ("ld", ctypes.c_int64), ("out_sample_stride", ctypes.c_int64),
                ("out_dtype", ctypes.c_int), ("kind", ctypes.c_int), ("taps", ctypes.c_int), ("rng", Rng)]


class Conv2dShape(ctypes.Structure):
    """bnn_conv2d_shape_t"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "C", "H", "W", "O", "KH", "KW", "stride_h", "stride_w", "pad_h", "pad_w",
                 "dil_h", "dil_w", "groups")]


_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f = ctypes.c_float
_rngp = ctypes.POINTER(Rng)

# name -> (restype, argtypes); every symbol include/bnn_hip.h declares.
SIGNATURES = {
    "bnn_abi_version": (_int, []),
    "bnn_arch": (ctypes.c_char_p, []),
    "bnn_last_error": (ctypes.c_char_p, []),
    "bnn_launch_count": (ctypes.c_uint64, []),
    "bnn_set_workspace": (_int, [_int, _p, _i64]),
    "bnn_check_device": (_int, [_int, _p]),
    "bnn_sample_affine_eps": (_int, [_p, _p, _p, _p, _i64, _int, _p]),
    "bnn_sample_affine_philox": (_int, [_p, _p, _p, _i64, _int, _i64, _int, _rngp, _p]),
    "bnn_eps_philox": (_int, [_p, _i64, _int, _i64, _rngp, _p]),
    "bnn_sigma": (_int, [_p, _p, _i64, _p]),
    "bnn_sample_affine_bwd": (_int, [_p, _i64, _p, _p, _i64, _rngp, _i64, _int, _p, _p, _int, _p]),
    "bnn_rng_advance": (_int, [_p, ctypes.c_uint32, _p]),
    "bnn_prune_score": (_int, [_p, _p, _p, _i64, _p]),
    "bnn_kl_workspace_bytes": (_i64, [_int]),
    "bnn_kl_forward": (_int, [ctypes.POINTER(KlTensor), _int, _f, _p, _p, _p]),
    "bnn_kl_forward_partial": (_int, [ctypes.POINTER(KlTensor), _int, _p, _p]),
    "bnn_kl_backward": (_int, [ctypes.POINTER(KlTensor), _int, _f, _p, ctypes.POINTER(_p),
                               ctypes.POINTER(_p), _int, _p]),
    "bnn_linear_forward_sampled": (_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64,
                                          _i64, _int, _rngp, _rngp, _int, _int, _p]),
    "bnn_linear_forward_sampled_kl": (_int, [_p, _i64, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64,
                                             _i64, _int, _rngp, _rngp, _int, _int, ctypes.POINTER(KlTensor), _int, _p, _p]),
    "bnn_draw_multi": (_int, [ctypes.POINTER(DrawTensor), _int, _int, ctypes.POINTER(KlTensor), _int, _p, _p]),
    "bnn_dense_forward": (_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _int, _int, _p]),
    "bnn_dense_head_parts": (_int, [_i64, _i64, _int]),
    "bnn_dense_forward_head": (_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _i64,
                                      _p, _i64, _i64, _i64, _int, _int, _p]),
    "bnn_dense_forward_x3_head": (_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64,
                                         _p, _i64, _i64, _p, _i64, _i64, _i64, _int, _int, _p]),
    "bnn_dense_forward_x3": (_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64,
                                    _i64, _i64, _i64, _int, _int, _p]),
    "bnn_split_bf16x3": (_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _p]),
    "bnn_transpose_bf16": (_int, [_p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "bnn_conv2d_dense_forward_x3": (_int, [_p, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, ctypes.POINTER(Conv2dShape), _int, _int, _p]),
    "bnn_conv2d_dense_forward": (_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, ctypes.POINTER(Conv2dShape), _int, _int, _p]),
    "bnn_conv2d_flipout_forward": (_int, [_p, _p, _i64, _p, _p, _p, ctypes.POINTER(Conv2dShape), _int, _p]),
    "bnn_conv2d_flipout_forward_x3": (_int, [_p, _p, _i64, _i64, _p, _p, _p, ctypes.POINTER(Conv2dShape), _int, _p]),
    "bnn_linear_forward": (_int, [_p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64,
                                  _int, _int, _int, _p]),
    "bnn_linear_backward_input_sampled": (_int, [_p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int,
                                                 _rngp, _int, _int, _p]),
    "bnn_linear_backward_input": (_int, [_p, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _int, _int, _p]),
    "bnn_linear_backward_weight_sampled": (_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64,
                                                  _i64, _int, _rngp, _rngp, ctypes.POINTER(KlFuse), _int, _int, _int, _p]),
    "bnn_linear_backward_narrow_sampled": (_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p,
                                                  _i64, _i64, _i64, _int, _rngp, _rngp, ctypes.POINTER(KlFuse), _int, _int, _p]),
    "bnn_linear_backward_weight": (_int, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _int, _int,
                                          _int, _int, _p]),
    "bnn_colsum": (_int, [_p, _i64, _i64, _p, _i64, _i64, _int, _int, _p]),
    "bnn_relu_backward": (_int, [_p, _p, _p, _i64, _int, _p]),
    "bnn_adam_step": (_int, [ctypes.POINTER(AdamTensor), _int, _f, _f, _f, _f, _f, _p, _p]),
    "bnn_adam_step_advance": (_int, [ctypes.POINTER(AdamTensor), _int, _f, _f, _f, _f, _f, _p, _p, ctypes.c_uint32, _p]),
    "bnn_xent_workspace_bytes": (_i64, [_i64]),
    "bnn_softmax_xent": (_int, [_p, _p, _i64, _int, _p, _p, _p, _p]),
    "bnn_conv2d_im2col": (_int, [_p, _i64, ctypes.POINTER(Conv2dShape), _int, _p, _int, _p]),
    "bnn_conv2d_col2im": (_int, [_p, ctypes.POINTER(Conv2dShape), _int, _int, _p, _p]),
    "bnn_nchw_to_rows": (_int, [_p, _i64, _int, _int, _p, _int, _p]),
    "bnn_conv2d_workspace_bytes": (_i64, [ctypes.POINTER(Conv2dShape), _int, _int]),
    "bnn_conv2d_forward_sampled": (_int, [_p, _i64, _p, _p, _p, _p, _p, _i64,
                                          ctypes.POINTER(Conv2dShape), _int, _rngp, _rngp, _int, _int, _p, _i64, _p]),
    "bnn_conv2d_forward": (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, ctypes.POINTER(Conv2dShape),
                                  _int, _int, _int, _p, _i64, _p]),
    "bnn_diag_sampler": (_int, [_p, _int, _int, _int, _p]),
    "bnn_diag_astream": (_int, [_p, _int, _int, _int, _int, _int, _int, _int, _p, _p]),
    "bnn_mc_sum": (_int, [_p, _i64, _int, _i64, _f, _p, _int, _p, ctypes.c_uint32, _p]),
    "bnn_mc_sum_kl": (_int, [_p, _i64, _int, _i64, _f, _p, _int, _p, ctypes.c_uint32, ctypes.POINTER(KlTensor), _int, _f,
                             _p, _p, _p]),
}

_lib = None


def load():
    """Load libbnn_hip.so (after torch, so that its libamdhip64.so.7 dependency binds to
    the HIP runtime torch already loaded).  Raises BnnHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BnnHipError(
            "libbnn_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bayesianneuralnetworks_amd/csrc`. There is no fallback path." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise BnnHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.bnn_abi_version() != ABI_VERSION:
        raise BnnHipError("libbnn_hip.so has ABI version %d, this package binds version %d (bnn_rng_t carries the generator id "
                          "since version 2): rebuild with `make -C bayesianneuralnetworks_amd/csrc`" % (lib.bnn_abi_version(), ABI_VERSION))
    _lib = _DeviceGuarded(lib)
    return _lib


class _StreamPtr(ctypes.c_void_p):
    """hipStream_t of torch's current stream on `device_index` (what stream_ptr returns)."""
    device_index = None


class _DeviceGuarded:
    """The loaded library; a call whose stream argument belongs to another device than the current one runs with
    that device current (kernels launch on -- and use the registered workspace of -- the current device, so a model
    on cuda:1 must not launch while cuda:0 is current).  The common case (same device) costs one comparison."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            st = args[-1] if args else None
            if isinstance(st, _StreamPtr) and st.device_index is not None and st.device_index != torch.cuda.current_device():
                with torch.cuda.device(st.device_index):
                    return fn(*args)
            return fn(*args)

        call.__name__ = name
        setattr(self, name, call)
        return call


_workspaces = {}


def ensure_workspace(device):
    """Register a zeroed 4 MiB scratch tensor for `device` once (split-K tickets + slabs)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _workspaces:
        ws = torch.zeros(4 << 20, dtype=torch.uint8, device=torch.device("cuda", idx))
        torch.cuda.synchronize(idx)
        check(load().bnn_set_workspace(idx, ctypes.c_void_p(ws.data_ptr()), ws.numel()), "bnn_set_workspace")
        _workspaces[idx] = ws
    return _workspaces[idx]


def check(rc, what):
    if rc != 0:
        msg = load().bnn_last_error().decode("utf-8", "replace")
        raise BnnHipError("%s failed (code %d): %s" % (what, rc, msg))


def stream_ptr(device=None):
    """torch's current stream on `device` as the C-ABI's `stream` argument (it remembers the device: see _DeviceGuarded)."""
    sp = _StreamPtr(torch.cuda.current_stream(device).cuda_stream)
    if device is not None:
        d = torch.device(device)
        sp.device_index = d.index if d.index is not None else torch.cuda.current_device()
    return sp


def check_device(device=None):
    """Synchronise `device`'s current stream and raise BnnHipError if a kernel reported an internal error since the
    last check (the device error word, include/bnn_hip.h: bnn_check_device).  A check point, not a per-launch call."""
    d = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    check(load().bnn_check_device(idx, stream_ptr(torch.device("cuda", idx))), "bnn_check_device")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_cuda_act(t, name, contiguous=True):
    """activation tensor: CUDA fp32 or bf16, contiguous unless the caller has checked its row layout itself"""
    if not t.is_cuda:
        raise BnnHipError("%s must be a CUDA/HIP tensor" % name)
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise BnnHipError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))
    if contiguous and not t.is_contiguous():
        raise BnnHipError("%s must be contiguous" % name)


def require_cuda_f32(t, name):
    if not t.is_cuda:
        raise BnnHipError("%s must be a CUDA/HIP tensor" % name)
    if t.dtype != torch.float32:
        raise BnnHipError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise BnnHipError("%s must be contiguous" % name)
