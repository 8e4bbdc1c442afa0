"""ctypes binding of ``csrc/libsapcu_hip.so`` (the C ABI declared in ``include/sapcu.h``).

There is no CPU fallback: if the shared library is missing or fails to load, every product
entry point raises ``SapcuLibraryError`` — build it with ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C <package>/csrc``.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAPCU_LIB_PATH: another build of the same library (e.g. csrc/libsapcu_hip_exact.so, `make -C csrc exact`: every neuron update in
# the reference's operation order); read once, at import
LIB_PATH = os.environ.get("SAPCU_LIB_PATH") or os.path.join(_HERE, "csrc", "libsapcu_hip.so")
EXACT_LIB_PATH = os.path.join(_HERE, "csrc", "libsapcu_hip_exact.so")

FN_TAPS = ("stem", "block1", "block2", "block3", "pooled", "enc", "logits")
FD_TAPS = ("fused0", "spikes", "knn", "pooled", "enc", "x0")
KIND_FN, KIND_FD = 0, 1
ABI_VERSION = 2        # include/sapcu.h SAPCU_ABI_VERSION this binding was written against (2: six fd taps)


class SapcuLibraryError(RuntimeError):
    """The HIP library is absent or unloadable (never silently replaced by a CPU path)."""


class SapcuError(RuntimeError):
    """A C entry point returned a non-zero status: args = (code, text)."""


_SIGNATURES = {
    "sapcu_abi_version": (c_int, []),
    "sapcu_last_error": (c_char_p, []),
    "sapcu_knn_gather_f64": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sapcu_gather_rotate_f64": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sapcu_displace_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "sapcu_fps_workspace_bytes": (c_int64, [c_int64]),
    "sapcu_fps_f32": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "sapcu_dense_seeds_host": (c_int, [c_void_p, c_int64, c_double, c_void_p, c_int64, POINTER(c_int64)]),
    "sapcu_lif_train_forward": (c_int, [c_void_p, c_int64, c_int, c_int] + [c_void_p] * 4 + [c_void_p, c_void_p]),
    "sapcu_lif_train_workspace_bytes": (c_int64, [c_int64, c_int]),
    "sapcu_lif_train_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int] + [c_void_p] * 4 + [c_void_p] * 5 +
                                 [c_void_p, c_int64, c_void_p]),
    "sapcu_train_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "sapcu_bn_train_forward": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float] + [c_void_p] * 4 + [c_void_p, c_int64, c_void_p]),
    "sapcu_bn_train_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int] + [c_void_p] * 3 + [c_void_p] * 3 + [c_void_p, c_int64, c_void_p]),
    "sapcu_conv1x1_wgrad_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sapcu_gemm_bf16": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "sapcu_wgrad_bf16_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "sapcu_conv1x1_wgrad_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sapcu_softmax_agg_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "sapcu_softmax_agg_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float,
                                          c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sapcu_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "sapcu_scatter_add_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "sapcu_scatter_add_rows_grouped": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sapcu_group_max_forward": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sapcu_group_max_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sapcu_neuron_selfloop": (c_int, [c_void_p, c_int64, c_int, c_int] + [c_void_p] * 6 + [c_void_p] * 4 + [c_void_p]),
    "sapcu_neuron_drive": (c_int, [c_void_p, c_int64, c_int, c_int] + [c_void_p] * 6 + [c_int] + [c_void_p] * 5 + [c_void_p]),
    "sapcu_patch_knn": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sapcu_model_create": (c_int, [c_int, POINTER(c_int32), c_int, c_void_p, c_int64, POINTER(c_int64), c_int,
                                   POINTER(c_void_p)]),
    "sapcu_model_destroy": (c_int, [c_void_p]),
    "sapcu_workspace_bytes": (c_int64, [c_void_p, c_int64, c_int]),
    "sapcu_model_gate_violations": (c_int, [c_void_p, POINTER(c_int)]),
    "sapcu_l2_normalize3": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sapcu_fn_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 POINTER(c_void_p), c_void_p]),
    "sapcu_fd_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                 POINTER(c_void_p), c_void_p]),
    "sapcu_gemm_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_void_p, c_int, c_int, c_void_p]),
    "sapcu_to_split_rows": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sapcu_posenc_gemm_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sapcu_model_gemm_mode": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "sapcu_model_fused_blocks": (c_int, [c_void_p, c_int, POINTER(c_int)]),
    "sapcu_fn_edge_chain_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "sapcu_fn_edge_chain_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int] + [c_void_p] * 12 + [c_int, c_int, c_void_p, c_void_p,
                                        c_int64, c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


def load(path=None):
    """Load (once) and return the ctypes library; raises SapcuLibraryError when unavailable."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise SapcuLibraryError(
            "HIP library not built: %s is missing. Run `make -C %s` (needs hipcc, gfx950). "
            "There is no CPU fallback." % (p, os.path.dirname(p)))
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:  # missing ROCm runtime etc.
        raise SapcuLibraryError("cannot load %s: %s" % (p, e)) from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SapcuLibraryError("%s does not export %s" % (p, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.sapcu_abi_version() != ABI_VERSION:
        raise SapcuLibraryError("ABI version mismatch: library %d, binding %d" % (lib.sapcu_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def check(rc):
    if rc != 0:
        text = load().sapcu_last_error()
        raise SapcuError(int(rc), text.decode("utf-8", "replace") if text else "")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
