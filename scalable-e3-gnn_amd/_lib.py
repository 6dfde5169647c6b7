"""ctypes binding of ``libe3gnn_hip.so`` (C ABI declared in ``include/e3gnn.h``).

There is no fallback: if the shared library is missing or a call fails, a ``RuntimeError`` is raised.
``torch`` is imported first so that the HIP runtime already mapped by PyTorch-ROCm
(``libamdhip64.so``, same SONAME) is the one the library binds to — device pointers and streams then
belong to a single runtime.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libe3gnn_hip.so")

E3_OK = 0
E3_F32, E3_F64, E3_BF16 = 0, 1, 2
CLS_NAMES = ("l0e", "l0o", "l1e", "l1o")

_lib = None

VoidP4 = c_void_p * 4

# name -> (restype, argtypes); mirrors include/e3gnn.h one to one
SIGNATURES = {
    "e3_abi_version": (c_int, []),
    "e3_status_string": (c_char_p, [c_int]),
    "e3_last_hip_error": (c_char_p, []),
    "e3_l1tp_plan_create": (c_int, [POINTER(c_int32), c_int, POINTER(c_int32), c_int, POINTER(c_void_p)]),
    "e3_l1tp_plan_destroy": (c_int, [c_void_p]),
    "e3_l1tp_in1_dim": (c_int, [c_void_p]),
    "e3_l1tp_out_dim": (c_int, [c_void_p]),
    "e3_l1tp_weight_shape": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]),
    "e3_l1tp_norm_len": (c_int, [c_void_p, c_int]),
    "e3_l1tp_packed_bytes": (c_int64, [c_void_p, c_int]),
    "e3_l1tp_pack_weights": (c_int, [c_void_p, VoidP4, VoidP4, c_int, c_void_p, c_void_p]),
    "e3_l1tp_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                c_int64, c_int, c_int, c_void_p]),
    "e3_l1tp_backward_workspace_bytes": (c_int64, [c_void_p, c_int64, c_int]),
    "e3_rg_grid": (c_int, [c_void_p]),
    "e3_rg_workspace_bytes": (c_int64, [c_int64, c_void_p]),
    "e3_rg_sort_count": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_void_p]),
    "e3_rg_fill": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "e3_edge_geometry": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e3_gather_concat": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                 c_int64, c_void_p]),
    "e3_gate": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "e3_segment_sum": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "e3_split_edges_workspace_bytes": (c_int64, [c_int64]),
    "e3_split_edges": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e3_tp_plan_create": (c_int, [POINTER(c_int32), c_int, c_int, POINTER(c_int32), c_int, POINTER(c_void_p)]),
    "e3_tp_plan_destroy": (c_int, [c_void_p]),
    "e3_tp_in1_dim": (c_int, [c_void_p]),
    "e3_tp_in2_dim": (c_int, [c_void_p]),
    "e3_tp_out_dim": (c_int, [c_void_p]),
    "e3_tp_weight_shape": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]),
    "e3_tp_norm_len": (c_int, [c_void_p, c_int]),
    "e3_tp_packed_bytes": (c_int64, [c_void_p, c_int]),
    "e3_tp_pack_weights": (c_int, [c_void_p, c_void_p * 6, c_void_p * 6, c_int, c_void_p, c_void_p]),
    "e3_tp_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                              c_int, c_void_p, c_int, c_void_p]),
    "e3_tp_forward_fused": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                    c_int, c_int, c_void_p, c_void_p]),
    "e3_msg_plan_create": (c_int, [c_int, c_int, POINTER(c_void_p)]),
    "e3_msg_plan_destroy": (c_int, [c_void_p]),
    "e3_msg_packed_bytes": (c_int64, [c_void_p]),
    "e3_msg_premix_floats_per_node": (c_int64, [c_void_p]),
    "e3_msg_weight_shape": (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "e3_msg_supports": (c_int, [c_void_p, c_int]),
    "e3_msg_pack_weights": (c_int, [c_void_p, c_void_p * 3, c_void_p * 3, c_void_p * 3, c_void_p * 3, c_int, c_void_p,
                                    c_void_p]),
    "e3_msg_premix": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "e3_msg_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "e3_edge_geometry_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "e3_gather_concat_backward": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                          c_void_p, c_void_p]),
    "e3_gate_blocks_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                        POINTER(c_int32), POINTER(c_int32), c_void_p]),
    "e3_segment_sum_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "e3_pow2_scale": (c_int, [c_void_p, POINTER(c_int64), c_int, c_int, c_void_p, c_void_p]),
    "e3_add_pow2_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "e3_tp_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                               c_int64, c_void_p, c_int64, c_void_p * 6, c_int64, c_int, c_void_p]),
    "e3_tp_backward_weights": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                       c_int64, c_int, c_void_p]),
    "e3_tp_backward_operands": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_void_p, c_int64, c_int, c_void_p]),
    "e3_tp_backward_contract": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                        c_int64, c_int64, c_int, c_void_p]),
    "e3_tp_forward_fused_epilogue": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                             c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "e3_tp_fused_supported": (c_int, [c_void_p, c_int]),
    "e3_tp_last_fused_kernel": (ctypes.c_char_p, []),
    "e3_tp_forward_fused_scatter": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "e3_segment_sum_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "e3_edge_geometry_l2": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "e3_gate_blocks": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, POINTER(c_int32),
                               POINTER(c_int32), c_void_p]),
    "e3_l1tp_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, VoidP4, VoidP4, c_void_p, c_int64,
                                 c_void_p, c_int64, c_void_p, VoidP4, c_void_p, c_int64, c_int, c_void_p]),
}


def load():
    """Load (once) and return the ctypes handle.  Raises RuntimeError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C scalable-e3-gnn_amd/csrc`). "
            "There is no CPU fallback.")
    import torch  # noqa: F401  (maps PyTorch's libamdhip64 before ours is resolved)

    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue  # entry points that are declared but land later are reported by tests
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = ""):
    if status != E3_OK:
        lib = load()
        msg = lib.e3_status_string(status).decode()
        if status == 5:
            msg += " — " + lib.e3_last_hip_error().decode()
        raise RuntimeError(f"libe3gnn_hip: {what}: {msg} (status {status})")


def blocks_array(blocks):
    flat = [int(v) for b in blocks for v in b]
    return (c_int32 * len(flat))(*flat), len(blocks)


def ptr4(tensors):
    """4-entry void* array from a list of 4 optional tensors."""
    return VoidP4(*[(t.data_ptr() if (t is not None and t.numel() > 0) else None) for t in tensors])


def dtype_code(dtype) -> int:
    import torch

    try:
        return {torch.float32: E3_F32, torch.float64: E3_F64, torch.bfloat16: E3_BF16}[dtype]
    except KeyError:
        raise RuntimeError(f"libe3gnn_hip supports float32 / float64 / bfloat16, got {dtype}") from None


class DevicePlans:
    """Plan handles of one operator, one per device (a C plan keeps its tables on the device that was current at its
    first use and rejects calls from any other).  Holds only the constructor arguments as state: ``copy.deepcopy``,
    ``pickle`` and ``torch.save(module)`` rebuild an empty holder, and handles are created lazily -- two owners never
    share (and double-free) a handle."""

    def __init__(self, create_name: str, destroy_name: str, *args):
        self._create, self._destroy_name, self._args = create_name, destroy_name, args
        self._handles = {}

    def _make(self):
        lib = load()
        cargs = []
        for a in self._args:
            if isinstance(a, (list, tuple)):
                arr, n = blocks_array(a)
                cargs += [arr, n]
            else:
                cargs.append(int(a))
        h = c_void_p()
        check(getattr(lib, self._create)(*cargs, ctypes.byref(h)), self._create)
        return h

    def handle(self, device=None) -> c_void_p:
        """The handle for ``device`` (a torch.device / index; None = a host-only handle for shape queries)."""
        key = -1
        if device is not None:
            import torch
            key = torch.device(device).index
            if key is None:
                key = torch.cuda.current_device()
        h = self._handles.get(key)
        if h is None:
            h = self._handles[key] = self._make()
        return h

    def __deepcopy__(self, memo):
        return DevicePlans(self._create, self._destroy_name, *self._args)

    def __reduce__(self):
        return (DevicePlans, (self._create, self._destroy_name) + tuple(self._args))

    def __del__(self):
        try:
            lib = load()
            for h in self._handles.values():
                if h:
                    getattr(lib, self._destroy_name)(h)
            self._handles = {}
        except Exception:
            pass
