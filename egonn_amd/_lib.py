"""ctypes binding of libegonn_hip.so (C ABI: include/egonn_hip.h).

The product path has NO fallback: if the HIP library is missing, or no MI355X is visible, every entry
point raises.  PyTorch is used only as the owner of device memory and of the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libegonn_hip.so")

QUANT_CARTESIAN, QUANT_POLAR = 0, 1
FLAG_DISABLE_GLOBAL, FLAG_DISABLE_LOCAL, FLAG_IGNORE_KP_REGRESSOR, FLAG_BF16 = 1, 2, 4, 8
FLAG_POOL_SPOC, FLAG_POOL_MAC = 16, 32

# every symbol include/egonn_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGS = [
    ("egonn_ctx_create", C.c_int, [C.POINTER(_P), C.c_int, C.c_int]),
    ("egonn_ctx_destroy", None, [_P]),
    ("egonn_last_error", C.c_char_p, []),
    ("egonn_debug_set_naive_conv", C.c_int, [_P, C.c_int]),
    ("egonn_debug_set_ksplit", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("egonn_debug_keep_level_features", C.c_int, [_P, C.c_int]),
    ("egonn_debug_set_trace", C.c_int, [_P]),
    ("egonn_prepare_maps", C.c_int, [_P, C.c_int, _P]),
    ("egonn_debug_rowgroup_tables", C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int64), _P]),
    ("egonn_voxelize", C.c_int, [_P, _P, C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    ("egonn_ctx_reserve", C.c_int, [_P, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    ("egonn_voxelize_device", C.c_int, [_P, _P, C.c_int64, _P, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    ("egonn_plan_status", C.c_int, [_P, _P]),
    ("egonn_ctx_set_exact_fp32", C.c_int, [_P, C.c_int]),
    ("egonn_ctx_set_operand_autoscale", C.c_int, [_P, C.c_int]),
    ("egonn_level_capacity", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    ("egonn_graph_begin", C.c_int, [_P]),
    ("egonn_graph_end", C.c_int, [_P, C.POINTER(_P)]),
    ("egonn_graph_launch", C.c_int, [_P, _P]),
    ("egonn_graph_destroy", None, [_P]),
    ("egonn_coords_set", C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    ("egonn_level_count", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    ("egonn_level_batch_offsets", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    ("egonn_level_coords", C.c_int, [_P, C.c_int, _P, _P]),
    ("egonn_input_index", C.c_int, [_P, _P, _P]),
    ("egonn_conv", C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, _P]),
    ("egonn_conv_transpose", C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P]),
    ("egonn_sparse_conv", C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    ("egonn_map_groups", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P]),
    ("egonn_global_avg_pool", C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P]),
    ("egonn_bn_fold", C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int, _P, _P, _P]),
    ("egonn_block_tail", C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, _P]),
    ("egonn_gem", C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P]),
    ("egonn_add", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("egonn_gather_input", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("egonn_model_create", C.c_int, [C.POINTER(_P)]),
    ("egonn_model_destroy", None, [_P]),
    ("egonn_model_set_tensor", C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64)]),
    ("egonn_model_finalize", C.c_int, [_P, _P]),
    ("egonn_forward", C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_float), C.c_int, _P, _P, _P, _P, _P]),
    ("egonn_forward_level_features", C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    ("egonn_select_keypoints", C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P]),
    ("egonn_topk_rows", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    ("egonn_triplet_loss_scratch_floats", C.c_int64, [C.c_int]),
    ("egonn_triplet_loss", C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_float, _P, _P, _P, _P, _P]),
    ("egonn_nn_search", C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P, _P, _P]),
    ("egonn_matrix_min", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    ("egonn_softmax_cross_entropy", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    ("egonn_dense", C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    ("egonn_dense_backward_weight", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int64, _P, _P, C.c_int64, _P]),
    ("egonn_conv_backward_weight", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P,
                                             C.c_int64, _P]),
    ("egonn_col_stats", C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int64, C.c_int, _P, _P, C.c_int64, _P]),
    ("egonn_bn_train_finalize", C.c_int, [_P, _P, C.c_double, C.c_int, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P]),
    ("egonn_bn_backward_finalize", C.c_int, [_P, _P, C.c_double, C.c_int, _P, _P, _P, _P, _P]),
    ("egonn_affine_act", C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P]),
    ("egonn_affine3", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int, _P, _P]),
    ("egonn_relu_backward", C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P]),
    ("egonn_eca_gate", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    ("egonn_eca_gate_backward", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    ("egonn_act_backward", C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P]),
    ("egonn_l2_normalize", C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P]),
    ("egonn_gate_residual", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    ("egonn_gate_residual_backward", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P]),
    ("egonn_segment_sums", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, C.c_int64, _P]),
    ("egonn_segment_broadcast", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    ("egonn_gem_backward", C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P]),
    ("egonn_filter_points_scratch_ints", C.c_int64, [C.c_int64]),
    ("egonn_filter_points", C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P,
                                      C.c_int64, _P]),
    ("egonn_knn", C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P, C.c_int64, _P]),
    ("egonn_recall_counts", C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    ("egonn_profile_enable", C.c_int, [_P, C.c_int, C.c_char_p]),
    ("egonn_profile_fetch", C.c_int, [_P, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_float),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGS]

_lib = None


def load() -> C.CDLL:
    """Load libegonn_hip.so; loud failure if it has not been built (python __graft_entry__.py / make)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"egonn_amd: HIP library not found at {LIB_PATH}. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            f"There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in _SIGS:
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _err(lib) -> str:
    msg = lib.egonn_last_error()
    return msg.decode() if msg else "unknown error"


class EgonnError(RuntimeError):
    """non-zero status of a libegonn_hip entry point; `.code` is the C status (include/egonn_hip.h EGONN_STATUS_*)"""

    def __init__(self, msg: str, code: int):
        super().__init__(msg)
        self.code = int(code)


class CapacityError(EgonnError):
    """status 5: the batch did not fit the capacities of egonn_ctx_reserve (the exact-size eager path still works)"""


class Fp16RangeError(EgonnError):
    """status 6: an fp32 sparse convolution on the fp16-split pipe met a non-finite accumulator (activation beyond +-65504 or
    non-finite input): the batch's outputs are invalid; Context.set_exact_fp32(True) and run it again"""


def check(rc: int):
    if rc != 0:
        cls = CapacityError if rc == 5 else (Fp16RangeError if rc == 6 else EgonnError)
        raise cls(f"libegonn_hip: {_err(load())} (code {rc})", rc)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("egonn_amd: no HIP device visible (torch.cuda.is_available() is False); "
                           "the descriptor-extraction path runs on MI355X only — there is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


class Context:
    """egonn_ctx: coordinate plan + workspace of one device."""

    def __init__(self, device: Optional[torch.device] = None, coord_bits: int = 16):
        self.lib = load()
        self.device = torch.device(device) if device is not None else require_gpu()
        if self.device.type != "cuda":
            raise RuntimeError(f"egonn_amd: device {self.device} is not a HIP device; there is no CPU fallback.")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        h = _P()
        check(self.lib.egonn_ctx_create(C.byref(h), self.device.index, coord_bits))
        self.h = h
        self.batch_size = 0
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.egonn_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ plan
    def voxelize(self, points: torch.Tensor, scan_offsets: Sequence[int], mode: int, step: Sequence[float]):
        assert points.is_cuda and points.dtype == torch.float32 and points.is_contiguous()
        assert points.dim() == 2 and points.shape[1] == 3
        B = len(scan_offsets) - 1
        off = (C.c_int64 * (B + 1))(*[int(o) for o in scan_offsets])
        st = (C.c_float * 3)(*([float(s) for s in step] + [0.0, 0.0])[:3])
        with torch.cuda.device(self.device):
            check(self.lib.egonn_voxelize(self.h, points.data_ptr(), off, B, mode, st, _stream()))
        self.batch_size = B

    def reserve(self, max_points: int, batch_size: int, level_capacity: Optional[Sequence[int]] = None):
        """egonn_ctx_reserve: fixed capacities => later voxelize_device plans neither allocate nor synchronise."""
        caps = None
        if level_capacity is not None:
            assert len(level_capacity) == 8
            caps = (C.c_int64 * 8)(*[int(v) for v in level_capacity])
        with torch.cuda.device(self.device):
            check(self.lib.egonn_ctx_reserve(self.h, int(max_points), int(batch_size), caps))
        self.batch_size = int(batch_size)

    def voxelize_device(self, points: torch.Tensor, scan_offsets: torch.Tensor, batch_size: int, mode: int,
                        step: Sequence[float]):
        """points (n_rows,3) f32 and scan_offsets (B+1,) int64 both on the device; no host synchronisation."""
        assert points.is_cuda and points.dtype == torch.float32 and points.is_contiguous() and points.shape[1] == 3
        assert scan_offsets.is_cuda and scan_offsets.dtype == torch.int64 and scan_offsets.numel() == batch_size + 1
        st = (C.c_float * 3)(*([float(s) for s in step] + [0.0, 0.0])[:3])
        with torch.cuda.device(self.device):
            check(self.lib.egonn_voxelize_device(self.h, points.data_ptr(), points.shape[0], scan_offsets.data_ptr(),
                                                 int(batch_size), mode, st, _stream()))
        self.batch_size = int(batch_size)

    def plan_status(self):
        """[SYNC] raises if the latest (replayed) plan left the coordinate range or the reserved capacities."""
        with torch.cuda.device(self.device):
            check(self.lib.egonn_plan_status(self.h, _stream()))

    def level_capacity(self, level: int) -> int:
        n = C.c_int64()
        check(self.lib.egonn_level_capacity(self.h, level, C.byref(n)))
        return n.value

    def coords_set(self, coords: torch.Tensor, batch_size: int):
        assert coords.is_cuda and coords.dtype == torch.int32 and coords.is_contiguous()
        assert coords.dim() == 2 and coords.shape[1] == 4
        with torch.cuda.device(self.device):
            check(self.lib.egonn_coords_set(self.h, coords.data_ptr(), coords.shape[0], int(batch_size), _stream()))
        self.batch_size = int(batch_size)

    def level_count(self, level: int) -> int:
        n = C.c_int64()
        check(self.lib.egonn_level_count(self.h, level, C.byref(n)))
        return n.value

    def level_batch_offsets(self, level: int) -> List[int]:
        off = (C.c_int64 * (self.batch_size + 1))()
        check(self.lib.egonn_level_batch_offsets(self.h, level, off))
        return list(off)

    def level_coords(self, level: int) -> torch.Tensor:
        out = torch.empty((self.level_count(level), 4), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_level_coords(self.h, level, out.data_ptr(), _stream()))
        return out

    def input_index(self) -> torch.Tensor:
        out = torch.empty((self.level_count(0),), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_input_index(self.h, out.data_ptr(), _stream()))
        return out

    # ------------------------------------------------------------------ operators
    def conv(self, level_in: int, level_out: int, kernel_size: int, x: torch.Tensor, kernel: torch.Tensor,
             scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, relu: bool = False):
        kernel = _dev_f32(kernel, self.device)
        cin, cout = kernel.shape[-2], kernel.shape[-1]
        if x is None:
            assert kernel_size == 5 and cin == 1, "x=None (all-ones features) is the k=5 input layer only"
        else:
            x = _dev_f32(x, self.device)
            assert x.shape == (self.level_count(level_in), cin), (x.shape, self.level_count(level_in), cin)
        out = torch.empty((self.level_count(level_out), cout), dtype=torch.float32, device=self.device)
        sc = None if scale is None else _dev_f32(scale, self.device)
        sh = None if shift is None else _dev_f32(shift, self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_conv(self.h, level_in, level_out, kernel_size, _ptr(x), cin, kernel.data_ptr(),
                                      cout, _ptr(sc), _ptr(sh), int(relu), out.data_ptr(), _stream()))
        return out

    def conv_transpose(self, level_in: int, x: torch.Tensor, kernel: torch.Tensor):
        x = _dev_f32(x, self.device)
        kernel = _dev_f32(kernel, self.device)
        cin, cout = kernel.shape[-2], kernel.shape[-1]
        assert x.shape == (self.level_count(level_in), cin)
        out = torch.empty((self.level_count(level_in - 1), cout), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_conv_transpose(self.h, level_in, x.data_ptr(), cin, kernel.data_ptr(), cout,
                                                out.data_ptr(), _stream()))
        return out

    def sparse_conv(self, map_kind: int, level_out: int, x: torch.Tensor, kernel: torch.Tensor, scale=None, shift=None,
                    relu: bool = False, group_sums: bool = False):
        """egonn_sparse_conv: x fp32 or bf16 (the output has x's dtype).  map_kind 0: k=3, 1: k=2/s=2 into level_out,
        2: transposed onto level_out.  Returns out, or (out, sums) with the per-group column sums."""
        assert x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous()
        kernel = _dev_f32(kernel, self.device)
        cin, cout = kernel.shape[-2], kernel.shape[-1]
        assert x.shape[1] == cin
        out = torch.empty((self.level_count(level_out), cout), dtype=x.dtype, device=self.device)
        sc = None if scale is None else _dev_f32(scale, self.device)
        sh = None if shift is None else _dev_f32(shift, self.device)
        sums = None
        if group_sums:
            sums = torch.zeros((self.map_groups(map_kind, level_out)[0], cout), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_sparse_conv(self.h, map_kind, level_out, x.data_ptr(), cin, kernel.data_ptr(), cout,
                                             int(x.dtype == torch.bfloat16), _ptr(sc), _ptr(sh), int(relu), out.data_ptr(),
                                             _ptr(sums), _stream()))
        return (out, sums) if group_sums else out

    def prepare_maps(self, with_level0_transpose: bool = False):
        """row-group tables of every kernel map of the plan in ONE launch (training steps, operator sequences)"""
        with torch.cuda.device(self.device):
            check(self.lib.egonn_prepare_maps(self.h, int(with_level0_transpose), _stream()))

    def map_groups(self, map_kind: int, level_out: int):
        n = C.c_int64()
        first = (C.c_int64 * (self.batch_size + 1))()
        with torch.cuda.device(self.device):
            check(self.lib.egonn_map_groups(self.h, map_kind, level_out, C.byref(n), first, _stream()))
        return n.value, list(first)

    def rowgroup_tables(self, map_kind: int, level_out: int, with_rows: bool = True):
        """measurement hook: (gmask [groups] int32 view of u32, snbr [groups, K, 16] int32) of a map's row-group tables."""
        ng, _ = self.map_groups(map_kind, level_out)
        K = 27 if map_kind == 0 else 8
        gm = torch.empty(ng, dtype=torch.int32, device=self.device)
        sn = torch.empty((ng, K, 16), dtype=torch.int32, device=self.device) if with_rows else None
        n = C.c_int64()
        with torch.cuda.device(self.device):
            check(self.lib.egonn_debug_rowgroup_tables(self.h, map_kind, level_out, gm.data_ptr(), sn.data_ptr() if with_rows else None,
                                                       ng, C.byref(n), _stream()))
        return gm, sn

    def set_naive_conv(self, on: bool):
        """tests only: route this context's sparse convolutions through the plain (non-MFMA) kernel."""
        check(self.lib.egonn_debug_set_naive_conv(self.h, int(on)))

    def keep_level_features(self, on: bool):
        """tests only: materialise every level's block output (forward_level_features(1) then works)."""
        check(self.lib.egonn_debug_keep_level_features(self.h, int(bool(on))))

    def set_exact_fp32(self, on: bool):
        """fp32 maps: True = every sparse convolution on the exact fp32 kernels (full fp32 range); False (default) = levels <= 5 on
        the fp16-split matrix pipe (|activation| < 65504, guarded: plan_status raises with code 6)."""
        check(self.lib.egonn_ctx_set_exact_fp32(self.h, int(bool(on))))

    def set_operand_autoscale(self, on: bool):
        """fp16-split convolutions scale their input by a power of two per launch (max |in| -> [2^13, 2^14)): small operands
        (input gradients) keep their low parts.  Eager plans only."""
        check(self.lib.egonn_ctx_set_operand_autoscale(self.h, int(bool(on))))

    def set_ksplit(self, map_class: int, level: int, kparts: int = -1, kw: int = -1, col_parts: int = -1):
        """tests / A-B only: offset-split rule of the fp32 sparse convolutions (map_class 0: k=3 maps, 1: 8-slot maps); -1 keeps a field."""
        check(self.lib.egonn_debug_set_ksplit(self.h, int(map_class), int(level), int(kparts), int(kw), int(col_parts)))

    def global_avg_pool(self, level: int, x: torch.Tensor):
        x = _dev_f32(x, self.device)
        out = torch.empty((self.batch_size, x.shape[1]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_global_avg_pool(self.h, level, x.data_ptr(), x.shape[1], out.data_ptr(), _stream()))
        return out

    def bn_fold(self, bn: torch.nn.BatchNorm1d):
        """(scale, shift) of an eval-mode BatchNorm1d, computed on the device by the library."""
        c = bn.num_features
        w, b, rm, rv = (_dev_f32(t.detach(), self.device) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        scale = torch.empty(c, dtype=torch.float32, device=self.device)
        shift = torch.empty(c, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_bn_fold(w.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), float(bn.eps), c,
                                         scale.data_ptr(), shift.data_ptr(), _stream()))
        return scale, shift

    def block_tail(self, level: int, x: torch.Tensor, residual: torch.Tensor, eca_weight: Optional[torch.Tensor] = None):
        x, residual = _dev_f32(x, self.device), _dev_f32(residual, self.device)
        assert x.shape == residual.shape == (self.level_count(level), x.shape[1])
        out = torch.empty_like(x)
        ew = None if eca_weight is None else _dev_f32(eca_weight.detach().reshape(-1), self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_block_tail(self.h, level, x.data_ptr(), residual.data_ptr(), x.shape[1], _ptr(ew),
                                            0 if ew is None else ew.numel(), out.data_ptr(), _stream()))
        return out

    def add(self, a: torch.Tensor, b: torch.Tensor):
        a, b = _dev_f32(a, self.device), _dev_f32(b, self.device)
        assert a.shape == b.shape
        out = torch.empty_like(a)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_add(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), _stream()))
        return out

    def gather_input(self, feats: torch.Tensor):
        feats = _dev_f32(feats, self.device)
        out = torch.empty((self.level_count(0), feats.shape[1]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_gather_input(self.h, feats.data_ptr(), feats.shape[1], out.data_ptr(), _stream()))
        return out

    def gem(self, level: int, x: torch.Tensor, p: torch.Tensor):
        x = _dev_f32(x, self.device)
        pp = _dev_f32(p.detach().reshape(-1), self.device)
        out = torch.empty((self.batch_size, x.shape[1]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_gem(self.h, level, x.data_ptr(), x.shape[1], pp.data_ptr(), out.data_ptr(), _stream()))
        return out

    # ------------------------------------------------------------------ training-mode operators (egonn_amd/train.py)
    def scratch(self, nfloats: int) -> torch.Tensor:
        """caller-owned scratch for the two-stage reductions (grown on demand, reused)."""
        buf = getattr(self, "_scratch", None)
        if buf is None or buf.numel() < nfloats:
            buf = torch.empty(max(int(nfloats), 1 << 24), dtype=torch.float32, device=self.device)
            self._scratch = buf
        return buf

    def _call(self, fn, *args):
        idx = self.device.index
        if idx is None or torch.cuda.current_device() == idx:       # common case: no device switch needed
            rc = fn(*args, torch.cuda.current_stream().cuda_stream)
        else:
            with torch.cuda.device(self.device):
                rc = fn(*args, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            check(rc)

    def dense(self, x, weight, out_in: bool, bias=None, act: int = 0):
        x, weight = _dev_f32(x, self.device), _dev_f32(weight, self.device)
        cin, cout = (weight.shape[1], weight.shape[0]) if out_in else (weight.shape[0], weight.shape[1])
        assert x.dim() == 2 and x.shape[1] == cin, (x.shape, weight.shape, out_in)
        b = None if bias is None else _dev_f32(bias, self.device)
        out = torch.empty((x.shape[0], cout), dtype=torch.float32, device=self.device)
        self._call(self.lib.egonn_dense, x.data_ptr(), x.shape[0], cin, weight.data_ptr(), int(out_in), _ptr(b), cout,
                   act, out.data_ptr())
        return out

    def dense_backward_weight(self, a, b):
        """a^T b over the rows: (ca, cb)."""
        a, b = _dev_f32(a, self.device), _dev_f32(b, self.device)
        assert a.shape[0] == b.shape[0]
        ca, cb = a.shape[1], b.shape[1]
        out = torch.empty((ca, cb), dtype=torch.float32, device=self.device)
        sc = self.scratch(64 * ca * cb)
        self._call(self.lib.egonn_dense_backward_weight, a.data_ptr(), ca, b.data_ptr(), cb, a.shape[0], out.data_ptr(),
                   sc.data_ptr(), sc.numel())
        return out

    def conv_backward_weight(self, level_in, level_out, kernel_size, transposed, x, grad_out, kernel_shape):
        g = _dev_f32(grad_out, self.device)
        xx = None if x is None else _dev_f32(x, self.device)
        cin, cout = kernel_shape[-2], kernel_shape[-1]
        out = torch.empty(tuple(kernel_shape), dtype=torch.float32, device=self.device)
        sc = self.scratch(16 * out.numel())
        self._call(self.lib.egonn_conv_backward_weight, self.h, level_in, level_out, kernel_size, int(transposed), _ptr(xx),
                   cin, g.data_ptr(), cout, out.data_ptr(), sc.data_ptr(), sc.numel())
        return out

    def col_stats(self, mode: int, a, b=None, mask=None, mean=None):
        a = _dev_f32(a, self.device)
        n, c = a.shape
        out = torch.empty((2, c), dtype=torch.float32, device=self.device)
        sc = self.scratch(2 * c * max(1024, n // 512 + 2))
        self._call(self.lib.egonn_col_stats, mode, a.data_ptr(), _ptr(b), _ptr(mask), _ptr(mean), n, c, out.data_ptr(),
                   sc.data_ptr(), sc.numel())
        return out

    def affine_act(self, x, scale, shift, relu: bool):
        x = _dev_f32(x, self.device)
        out = torch.empty_like(x)
        self._call(self.lib.egonn_affine_act, x.data_ptr(), scale.data_ptr(), shift.data_ptr(), x.shape[0], x.shape[1],
                   int(relu), out.data_ptr())
        return out

    def affine3(self, g, mask, x, A, B, Cc):
        g, x = _dev_f32(g, self.device), _dev_f32(x, self.device)
        out = torch.empty_like(x)
        self._call(self.lib.egonn_affine3, g.data_ptr(), _ptr(mask), x.data_ptr(), A.data_ptr(), B.data_ptr(),
                   Cc.data_ptr(), x.shape[0], x.shape[1], out.data_ptr())
        return out

    def relu_backward(self, grad_out, out):
        g = _dev_f32(grad_out, self.device)
        dx = torch.empty_like(g)
        self._call(self.lib.egonn_relu_backward, g.data_ptr(), out.data_ptr(), g.shape[0], g.shape[1], dx.data_ptr())
        return dx

    def act_backward(self, act: int, grad_out, out):
        g = _dev_f32(grad_out, self.device)
        dx = torch.empty_like(g)
        self._call(self.lib.egonn_act_backward, act, g.data_ptr(), out.data_ptr(), g.shape[0], g.shape[1], dx.data_ptr())
        return dx

    def l2_normalize(self, x, grad_out=None):
        x = _dev_f32(x, self.device)
        g = None if grad_out is None else _dev_f32(grad_out, self.device)
        out = torch.empty_like(x)
        self._call(self.lib.egonn_l2_normalize, x.data_ptr(), _ptr(g), x.shape[0], x.shape[1], out.data_ptr())
        return out

    def gate_residual(self, level, x, gate, residual, relu: bool = True):
        x = _dev_f32(x, self.device)
        assert x.shape[0] == self.level_count(level)
        out = torch.empty_like(x)
        self._call(self.lib.egonn_gate_residual, self.h, level, x.data_ptr(), _ptr(gate), _ptr(residual), x.shape[1],
                   int(relu), out.data_ptr())
        return out

    def gate_residual_backward(self, level, grad_out, out, gate, want_residual: bool = True):
        g = _dev_f32(grad_out, self.device)
        dx = torch.empty_like(g)
        dres = torch.empty_like(g) if want_residual else None
        self._call(self.lib.egonn_gate_residual_backward, self.h, level, g.data_ptr(), _ptr(out), _ptr(gate), g.shape[1],
                   dx.data_ptr(), _ptr(dres))
        return dx, dres

    def segment_sums(self, level, mode, a, b=None, x2=None, p=None):
        a = _dev_f32(a, self.device)
        c = a.shape[1]
        out = torch.empty((self.batch_size, c), dtype=torch.float32, device=self.device)
        sc = self.scratch(32 * self.batch_size * c)
        self._call(self.lib.egonn_segment_sums, self.h, level, mode, a.data_ptr(), _ptr(b), _ptr(x2), _ptr(p), c,
                   out.data_ptr(), sc.data_ptr(), sc.numel())
        return out

    def segment_broadcast(self, level, v, mean: bool):
        v = _dev_f32(v, self.device)
        out = torch.empty((self.level_count(level), v.shape[1]), dtype=torch.float32, device=self.device)
        self._call(self.lib.egonn_segment_broadcast, self.h, level, v.data_ptr(), v.shape[1], int(mean), out.data_ptr())
        return out

    def gem_backward(self, level, x, coef, p):
        x, coef = _dev_f32(x, self.device), _dev_f32(coef, self.device)
        dx = torch.empty_like(x)
        self._call(self.lib.egonn_gem_backward, self.h, level, x.data_ptr(), coef.data_ptr(), p.data_ptr(), x.shape[1],
                   dx.data_ptr())
        return dx

    def forward_level_features(self, level: int, channels: int) -> torch.Tensor:
        out = torch.empty((self.level_count(level), channels), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.egonn_forward_level_features(self.h, level, out.data_ptr(), channels, _stream()))
        return out

    def select_keypoints(self, sigma: torch.Tensor, keypoints: torch.Tensor, descriptors: torch.Tensor, n_k: int):
        B = self.batch_size
        dev = self.device
        sel_kp = torch.empty((B, n_k, 3), dtype=torch.float32, device=dev)
        sel_desc = torch.empty((B, n_k, descriptors.shape[1]), dtype=torch.float32, device=dev)
        sel_rows = torch.empty((B, n_k), dtype=torch.int32, device=dev)
        sel_count = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(self.lib.egonn_select_keypoints(self.h, sigma.data_ptr(), keypoints.data_ptr(),
                                                  descriptors.data_ptr(), n_k, sel_kp.data_ptr(), sel_desc.data_ptr(),
                                                  sel_rows.data_ptr(), sel_count.data_ptr(), _stream()))
        return sel_kp, sel_desc, sel_rows, sel_count


    # ------------------------------------------------------------------ launch timing
    def profile_enable(self, mode: int, filt: str = ""):
        check(self.lib.egonn_profile_enable(self.h, mode, filt.encode()))

    def profile_fetch(self, cap: int = 4096):
        n = C.c_int()
        names = C.create_string_buffer(cap * 64)
        ms = (C.c_float * cap)()
        by = (C.c_double * cap)()
        fl = (C.c_double * cap)()
        with torch.cuda.device(self.device):
            check(self.lib.egonn_profile_fetch(self.h, cap, C.byref(n), names, ms, by, fl, _stream()))
        out = []
        for i in range(n.value):
            nm = names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode()
            out.append((nm, float(ms[i]), float(by[i]), float(fl[i])))
        return out


class ModelHandle:
    """egonn_model: weights registered under the reference's state_dict keys."""

    def __init__(self):
        self.lib = load()
        h = _P()
        check(self.lib.egonn_model_create(C.byref(h)))
        self.h = h
        self._keep = {}

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.egonn_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_tensor(self, key: str, t: torch.Tensor):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), key
        shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        check(self.lib.egonn_model_set_tensor(self.h, key.encode(), t.data_ptr(), t.dim(), shape))
        self._keep[key] = t

    def finalize(self):
        check(self.lib.egonn_model_finalize(self.h, _stream()))
