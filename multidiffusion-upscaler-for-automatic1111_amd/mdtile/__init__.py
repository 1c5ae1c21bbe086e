"""
mdtile -- ctypes binding of libmdtile.so (include/mdtile.h) for the PyTorch-ROCm host.

PyTorch is plumbing here: it owns device memory and the HIP stream; every hot-path computation happens in the
hand-written gfx950 kernels behind the C ABI.  There is NO CPU / eager-torch fallback: if the shared library is
missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p
from typing import List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdtile.so")

OK, E_ARG, E_HIP, E_LIMIT = 0, -1, -2, -3
DT_F32, DT_F16, DT_BF16 = 0, 1, 2
METHOD_MD, METHOD_MOD = 0, 1
REGION_BG, REGION_FG = 0, 1
BLEND_PARTIAL, BLEND_TILE_RANGE, BLEND_PACKED = 1, 2, 4
CONV_UPSAMPLE2X = 1
CONV_REC_ONE_BLOCK, CONV_REC_TWO_BLOCKS = 4, 8      # call_rec(family=...): name the record-conv kernel family (tests / probes); 0 = per launch
CONV_EXACT_F32 = 2
ATTN_EXACT_F32 = 1
ATTN_V_CHANNEL_MAJOR = 2
MAX_BATCHES, MAX_REGIONS = 320, 16

_DTYPES = {torch.float32: DT_F32, torch.float16: DT_F16, torch.bfloat16: DT_BF16}


class MdtileError(RuntimeError):
    pass


class _Region(ctypes.Structure):
    _fields_ = [("x", c_int), ("y", c_int), ("w", c_int), ("h", c_int), ("mode", c_int), ("_pad", c_int),
                ("out", c_void_p), ("weight", c_void_p)]


class _P2P(ctypes.Structure):
    _fields_ = [("peer", c_int), ("send", c_void_p), ("send_bytes", c_size_t), ("recv", c_void_p), ("recv_bytes", c_size_t)]


class _BlendArgs(ctypes.Structure):
    _fields_ = [("method", c_int), ("dtype", c_int), ("N", c_int), ("C", c_int), ("flags", c_int),
                ("tile_lo", c_int), ("tile_hi", c_int), ("row_lo", c_int), ("row_hi", c_int),
                ("d_weights", c_void_p), ("d_tile_w", c_void_p), ("d_rescale", c_void_p), ("d_x_out", c_void_p)]


# every symbol include/mdtile.h declares: name -> (restype, argtypes)
_IP = POINTER(c_int)
_SIGNATURES = {
    "mdtile_version": (c_int, []),
    "mdtile_last_error": (c_char_p, []),
    "mdtile_set_precision": (c_int, [c_int]),
    "mdtile_get_precision": (c_int, []),
    "mdtile_plan_create": (c_void_p, [c_int] * 7),
    "mdtile_plan_destroy": (None, [c_void_p]),
    "mdtile_plan_info": (c_int, [c_void_p, _IP]),
    "mdtile_plan_bboxes": (c_int, [c_void_p, _IP]),
    "mdtile_gaussian_weights": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "mdtile_feather_mask": (c_int, [c_int, c_int, c_double, c_void_p, c_void_p]),
    "mdtile_weight_map_add_grid": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdtile_weight_map_add_rect": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p]),
    "mdtile_reciprocal": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdtile_rect_mul_canvas": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_gather": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "mdtile_gather_all": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, POINTER(c_void_p), c_int, c_void_p]),
    "mdtile_gather_range": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mdtile_gather_rect": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdtile_stream_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdtile_blend": (c_int, [c_void_p, POINTER(_BlendArgs), POINTER(c_void_p), c_int, POINTER(_Region), c_int, c_void_p]),
    "mdtile_blend_finalize": (c_int, [c_void_p, POINTER(_BlendArgs), c_void_p, POINTER(_Region), c_int, c_void_p]),
    "mdtile_region_noise": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(_Region), c_int, c_void_p]),
    "mdtile_noise_inverse_blend": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(_Region), c_int, c_void_p]),
    "mdtile_gather_rects": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, _IP, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdtile_shard_init": (c_void_p, [c_int, _IP]),
    "mdtile_shard_unique_id": (c_int, [c_void_p]),
    "mdtile_shard_init_rank": (c_void_p, [c_int, c_int, c_void_p, c_int]),
    "mdtile_shard_probe_rank": (c_int, [c_int, c_int, c_void_p, c_int, c_double]),
    "mdtile_shard_destroy": (None, [c_void_p]),
    "mdtile_shard_info": (c_int, [c_void_p, _IP]),
    "mdtile_shard_stream": (c_void_p, [c_void_p, c_int]),
    "mdtile_halo_scratch_bytes": (c_size_t, [c_int, c_int, _IP, c_int, c_int, c_int]),
    "mdtile_halo_exchange": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_int, _IP, POINTER(c_void_p)]),
    "mdtile_allreduce_stats": (c_int, [c_void_p, POINTER(c_void_p), c_int, POINTER(c_void_p)]),
    "mdtile_shard_bcast": (c_int, [c_void_p, POINTER(c_void_p), c_size_t, c_int, POINTER(c_void_p)]),
    "mdtile_shard_p2p": (c_int, [c_void_p, POINTER(POINTER(_P2P)), _IP, POINTER(c_void_p)]),
    "mdtile_shard_allgather": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, POINTER(c_void_p)]),
    "mdtile_shard_selfcheck_bytes": (c_size_t, [c_void_p]),
    "mdtile_shard_selfcheck": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p)]),
    "mdtile_window_blend": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_dilated_gather": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, _IP, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_demofusion_combine": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "mdtile_depthwise_blur": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_restandardize": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdtile_vae_split_tiles": (c_int, [c_int, c_int, c_int, c_int, _IP, _IP, c_int]),
    "mdtile_vae_best_tile_size": (c_int, [c_int, c_int]),
    "mdtile_gn_stats_ws_size": (c_size_t, [c_int, c_int]),
    "mdtile_gn_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdtile_gn_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mdtile_gn_apply": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_float, c_int, c_void_p]),
    "mdtile_silu": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdtile_tanh": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdtile_add": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdtile_conv_packed_size": (c_size_t, [c_int, c_int, c_int]),
    "mdtile_conv_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mdtile_conv2d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_int, c_void_p]),
    "mdtile_conv2d_down2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_gn_sums": (c_int, [c_void_p, c_int, c_int, c_size_t, c_size_t, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "mdtile_gn_from_sums": (c_int, [c_void_p, ctypes.c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "mdtile_vae_attn_qk_ws_size": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mdtile_vae_attn_qk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "mdtile_gn_coeffs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "mdtile_conv2d_gn_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mdtile_conv2d_gn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p]),
    "mdtile_conv_stats_ws_size": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mdtile_conv2d_gn_stats_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mdtile_conv2d_gn_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdtile_conv2d_rec_stats_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mdtile_conv2d_rec_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdtile_rec_size": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mdtile_rec_from_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_rec_to_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdtile_conv2d_rec_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mdtile_conv2d_rec": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_void_p]),
    "mdtile_upconv2d_rec_window": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_int, c_int,
                                                                                                                  c_int, c_void_p]),
    "mdtile_vae_attn_ws_size": (c_size_t, [c_int, c_int, c_int]),
    "mdtile_vae_attn_takes_channel_major": (c_int, [c_int, c_int]),
    "mdtile_vae_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdtile_crop_store": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _IP, _IP, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mdtile_vae_fast_size": (c_int, [c_int, c_int, c_int, _IP, _IP]),
    "mdtile_vae_fast_ws_size": (c_size_t, [c_int]),
    "mdtile_vae_fast_input": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load libmdtile.so (once).  Raises loudly when it is absent -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MdtileError(f"{LIB_PATH} not found: build it with `python -m mdtile.build` (hipcc, gfx950). "
                          "This extension has no CPU/eager fallback.")
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export what the header declares
        fn.restype, fn.argtypes = res, args
    if L.mdtile_version() != 100:
        raise MdtileError(f"libmdtile.so version {L.mdtile_version()} != binding version 100")
    _lib = L
    return L


PRECISION_BF16X3, PRECISION_F32 = 0, 1


def set_precision(mode: int) -> None:
    """PRECISION_BF16X3 (default: split-bf16 matrix-core kernels) or PRECISION_F32 (exact-fp32 MFMA kernels everywhere)."""
    _check(lib().mdtile_set_precision(int(mode)), "mdtile_set_precision")


def get_precision() -> int:
    return int(lib().mdtile_get_precision())


def exported_symbols() -> List[str]:
    return list(_SIGNATURES)


def _check(rc: int, what: str):
    if rc != OK:
        raise MdtileError(f"{what} failed (rc={rc}): {lib().mdtile_last_error().decode(errors='replace')}")


def require_device(dev) -> None:
    """The engine only runs on the GPU: callers check up front instead of failing inside the first kernel wrapper."""
    if torch.device(dev).type != "cuda":
        raise MdtileError(f"the mdtile engine needs its tensors on the GPU, got device {dev} (no CPU fallback exists; "
                          "for Tiled VAE enable 'Move VAE to GPU')")


def _dev_tensor(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if t.device.type != "cuda":
        raise MdtileError(f"{name} lives on {t.device}; the mdtile engine only runs on the GPU (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise MdtileError(f"{name} has dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise MdtileError(f"{name} must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DTYPES[dt]
    except KeyError:
        raise MdtileError(f"unsupported dtype {dt}") from None


# ---------------------------------------------------------------------------------------------------------------------
class Plan:
    """Grid plan == split_bboxes + init_grid_bbox (tile_utils/utils.py:160-177, abstractdiffusion.py:173-186)."""

    def __init__(self, w: int, h: int, tile_w: int, tile_h: int, overlap: int, tile_bs: int, clamp: bool = True):
        L = lib()
        self._h = L.mdtile_plan_create(int(w), int(h), int(tile_w), int(tile_h), int(overlap), int(tile_bs), int(clamp))
        if not self._h:
            raise MdtileError("mdtile_plan_create: " + L.mdtile_last_error().decode(errors="replace"))
        info = (c_int * 8)()
        _check(L.mdtile_plan_info(self._h, info), "mdtile_plan_info")
        (self.cols, self.rows, self.num_tiles, self.num_batches, self.tile_bs, self.tile_w, self.tile_h,
         self.overlap) = list(info)
        self.w, self.h = int(w), int(h)
        buf = (c_int * (4 * self.num_tiles))()
        _check(L.mdtile_plan_bboxes(self._h, buf), "mdtile_plan_bboxes")
        flat = list(buf)
        self.bboxes = [tuple(flat[4 * i:4 * i + 4]) for i in range(self.num_tiles)]  # (x, y, w, h), upstream order
        self.batches = [self.bboxes[i * self.tile_bs:(i + 1) * self.tile_bs] for i in range(self.num_batches)]

    @property
    def handle(self):
        return self._h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.mdtile_plan_destroy(h)


# ---- maps ------------------------------------------------------------------------------------------------------------
def gaussian_weights(tile_w: int, tile_h: int, device) -> torch.Tensor:
    out = torch.empty((tile_h, tile_w), dtype=torch.float32, device=device)
    _dev_tensor(out, "out")
    _check(lib().mdtile_gaussian_weights(tile_w, tile_h, _p(out), _stream()), "mdtile_gaussian_weights")
    return out


def feather_mask(w: int, h: int, ratio: float, device) -> torch.Tensor:
    out = torch.empty((h, w), dtype=torch.float32, device=device)
    _dev_tensor(out, "out")
    _check(lib().mdtile_feather_mask(w, h, float(ratio), _p(out), _stream()), "mdtile_feather_mask")
    return out


def weight_map_add_grid(plan: Plan, tile_w: Optional[torch.Tensor], weights: torch.Tensor) -> None:
    _dev_tensor(weights, "weights", torch.float32)
    if tile_w is not None:
        _dev_tensor(tile_w, "tile_w", torch.float32)
        assert tuple(tile_w.shape[-2:]) == (plan.tile_h, plan.tile_w)
    assert weights.numel() == plan.w * plan.h
    _check(lib().mdtile_weight_map_add_grid(plan.handle, _p(tile_w), _p(weights), _stream()), "mdtile_weight_map_add_grid")


def weight_map_add_rect(weights: torch.Tensor, x: int, y: int, w: int, h: int, rect_w: Optional[torch.Tensor] = None,
                        scalar: float = 1.0) -> None:
    _dev_tensor(weights, "weights", torch.float32)
    H, W = weights.shape[-2:]
    if rect_w is not None:
        _dev_tensor(rect_w, "rect_w", torch.float32)
        assert rect_w.numel() == w * h
    _check(lib().mdtile_weight_map_add_rect(_p(weights), W, H, x, y, w, h, _p(rect_w), scalar, _stream()),
           "mdtile_weight_map_add_rect")


def region_noise(noise: torch.Tensor, regions) -> torch.Tensor:
    """In-place per-region noise paste (tilediffusion.py:486-529).  noise: [N,C,H,W] fp32 on the GPU;
    regions: [(x, y, w, h, mode, rand)] with rand = that region's own noise [1,C,h,w] fp32 (same device), in list order."""
    _dev_tensor(noise, "noise", torch.float32)
    N, C, H, W = noise.shape
    arr = (_Region * max(1, len(regions)))()
    keep = []
    for i, (x, y, w, h, mode, rand) in enumerate(regions):
        _dev_tensor(rand, f"regions[{i}].rand", torch.float32)
        assert tuple(rand.shape) == (1, C, h, w), f"region {i}: noise shape {tuple(rand.shape)} != (1, {C}, {h}, {w})"
        keep.append(rand)
        arr[i] = _Region(int(x), int(y), int(w), int(h), int(mode), 0, rand.data_ptr(), None)
    _check(lib().mdtile_region_noise(_p(noise), N, C, H, W, arr, len(regions), _stream()), "mdtile_region_noise")
    return noise


def noise_inverse_blend(noise: torch.Tensor, inverse_noise: torch.Tensor, renoise_mask: torch.Tensor, regions=()) -> torch.Tensor:
    """Renoise composite of Noise Inversion (abstractdiffusion.py:651-676).  noise / inverse_noise [N,C,H,W] fp32, renoise_mask
    [H,W] fp32; regions: [(x, y, w, h, mode, feather [h,w] fp32 or None)] -- pass them only when the grid is disabled (:658)."""
    _dev_tensor(noise, "noise", torch.float32)
    _dev_tensor(inverse_noise, "inverse_noise", torch.float32)
    _dev_tensor(renoise_mask, "renoise_mask", torch.float32)
    N, C, H, W = noise.shape
    assert inverse_noise.shape == noise.shape and tuple(renoise_mask.shape) == (H, W)
    arr = (_Region * max(1, len(regions)))()
    keep = []
    for i, (x, y, w, h, mode, feather) in enumerate(regions):
        fp = 0
        if feather is not None:
            _dev_tensor(feather, f"regions[{i}].feather", torch.float32)
            assert feather.numel() == w * h
            keep.append(feather)
            fp = feather.data_ptr()
        arr[i] = _Region(int(x), int(y), int(w), int(h), int(mode), 0, None, fp)
    out = torch.empty_like(noise)
    _check(lib().mdtile_noise_inverse_blend(_p(noise), _p(inverse_noise), _p(renoise_mask), _p(out), N, C, H, W, arr, len(regions), _stream()),
           "mdtile_noise_inverse_blend")
    return out


def gather_rects(x_in: torch.Tensor, rects_xy: Sequence[Tuple[int, int]], w: int, h: int, repeat: int = 1, tile_major: bool = True) -> torch.Tensor:
    """cat([x_in[:, :, y:y+h, x:x+w] for (x, y) in rects_xy]) repeated for the sampler's batch copies in one launch
    (ControlNet / StableSR tile slicing, abstractdiffusion.py:475-544, 548-588).  tile_major: every row's copies are consecutive
    (k-diffusion); else the whole batch is repeated (DDIM)."""
    _dev_tensor(x_in, "x_in")
    N, C, H, W = x_in.shape
    n = len(rects_xy)
    flat = (c_int * (2 * n))(*[int(v) for xy in rects_xy for v in xy])
    out = torch.empty((n * N * repeat, C, h, w), dtype=x_in.dtype, device=x_in.device)
    _check(lib().mdtile_gather_rects(dtype_code(x_in.dtype), N, C, W, H, _p(x_in), flat, n, int(w), int(h), int(repeat), int(tile_major),
                                     _p(out), _stream()), "mdtile_gather_rects")
    return out


def reciprocal(x: torch.Tensor) -> torch.Tensor:
    _dev_tensor(x, "x", torch.float32)
    out = torch.empty_like(x)
    _check(lib().mdtile_reciprocal(_p(x), _p(out), x.numel(), _stream()), "mdtile_reciprocal")
    return out


def rect_mul_canvas(rect_w: torch.Tensor, canvas: torch.Tensor, x: int, y: int, w: int, h: int) -> None:
    _dev_tensor(rect_w, "rect_w", torch.float32)
    _dev_tensor(canvas, "canvas", torch.float32)
    H, W = canvas.shape[-2:]
    assert rect_w.numel() == w * h
    _check(lib().mdtile_rect_mul_canvas(_p(rect_w), _p(canvas), W, H, x, y, w, h, _stream()), "mdtile_rect_mul_canvas")


# ---- gather ----------------------------------------------------------------------------------------------------------
def gather(plan: Plan, x_in: torch.Tensor, batch_id: int) -> torch.Tensor:
    """x_tile = cat([x_in[bbox.slicer] for bbox in batch]) -- tile-major (multidiffusion.py:155)."""
    _dev_tensor(x_in, "x_in")
    N, C, H, W = x_in.shape
    assert (H, W) == (plan.h, plan.w)
    nb = len(plan.batches[batch_id])
    out = torch.empty((nb * N, C, plan.tile_h, plan.tile_w), dtype=x_in.dtype, device=x_in.device)
    _check(lib().mdtile_gather(plan.handle, dtype_code(x_in.dtype), N, C, _p(x_in), batch_id, _p(out), _stream()), "mdtile_gather")
    return out


def gather_all(plan: Plan, x_in: torch.Tensor, out: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """All tile batches in one launch."""
    _dev_tensor(x_in, "x_in")
    N, C, H, W = x_in.shape
    assert (H, W) == (plan.h, plan.w)
    if out is None and len(plan.batches) > MAX_BATCHES:
        # beyond the pointer table of the kernel arguments: one tile-major buffer, the batches are views into it
        packed = torch.empty((plan.num_tiles * N, C, plan.tile_h, plan.tile_w), dtype=x_in.dtype, device=x_in.device)
        gather_range(plan, x_in, packed, 0, plan.num_tiles)
        return list(packed.split([len(b) * N for b in plan.batches], dim=0))
    if out is None:
        out = [torch.empty((len(b) * N, C, plan.tile_h, plan.tile_w), dtype=x_in.dtype, device=x_in.device)
               for b in plan.batches]
    ptrs = (c_void_p * len(out))(*[t.data_ptr() for t in out])
    _check(lib().mdtile_gather_all(plan.handle, dtype_code(x_in.dtype), N, C, _p(x_in), ptrs, len(out), _stream()),
           "mdtile_gather_all")
    return list(out)


def gather_range(plan: Plan, x_in: torch.Tensor, packed: torch.Tensor, tile_lo: int, tile_hi: int) -> torch.Tensor:
    """Tiles [tile_lo, tile_hi) into the packed [T*N, C, th, tw] buffer (rows of other tiles are left untouched)."""
    _dev_tensor(x_in, "x_in")
    _dev_tensor(packed, "packed", x_in.dtype)
    N, C, H, W = x_in.shape
    assert (H, W) == (plan.h, plan.w) and packed.numel() == plan.num_tiles * N * C * plan.tile_h * plan.tile_w
    _check(lib().mdtile_gather_range(plan.handle, dtype_code(x_in.dtype), N, C, _p(x_in), _p(packed), tile_lo, tile_hi, _stream()),
           "mdtile_gather_range")
    return packed


def gather_rect(x_in: torch.Tensor, x: int, y: int, w: int, h: int) -> torch.Tensor:
    _dev_tensor(x_in, "x_in")
    N, C, H, W = x_in.shape
    out = torch.empty((N, C, h, w), dtype=x_in.dtype, device=x_in.device)
    _check(lib().mdtile_gather_rect(dtype_code(x_in.dtype), N, C, W, H, _p(x_in), x, y, w, h, _p(out), _stream()),
           "mdtile_gather_rect")
    return out


# ---- blend -----------------------------------------------------------------------------------------------------------
class RegionSpec:
    """One custom region for the blend: rect, mode, its model output and (optional) weight / feather map."""

    __slots__ = ("x", "y", "w", "h", "mode", "out", "weight")

    def __init__(self, x, y, w, h, mode, out, weight=None):
        self.x, self.y, self.w, self.h, self.mode, self.out, self.weight = x, y, w, h, mode, out, weight


def _regions_array(regions: Sequence[RegionSpec], dtype):
    if len(regions) > MAX_REGIONS:
        raise MdtileError(f"{len(regions)} regions > {MAX_REGIONS}")
    arr = (_Region * max(1, len(regions)))()
    keep = []
    for i, r in enumerate(regions):
        out = _dev_tensor(r.out, f"region[{i}].out", dtype)
        keep.append(out)
        wp = 0
        if r.weight is not None:
            wt = _dev_tensor(r.weight, f"region[{i}].weight", torch.float32)
            assert wt.numel() == r.w * r.h
            keep.append(wt)
            wp = wt.data_ptr()
        assert tuple(out.shape[-2:]) == (r.h, r.w), f"region[{i}] output {tuple(out.shape)} vs rect {r.h}x{r.w}"
        arr[i] = _Region(r.x, r.y, r.w, r.h, r.mode, 0, out.data_ptr(), wp)
    return arr, keep


def _as_one_buffer(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """The tensors of `parts` as one contiguous [sum rows, ...] tensor: zero-copy when they are back-to-back views of one storage."""
    first = parts[0]
    at, ok = first.data_ptr(), True
    for t in parts:
        ok = ok and t.is_contiguous() and t.data_ptr() == at and t.shape[1:] == first.shape[1:] and t.dtype == first.dtype
        at += t.numel() * t.element_size()
    rows = sum(t.shape[0] for t in parts)
    if ok and first.untyped_storage().nbytes() - first.storage_offset() * first.element_size() >= at - first.data_ptr():
        return first.as_strided((rows,) + tuple(first.shape[1:]), first.stride())
    return torch.cat(list(parts), dim=0)


class BlendCall:
    """A fully marshalled mdtile_blend call: the argument struct, the batch-pointer array and the region array are built ONCE; calling
    the object only does the C call (a few microseconds of host time instead of rebuilding ~10 ctypes objects per evaluation -- the
    blend itself runs 20 us on an 8K canvas).  Valid while the tensors it was built from stay alive at the same addresses (sampler
    loops that write the model outputs into fixed buffers; bench.py)."""

    __slots__ = ("plan", "out", "_a", "_ptrs", "_n", "_rarr", "_nreg", "_keep", "_fn")

    def __init__(self, plan: Plan, method: int, batch_out: Sequence[torch.Tensor], N: int, C: int, *, weights=None, tile_w=None, rescale=None,
                 regions: Sequence[RegionSpec] = (), out: Optional[torch.Tensor] = None, dtype=None, device=None, partial: bool = False,
                 tile_range=None, row_range=None, packed: bool = False):
        if len(batch_out):
            dtype, device = batch_out[0].dtype, batch_out[0].device
        elif len(regions):
            dtype, device = regions[0].out.dtype, regions[0].out.device
        assert dtype is not None and device is not None
        for i, t in enumerate(batch_out):
            _dev_tensor(t, f"batch_out[{i}]", dtype)
        if not packed and len(batch_out) > MAX_BATCHES:
            # more tile batches than pointers ride in the kernel arguments (e.g. a 1024^2 latent at tile 96 / overlap 48 / batch 1 =
            # 441): hand the kernel ONE tile-major buffer instead.  Batches are consecutive runs of the tile list, so the views
            # gather_all returns in this situation are already one buffer; anything else costs one concatenation.
            batch_out, packed = [_as_one_buffer(batch_out)], True
        flags = (BLEND_PARTIAL if partial else 0) | (BLEND_PACKED if packed else 0) | (BLEND_TILE_RANGE if tile_range is not None else 0)
        if out is None:
            out = torch.empty((N, C, plan.h, plan.w), dtype=torch.float32 if partial else dtype, device=device)
        _dev_tensor(out, "out", torch.float32 if partial else dtype)
        for nm, t in (("weights", weights), ("tile_w", tile_w), ("rescale", rescale)):
            if t is not None:
                _dev_tensor(t, nm, torch.float32)
        self.plan, self.out = plan, out
        self._a = _BlendArgs(method, dtype_code(dtype), N, C, flags, *(tile_range or (0, 0)), *(row_range or (0, 0)),
                             _p(weights).value, _p(tile_w).value, _p(rescale).value, out.data_ptr())
        self._ptrs = (c_void_p * max(1, len(batch_out)))(*[t.data_ptr() for t in batch_out])
        self._n = len(batch_out)
        self._rarr, keep = _regions_array(regions, dtype)
        self._nreg = len(regions)
        self._keep = (list(batch_out), weights, tile_w, rescale, keep)
        self._fn = lib().mdtile_blend

    def __call__(self) -> torch.Tensor:
        rc = self._fn(self.plan.handle, ctypes.byref(self._a), self._ptrs, self._n, self._rarr, self._nreg, _stream())
        if rc != OK:
            _check(rc, "mdtile_blend")
        return self.out


def blend(plan: Plan, method: int, batch_out: Sequence[torch.Tensor], N: int, C: int, *, weights=None, tile_w=None,
          rescale=None, regions: Sequence[RegionSpec] = (), out: Optional[torch.Tensor] = None, dtype=None, device=None,
          partial: bool = False, tile_range=None, row_range=None, packed: bool = False) -> torch.Tensor:
    """One fused launch replacing the scatter/normalise/feather op sequence of sample_one_step / apply_model_hijack."""
    return BlendCall(plan, method, batch_out, N, C, weights=weights, tile_w=tile_w, rescale=rescale, regions=regions, out=out, dtype=dtype,
                     device=device, partial=partial, tile_range=tile_range, row_range=row_range, packed=packed)()


class StreamCopyCall:
    """Marshalled mdtile_stream_copy (src -> dst, same size, contiguous, 16-byte aligned): the measurement floor bench.py times beside the
    blend kernel, launched the same way (see BlendCall)."""

    __slots__ = ("_args", "_fn", "_keep")

    def __init__(self, src: torch.Tensor, dst: torch.Tensor):
        _dev_tensor(src, "src")
        _dev_tensor(dst, "dst")
        nbytes = src.numel() * src.element_size()
        assert nbytes == dst.numel() * dst.element_size() and nbytes % 16 == 0
        self._args = (c_void_p(src.data_ptr()), c_void_p(dst.data_ptr()), c_size_t(nbytes))
        self._keep = (src, dst)
        self._fn = lib().mdtile_stream_copy

    def __call__(self) -> None:
        rc = self._fn(*self._args, _stream())
        if rc != OK:
            _check(rc, "mdtile_stream_copy")


def stream_copy(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    StreamCopyCall(src, dst)()
    return dst


class GatherRangeCall:
    """Marshalled mdtile_gather_range (tiles [tile_lo, tile_hi) into a packed buffer), see BlendCall."""

    __slots__ = ("_args", "_fn", "_keep", "packed")

    def __init__(self, plan: Plan, x_in: torch.Tensor, packed: torch.Tensor, tile_lo: int, tile_hi: int):
        _dev_tensor(x_in, "x_in")
        _dev_tensor(packed, "packed", x_in.dtype)
        N, C, H, W = x_in.shape
        assert (H, W) == (plan.h, plan.w) and packed.numel() == plan.num_tiles * N * C * plan.tile_h * plan.tile_w
        self._args = (plan.handle, dtype_code(x_in.dtype), N, C, _p(x_in), _p(packed), int(tile_lo), int(tile_hi))
        self._fn = lib().mdtile_gather_range
        self._keep, self.packed = (plan, x_in), packed

    def __call__(self) -> torch.Tensor:
        rc = self._fn(*self._args, _stream())
        if rc != OK:
            _check(rc, "mdtile_gather_range")
        return self.packed


def blend_finalize(plan: Plan, method: int, partial: torch.Tensor, *, weights=None, regions: Sequence[RegionSpec] = (),
                   out: Optional[torch.Tensor] = None, dtype=torch.float32, row_range=None) -> torch.Tensor:
    _dev_tensor(partial, "partial", torch.float32)
    N, C = partial.shape[:2]
    if out is None:
        out = torch.empty((N, C, plan.h, plan.w), dtype=dtype, device=partial.device)
    _dev_tensor(out, "out", dtype)
    a = _BlendArgs(method, dtype_code(dtype), N, C, 0, 0, 0, *(row_range or (0, 0)), _p(weights).value, 0, 0, out.data_ptr())
    rarr, _keep = _regions_array(regions, dtype)
    _check(lib().mdtile_blend_finalize(plan.handle, ctypes.byref(a), _p(partial), rarr, len(regions), _stream()),
           "mdtile_blend_finalize")
    return out


# ---- tiled VAE -------------------------------------------------------------------------------------------------------
def vae_split_tiles(h: int, w: int, tile_size: int, is_decoder: bool = True):
    """split_tiles (scripts/tilevae.py:405-462): ([x1,x2,y1,y2] input bboxes, output bboxes).  Host ints only."""
    L = lib()
    n = L.mdtile_vae_split_tiles(h, w, tile_size, int(is_decoder), None, None, 0)
    if n < 0:
        _check(n, "mdtile_vae_split_tiles")
    ins, outs = (c_int * (4 * n))(), (c_int * (4 * n))()
    rc = L.mdtile_vae_split_tiles(h, w, tile_size, int(is_decoder), ins, outs, n)
    if rc < 0:
        _check(rc, "mdtile_vae_split_tiles")
    fi, fo = list(ins), list(outs)
    return [fi[4 * i:4 * i + 4] for i in range(n)], [fo[4 * i:4 * i + 4] for i in range(n)]


def vae_best_tile_size(lowerbound: int, upperbound: int) -> int:
    """get_best_tile_size (scripts/tilevae.py:390-403).  Host ints only."""
    return int(lib().mdtile_vae_best_tile_size(int(lowerbound), int(upperbound)))


def gn_stats(x: torch.Tensor, groups: int = 32):
    """get_var_mean (tilevae.py:207-215) -> (var, mean), each [B*groups]."""
    _dev_tensor(x, "x", torch.float32)
    B, C = x.shape[:2]
    HW = x.numel() // (B * C)
    mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    ws = torch.empty(lib().mdtile_gn_stats_ws_size(B, groups) // 8, dtype=torch.float64, device=x.device)
    _check(lib().mdtile_gn_stats(_p(x), B, C, HW, groups, _p(mean), _p(var), _p(ws), _stream()), "mdtile_gn_stats")
    return var, mean


def gn_pool(means: torch.Tensor, vars_: torch.Tensor, pixels: Sequence[int]):
    """GroupNormParam.summary (tilevae.py:320-335) -> (var, mean)."""
    _dev_tensor(means, "means", torch.float32)
    _dev_tensor(vars_, "vars", torch.float32)
    T, BG = means.shape
    assert len(pixels) == T
    mean = torch.empty(BG, dtype=torch.float32, device=means.device)
    var = torch.empty_like(mean)
    # p_i exactly as upstream forms them (fp32, tilevae.py:328-331); T host floats -- control data, not tensor math
    px = torch.tensor([int(p) for p in pixels], dtype=torch.float32) / max(pixels)
    frac = (px / torch.sum(px)).to(means.device)
    _check(lib().mdtile_gn_pool(_p(means), _p(vars_), _p(frac), T, BG, _p(mean), _p(var), _stream()), "mdtile_gn_pool")
    return var, mean


def gn_apply(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, gamma=None, beta=None, groups: int = 32,
             eps: float = 1e-6, silu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """custom_group_norm (+ fused SiLU) (tilevae.py:218-245, 102-104)."""
    _dev_tensor(x, "x", torch.float32)
    B, C = x.shape[:2]
    HW = x.numel() // (B * C)
    if out is None:
        out = torch.empty_like(x)
    for nm, t in (("mean", mean), ("var", var)):
        _dev_tensor(t, nm, torch.float32)
        assert t.numel() == B * groups
    for nm, t in (("gamma", gamma), ("beta", beta)):
        if t is not None:
            _dev_tensor(t, nm, torch.float32)
            assert t.numel() == C
    _check(lib().mdtile_gn_apply(_p(x), _p(out), B, C, HW, groups, _p(mean), _p(var), _p(gamma), _p(beta), eps, int(silu),
                                 _stream()), "mdtile_gn_apply")
    return out


def gn_sums(x: torch.Tensor, row_lo: int, row_hi: int, groups: int = 32) -> torch.Tensor:
    """fp64 [B*groups, 2] (sum, sum of squares) over rows [row_lo, row_hi) of every plane of x [B, C, H, W] (the rows a rank
    owns; halo rows excluded).  Piece of get_var_mean (tilevae.py:207-215) for a row-split activation."""
    _dev_tensor(x, "x", torch.float32)
    B, C, H, W = x.shape
    assert 0 <= row_lo < row_hi <= H
    sums = torch.empty((B * groups, 2), dtype=torch.float64, device=x.device)
    ws = torch.empty(lib().mdtile_gn_stats_ws_size(B, groups) // 8, dtype=torch.float64, device=x.device)
    _check(lib().mdtile_gn_sums(_p(x), B, C, H * W, row_lo * W, (row_hi - row_lo) * W, groups, _p(sums), _p(ws), _stream()), "mdtile_gn_sums")
    return sums


def gn_from_sums(sums: torch.Tensor, count: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(var, mean) from all-reduced fp64 sums; count = elements per (sample, group) over every rank."""
    _dev_tensor(sums, "sums", torch.float64)
    BG = sums.shape[0]
    mean = torch.empty(BG, dtype=torch.float32, device=sums.device)
    var = torch.empty_like(mean)
    _check(lib().mdtile_gn_from_sums(_p(sums), float(count), BG, _p(mean), _p(var), _stream()), "mdtile_gn_from_sums")
    return var, mean


def gn_coeffs(mean: torch.Tensor, var: torch.Tensor, gamma, beta, C: int, groups: int = 32, eps: float = 1e-6) -> torch.Tensor:
    """[B, 2, C] per-channel (a, s) of a fixed-statistics GroupNorm: a = gamma / sqrt(var + eps), s = beta - mean * a --
    the operand of the fused pre-activation conv (`PackedConv.__call__(..., pre_gn=coef)`)."""
    _dev_tensor(mean, "mean", torch.float32)
    _dev_tensor(var, "var", torch.float32)
    B = mean.numel() // groups
    assert mean.numel() == B * groups and var.numel() == B * groups
    for nm, t in (("gamma", gamma), ("beta", beta)):
        if t is not None:
            _dev_tensor(t, nm, torch.float32)
            assert t.numel() == C
    coef = torch.empty((B, 2, C), dtype=torch.float32, device=mean.device)
    _check(lib().mdtile_gn_coeffs(_p(mean), _p(var), _p(gamma), _p(beta), B, C, groups, eps, _p(coef), _stream()), "mdtile_gn_coeffs")
    return coef


class RecImage:
    """Split-bf16 record image of an activation [B, C, H, W] (include/mdtile.h "Record-image conv path"): the form the record
    conv kernels read by DMA.  `data` is an opaque int32 buffer of mdtile_rec_size bytes."""

    __slots__ = ("data", "shape")

    def __init__(self, shape, device):
        B, C, H, W = (int(v) for v in shape)
        n = lib().mdtile_rec_size(B, C, H, W)
        if n == 0:
            raise MdtileError(f"no record image for shape {tuple(shape)} (C % 32 == 0 required)")
        self.shape = (B, C, H, W)
        self.data = torch.empty(n // 4, dtype=torch.int32, device=device)

    REC_COL0 = 7          # csrc/conv_rec_common.h: column of the left border record; pixel x sits at column x + 8

    @staticmethod
    def pitch(W: int) -> int:
        """records per row of a plane: W + 2 logical columns behind 7 columns of padding, rounded up to whole 128-byte lines"""
        return (int(W) + 2 + RecImage.REC_COL0 + 7) & ~7

    def records(self) -> torch.Tensor:
        """[B, 2 (hi | lo), C / 8, H + 2, W + 2, 4] int32 view of the LOGICAL image: border records included, the row padding (never
        written, never read) left out -- what two record images are compared on."""
        B, C, H, W = self.shape
        full = self.data.view(B, 2, C // 8, H + 2, RecImage.pitch(W), 4)
        return full[:, :, :, :, RecImage.REC_COL0:RecImage.REC_COL0 + W + 2, :]

    # the image is batch-major (rec[b][hl][C/8][H+2][pitch]): samples can be cut out of / stacked into it without touching the records
    def batch_slice(self, b0: int, b1: int) -> "RecImage":
        B, C, H, W = self.shape
        assert 0 <= b0 < b1 <= B
        per = self.data.numel() // B
        r = RecImage.__new__(RecImage)
        r.shape, r.data = (b1 - b0, C, H, W), self.data[b0 * per:b1 * per]
        return r

    @staticmethod
    def cat(recs: Sequence["RecImage"]) -> "RecImage":
        assert recs and all(r.shape[1:] == recs[0].shape[1:] for r in recs)
        if len(recs) == 1:
            return recs[0]
        r = RecImage.__new__(RecImage)
        r.shape = (sum(q.shape[0] for q in recs),) + tuple(recs[0].shape[1:])
        r.data = torch.cat([q.data for q in recs])
        return r

    def to_f32(self) -> torch.Tensor:
        B, C, H, W = self.shape
        out = torch.empty(self.shape, dtype=torch.float32, device=self.data.device)
        _check(lib().mdtile_rec_to_f32(_p(self.data), _p(out), B, C, H, W, _stream()), "mdtile_rec_to_f32")
        return out


def rec_from_f32(x: torch.Tensor, coef: Optional[torch.Tensor] = None) -> RecImage:
    """split(silu(a x + s)) with coef = gn_coeffs(...) [B, 2, C], or split(x) when coef is None."""
    _dev_tensor(x, "x", torch.float32)
    B, C, H, W = x.shape
    rec = RecImage(x.shape, x.device)
    if coef is not None:
        _dev_tensor(coef, "coef", torch.float32)
        assert tuple(coef.shape) == (B, 2, C)
    _check(lib().mdtile_rec_from_f32(_p(x), _p(coef), _p(rec.data), B, C, H, W, _stream()), "mdtile_rec_from_f32")
    return rec


def silu(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev_tensor(x, "x", torch.float32)
    out = torch.empty_like(x) if out is None else out
    _check(lib().mdtile_silu(_p(x), _p(out), x.numel(), _stream()), "mdtile_silu")
    return out


def tanh(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decoder.tanh_out (the queue's last task, scripts/tilevae.py:192-193)."""
    _dev_tensor(x, "x", torch.float32)
    out = torch.empty_like(x) if out is None else out
    _check(lib().mdtile_tanh(_p(x), _p(out), x.numel(), _stream()), "mdtile_tanh")
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev_tensor(a, "a", torch.float32)
    _dev_tensor(b, "b", torch.float32)
    assert a.shape == b.shape
    out = torch.empty_like(a) if out is None else out
    _check(lib().mdtile_add(_p(a), _p(b), _p(out), a.numel(), _stream()), "mdtile_add")
    return out


class PackedConv:
    """Weights of one nn.Conv2d (1x1 or 3x3, stride 1, 'same' padding) re-laid out once for the MFMA conv kernel."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        _dev_tensor(weight, "weight", torch.float32)
        self.cout, self.cin, kh, kw = weight.shape
        assert kh == kw and kh in (1, 3)
        self.ksize = kh
        n = lib().mdtile_conv_packed_size(self.cout, self.cin, self.ksize)
        self.packed = torch.empty(n, dtype=torch.float32, device=weight.device)
        _check(lib().mdtile_conv_pack(_p(weight), _p(self.packed), self.cout, self.cin, self.ksize, _stream()), "mdtile_conv_pack")
        self.bias = None if bias is None else _dev_tensor(bias.detach().contiguous(), "bias", torch.float32)
        # the record kernels fetch the bias of a whole 32-cout tile: pad narrow convs (conv_out) with zeros
        self.bias_rec = self.bias
        if self.bias is not None and self.cout % 32:
            self.bias_rec = torch.zeros((self.cout + 31) // 32 * 32, dtype=torch.float32, device=weight.device)
            self.bias_rec[:self.cout] = self.bias

    def down2(self, x: torch.Tensor) -> torch.Tensor:
        """ldm Downsample: conv3x3 stride 2 over pad(x, right 1, bottom 1) (encoder 'downsample' task)."""
        _dev_tensor(x, "x", torch.float32)
        B, cin, H, W = x.shape
        assert cin == self.cin and self.ksize == 3 and H >= 2 and W >= 2
        y = torch.empty((B, self.cout, (H - 2) // 2 + 1, (W - 2) // 2 + 1), dtype=torch.float32, device=x.device)
        _check(lib().mdtile_conv2d_down2(_p(x), _p(self.packed), _p(self.bias), _p(y), B, self.cin, self.cout, H, W, _stream()),
               "mdtile_conv2d_down2")
        return y

    def fuses_pre_gn(self, upsample2x: bool = False, token_major: bool = False, exact: bool = False) -> bool:
        """True when this conv has a kernel that applies GroupNorm + SiLU to its input on load (mdtile_conv2d_gn)."""
        flags = (CONV_UPSAMPLE2X if upsample2x else 0) | (CONV_EXACT_F32 if exact else 0)
        return bool(lib().mdtile_conv2d_gn_supported(self.cout, self.cin, self.ksize, flags, int(token_major)))

    def takes_rec(self, upsample2x: bool = False) -> bool:
        """True when the record-image kernels (mdtile_conv2d_rec) take this conv."""
        return bool(lib().mdtile_conv2d_rec_supported(self.cout, self.cin, self.ksize, CONV_UPSAMPLE2X if upsample2x else 0))

    def call_rec(self, x: RecImage, residual: Optional[torch.Tensor] = None, upsample2x: bool = False, want_f32: bool = True,
                 want_rec: bool = False, rec_coef: Optional[torch.Tensor] = None, window: Optional[Tuple[int, int, int, int]] = None,
                 family: int = 0):
        """y = conv(x_rec) + bias (+ residual) -> (fp32 NCHW or None, RecImage or None), always that pair (the statistics-leaving form is
        call_rec_stats).  The record output is
        split(silu(a y + s)) with rec_coef = gn_coeffs(...) of the NEXT norm, or split(y) when rec_coef is None.
        window = (y0, x0, h, w) in INPUT pixels (upsample2x only; y0, x0: ints, or one int per image): the conv of that window of x,
        outputs [B, cout, 2h, 2w] (mdtile_upconv2d_rec_window: live-window narrowing of a decoder tile)."""
        B, cin, H, W = x.shape
        assert cin == self.cin and (want_f32 or want_rec)
        if window is not None:
            assert upsample2x and residual is None, "a window is taken by the upsample conv only"
            y0, x0, h, w = window
            y0 = [int(y0)] * B if isinstance(y0, int) else [int(v) for v in y0]       # one origin for every image, or one per image
            x0 = [int(x0)] * B if isinstance(x0, int) else [int(v) for v in x0]
            assert len(y0) == len(x0) == B, f"{B} images but {len(y0)} / {len(x0)} window origins"
            h, w = int(h), int(w)
            y = torch.empty((B, self.cout, 2 * h, 2 * w), dtype=torch.float32, device=x.data.device) if want_f32 else None
            yr = RecImage((B, self.cout, 2 * h, 2 * w), x.data.device) if want_rec else None
            if rec_coef is not None:
                _dev_tensor(rec_coef, "rec_coef", torch.float32)
                assert want_rec and tuple(rec_coef.shape) == (B, 2, self.cout)
            _check(lib().mdtile_upconv2d_rec_window(_p(x.data), _p(self.packed), _p(self.bias_rec), _p(y), None if yr is None else _p(yr.data),
                                                    _p(rec_coef), B, self.cin, self.cout, H, W, (c_int * B)(*y0), (c_int * B)(*x0), h, w, int(family), _stream()),
                   "mdtile_upconv2d_rec_window")
            return y, yr
        if upsample2x:
            H, W = 2 * H, 2 * W
        y = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=x.data.device) if want_f32 else None
        yr = RecImage((B, self.cout, H, W), x.data.device) if want_rec else None
        if residual is not None:
            _dev_tensor(residual, "residual", torch.float32)
            assert tuple(residual.shape) == (B, self.cout, H, W)
        if rec_coef is not None:
            _dev_tensor(rec_coef, "rec_coef", torch.float32)
            assert want_rec and tuple(rec_coef.shape) == (B, 2, self.cout)
        _check(lib().mdtile_conv2d_rec(_p(x.data), _p(self.packed), _p(self.bias_rec), _p(residual), _p(y), None if yr is None else _p(yr.data),
                                       _p(rec_coef), B, self.cin, self.cout, H, W, (CONV_UPSAMPLE2X if upsample2x else 0) | int(family), _stream()),
               "mdtile_conv2d_rec")
        return y, yr

    def call_rec_stats(self, x: RecImage, residual: Optional[torch.Tensor] = None, upsample2x: bool = False, family: int = 0, groups: int = 32):
        """y = conv(x_rec) + bias (+ residual) as fp32 NCHW AND get_var_mean(y, groups) from the conv's own epilogue
        (mdtile_conv2d_rec_stats; leaves_stats(groups, upsample2x, rec=True) says where) -> (y, (var, mean)), always that pair."""
        B, cin, H, W = x.shape
        assert cin == self.cin
        if upsample2x:
            H, W = 2 * H, 2 * W
        y = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=x.data.device)
        if residual is not None:
            _dev_tensor(residual, "residual", torch.float32)
            assert tuple(residual.shape) == (B, self.cout, H, W)
        mean, var, ws = self._stats_buffers(B, H, W, int(groups), x.data.device)
        _check(lib().mdtile_conv2d_rec_stats(_p(x.data), _p(self.packed), _p(self.bias_rec), _p(residual), _p(y), B, self.cin, self.cout, H, W,
                                             (CONV_UPSAMPLE2X if upsample2x else 0) | int(family), int(groups), _p(mean), _p(var), _p(ws),
                                             _stream()), "mdtile_conv2d_rec_stats")
        return y, (var, mean)

    def call_stats(self, x: torch.Tensor, pre_gn: torch.Tensor, residual: Optional[torch.Tensor] = None, groups: int = 32):
        """y = conv(silu(a * x + s)) (+ residual) on the fp32 hand-over kernel AND get_var_mean(y, groups) from its epilogue
        (mdtile_conv2d_gn_stats; leaves_stats(groups) says where) -> (y, (var, mean)), always that pair."""
        _dev_tensor(x, "x", torch.float32)
        _dev_tensor(pre_gn, "pre_gn", torch.float32)
        B, cin, H, W = x.shape
        assert cin == self.cin and tuple(pre_gn.shape) == (B, 2, self.cin)
        y = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=x.device)
        if residual is not None:
            _dev_tensor(residual, "residual", torch.float32)
            assert residual.shape == y.shape
        mean, var, ws = self._stats_buffers(B, H, W, int(groups), x.device)
        _check(lib().mdtile_conv2d_gn_stats(_p(x), _p(pre_gn), _p(self.packed), _p(self.bias), _p(residual), _p(y), B, self.cin, self.cout,
                                            H, W, self.ksize, 0, int(groups), _p(mean), _p(var), _p(ws), _stream()), "mdtile_conv2d_gn_stats")
        return y, (var, mean)

    def leaves_stats(self, groups: int = 32, upsample2x: bool = False, rec: bool = False) -> bool:
        """True when this conv has a kernel whose epilogue also leaves the GroupNorm statistics of its output (slow mode: the producer of
        a pooled norm's input): call_stats(x, pre_gn, ...), or call_rec_stats(...) with rec=True."""
        if rec:
            return bool(lib().mdtile_conv2d_rec_stats_supported(self.cout, self.cin, self.ksize, CONV_UPSAMPLE2X if upsample2x else 0, int(groups)))
        return bool(not upsample2x and lib().mdtile_conv2d_gn_stats_supported(self.cout, self.cin, self.ksize, 0, int(groups)))

    def _stats_buffers(self, B: int, H: int, W: int, groups: int, device):
        """(mean, var) rows of one call -- fresh: the caller keeps them (GroupNormParam pools the rows of several tiles) -- and the partials
        workspace, which only lives inside the call (launches of one stream run in order): ONE buffer per (shape, device), up to ~10 MB."""
        mean = torch.empty(B * groups, dtype=torch.float32, device=device)
        key = (B, H, W, groups, str(device))
        cache = self.__dict__.setdefault("_stats_ws", {})
        if key not in cache:
            if len(cache) >= 8:
                cache.clear()
            cache[key] = torch.empty((lib().mdtile_conv_stats_ws_size(B, self.cout, H, W, groups) + 7) // 8, dtype=torch.float64, device=device)
        return mean, torch.empty_like(mean), cache[key]

    def __call__(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, upsample2x: bool = False,
                 token_major: bool = False, exact: bool = False, pre_gn: Optional[torch.Tensor] = None):
        """exact=True forces the exact-fp32 MFMA kernel; by default 3x3 convs with cin % 16 == 0 run on the split-bf16
        ("bf16x3") matrix-core kernel: fp32 accumulate, ~1e-5 relative to fp32.
        pre_gn = gn_coeffs(...) [B, 2, cin]: y = conv(silu(a * x + s)) -- GroupNorm + SiLU fused into the input staging
        (only where fuses_pre_gn() says so)."""
        _dev_tensor(x, "x", torch.float32)
        B, cin, H, W = x.shape
        assert cin == self.cin, f"conv expects {self.cin} input channels, got {cin}"
        if upsample2x:
            H, W = 2 * H, 2 * W
        shape = (B, H * W, self.cout) if token_major else (B, self.cout, H, W)
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
        if residual is not None:
            _dev_tensor(residual, "residual", torch.float32)
            assert residual.shape == y.shape
        if pre_gn is not None:
            _dev_tensor(pre_gn, "pre_gn", torch.float32)
            assert tuple(pre_gn.shape) == (B, 2, self.cin) and not token_major and not upsample2x
            _check(lib().mdtile_conv2d_gn(_p(x), _p(pre_gn), _p(self.packed), _p(self.bias), _p(residual), _p(y), B, self.cin, self.cout,
                                          H, W, self.ksize, CONV_EXACT_F32 if exact else 0, _stream()), "mdtile_conv2d_gn")
            return y
        _check(lib().mdtile_conv2d(_p(x), _p(self.packed), _p(self.bias), _p(residual), _p(y), B, self.cin, self.cout, H, W,
                                   self.ksize, (CONV_UPSAMPLE2X if upsample2x else 0) | (CONV_EXACT_F32 if exact else 0),
                                   int(token_major), _stream()), "mdtile_conv2d")
        return y


def vae_attn(q: torch.Tensor, k: torch.Tensor, v_tok: torch.Tensor, scale: float, exact: bool = False, v_channel_major: bool = False) -> torch.Tensor:
    """Single-head attention core (tile_utils/attn.py:55-70).  q,k: [B,C,T]; v_tok: [B,T,C] (or [B,C,T] with v_channel_major); returns
    [B,C,T].  Default: split-bf16 matrix-core flash kernel (fp32 accumulate / softmax, ~1e-5 relative); exact=True: fp32 MFMA."""
    _dev_tensor(q, "q", torch.float32)
    _dev_tensor(k, "k", torch.float32)
    _dev_tensor(v_tok, "v", torch.float32)
    B, C, T = q.shape
    assert k.shape == q.shape and tuple(v_tok.shape) == ((B, C, T) if v_channel_major else (B, T, C))
    out = torch.empty_like(q)
    ws_bytes = 0 if exact else lib().mdtile_vae_attn_ws_size(B, C, T)
    ws = torch.empty(max(1, (ws_bytes + 3) // 4), dtype=torch.float32, device=q.device)
    flags = (ATTN_EXACT_F32 if exact else 0) | (ATTN_V_CHANNEL_MAJOR if v_channel_major else 0)
    _check(lib().mdtile_vae_attn(_p(q), _p(k), _p(v_tok), _p(out), B, C, T, scale, flags, _p(ws), _stream()), "mdtile_vae_attn")
    return out


def v_channel_major_ok(C: int = 512, exact: bool = False) -> bool:
    """True when vae_attn(C channels) takes a channel-major v (the split-bf16 kernel; the exact-fp32 kernel wants it token-major).
    Asked of the library: the answer uses the very predicate mdtile_vae_attn dispatches on (precision mode, MDTILE_ATTN_MODE, C)."""
    return bool(lib().mdtile_vae_attn_takes_channel_major(int(C), ATTN_EXACT_F32 if exact else 0))


def vae_attn_qk(q: torch.Tensor, k: torch.Tensor, v_tok: torch.Tensor, scale: float) -> torch.Tensor:
    """Attention of a band of queries q [B,C,Tq] against keys k [B,C,Tk] / values v_tok [B,Tk,C]; returns [B,C,Tq]."""
    _dev_tensor(q, "q", torch.float32)
    _dev_tensor(k, "k", torch.float32)
    _dev_tensor(v_tok, "v", torch.float32)
    B, C, Tq = q.shape
    Tk = k.shape[2]
    assert k.shape[:2] == (B, C) and tuple(v_tok.shape) == (B, Tk, C)
    out = torch.empty_like(q)
    ws = torch.empty(max(1, (lib().mdtile_vae_attn_qk_ws_size(B, C, Tq, Tk) + 3) // 4), dtype=torch.float32, device=q.device)
    _check(lib().mdtile_vae_attn_qk(_p(q), _p(k), _p(v_tok), _p(out), B, C, Tq, Tk, scale, _p(ws), _stream()), "mdtile_vae_attn_qk")
    return out


def crop_store(tile: torch.Tensor, in_bbox, out_bbox, result: torch.Tensor, is_decoder: bool = True) -> None:
    """result[..., out_bbox] = crop_valid_region(tile) (tilevae.py:248-259, 630-632)."""
    _dev_tensor(tile, "tile", torch.float32)
    _dev_tensor(result, "result", torch.float32)
    N, C, th, tw = tile.shape
    assert result.shape[:2] == (N, C)
    ib, ob = (c_int * 4)(*in_bbox), (c_int * 4)(*out_bbox)
    _check(lib().mdtile_crop_store(_p(tile), N, C, th, tw, ib, ob, int(is_decoder), _p(result), result.shape[2], result.shape[3],
                                   _stream()), "mdtile_crop_store")


def vae_fast_input(z: torch.Tensor, tile_size: int) -> torch.Tensor:
    """Fast-mode estimator input (tilevae.py:545-559): downsample to <= tile_size, re-standardise, clamp."""
    _dev_tensor(z, "z", torch.float32)
    N, C, H, W = z.shape
    oh, ow = c_int(), c_int()
    _check(lib().mdtile_vae_fast_size(H, W, int(tile_size), ctypes.byref(oh), ctypes.byref(ow)), "mdtile_vae_fast_size")
    out = torch.empty((N, C, oh.value, ow.value), dtype=torch.float32, device=z.device)
    ws = torch.empty((lib().mdtile_vae_fast_ws_size(C) + 7) // 8, dtype=torch.float64, device=z.device)
    _check(lib().mdtile_vae_fast_input(_p(z), N, C, H, W, int(tile_size), _p(out), _p(ws), _stream()), "mdtile_vae_fast_input")
    return out


# ---- DemoFusion (tile_methods/demofusion.py:219-324) -----------------------------------------------------------------------
class WindowSet:
    """The jittered local windows of one DemoFusion phase on the device: origins (row-major over rows x cols), their nominal
    (un-jittered) grid and the jitter range J (every origin lies in [nominal, nominal + 2 J])."""

    def __init__(self, origins: Sequence[Tuple[int, int]], nomx: Sequence[int], nomy: Sequence[int], jitter: int, window: int, device):
        assert len(origins) == len(nomx) * len(nomy)
        self.rows, self.cols, self.jitter, self.window = len(nomy), len(nomx), int(jitter), int(window)
        self.origins = [(int(x), int(y)) for x, y in origins]
        self.xy = torch.tensor([v for o in self.origins for v in o], dtype=torch.int32, device=device)
        self.nomx = torch.tensor([int(v) for v in nomx], dtype=torch.int32, device=device)
        self.nomy = torch.tensor([int(v) for v in nomy], dtype=torch.int32, device=device)


def window_blend(tiles: torch.Tensor, ws: WindowSet, N: int, C: int, Hp: int, Wp: int) -> torch.Tensor:
    """Count-averaged sum of the window outputs (demofusion.py:244-257).  tiles [T*N, C, window, window], tile-major."""
    _dev_tensor(tiles, "tiles")
    assert tuple(tiles.shape) == (ws.rows * ws.cols * N, C, ws.window, ws.window)
    out = torch.empty((N, C, Hp, Wp), dtype=tiles.dtype, device=tiles.device)
    _check(lib().mdtile_window_blend(dtype_code(tiles.dtype), _p(tiles), _p(out), _p(ws.xy), _p(ws.nomx), _p(ws.nomy), ws.rows, ws.cols, ws.jitter,
                                     ws.window, N, C, Hp, Wp, _stream()), "mdtile_window_blend")
    return out


def dilated_gather(x: torch.Tensor, x_filtered: Optional[torch.Tensor], num_from_x: int, cells: Sequence[Tuple[int, int]], S: int, jitter: int,
                   h0: int, w0: int) -> torch.Tensor:
    """cat([src[:, :, by+J:Wp-J:S, bx+J:Wp-J:S] for (bx, by) in cells]) with src = x for the first num_from_x cells, x_filtered after."""
    _dev_tensor(x, "x")
    N, C, Hp, Wp = x.shape
    if x_filtered is not None:
        _dev_tensor(x_filtered, "x_filtered", x.dtype)
        assert x_filtered.shape == x.shape
    flat = (c_int * (2 * len(cells)))(*[int(v) for c in cells for v in c])
    out = torch.empty((len(cells) * N, C, h0, w0), dtype=x.dtype, device=x.device)
    _check(lib().mdtile_dilated_gather(dtype_code(x.dtype), _p(x), _p(x_filtered), int(num_from_x), _p(out), flat, len(cells), N, C, Hp, Wp, int(S),
                                       int(jitter), int(h0), int(w0), _stream()), "mdtile_dilated_gather")
    return out


def demofusion_combine(x_local: torch.Tensor, global_out: torch.Tensor, S: int, jitter: int, mixture: bool, c2: float) -> torch.Tensor:
    """x_local * (1 - c2) + scatter(global_out) * c2 (demofusion.py:284-322).  global_out [cells*N, C, h0, w0] in cell-list order."""
    _dev_tensor(x_local, "x_local")
    _dev_tensor(global_out, "global_out", x_local.dtype)
    N, C, Hp, Wp = x_local.shape
    h0, w0 = global_out.shape[2:]
    assert global_out.shape[0] == S * S * N * (2 if mixture else 1)
    out = torch.empty_like(x_local)
    _check(lib().mdtile_demofusion_combine(dtype_code(x_local.dtype), _p(x_local), _p(global_out), _p(out), N, C, Hp, Wp, int(S), int(jitter), h0, w0,
                                           int(bool(mixture)), float(c2), _stream()), "mdtile_demofusion_combine")
    return out


def depthwise_blur(x: torch.Tensor, kernel2d: torch.Tensor) -> torch.Tensor:
    """F.conv2d(x, kernel2d[None, None].repeat(C, 1, 1, 1), padding=K // 2, groups=C)  (demofusion.py:173-178)."""
    _dev_tensor(x, "x")
    _dev_tensor(kernel2d, "kernel2d", torch.float32)
    N, C, H, W = x.shape
    K = kernel2d.shape[-1]
    assert tuple(kernel2d.shape) == (K, K) and K % 2 == 1
    out = torch.empty_like(x)
    _check(lib().mdtile_depthwise_blur(dtype_code(x.dtype), _p(x), _p(kernel2d), _p(out), N * C, H, W, K, _stream()), "mdtile_depthwise_blur")
    return out


def moments(x: torch.Tensor) -> torch.Tensor:
    """fp64 [2] = (mean, unbiased std) of the whole tensor, on the device, without a host sync (two-stage fp64 sums of mdtile_gn_sums)."""
    xf = x if x.dtype == torch.float32 else x.float()
    n = xf.numel()
    sums = gn_sums(xf.contiguous().view(1, 1, 1, n), 0, 1, groups=1)[0]          # (sum, sum of squares)
    mean = sums[0] / n
    var = (sums[1] - sums[0] * sums[0] / n) / (n - 1)
    return torch.stack([mean, torch.sqrt(torch.clamp(var, min=0.0))])


def restandardize(x: torch.Tensor, stats4: torch.Tensor) -> torch.Tensor:
    """(x - stats4[0]) / stats4[1] * stats4[3] + stats4[2]; stats4 fp32 [4] on the device."""
    _dev_tensor(x, "x")
    _dev_tensor(stats4, "stats4", torch.float32)
    out = torch.empty_like(x)
    _check(lib().mdtile_restandardize(dtype_code(x.dtype), _p(x), _p(stats4), _p(out), x.numel(), _stream()), "mdtile_restandardize")
    return out


# ---- multi-GPU -------------------------------------------------------------------------------------------------------
class Shard:
    """Shard context of the C ABI (include/mdtile.h "Multi-GPU"): one RCCL communicator + one stream per local rank.
    Shard(dev_ids=[0, 1, ...])            single process, one rank per device (the form usable inside a webui process)
    Shard(nranks=N, rank=r, uid=bytes)    one rank of a process-per-GPU job; uid = Shard.unique_id() of rank 0, distributed by the host"""

    def __init__(self, dev_ids: Optional[Sequence[int]] = None, nranks: int = 0, rank: int = 0, uid: Optional[bytes] = None,
                 device: Optional[int] = None):
        L = lib()
        if dev_ids is not None:
            arr = (c_int * len(dev_ids))(*[int(d) for d in dev_ids])
            self._h = L.mdtile_shard_init(len(dev_ids), arr)
            self.devices = [int(d) for d in dev_ids]
        else:
            assert uid is not None and len(uid) == 128
            dev = torch.cuda.current_device() if device is None else int(device)
            self._uid = ctypes.create_string_buffer(bytes(uid), 128)
            self._h = L.mdtile_shard_init_rank(int(nranks), int(rank), self._uid, dev)
            self.devices = [dev]
        if not self._h:
            raise MdtileError("mdtile_shard_init: " + L.mdtile_last_error().decode(errors="replace"))
        info = (c_int * 4)()
        _check(L.mdtile_shard_info(self._h, info), "mdtile_shard_info")
        self.nranks, self.nlocal, self.first, self.rccl = info[0], info[1], info[2], bool(info[3])

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _check(lib().mdtile_shard_unique_id(buf), "mdtile_shard_unique_id")
        return buf.raw

    @staticmethod
    def probe(nranks: int, rank: int, uid: bytes, device: int, timeout_s: float = 60.0) -> None:
        """Interruptible bring-up probe (mdtile_shard_probe_rank): the rendezvous of a process-per-GPU communicator on a non-blocking
        communicator that is aborted on the deadline and thrown away on success.  Raises MdtileError (timeout included); never leaves
        the calling thread inside RCCL.  `uid` must be an id of its own (not the one the real communicator is built from)."""
        buf = ctypes.create_string_buffer(bytes(uid), 128)
        _check(lib().mdtile_shard_probe_rank(int(nranks), int(rank), buf, int(device), float(timeout_s)), "mdtile_shard_probe_rank")

    def _streams(self, streams):
        if streams is None:
            return None
        return (c_void_p * self.nlocal)(*[int(s) for s in streams])

    def stream(self, local_rank: int) -> int:
        return int(lib().mdtile_shard_stream(self._h, int(local_rank)) or 0)

    def halo_scratch(self, band_rows: Sequence[int], N: int, C: int, W: int) -> List[torch.Tensor]:
        """One scratch buffer per local rank for halo_exchange (send + receive copies of the shared slabs)."""
        arr = (c_int * len(band_rows))(*[int(v) for v in band_rows])
        out = []
        for i, dev in enumerate(self.devices):
            n = lib().mdtile_halo_scratch_bytes(self.nranks, self.first + i, arr, N, C, W)
            out.append(torch.empty(max(1, n // 4), dtype=torch.float32, device=torch.device("cuda", dev)))
        return out

    def halo_exchange(self, partials: Sequence[torch.Tensor], scratch: Sequence[torch.Tensor], band_rows: Sequence[int], streams=None) -> None:
        """In place: every partial canvas ([N,C,H,W] fp32, one per local rank) becomes complete on the rows its band touches."""
        assert len(partials) == self.nlocal == len(scratch)
        N, C, H, W = partials[0].shape
        for t in partials:
            _dev_tensor(t, "partial", torch.float32)
        pp = (c_void_p * self.nlocal)(*[t.data_ptr() for t in partials])
        ss = (c_void_p * self.nlocal)(*[t.data_ptr() for t in scratch])
        arr = (c_int * len(band_rows))(*[int(v) for v in band_rows])
        _check(lib().mdtile_halo_exchange(self._h, pp, ss, N, C, H, W, arr, self._streams(streams)), "mdtile_halo_exchange")

    def allreduce_stats(self, bufs: Sequence[torch.Tensor], streams=None) -> None:
        for t in bufs:
            _dev_tensor(t, "buf", torch.float64)
        pp = (c_void_p * self.nlocal)(*[t.data_ptr() for t in bufs])
        _check(lib().mdtile_allreduce_stats(self._h, pp, bufs[0].numel(), self._streams(streams)), "mdtile_allreduce_stats")

    def bcast(self, bufs: Sequence[torch.Tensor], root: int, streams=None) -> None:
        pp = (c_void_p * self.nlocal)(*[t.data_ptr() for t in bufs])
        _check(lib().mdtile_shard_bcast(self._h, pp, bufs[0].numel() * bufs[0].element_size(), int(root), self._streams(streams)), "mdtile_shard_bcast")

    def p2p(self, ops: Sequence[Sequence[Tuple[int, Optional[torch.Tensor], Optional[torch.Tensor]]]], streams=None) -> None:
        """Grouped point-to-point: ops[i] = [(peer rank, tensor to send or None, tensor to receive into or None), ...] of local
        rank i; every tensor contiguous on that rank's device.  One ncclGroup for the whole call."""
        assert len(ops) == self.nlocal
        arrs, keep = [], []
        for lst in ops:
            a = (_P2P * max(1, len(lst)))()
            for k, (peer, snd, rcv) in enumerate(lst):
                for nm, t in (("send", snd), ("recv", rcv)):
                    if t is not None:
                        _dev_tensor(t, nm)
                keep.append((snd, rcv))
                a[k] = _P2P(int(peer), None if snd is None else snd.data_ptr(), 0 if snd is None else snd.numel() * snd.element_size(),
                            None if rcv is None else rcv.data_ptr(), 0 if rcv is None else rcv.numel() * rcv.element_size())
            arrs.append(a)
        pp = (POINTER(_P2P) * self.nlocal)(*[ctypes.cast(a, POINTER(_P2P)) for a in arrs])
        cnt = (c_int * self.nlocal)(*[len(lst) for lst in ops])
        _check(lib().mdtile_shard_p2p(self._h, pp, cnt, self._streams(streams)), "mdtile_shard_p2p")

    def allgather(self, sends: Sequence[torch.Tensor], streams=None) -> List[torch.Tensor]:
        """Every rank contributes a tensor of one common shape; returns per local rank the [nranks, *shape] stack in rank order."""
        outs = []
        for t in sends:
            _dev_tensor(t, "send")
            outs.append(torch.empty((self.nranks,) + tuple(t.shape), dtype=t.dtype, device=t.device))
        ss = (c_void_p * self.nlocal)(*[t.data_ptr() for t in sends])
        rr = (c_void_p * self.nlocal)(*[t.data_ptr() for t in outs])
        _check(lib().mdtile_shard_allgather(self._h, ss, rr, sends[0].numel() * sends[0].element_size(), self._streams(streams)), "mdtile_shard_allgather")
        return outs

    def selfcheck(self, streams=None) -> None:
        """Bring-up check (all-reduce, broadcast, grouped ring send / receive, all-gather; verified on the host).  Raises on failure."""
        n = int(lib().mdtile_shard_selfcheck_bytes(self._h))
        bufs = [torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", d)) for d in self.devices]
        pp = (c_void_p * self.nlocal)(*[t.data_ptr() for t in bufs])
        _check(lib().mdtile_shard_selfcheck(self._h, pp, self._streams(streams)), "mdtile_shard_selfcheck")

    def destroy(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.mdtile_shard_destroy(h)

    def __del__(self):
        self.destroy()
