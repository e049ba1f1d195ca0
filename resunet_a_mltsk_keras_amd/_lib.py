"""ctypes binding of librua_hip.so (include/rua_hip.h).  There is NO fallback: if the library
is missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RUA_LIB_PATH") or os.path.join(HERE, "librua_hip.so")   # RUA_LIB_PATH: experiment builds only

RUA_F32, RUA_BF16 = 0, 1
RUA_MAX_SEG, RUA_MAX_BRANCH, RUA_MAX_WGRAD_GROUP = 6, 4, 8
LOSS_TANIMOTO, LOSS_WCE, LOSS_CE_LOGITS, LOSS_BCE_LOGITS, LOSS_MSE = 0, 1, 2, 3, 4
ACT_NONE, ACT_SOFTMAX, ACT_SIGMOID = 0, 1, 2

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


class ConvSeg(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("C", i32), ("Hs", i32), ("Ws", i32), ("up_shift", i32), ("dil", i32), ("taps", i32)]


class ConvDesc(C.Structure):
    _fields_ = [("seg", ConvSeg * RUA_MAX_SEG), ("nseg", i32), ("N", i32), ("H", i32), ("W", i32), ("Cout", i32),
                ("stride", i32), ("dtype", i32), ("bias", vp), ("aux", vp), ("aux_mode", i32), ("mscale", vp),
                ("mshift", vp), ("out_relu", i32), ("accumulate", i32), ("y", vp), ("out_stride", i32), ("OH", i32),
                ("OW", i32), ("stats", vp), ("stats_mode", i32), ("workspace", vp), ("workspace_bytes", i64), ("stats_replicas", i32),
                ("bias_more", vp * 3), ("in_scale", vp), ("in_shift", vp), ("in_relu", i32), ("pad_fold", i32), ("in_fold", vp)]


class BnFold(C.Structure):
    _fields_ = [("stats", vp), ("replicas", i32), ("pad", i32), ("count", f64), ("bessel_n", f64), ("eps", f32), ("momentum", f32),
                ("gamma", vp), ("beta", vp), ("moving_mean", vp), ("moving_var", vp), ("scale", vp), ("shift", vp), ("mean", vp), ("rstd", vp)]


class WgradDesc(C.Structure):
    _fields_ = [("a", vp), ("C", i32), ("Hs", i32), ("Ws", i32), ("dy", vp), ("Cout", i32), ("H", i32), ("W", i32),
                ("N", i32), ("stride", i32), ("dil", i32), ("taps", i32), ("dtype", i32), ("dw", vp), ("workspace", vp), ("workspace_bytes", i64),
                ("in_scale", vp), ("in_shift", vp), ("in_relu", i32), ("defer", i32), ("group_members", i32), ("pad_group", i32), ("overwrite_dev", vp)]


class WgradPending(C.Structure):
    _fields_ = [("kind", i32), ("parts", i32), ("n", i64), ("partials", vp), ("dw", vp), ("CC", i32), ("blocks", i32),
                ("block_begin", i32), ("pad", i32), ("overwrite_dev", vp)]


class TaniHead(C.Structure):
    _fields_ = [("sums", vp), ("replicas", i32), ("B", i32), ("C", i32), ("grad_scale", f32), ("loss_out", vp), ("coef", vp), ("per_sample", vp)]


class DzHead(C.Structure):
    _fields_ = [("kind", i32), ("act", i32), ("p", vp), ("y", vp), ("coef", vp), ("class_w", vp), ("grad_scale", f32), ("B", i32), ("HW", i64),
                ("C", i32), ("pad", i32), ("dz", vp)]


class BnBranch(C.Structure):
    _fields_ = [("gamma", vp), ("beta", vp), ("moving_mean", vp), ("moving_var", vp), ("scale", vp), ("shift", vp),
                ("mean", vp), ("rstd", vp), ("out", vp), ("stats", vp), ("replicas", i32), ("pad", i32), ("out_stats", vp)]


class BnFwdDesc(C.Structure):
    _fields_ = [("x", vp), ("M", i64), ("C", i32), ("dtype", i32), ("nb", i32), ("relu", i32), ("training", i32), ("replicas", i32),
                ("stats", vp), ("count", f64), ("bessel_n", f64), ("momentum", f32), ("eps", f32), ("br", BnBranch * RUA_MAX_BRANCH)]


class BnBwdBranch(C.Structure):
    _fields_ = [("g", vp), ("stats2", vp), ("replicas", i32), ("stats2_out", i32), ("gamma", vp), ("mean", vp), ("rstd", vp),
                ("scale", vp), ("shift", vp), ("dgamma", vp), ("dbeta", vp)]


class BnBwdDesc(C.Structure):
    _fields_ = [("x", vp), ("dskip", vp), ("dx", vp), ("M", i64), ("C", i32), ("dtype", i32), ("nb", i32), ("masked", i32),
                ("accumulate", i32), ("pad", i32), ("count", f64), ("br", BnBwdBranch * RUA_MAX_BRANCH),
                ("skip_stats", vp), ("skip_replicas", i32), ("pad2", i32), ("dx_stats", vp), ("dx_replicas", i32), ("pad3", i32)]


class WprepItem(C.Structure):
    _fields_ = [("src_off", i64), ("dst_off", i64), ("taps", i32), ("Cout", i32), ("C", i32), ("pad", i32)]


PP = C.POINTER(vp)

_SIGS = {
    "rua_version": ([], i32),
    "rua_device_info": ([C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32], i32),
    "rua_conv_fwd": ([C.POINTER(ConvDesc), vp], i32),
    "rua_conv_fwd_group": ([C.POINTER(ConvDesc), i32, vp], i32),
    "rua_conv_smem_bytes": ([C.POINTER(ConvDesc)], i32),
    "rua_conv_tile_bn": ([C.POINTER(ConvDesc)], i32),
    "rua_conv_tile_bm": ([C.POINTER(ConvDesc)], i32),
    "rua_conv_kernel_id": ([C.POINTER(ConvDesc)], i32),
    "rua_conv_fused_input_ok": ([C.POINTER(ConvDesc)], i32),
    "rua_conv_last_ksplit": ([], i32),
    "rua_conv_group_last_grids": ([], i32),
    "rua_conv_group_band_ok": ([C.POINTER(ConvDesc), i32], i32),
    "rua_conv_group_last_band": ([], i32),
    "rua_conv_group_last_chain": ([], i32),
    "rua_conv_fwd_sum": ([C.POINTER(ConvDesc), i32, vp], i32),
    "rua_conv_sum_last_kernel": ([], i32),
    "rua_conv_sum_kernel": ([C.POINTER(ConvDesc), i32], i32),
    "rua_conv_wgrad_group": ([C.POINTER(WgradDesc), i32, vp], i32),
    "rua_wgrad_group_last_grids": ([], i32),
    "rua_profile_mid_event": ([vp], None),
    "rua_profile_mid_event_fired": ([], i32),
    "rua_prof_event_create": ([], vp),
    "rua_prof_event_record": ([vp, vp], i32),
    "rua_prof_event_elapsed_us": ([vp, vp, C.POINTER(f64)], i32),
    "rua_prof_event_destroy": ([vp], None),
    "rua_graph_kernel_nodes": ([vp, C.POINTER(i32), C.POINTER(i32)], i32),
    "rua_conv_workspace_bytes": ([C.POINTER(ConvDesc)], i64),
    "rua_conv_wgrad": ([C.POINTER(WgradDesc), vp], i32),
    "rua_wgrad_workspace_bytes": ([C.POINTER(WgradDesc)], i64),
    "rua_wgrad_kind": ([C.POINTER(WgradDesc)], i32),
    "rua_wgrad_img_kind": ([C.POINTER(WgradDesc)], i32),
    "rua_wgrad_plan": ([C.POINTER(WgradDesc), C.POINTER(WgradPending)], i32),
    "rua_wgrad_reduce_batch": ([vp, i32, i32, vp], i32),
    "rua_weight_prep": ([vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "rua_weight_prep_dgrad": ([vp, vp, vp, i32, i32, vp, i32, i32, vp], i32),
    "rua_wprep_blocks": ([i32, i32, i32], i32),
    "rua_stem_fwd": ([vp, vp, vp, vp, i64, i32, i32, i32, vp], i32),
    "rua_stem_fwd_stats": ([vp, vp, vp, vp, i64, i32, i32, i32, vp, i32, vp], i32),
    "rua_stem_bwd": ([vp, vp, vp, vp, i64, i32, i32, i32, vp], i32),
    "rua_stem_fwd_pack": ([vp, vp, vp, vp, i64, i32, i32, i32, vp, i32, vp, vp], i32),
    "rua_stem_bwd_fold": ([vp, vp, vp, i32, i32, vp], i32),
    "rua_head_fwd": ([vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp], i32),
    "rua_head_fwd_loss": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, i32, i32, vp], i32),
    "rua_head_fwd_loss_rep": ([vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, i32, i32, i32, vp], i32),
    "rua_head_bwd": ([vp, vp, vp, vp, i32, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp], i32),
    "rua_head_bwd_sums": ([vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp], i32),
    "rua_col_stats": ([vp, i64, i32, vp, i32, i32, vp], i32),
    "rua_col_stats2": ([vp, vp, vp, vp, i32, i64, i32, vp, i32, i32, vp], i32),
    "rua_stats_replicas": ([i64], i32),
    "rua_bn_finalize": ([vp, i32, f64, f64, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, i32, vp], i32),
    "rua_bn_apply": ([vp, i32, PP, PP, i32, PP, i64, i32, i32, vp], i32),
    "rua_bn_bwd_finalize": ([vp, i32, f64, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp], i32),
    "rua_bn_bwd_apply": ([i32, PP, PP, PP, PP, PP, PP, i32, vp, vp, vp, i32, i64, i32, i32, vp], i32),
    "rua_stats_to_f32": ([vp, i32, i32, PP, i32, vp], i32),
    "rua_bn_fwd": ([C.POINTER(BnFwdDesc), vp], i32),
    "rua_bn_fwd_group": ([C.POINTER(BnFwdDesc), i32, vp], i32),
    "rua_bn_fwd_group_last_grids": ([], i32),
    "rua_bn_bwd": ([C.POINTER(BnBwdDesc), vp], i32),
    "rua_bn_bwd_group": ([C.POINTER(BnBwdDesc), i32, vp], i32),
    "rua_bn_bwd_group_last_grids": ([], i32),
    "rua_maxpool_fwd": ([vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "rua_maxpool_bwd": ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "rua_sumpool": ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "rua_maxpool_derive": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "rua_maxpool_bwd_multi": ([i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "rua_sumpool_pyramid": ([vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "rua_upsample_nearest": ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "rua_add_n": ([i32, PP, vp, i32, i64, i32, vp], i32),
    "rua_relu_mask": ([vp, vp, i64, i32, vp], i32),
    "rua_relu": ([vp, vp, i64, i32, vp], i32),
    "rua_cast_f32_to": ([vp, vp, i64, i32, vp], i32),
    "rua_cast_to_f32": ([vp, vp, i64, i32, vp], i32),
    "rua_fill_zero": ([vp, i64, vp], i32),
    "rua_tanimoto_sums": ([vp, vp, i32, i64, i32, vp, vp], i32),
    "rua_tanimoto_finalize": ([vp, i32, i64, i32, f32, vp, vp, vp, vp], i32),
    "rua_tanimoto_finalize_rep": ([vp, i32, i32, i64, i32, f32, vp, vp, vp, vp], i32),
    "rua_tanimoto_finalize_multi": ([vp, i32, vp], i32),
    "rua_head_dz_multi": ([vp, i32, vp], i32),
    "rua_tanimoto_ratio": ([vp, i32, i32, vp, vp, vp], i32),
    "rua_pixel_loss": ([i32, vp, vp, vp, vp, i64, i32, vp, vp, vp], i32),
    "rua_head_dz": ([i32, i32, vp, vp, vp, vp, f32, i32, i64, i32, vp, vp], i32),
    "rua_seg_metrics": ([vp, vp, i64, i32, vp, vp], i32),
    "rua_lr_step": ([vp, vp, i32, f64, f64, vp], i32),
    "rua_adam_step": ([vp, vp, vp, vp, i64, f32, vp, f32, f32, f32, f32, i32, vp], i32),
    "rua_sgd_step": ([vp, vp, vp, i64, f32, vp, f32, f32, i32, vp], i32),
    "rua_adam_step_w": ([vp, vp, vp, vp, i64, f32, vp, f32, f32, f32, f32, i32, vp, vp], i32),
    "rua_sgd_step_w": ([vp, vp, vp, i64, f32, vp, f32, f32, i32, vp, vp], i32),
    "rua_set_tuning": ([C.c_char_p, i64], i32),
    "rua_get_tuning": ([C.c_char_p, C.POINTER(i64)], i32),
    "rua_tuning_key": ([i32], C.c_char_p),
}

EXPORTED_SYMBOLS = sorted(list(_SIGS) + ["rua_last_error"])


class RuaError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuaError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -m resunet_a_mltsk_keras_amd.build` (needs hipcc). There is no CPU fallback.")
        self.dll = C.CDLL(LIB_PATH)
        self.dll.rua_last_error.restype = C.c_char_p
        self.dll.rua_last_error.argtypes = []
        for name, (args, res) in _SIGS.items():
            fn = getattr(self.dll, name)
            fn.argtypes = args
            fn.restype = res
            setattr(self, "_" + name, fn)

        # Experiment switches: the library itself never reads the environment (include/rua_hip.h, rua_set_tuning); this
        # host-side shim forwards RUA_TUNE_<KEY>=<int> variables once at load time so A/B runs need no code change.
        i = 0
        while True:
            key = self.dll.rua_tuning_key(i) if hasattr(self.dll, "rua_tuning_key") else None
            if not key:
                break
            v = os.environ.get("RUA_TUNE_" + key.decode().upper())
            if v is not None:
                self.check(self.dll.rua_set_tuning(key, int(v)), "rua_set_tuning")
            i += 1

    def set_tuning(self, **kv):
        for k, v in kv.items():
            self.check(self.dll.rua_set_tuning(k.encode(), int(v)), "rua_set_tuning")

    def get_tuning(self, key: str) -> int:
        v = i64()
        self.check(self.dll.rua_get_tuning(key.encode(), C.byref(v)), "rua_get_tuning")
        return int(v.value)

    def raw(self, name):
        return getattr(self, "_" + name)

    def check(self, rc, name):
        if rc != 0:
            raise RuaError(f"{name} failed ({rc}): {self.dll.rua_last_error().decode()}")

    def call(self, name, *args):
        self.check(getattr(self, "_" + name)(*args), name)


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr_array(ptrs):
    """ctypes void*[n] from a list of ints (kept alive by the caller)."""
    arr = (vp * len(ptrs))(*[vp(p) for p in ptrs])
    return arr
