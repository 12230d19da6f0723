"""ctypes binding of libswapnet_hip.so (include/swapnet_hip.h).

The product path loads exactly one library: swapnet_amd/csrc/libswapnet_hip.so, built for
gfx950 by `__graft_entry__.build()` / `python -m swapnet_amd.build`.  There is no CPU
fallback: if the library is missing, or no HIP device is visible, loading / context creation
raises.  (`Lib(path)` accepts an explicit path so that the CI-only host simulator under
tests/hostsim can exercise the same binding; the package itself never passes one.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libswapnet_hip.so")

LOSS_NAMES = ("D", "D_real", "D_fake", "G", "G_gan", "G_ce", "G_l1", "G_content", "G_style", "D_gp")


class SwnHyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "d_lr", "weight_decay", "d_weight_decay", "b1", "b2",
                                         "lambda_gan", "lambda_ce", "lambda_l1", "lambda_content",
                                         "lambda_style")] + [("gan_mode", C.c_int), ("warp_mode_ce", C.c_int), ("grad_scale", C.c_float),
                                                                           ("d_b1", C.c_float), ("d_b2", C.c_float),
                                                                           ("gp_mode", C.c_int), ("lambda_gp", C.c_float)]


class SwapnetHipError(RuntimeError):
    pass


_vp, _fp, _i, _f = C.c_void_p, C.c_void_p, C.c_int, C.c_float     # device float* travel as void*

_PROTOS = {
    "swn_abi_version": ([], _i),
    "swn_is_device_build": ([], _i),
    "swn_ctx_create": ([_i, _vp, _i, C.c_size_t, C.POINTER(_vp)], _i),
    "swn_ctx_destroy": ([_vp], _i),
    "swn_ctx_sync": ([_vp], _i),
    "swn_ctx_set_overlap": ([_vp, _i], _i),
    "swn_ctx_set_patchgan_layers": ([_vp, _i], _i),
    "swn_ctx_bytes_allocated": ([_vp, C.POINTER(C.c_size_t)], _i),
    "swn_prof_enable": ([_i], _i),
    "swn_prof_reset": ([], _i),
    "swn_prof_report": ([C.c_char_p, _i], _i),
    "swn_probe_mfma": ([_vp, _i, _i, C.POINTER(C.c_float)], _i),
    "swn_route_trace": ([_i], _i),
    "swn_route_report": ([C.c_char_p, _i], _i),
    "swn_warp_model_create": ([_vp, _i, _i, _i, _i, _f, C.POINTER(_vp)], _i),
    "swn_texture_model_create": ([_vp, _i, _i, _i, _i, _i, C.POINTER(_vp)], _i),
    "swn_warp_model_create_ex": ([_vp, _i, _i, _i, _i, _f, _i, _i, C.POINTER(_vp)], _i),
    "swn_texture_model_create_ex": ([_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)], _i),
    "swn_model_create_shared": ([_vp, _i, _i, _i, C.POINTER(_vp)], _i),
    "swn_model_destroy": ([_vp], _i),
    "swn_model_set_hyper": ([_vp, C.POINTER(SwnHyper)], _i),
    "swn_model_param_count": ([_vp, _i, C.POINTER(_i)], _i),
    "swn_model_param_info": ([_vp, _i, _i, C.c_char_p, _i, C.POINTER(_i * 4), C.POINTER(_i)], _i),
    "swn_model_param_set": ([_vp, _i, _i, C.c_char_p, _fp], _i),
    "swn_model_param_get": ([_vp, _i, _i, C.c_char_p, _fp], _i),
    "swn_model_optim_step_get": ([_vp, _i, C.POINTER(_i)], _i),
    "swn_model_optim_step_set": ([_vp, _i, _i], _i),
    "swn_model_set_input": ([_vp, _i, _fp, _i, _i, _i, _i], _i),
    "swn_model_set_input_labels": ([_vp, _i, _vp, _i, _i, _i], _i),
    "swn_model_get_output": ([_vp, _i, _fp], _i),
    "swn_model_get_tap": ([_vp, _i, C.c_char_p, _fp, C.POINTER(_i * 4)], _i),
    "swn_model_get_tap_grad": ([_vp, _i, C.c_char_p, _fp, C.POINTER(_i * 4)], _i),
    "swn_model_dropout_sites": ([_vp, _i, C.POINTER(_i)], _i),
    "swn_model_act_sites": ([_vp, _i, C.POINTER(_i)], _i),
    "swn_model_act_pattern": ([_vp, _i, _i, _vp, C.POINTER(_i * 4), C.POINTER(_i)], _i),
    "swn_model_dropout_mask": ([_vp, _i, _i, C.c_uint64, _fp, C.POINTER(_i * 4), C.POINTER(_f)], _i),
    "swn_pipeline_create": ([_vp, _vp, C.POINTER(_vp)], _i),
    "swn_pipeline_destroy": ([_vp], _i),
    "swn_pipeline_run": ([_vp, _i, C.POINTER(_i)], _i),
    "swn_pipeline_labels": ([_vp, C.POINTER(_vp)], _i),
    "swn_model_set_style_context": ([_vp, _fp, _fp, _i, _i], _i),
    "swn_model_set_gp_random": ([_vp, _fp, _fp], _i),
    "swn_model_discriminate": ([_vp, _fp, _fp], _i),
    "swn_model_perceptual": ([_vp, _fp, _fp, _i, _fp, _f, _f, _fp], _i),
    "swn_model_forward": ([_vp, _i, C.c_uint64], _i),
    "swn_model_backward_D": ([_vp, _f, _f], _i),
    "swn_model_backward_G": ([_vp, _f], _i),
    "swn_model_backward_G_parts": ([_vp, C.POINTER(C.c_int)], _i),
    "swn_model_backward_G_part": ([_vp, _f, _i, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], _i),
    "swn_model_optimizer_step": ([_vp, _i], _i),
    "swn_model_optimizer_step_range": ([_vp, _i, C.c_size_t, C.c_size_t, _i], _i),
    "swn_model_step": ([_vp, C.POINTER(_f * 3), _i, C.c_uint64], _i),
    "swn_model_step_captured": ([_vp, C.POINTER(_f * 3), _i, C.c_uint64], _i),
    "swn_model_step_dp": ([_vp, C.POINTER(_f * 3), _i, C.c_uint64, _i], _i),
    "swn_ctx_attach_comm": ([_vp, _vp, _vp, _i], _i),
    "swn_model_get_losses": ([_vp, C.POINTER(_f), _i], _i),
    "swn_model_grad_arena": ([_vp, _i, C.POINTER(_vp), C.POINTER(C.c_size_t)], _i),
    "swn_model_weight_arena": ([_vp, _i, C.POINTER(_vp), C.POINTER(C.c_size_t)], _i),
    "swn_model_arena": ([_vp, _i, _i, C.POINTER(_vp), C.POINTER(C.c_size_t)], _i),
    "swn_op_roi_align": ([_vp, _fp, _i, _i, _i, _i, _fp, _i, _i, _i, _fp], _i),
    "swn_op_roi_align_indices": ([_vp, _fp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "swn_op_decode_labels": ([_vp, _fp, _i, _i, _i, _i, _vp], _i),
    "swn_op_argmax_labels": ([_vp, _fp, _i, _i, _i, _i, _vp], _i),
    "swn_op_labels_to_onehot": ([_vp, _vp, _i, _i, _i, _i, _fp], _i),
    "swn_op_conv": ([_vp, _i, _i, _i, _i, _fp, _i, _i, _i, _i, _fp, _i, _fp, _i, _fp], _i),
    "swn_op_instance_norm_act": ([_vp, _fp, _i, _i, _i, _i, _i, _fp], _i),
    "swn_op_instance_norm_act_bwd": ([_vp, _fp, _fp, _i, _i, _i, _i, _i, _fp], _i),
    "swn_op_affine_gather": ([_vp, _fp, _fp, _i, _i, _i, _i, _vp, _i], _i),
    "swn_op_gan_loss": ([_vp, _i, _fp, _i, _i, _i, _i, _f, _i, _f, _fp, _fp], _i),
    "swn_op_norm_act_bwd2": ([_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp], _i),
    "swn_op_norm_act_dropout": ([_vp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, C.c_uint64, _fp, _fp, _fp], _i),
    "swn_op_adamw": ([_vp, _fp, _fp, _fp, _fp, C.c_size_t, _f, _f, _f, _f, _f, _i], _i),
}


class Lib:
    """Loaded shared library + error translation.  Every export declared in
    include/swapnet_hip.h must resolve, otherwise loading fails."""

    def __init__(self, path=None):
        self.path = path or LIB_PATH
        if not os.path.exists(self.path):
            raise SwapnetHipError(
                f"{self.path} not found: build it with `python -m swapnet_amd.build` "
                "(hipcc --offload-arch=gfx950).  swapnet_amd has no CPU fallback.")
        # torch first: it ships its own libamdhip64 / libhsa-runtime64, and the library's NEEDED
        # libamdhip64.so.7 must resolve to that already-loaded copy.  Loaded the other way round the
        # process ends up with two HSA runtimes and the second to initialise sees no device.
        import torch  # noqa: F401
        self.dll = C.CDLL(self.path)
        self.dll.swn_last_error.restype = C.c_char_p
        self.dll.swn_last_error.argtypes = []
        for name, (args, res) in _PROTOS.items():
            fn = getattr(self.dll, name)          # AttributeError if a symbol is missing
            fn.argtypes = args
            fn.restype = res
        self.is_device = bool(self.dll.swn_is_device_build())

    def last_error(self):
        return (self.dll.swn_last_error() or b"").decode()

    def call(self, name, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            msg = self.last_error()
            # same exception types the reference raises (SURVEY.md 8(b) "Error conventions")
            if "not implemented" in msg or "not recognized" in msg:
                raise NotImplementedError(msg)
            if rc == 1:
                raise ValueError(msg)
            raise SwapnetHipError(msg)
        return rc

    @staticmethod
    def exported_symbols():
        return ["swn_last_error"] + list(_PROTOS)


_default = None


def lib():
    """The product library (HIP).  Raises if it has not been built."""
    global _default
    if _default is None:
        _default = Lib()
        if not _default.is_device:
            raise SwapnetHipError("libswapnet_hip.so is not a device build")
    return _default


def ptr(t):
    """Raw data pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())
