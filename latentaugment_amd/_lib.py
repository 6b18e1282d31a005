"""ctypes binding of liblatentaug_hip.so (the C ABI declared in include/latentaug_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing, or a call fails, this module
raises (`LatentAugHipError`) -- it never silently computes elsewhere.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblatentaug_hip.so')


class LatentAugHipError(RuntimeError):
    pass


class FeatOp(C.Structure):
    """Mirror of `la_feat_op`."""
    _fields_ = [('kind', C.c_int), ('cin', C.c_int), ('cout', C.c_int)]


class OptConfig(C.Structure):
    """Mirror of `la_opt_config` (include/latentaug_hip.h)."""
    _fields_ = [
        ('steps', C.c_int), ('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
        ('w_latent', C.c_float), ('w_pix', C.c_float), ('w_disc', C.c_float), ('w_lpips', C.c_float),
        ('criterion_mode', C.c_int), ('soft_aug', C.c_int), ('alpha', C.c_float),
        ('loop_noise_mode', C.c_int), ('final_noise_mode', C.c_int), ('norm_batch', C.c_int),
        ('crop', C.c_int), ('crop_off', C.c_int),
    ]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes); kept in one table so tests can check every header symbol is exported
SIGNATURES = {
    'la_last_error': (C.c_char_p, []),
    'la_abi_version': (_I, []),
    'la_bias_act_f32': (_I, [_P, _P, _P, _L, _L, _I, _I, _F, _F, _F, _P]),
    'la_bias_act_grad_f32': (_I, [_P, _P, _P, _P, _L, _L, _I, _I, _F, _F, _F, _P]),
    'la_bias_act_ex_f32': (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _F, _F, _F, _P]),
    'la_bias_sum_f32': (_I, [_P, _P, _L, _L, _I, _P]),
    'la_upfirdn2d_out_size': (_I, [_I] * 6),
    'la_upfirdn2d_f32': (_I, [_P, _P, _P] + [_I] * 15 + [_F, _P]),
    'la_pack_conv_weights_f32': (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    'la_modconv3x3_fwd_f32': (_I, [_P, _L, _P, _P, _I, _P, _I, _P, _I, _P, _L, _F, _P, _I, _F, _F, _F, _P, _P, _Z, _I, _I, _I, _I, _P]),
    'la_modconv3x3_up2_fwd_f32': (_I, [_P, _L, _P, _P, _I, _P, _I, _P, _I, _P, _L, _F, _P, _I, _F, _F, _F, _P, _P, _P, _P, _Z, _I,
                                       _I, _I, _I, _P]),
    'la_modconv3x3_bwd_f32': (_I, [_P, _P, _P, _I, _P, _I, _P, _L, _P, _P, _P, _Z, _I, _I, _I, _I, _P]),
    'la_modconv3x3_up2_bwd_f32': (_I, [_P, _P, _P, _I, _P, _I, _P, _L, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _P]),
    'la_modconv_ds_tiles': (_I, [_I]),
    'la_modconv_workspace_bytes': (_Z, [_I, _I, _I, _I, _I]),
    'la_modconv_bf16_pack_bytes': (_Z, [_I, _I, _I, _I]),
    'la_pack_conv_weights_bf16_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'la_pairwise_l2_workspace_floats': (_L, [_I, _L]),
    'la_pairwise_l2_f32': (_I, [_P, _I, _P, _L, _L, _P, _P, _P, _P]),
    'la_center_crop_f32': (_I, [_P, _P, _L, _I, _I, _I, _P]),
    'la_adam_step_f32': (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _P]),
    'la_noise_normal_f32': (_I, [_P, _L, _L, C.c_ulonglong, C.c_uint, _L, _P]),
    'la_synth_num_ws': (_I, [_I]),
    'la_synth_num_params': (_I, [_I]),
    'la_synth_workspace_bytes': (_Z, [_I, _I, _I, _P, _I]),
    'la_synth_create': (_I, [_I, _I, _I, _P, _F, _P, _I, _P, _I, _P, _I, _I, _I, _P, _Z, _P, _P]),
    'la_synth_destroy': (None, [_P]),
    'la_synth_set_precision': (_I, [_P, _I]),
    'la_synth_set_operand_scale': (_I, [_P, _I]),
    'la_synth_set_row_window': (_I, [_P, _I, _I]),
    'la_synth_set_col_window': (_I, [_P, _I, _I]),
    'la_synth_get_precision': (_I, [_P]),
    'la_synth_forward': (_I, [_P, _P, _L, _L, _I, _I, _P, _P, _P]),
    'la_synth_backward': (_I, [_P, _P, _P, _P]),
    'la_synth_image': (_P, [_P]),
    'la_synth_block_image': (_P, [_P, _I]),
    'la_synth_layer_output': (_P, [_P, _I]),
    'la_synth_styles': (_P, [_P]),
    'la_synth_style_grads': (_P, [_P]),
    'la_synth_style_rows': (_I, [_P]),
    'la_latent_opt_workspace_bytes': (_Z, [_I, _I, _I, _P, _L, _L, _I]),
    'la_latent_opt_create': (_I, [_P, _I, _I, _I, _P, _P, _L, _P, _L, _I, _P, _Z, _P]),
    'la_latent_opt_destroy': (None, [_P]),
    'la_latent_opt_run': (_I, [_P, _P, _I, _P, _P, _P, _P, _P]),
    'la_fc_f32': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _I, _F, _F, _P]),
    'la_mapping_forward_f32': (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _P, _F, _I, _P, _P, _P]),
    'la_feature_moments_f64': (_I, [_P, _L, _I, _P, _P, _P]),
    'la_pr_workspace_floats': (_Z, [_L, _L]),
    'la_cdist_f16': (_I, [_P, _L, _P, _L, _I, _P, _P, _P]),
    'la_pr_kth_f16': (_I, [_P, _L, _P, _L, _I, _I, _P, _P, _P]),
    'la_pr_member_f16': (_I, [_P, _L, _P, _L, _I, _P, _P, _P, _P]),
    'la_disc_num_params': (_I, [_I]),
    'la_disc_workspace_bytes': (_Z, [_I, _I, _P, _I]),
    'la_disc_create': (_I, [_I, _I, _P, _F, _P, _I, _P, _I, _I, _P, _Z, _P, _P]),
    'la_disc_destroy': (None, [_P]),
    'la_disc_set_precision': (_I, [_P, _I]),
    'la_disc_forward': (_I, [_P, _P, _I, _P]),
    'la_disc_loss': (_I, [_P, _F, _I, _P, _P]),
    'la_disc_backward': (_I, [_P, _P, _P, _I, _P]),
    'la_disc_logits': (_P, [_P]),
    'la_latent_opt_set_disc': (_I, [_P, _P]),
    'la_feat_workspace_bytes': (_Z, [_I, _P, _I, _I, _I]),
    'la_feat_create': (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _Z, _P, _P]),
    'la_feat_destroy': (None, [_P]),
    'la_feat_num_features': (_I, [_P]),
    'la_feat_set_precision': (_I, [_P, _I]),
    'la_feat_forward': (_I, [_P, _P, _I, _P, _P]),
    'la_feat_backward': (_I, [_P, _P, _P, _P]),
    'la_crop_repeat_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    'la_crop_repeat_grad_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    'la_latent_opt_lpips_workspace_bytes': (_Z, [_I, _I, _I, _L, _I]),
    'la_latent_opt_set_lpips': (_I, [_P, _P, _P, _L, _I, _F, _F, _P, _Z]),
    'la_latent_opt_set_crop_pos': (_I, [_P, _I, _I]),
    'la_latent_opt_set_graph': (_I, [_P, _I]),
    'la_latent_opt_graph_state': (_I, [_P]),
    'la_latent_opt_set_trace': (_I, [_P, _P, _P]),
    'la_latent_opt_set_grad_trace': (_I, [_P, _P]),
    'la_latent_opt_set_overlap': (_I, [_P, _I]),
    'la_latent_opt_set_row_window': (_I, [_P, _I, _I]),
    'la_latent_opt_set_col_window': (_I, [_P, _I, _I]),
    'la_latent_opt_set_time_trace': (_I, [_P, _I]),
    'la_latent_opt_get_times': (_I, [_P, _P]),
    'la_latent_opt_set_lpips_preproc': (_I, [_P, _P, _P, _I]),
    'la_latent_opt_invalidate_banks': (_I, [_P]),
    'la_prof_begin': (_I, []),
    'la_prof_end': (_I, [_P, _P, _P, _P]),
    'la_prof_set_stride': (_I, [_I]),
    'la_prof_total_launches': (_L, []),
    'la_prof_num_classes': (_I, []),
    'la_prof_end_classes': (_I, [_P, _P, _P, _P, _I]),
}

_lib = None
LOADED_PATH = None
DEV_LIB_PATH = os.path.join(_HERE, 'liblatentaug_hip_dev.so')      # `make -C latentaugment_amd/csrc dev`: measurement tools only
_use_dev = False


def select_dev_build():
    """Measurement tools (scripts/) and the one test that compares two internal code paths call this BEFORE the first load():
    the process then runs on the development build (kernel-variant knobs + LA_* environment switches).  The package never does."""
    global _use_dev
    if _lib is not None and not _use_dev:
        raise LatentAugHipError('select_dev_build() must come before the library is first used')
    _use_dev = True


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime: import it FIRST so that liblatentaug_hip.so binds to the runtime that owns torch's
    # streams and allocations (loading ours first leaves two runtimes in the process and HIP calls fail with
    # "no ROCm-capable device is detected")
    import torch  # noqa: F401
    path = DEV_LIB_PATH if _use_dev else LIB_PATH
    if not os.path.isfile(path):
        raise LatentAugHipError(
            f'{path} not found: build it with `make -C latentaugment_amd/csrc{" dev" if _use_dev else ""}` (or __graft_entry__.build()). '
            'There is no CPU fallback for the latent-augmentation hot path.')
    lib = C.CDLL(path)
    global LOADED_PATH
    LOADED_PATH = os.path.realpath(path)      # what was actually dlopen'ed (tests check it is the in-tree product build)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if _use_dev:
        lib.la_dev_knob_set.restype = _I
        lib.la_dev_knob_set.argtypes = [_I, _I]
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().la_last_error()
        raise LatentAugHipError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


def require_gpu(t):
    """Product tensors must live on the GPU: this path has no host implementation."""
    if not t.is_cuda:
        raise LatentAugHipError('latentaugment_amd needs a ROCm device tensor (no CPU fallback); got ' + str(t.device))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
