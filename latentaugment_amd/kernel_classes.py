"""The ONE map from kernel names to the launch-profiler classes (csrc/la_common.h: LA_PC_*), shared by bench.py (per-class HIP-event
brackets, `roofline.classes`) and the rocprofv3 post-processing in scripts/ (kernel trace and PMC passes), so that "launches of a
class" means the same dispatches in both.  A class is a set of kernels that the library brackets with ONE la_prof_open / la_prof_close
pair per kernel launch; kernels outside every bracket (style / demod tables, criteria, optimiser) map to None."""

# order = the C enum (la_prof_end_classes fills arrays in this order)
CLASSES = ['conv_halo', 'conv_flat', 'conv_splitk', 'conv_f32', 'operand_prep', 'fir', 'seam_bwd', 'torgb_fwd', 'bank']

CLASS_KERNELS = {
    'conv_halo': 'la_conv_bf16_halo_kernel (stride-1 3x3 layers >= 64^2, fp32 input read directly)',
    'conv_flat': 'la_conv_bf16_kernel<split=false> (transposed-conv phases and stride-2 backward of the up-sampling layers)',
    'conv_splitk': 'la_conv_bf16_kernel<split=true> + la_conv_splitk_finish_kernel (layers <= 32^2)',
    'conv_f32': 'la_conv_igemm_kernel (exact fp32 MFMA)',
    'operand_prep': 'la_presplit_t_kernel / la_plane_absmax_kernel / la_xscale_kernel / la_xscale_pmax_kernel (fp16 operand scale and pre-split copy)',
    'fir': 'la_fir4x4_* / la_upfirdn2d_kernel / la_imgrad_pyramid_kernel (upfirdn2d family, fwd with the layer epilogue, adjoint, image-gradient pyramid)',
    'seam_bwd': 'la_seam_bwd_kernel (bias_act backward + ToRGB backward + demod-gradient reductions)',
    'torgb_fwd': 'la_torgb_fwd*_kernel (1x1 modulated ToRGB + clamp + skip add)',
    'bank': 'la_bank_* (criteria bank scans)',
}

# (substring of the demangled kernel name, class); first match wins
_RULES = [
    ('la_xscale_bound', None),                      # once per pass, with the style tables: outside the brackets
    ('la_conv_bf16_halo_kernel', 'conv_halo'),
    ('la_conv_splitk_finish', 'conv_splitk'),
    ('la_conv_igemm', 'conv_f32'),
    ('la_presplit', 'operand_prep'), ('la_plane_absmax', 'operand_prep'), ('la_act_grad_pmax', 'operand_prep'), ('la_xscale', 'operand_prep'),
    ('la_fir4x4', 'fir'), ('la_upfirdn2d_kernel', 'fir'), ('la_imgrad_pyramid', 'fir'),
    ('la_seam_bwd', 'seam_bwd'),
    ('la_torgb_fwd', 'torgb_fwd'),
    ('la_bank', 'bank'),
]
# a bracket of these classes holds more than one dispatch; its "launch" is the dispatch of the named kernel
LAUNCH_KERNEL = {'conv_splitk': 'la_conv_bf16_kernel', 'operand_prep': 'la_presplit'}


def class_of(kernel_name):
    """Class of a (demangled) kernel name, or None."""
    if 'la_conv_bf16_kernel' in kernel_name:
        return 'conv_splitk' if ', true,' in kernel_name else 'conv_flat'
    for pat, c in _RULES:
        if pat in kernel_name:
            return c
    return None
