"""ctypes binding of libsisr_hip.so (the C ABI in include/sisr_hip.h).

This is the only place the package touches native code.  There is no fallback: if the
library is missing, or a tensor is not on a HIP device, the call raises.  PyTorch is used
for device memory and streams only (``tensor.data_ptr()``, ``torch.cuda.current_stream()``).
"""
import collections
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_long, c_size_t, c_void_p

import torch  # noqa: F401  (must be imported first: libsisr_hip.so binds to torch's libamdhip64.so.7 by SONAME)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SISR_HIP_LIB") or os.path.join(_HERE, "libsisr_hip.so")  # override: A/B of two builds
_lib = None

P = c_void_p
_SIGS = {
    "sisr_pack_conv3x3": (c_int, [P, P, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_pack_conv3x3_both": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "sisr_conv3x3_c64_gap_parts": (c_int, [c_int, c_int]),
    "sisr_conv3x3_c64": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_int, c_int,
                                 c_int, c_int, c_int, P, c_int, P]),
    "sisr_ca_tail_bytes": (c_size_t, []),
    "sisr_wgrad3x3_c64_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "sisr_wgrad3x3_c64": (c_int, [P, P, P, P, P, P, c_float, P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P,
                                  c_int, c_int, P, c_size_t, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint64, P]),
    "sisr_wgrad_job_bytes": (c_size_t, []),
    "sisr_wgrad3x3_c64_batch_max": (c_int, []),
    "sisr_wgrad3x3_c64_batch_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sisr_wgrad3x3_c64_batch": (c_int, [P, c_int, P, P, P, c_size_t, c_int, c_int, c_int, P]),
    "sisr_conv3x3_cin3": (c_int, [P, P, c_int64, c_int64, c_int, P, P, P, c_int, c_int, c_int, c_int, P]),
    "sisr_conv3x3_cout3": (c_int, [P, P, P, c_int64, c_int64, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "sisr_corr3x3_c3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sisr_corr3x3_c3": (c_int, [P, P, P, c_float, P, c_int64, c_int64, c_int, c_int, P, P, c_size_t, c_int, c_int, c_int,
                                c_int, P]),
    "sisr_ca_gate_fwd": (c_int, [P, c_int, c_int, c_float, P, P, P, P, c_int, c_int, P, P, P, P, P, P]),
    "sisr_ca_gate_bwd_workspace_bytes": (c_size_t, [c_int]),
    "sisr_ca_gate_bwd": (c_int, [P, c_int, c_int, c_float, P, P, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "sisr_ca_gate_bwd_params_batch_max": (c_int, []),
    "sisr_ca_param_job_bytes": (c_size_t, []),
    "sisr_ca_gate_bwd_params_batch": (c_int, [P, c_int, c_int, c_int, P]),
    "sisr_meta_gate_fwd": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P, P]),
    "sisr_meta_gate_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sisr_meta_gate_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P, P]),
    "sisr_pa_fwd": (c_int, [P, P, P, P, P, P, c_long, c_int, c_int, P]),
    "sisr_pa_bwd_workspace_bytes": (c_size_t, [c_long]),
    "sisr_pa_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, P]),
    "sisr_gate_residual_fwd": (c_int, [P, P, P, P, P, c_int, c_long, c_int, P]),
    "sisr_gate_dg_parts": (c_int, [c_long]),
    "sisr_gate_dg_partial": (c_int, [P, P, P, c_int, c_long, c_int, P]),
    "sisr_sum_partials": (c_int, [P, c_int, c_int, c_int, c_float, P, P]),
    "sisr_l1_loss_workspace_bytes": (c_size_t, []),
    "sisr_l1_loss": (c_int, [P, P, c_long, P, P, P, P]),
    "sisr_crop_augment": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "sisr_adam_flat": (c_int, [P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_float, c_float, c_float, P]),
    "sisr_host_flags_alloc": (P, [c_int]),
    "sisr_host_flags_free": (None, [P]),
    "sisr_signal_host": (c_int, [P, P]),
    "sisr_stream_wait_flag": (c_int, [P, ctypes.c_uint, P]),
    "sisr_stream_spin_flag": (c_int, [P, ctypes.c_uint, P, P]),
}
OPTIONAL_SIGS = {  # only in libsisr_hip_diag.so (csrc/build.sh diag; select it with SISR_HIP_LIB)
    "sisr_diag_mfma_peak": (c_int, [c_int, c_int, P, P, P]),
    "sisr_diag_conv_occupancy": (c_int, [c_int]),
    "sisr_diag_conv_stamp": (None, [P]),
    "sisr_diag_mfma_fill": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P]),
    "sisr_diag_wgrad_stamp": (None, [P]),
}
_SIGS.update({
    "sisr_lam_workspace_bytes": (c_size_t, [c_int, c_int, c_long]),
    "sisr_lam_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_long, P]),
    "sisr_lam_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_long, P]),
    "sisr_csam_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "sisr_csam_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sisr_csam_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
})
_SIGS.update({  # bf16 matrix-core variants (csrc/conv3x3_mfma.hip, csrc/wgrad3x3_mfma.hip)
    "sisr_pack_conv3x3_bf16_both": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "sisr_conv3x3_c64_bf16": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_int, c_int,
                                 c_int, c_int, c_int, c_int, P]),
    "sisr_wgrad3x3_c64_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "sisr_wgrad3x3_c64_bf16": (c_int, [P, P, P, P, P, P, c_float, P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P,
                                  c_int, c_int, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
})
_SIGS.update({  # bf16 storage of the maps a residual group keeps (csrc/conv3x3_mfma.hip, wgrad3x3_mfma.hip, misc.hip)
    "sisr_conv3x3_c64_bf16s": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_int, c_int, c_int,
                                      c_int, P]),
    "sisr_wgrad3x3_c64_bf16s": (c_int, [P, P, P, P, P, P, c_float, P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P,
                                       c_int, c_int, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_f32_to_bf16": (c_int, [P, P, c_long, P]),
})
_SIGS.update({  # step-level launches (round 2): all conv weights / all meta gates of a network at once
    "sisr_pack_job_bytes": (c_size_t, []),
    "sisr_pack_conv3x3_many": (c_int, [P, c_int, c_int, c_int, P]),
    "sisr_meta_gate_many_fwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P, P]),
    "sisr_meta_gate_many_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sisr_meta_gate_many_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P]),
    "sisr_meta_gate_many_bwd_scatter": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P]),
})
_SIGS.update({  # generic gate MLP: the metadata-mixing QCALayer styles (csrc/attention.hip)
    "sisr_gate_mlp_desc_bytes": (c_size_t, []),
    "sisr_gate_mlp_fwd": (c_int, [P, P, P, c_int, P, P, P, P, P]),
    "sisr_gate_mlp_bwd": (c_int, [P, P, P, c_int, P, P, P, P, P, P, P, P, P, P]),
})
_SIGS.update({  # channel padding / RGB shuffle for the SRMD widening (csrc/misc.hip)
    "sisr_nchw_to_nhwc_pad": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_pad_oihw": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_shuffle_rgb": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_stack_maps": (c_int, [P, P, c_int, c_long, c_int, c_int, c_int, P]),
    "sisr_pixel_shuffle_cl": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
})
_SIGS.update({  # fp32 through the bf16 matrix cores: three-way operand split, six products (csrc/conv3x3_mfma.hip)
    "sisr_pack_conv3x3_x3_both": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "sisr_wgrad3x3_c64_x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "sisr_wgrad3x3_c64_x3": (c_int, [P, P, P, P, P, P, c_float, P, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, P,
                                c_int, c_int, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_conv3x3_c64_x3": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, P, P, c_float, c_int, P, P, P, P, c_int, c_int,
                               c_int, c_int, c_int, c_int, P]),
})
_SIGS.update({  # on-the-fly degradation (csrc/degrade.hip)
    "sisr_blur_quant": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "sisr_pil_resample": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_noise_quant": (c_int, [P, P, c_float, P, c_long, P]),
})
_SIGS.update({  # around the non-local attention (csrc/nonlocal.hip)
    "sisr_nl_project_fwd": (c_int, [P] * 8 + [c_long, P]),
    "sisr_nl_project_bwd_parts": (c_int, [c_long]),
    "sisr_nl_project_bwd": (c_int, [P] * 8 + [c_long, P]),
    "sisr_nl_split_pool_fwd": (c_int, [P, P, P, P, P, P]),
    "sisr_nl_split_pool_bwd": (c_int, [P, P, P, P, P, P, P]),
    "sisr_nl_output_fwd": (c_int, [P, P, P, P, P, P, P]),
    "sisr_nl_output_bwd_parts": (c_int, [P]),
    "sisr_nl_output_bwd": (c_int, [P, P, P, P, P, P, P]),
})
_SIGS.update({  # SPARNet pieces (csrc/sparnet.hip)
    "sisr_bn_workspace_bytes": (c_size_t, [c_long, c_int]),
    "sisr_nearest_up": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_conv3x3_c64_geo": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_wgrad3x3_c64_geo": (c_int, [P, P, P, P, P, c_int, c_int, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int,
                                      ctypes.c_uint64, P]),
    "sisr_wgrad_geo_job_bytes": (c_size_t, []),
    "sisr_wgrad3x3_c64_geo_batch_workspace_bytes": (c_size_t, [P, c_int]),
    "sisr_wgrad3x3_c64_geo_batch": (c_int, [P, c_int, P, c_size_t, P]),
    "sisr_group_norm_fwd": (c_int, [P, P, P, P, P, P, c_int, c_long, c_int, c_int, c_int, c_float, P]),
    "sisr_group_norm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_long, c_int, c_int, c_int, P]),
    "sisr_pixel_norm": (c_int, [P, P, P, c_long, c_int, c_int, P]),
    "sisr_act": (c_int, [P, P, P, P, P, c_long, c_int, c_int, c_int, c_int, P]),
    "sisr_spar3d": (c_int, [P, P, P, P, P, c_long, c_int, P]),
    "sisr_pad_reflect_up": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_crop_stride": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "sisr_bn_act_fwd": (c_int, [P] * 8 + [c_long, c_int, c_int, c_int, c_float, c_float, c_float, P, c_size_t, P]),
    "sisr_bn_act_bwd": (c_int, [P] * 9 + [c_long, c_int, c_int, c_float, P, c_size_t, P]),
    "sisr_spar_combine_fwd": (c_int, [P, P, P, P, P, c_long, c_int, c_int, P]),
    "sisr_spar_combine_bwd": (c_int, [P, P, P, P, P, c_long, c_int, c_int, P]),
})
_SIGS.update({  # SFTMD pieces (csrc/sft.hip)
    "sisr_sft_compose": (c_int, [P] * 12 + [c_int, c_int, P]),
    "sisr_sft_compose_record_bytes": (c_size_t, []),
    "sisr_sft_compose_many": (c_int, [P, c_int, P]),
    "sisr_sft_combine_fwd": (c_int, [P, c_long, P, P, P, c_long, c_long, c_int, P]),
    "sisr_sft_combine_bwd": (c_int, [P, c_long, P, c_long, P, P, P, c_long, c_int, P]),
    "sisr_map64": (c_int, [P, c_long, P, c_long, P, c_long, c_long, c_int, P]),
    "sisr_conv9_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "sisr_conv9_dgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "sisr_conv9_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sisr_conv9_wgrad": (c_int, [P, P, P, P, P, c_size_t, c_int, c_int, c_int, P]),
    "sisr_clamp01": (c_int, [P, P, P, c_long, c_int, P]),
})
_SIGS.update({  # SAN attention (csrc/san.hip)
    "sisr_covpool_workspace_bytes": (c_size_t, [c_int, c_long]),
    "sisr_covpool_fwd": (c_int, [P, P, P, P, c_int, c_long, c_int, P]),
    "sisr_sqrtm_saved_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sisr_sqrtm_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "sisr_sqrtm_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "sisr_soca_bwd_apply": (c_int, [P, P, P, P, P, P, c_int, c_long, c_int, P]),
    "sisr_nl_attn_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "sisr_nl_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
})


GM_MAXL = 4


class GateMlpDesc(ctypes.Structure):
    """Host mirror of sisr_gate_mlp (include/sisr_hip.h)."""
    _fields_ = [("w", c_void_p * GM_MAXL), ("b", c_void_p * GM_MAXL), ("nin", c_int * GM_MAXL), ("nout", c_int * GM_MAXL),
                ("cat", c_int * GM_MAXL), ("relu_in", c_int * GM_MAXL), ("act", c_int * GM_MAXL), ("L", c_int),
                ("M", c_int), ("C", c_int), ("final_mode", c_int)]


class CaParamJob(ctypes.Structure):
    """Host mirror of the records of sisr_ca_gate_bwd_params_batch (include/sisr_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("dz", "hid", "s", "dw1", "db1", "dw2", "db2")]


class WgradJob(ctypes.Structure):
    """Host mirror of sisr_wgrad_job (include/sisr_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("x", "dy", "dy_scale", "dy_shift", "dw", "dbias")]


class WgradGeoJob(ctypes.Structure):
    """Host mirror of sisr_wgrad_geo_job (include/sisr_hip.h)."""
    _fields_ = [(n, c_void_p) for n in ("x", "dy", "dw", "dbias")] + \
               [(n, c_int) for n in ("B", "H", "W", "cin", "cout", "up", "co_real", "ci_real")] + [("active_units", ctypes.c_uint64)]


class CaTail(ctypes.Structure):
    """Host mirror of sisr_ca_tail (include/sisr_hip.h): the channel-attention gate computed by the last-arriving
    workgroup of the conv launch that writes its partial sums."""
    _fields_ = [("backward", c_int), ("hidden", c_int), ("inv_hw", c_float)] + \
               [(n, c_void_p) for n in ("w1", "b1", "w2", "b2", "mul", "s", "hid", "ca", "s_out", "hid_out", "ca_out", "g_out",
                                        "shift", "dmul", "dw1", "db1", "dw2", "db2", "workspace", "counter", "head_part")] + \
               [("head_parts", c_int), ("head", c_int)]


_tail_counters = {}


def tail_counter(device, B):
    """Zero-initialised device words the tails count workgroups on (B per-sample words + one for the batch; every launch
    returns them to zero; one set per device and batch size: tails only run on the stream that drives the pass)."""
    key = (device.index, B)
    c = _tail_counters.get(key)
    if c is None:
        c = torch.zeros(B + 1, device=device, dtype=torch.int32)
        _tail_counters[key] = c
    return c.data_ptr()


class HipLibraryMissing(ImportError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the .so was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or csrc/build.sh).  This package has no CPU or PyTorch fallback.")
        L = ctypes.CDLL(LIB_PATH)
        lenient = "SISR_HIP_LIB" in os.environ  # an A/B or diagnostic build named explicitly may predate newer entry points
        for name, (res, args) in list(_SIGS.items()) + list(OPTIONAL_SIGS.items()):
            if (name in OPTIONAL_SIGS or lenient) and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGS) + sorted(k for k in OPTIONAL_SIGS if hasattr(lib(), k))


_ERR = {-1: "bad argument", -2: "misaligned pointer or stride (16 B required)", -4: "unsupported shape"}


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f"HIP launch failure (hipError {(-rc - 3) // 16})" if rc <= -3 else "error")
        raise RuntimeError(f"{what} failed: {msg} (rc={rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses non-HIP tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("sisr HIP kernels need tensors on a HIP device (cuda:N); got a CPU tensor. "
                           "There is no CPU fallback in this package.")
    if t.dtype != torch.float32:
        raise RuntimeError(f"sisr HIP kernels are fp32; got {t.dtype}")
    return t.data_ptr()


_counters = {}


def gate_counter(device):
    """The zero-initialised device word sisr_ca_gate_bwd counts its sample blocks on (returned to zero by every launch;
    one per device: the gate backward only ever runs on the stream that drives the backward pass)."""
    c = _counters.get(device.index)
    if c is None:
        c = torch.zeros(4, device=device, dtype=torch.int32)
        _counters[device.index] = c
    return c.data_ptr()


def ptr_any(t):
    """Device pointer of an fp32 OR bf16 map (None -> NULL): the bf16-storage entry points take both through float*."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"expected an fp32 or bf16 tensor on a HIP device; got {t.dtype} on {t.device}")
    return t.data_ptr()


def ptr_bf16(t):
    """Device pointer of a packed bf16 weight buffer."""
    if not t.is_cuda or t.dtype != torch.bfloat16:
        raise RuntimeError(f"expected a bf16 tensor on a HIP device; got {t.dtype} on {t.device}")
    return t.data_ptr()


_recent_copies = collections.deque(maxlen=32)


def ptr_c(t):
    """ptr(t.contiguous()) that is safe inside a call expression: if a copy had to be made it is kept referenced for
    the next few calls, so that another temporary of the same expression cannot be handed its block."""
    if t is None:
        return None
    c = t.contiguous()
    if c is not t:
        _recent_copies.append(c)
    return ptr(c)


def stream():
    return torch.cuda.current_stream().cuda_stream


_view_cache = {}
HUGE = 1 << 30


def view_plain(H, W, C, sB=None):
    """NHWC tensor with C (multiple of 64) channels; optional batch stride override (floats)."""
    key = ("p", H, W, C, sB)
    v = _view_cache.get(key)
    if v is None:
        v = (c_int64 * 6)(H * W * C if sB is None else sB, W * C, C, 0, 64, HUGE)
        _view_cache[key] = v
    return v


def view_pair(H, W, second_offset):
    """Two 64-channel NHWC tensors of identical layout read as ONE 128-channel map: chunk 0 at the base pointer, chunk 1
    `second_offset` floats away (a multiple of 4; the tensors may be separate allocations) -- torch.cat without the copy."""
    return (c_int64 * 6)(H * W * 64, W * 64, 64, second_offset, 0, 1)


def view_maps(H, W, nmaps, sB=None):
    """[B][nmaps][H][W][64] stack read as an NHWC tensor with nmaps*64 channels (chunk q = map q)."""
    key = ("m", H, W, nmaps, sB)
    v = _view_cache.get(key)
    if v is None:
        v = (c_int64 * 6)(nmaps * H * W * 64 if sB is None else sB, W * 64, 64, 0, H * W * 64, HUGE)
        _view_cache[key] = v
    return v


def view_shuffle(H, W, r):
    """The (B,H,W,64*r*r) result of a conv whose PixelShuffle(r) output is the [B][rH][rW][64] tensor."""
    key = ("s", H, W, r)
    v = _view_cache.get(key)
    if v is None:
        rH, rW = r * H, r * W
        v = (c_int64 * 6)(rH * rW * 64, r * rW * 64, r * 64, rW * 64, 64, r)
        _view_cache[key] = v
    return v


_workspaces = {}


def workspace(device, nbytes):
    """Grow-only per-device scratch (all kernels of a step run in order on one stream)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _workspaces[key] = buf
    return buf
