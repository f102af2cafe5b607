"""ctypes binding of libunivst_hip.so (include/univst.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.  torch is
used only as the owner of device memory and of the current HIP stream.
"""
import ctypes as C
import os
from typing import Optional

import torch

# UNIVST_LIB: another build of the same library (A/B measurements of kernel variants inside one process tree; tools/README.md)
_LIB_PATH = os.environ.get("UNIVST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libunivst_hip.so")
_lib = None


class UnetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("block_out_channels", C.c_int * 4),
                ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int), ("attention_heads", C.c_int * 4),
                ("norm_num_groups", C.c_int), ("norm_eps", C.c_float), ("flip_sin_to_cos", C.c_int),
                ("freq_shift", C.c_float)]


class VaeCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int), ("block_out_channels", C.c_int * 4),
                ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int)]


class PnP(C.Structure):
    _fields_ = [("registered", C.c_int), ("idx", C.c_int), ("eta1", C.c_float), ("eta2", C.c_float),
                ("alpha", C.c_float), ("gamma", C.c_float)]


class Sd3AttnWeights(C.Structure):
    _NAMES = ("to_q", "to_q_bias", "to_k", "to_k_bias", "to_v", "to_v_bias", "norm_q", "norm_k", "add_q", "add_q_bias", "add_k", "add_k_bias",
              "add_v", "add_v_bias", "norm_added_q", "norm_added_k", "to_out", "to_out_bias", "to_add_out", "to_add_out_bias")
    _fields_ = [(n, C.c_void_p) for n in _NAMES]


class Sd3GatedResidual(C.Structure):
    _fields_ = [("res_img", C.c_void_p), ("gate_img", C.c_void_p), ("res_txt", C.c_void_p), ("gate_txt", C.c_void_p), ("ld_gate_img", C.c_int64),
                ("ld_gate_txt", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int)
KVEXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64)

# name -> (restype, argtypes); kept in sync with include/univst.h (tests/test_abi.py checks both ways)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "univst_last_error": (C.c_char_p, []),
    "univst_abi_version": (_I, []),
    "univst_sd3_shift_window": (_I, [_I, C.c_double, C.c_double, C.POINTER(_I), C.POINTER(_F)]),
    "univst_unet_create": (_I, [C.POINTER(UnetCfg), C.POINTER(_P)]),
    "univst_unet_destroy": (_I, [_P]),
    "univst_unet_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_L), _I, _P]),
    "univst_unet_finalize": (_I, [_P, _P]),
    "univst_unet_reserve": (_I, [_P, _I, _I, _I, _I]),
    "univst_unet_forward": (_I, [_P, _P, _F, _P, _I, _I, _I, _I, _I, C.POINTER(PnP), _P, _P, _I, _P]),
    "univst_unet_set_comm": (_I, [_P, _I, _I, _P, _L, ALLREDUCE_FN, KVEXCHANGE_FN, _P]),
    "univst_unet_set_option": (_I, [_P, C.c_char_p, _I]),
    "univst_unet_query": (_I, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    "univst_comm_create": (_I, [_I, _I, _L, C.POINTER(_P)]),
    "univst_comm_handle_bytes": (_I, []),
    "univst_comm_export": (_I, [_P, _P]),
    "univst_comm_connect": (_I, [_P, _P]),
    "univst_comm_connect_local": (_I, [_P, C.POINTER(_P)]),
    "univst_comm_connect_emulated": (_I, [_P, C.c_double, C.c_double]),
    "univst_comm_query": (_I, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    "univst_comm_destroy": (_I, [_P]),
    "univst_comm_allreduce_f32": (_I, [_P, _P, _I, _P]),
    "univst_comm_status": (_I, [_P]),
    "univst_unet_set_comm_native": (_I, [_P, _P]),
    "univst_linear": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "univst_linear_ln": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P]),
    "univst_geglu_xres_permute": (_I, [_P, _P, _I, _I, _P]),
    "univst_frag_pack": (_I, [_P, _P, _I, _I, _P]),
    "univst_vae_create": (_I, [_P, C.POINTER(_P)]),
    "univst_vae_destroy": (_I, [_P]),
    "univst_vae_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64), _I, _P]),
    "univst_vae_finalize": (_I, [_P, _P]),
    "univst_vae_decode": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "univst_vae_encode": (_I, [_P, _P, _L, _I, _I, _P, _P]),
    "univst_attn2_fused_workspace_bytes": (_L, [_I, _I, _I]),
    "univst_attn12_fused": (_I, [_P, _L, _P, _P, _P, _L, _F, _P, _P, _P, _I, _P, _I, _I, _L, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    "univst_attn2_fused": (_I, [_P, _L, _P, _F, _P, _P, _P, _I, _P, _I, _I, _L, _P, _P, _P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    "univst_conv_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P]),
    "univst_conv_nhwc_tapinner": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P]),
    "univst_conv3x3_patch": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P]),
    "univst_groupnorm_fold_linear": (_I, [_P, _I, _L, _I, _I, _F, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "univst_linear_sets": (_I, [_P, _L, _P, _P, _I, _P, _L, _P, _L, _I, _I, _I, _P, _P]),
    "univst_groupnorm_workspace_bytes": (_L, [_L, _I, _I]),
    "univst_groupnorm_nhwc": (_I, [_P, _P, _I, _I, _L, _I, _I, _F, _P, _P, _I, _P, _P, _P]),
    "univst_layernorm": (_I, [_P, _P, _P, _P, _L, _I, _F, _P]),
    "univst_attention": (_I, [_P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "univst_attention_phase": (_I, [_P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "univst_sd3_joint_attention": (_I, [C.POINTER(Sd3AttnWeights), _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P,
                                        C.POINTER(Sd3GatedResidual), _P, _P]),
    "univst_sd3_adain_shift": (_I, [_P, _L, _I, _I, _I, _I, _F, _F, _F, _P, _P]),
    "univst_rmsnorm_heads": (_I, [_P, _L, _L, _I, _I, _P, _F, _P]),
    "univst_linear_gated": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P, _L, _I, _P]),
    "univst_adaln_modulate": (_I, [_P, _P, _P, _P, _L, _L, _L, _I, _F, _P, _P, _P, _P]),
    "univst_gate_residual": (_I, [_P, _P, _L, _P, _P, _L, _L, _I, _P]),
    "univst_activation": (_I, [_P, _P, _L, _I, _P]),
    "univst_timestep_embedding": (_I, [_P, _P, _I, _I, _I, _F, _F, _P]),
    "univst_sd3_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "univst_sd3_unpatchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "univst_axpbypcz": (_I, [_P, _P, _P, _P, _F, _F, _F, _L, _P]),
    "univst_attention_adain_shift": (_I, [_P, _L, _I, _I, _I, _F, _F, _F, _P, _P]),
    "univst_latent_adain": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "univst_latent_adain_stats": (_I, [_P, _P, _I, _I, _I, _P]),
    "univst_latent_adain_apply": (_I, [_P, _P, _P, _L, _P, _I, _I, _I, _P]),
    "univst_axpby": (_I, [_P, _P, _P, _F, _F, _L, _P]),
    "univst_mask_blend": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "univst_mask_resize": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "univst_maskprop_workspace_bytes": (_L, [_I, _I, _I]),
    "univst_maskprop_frame": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P]),
    "univst_maskprop_finalize": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "univst_warp_accumulate": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "univst_warp_window_key": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "univst_latent_window_smooth": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "univst_accumulate_u8": (_I, [_P, _P, _L, _P]),
    "univst_window_store": (_I, [_P, _F, _P, _L, _P]),
    "univst_debug_tr16": (_I, [_P, _P]),
    "univst_debug_delay_us": (_I, [C.c_double, _P]),
    "univst_profile_enable": (_I, [_I]),
    "univst_profile_symbols": (_I, [_I, C.c_char_p, _I]),
    "univst_profile_collect": (_I, [C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double), _I]),
    "univst_profile_collect_aux": (_I, [C.POINTER(C.c_double), _I]),
}


ABI_VERSION = 3      # include/univst.h UNIVST_ABI_VERSION


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build it with `python __graft_entry__.py build` (or `make`); "
                               "univst_amd has no CPU or eager-PyTorch fallback")
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.univst_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{_LIB_PATH} has ABI version {lib.univst_abi_version()}, this binding was written against {ABI_VERSION} "
                               "(include/univst.h UNIVST_ABI_VERSION): rebuild the library (`make`)")
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().univst_last_error().decode(errors="replace")
        raise RuntimeError(f"libunivst_hip {what} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f16(t: torch.Tensor, name="tensor"):
    if not (t.is_cuda and t.dtype == torch.float16 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous fp16 CUDA/HIP tensor, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t


# --------------------------------------------------------------------------- stand-alone operator wrappers
def linear(x, w, bias=None, residual=None, geglu=False, out=None):
    """y[M,N] = x[M,K] w[N,K]^T (+bias)(+residual); geglu: w/bias rows pre-interleaved (True / 1: [16 x | 16 gate] blocks; 2: the
    X-resident order for K = 320, see geglu_xres_permute), N/2 output columns."""
    _f16(x), _f16(w)
    M, K = x.shape
    N = w.shape[0]
    No = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, No, device=x.device, dtype=torch.float16)
    check(load().univst_linear(ptr(x), K, ptr(w), ptr(bias), ptr(residual), No, ptr(out), No, M, N, K, int(geglu),
                               stream_ptr()), "linear")
    return out


def linear_gated(x, w, bias=None, residual=None, act=None, gate=None, rows_per_gate=1, out=None):
    """y[M,N] = residual + gate[m // rows_per_gate] * act(x w^T + bias); act None or ACT_GELU_TANH; gate a [B, N] row view."""
    _f16(x), _f16(w)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float16)
    check(load().univst_linear_gated(ptr(x), K, ptr(w), ptr(bias), ptr(residual), N, ptr(out), N, M, N, K, -1 if act is None else act,
                                     ptr(gate), 0 if gate is None else _mod_ld(gate), rows_per_gate, stream_ptr()), "linear_gated")
    return out


def geglu_xres_permute(w):
    """GEGLU projection weight [N, K] (or bias [N]) from the checkpoint's [x rows | gate rows] order into the row order ``linear(...,
    geglu=2)`` expects (the X-resident kernel for K = 320)."""
    _f16(w)
    w = w.contiguous()
    out = torch.empty_like(w)
    check(load().univst_geglu_xres_permute(ptr(w), ptr(out), w.shape[0], 1 if w.dim() == 1 else w.shape[1], stream_ptr()), "geglu_xres_permute")
    return out


def frag_pack(w):
    """[N, K] weight -> MFMA A-operand order [N/16][K/32][64][8] (same numel)."""
    _f16(w)
    w = w.contiguous()
    out = torch.empty_like(w)
    check(load().univst_frag_pack(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "frag_pack")
    return out


def attn2_fused(x, wq_frag, kv, wo_frag, bias_o, rows_per_branch, heads, residual=None, ln=None, q_prescaled=False, stats_out=None, out=None,
                eps=1e-5):
    """The text cross-attention of a transformer block in one launch: x [M, C] (raw rows when ``ln = (stats [M, C/160, 2], wsum [C], lnb [C])``
    folds the LayerNorm), kv [B*T, 2C] text K | V rows -> to_out(attention) + bias + residual (default: x)."""
    _f16(x), _f16(wq_frag), _f16(kv), _f16(wo_frag)
    M, C_ = x.shape
    B = -(-M // rows_per_branch)
    T = kv.shape[0] // B
    residual = x if residual is None else residual
    out = torch.empty(M, C_, device=x.device, dtype=torch.float16) if out is None else out
    st, ws, lb = ln if ln is not None else (None, None, None)
    for t in (st, ws, lb, stats_out):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    wsb = torch.empty(max(int(load().univst_attn2_fused_workspace_bytes(B, heads, C_ // heads)), 16), device=x.device, dtype=torch.uint8)
    check(load().univst_attn2_fused(ptr(x), x.stride(0), ptr(st), eps, ptr(ws), ptr(lb), ptr(wq_frag), int(q_prescaled), ptr(kv), B, T,
                                    rows_per_branch, ptr(wo_frag), ptr(bias_o), ptr(residual), residual.stride(0), ptr(out), out.stride(0), M, C_,
                                    heads, ptr(stats_out), ptr(wsb), stream_ptr()), "attn2_fused")
    return out


def attn12_fused(attn_out, wp_frag, bias_p, residual_in, wq_frag, wsum, lnb, kv, wo_frag, bias_o, rows_per_branch, heads, q_prescaled=False, stats_out=None,
                 eps=1e-5):
    """attn1.to_out + residual -> LayerNorm -> text cross-attention -> to_out + residual in one launch (univst_attn12_fused)."""
    _f16(attn_out), _f16(wp_frag), _f16(residual_in), _f16(wq_frag), _f16(kv), _f16(wo_frag)
    M, C_ = attn_out.shape
    B = -(-M // rows_per_branch)
    T = kv.shape[0] // B
    out = torch.empty(M, C_, device=attn_out.device, dtype=torch.float16)
    wsb = torch.empty(max(int(load().univst_attn2_fused_workspace_bytes(B, heads, C_ // heads)), 16), device=attn_out.device, dtype=torch.uint8)
    check(load().univst_attn12_fused(ptr(attn_out), attn_out.stride(0), ptr(wp_frag), ptr(bias_p), ptr(residual_in), residual_in.stride(0), eps, ptr(wsum), ptr(lnb),
                                     ptr(wq_frag), int(q_prescaled), ptr(kv), B, T, rows_per_branch, ptr(wo_frag), ptr(bias_o), ptr(out), out.stride(0), M, C_, heads,
                                     ptr(stats_out), ptr(wsb), stream_ptr()), "attn12_fused")
    return out


def linear_ln(x, w, bias=None, residual=None, geglu=False, out=None, ln=None, stats_out=None, eps=1e-5):
    """``linear`` through the LayerNorm-fold entry: ``stats_out`` fp32 [M, N/160, 2] receives the row statistics of the output;
    ``ln = (stats [M, K/160, 2], wsum [N], lnb [N])`` folds LayerNorm(x) into this linear (w = gamma-scaled weight)."""
    _f16(x), _f16(w)
    M, K = x.shape
    N = w.shape[0]
    No = N // 2 if geglu else N
    if out is None:
        out = torch.empty(M, No, device=x.device, dtype=torch.float16)
    st, ws, lb = ln if ln is not None else (None, None, None)
    for t in (st, ws, lb, stats_out):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    check(load().univst_linear_ln(ptr(x), K, ptr(w), ptr(bias), ptr(residual), No, ptr(out), No, M, N, K, int(geglu), ptr(st), eps,
                                  ptr(ws), ptr(lb), ptr(stats_out), stream_ptr()), "linear_ln")
    return out


def conv_nhwc(x1, w, bias=None, x2=None, upsample=False, stride=1, rowbias=None, rows_per_rowbias=1, residual=None):
    """x1 [imgs,Hs,Ws,C1] (+ x2 [imgs,Hs,Ws,C2]) NHWC, w [Cout,taps,C1+C2] -> [imgs,Ho,Wo,Cout]."""
    _f16(x1), _f16(w)
    imgs, Hs, Ws, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cout, taps, _ = w.shape
    He, We = (Hs * 2, Ws * 2) if upsample else (Hs, Ws)
    Ho = (He - 1) // stride + 1 if taps == 9 else He
    Wo = (We - 1) // stride + 1 if taps == 9 else We
    out = torch.empty(imgs, Ho, Wo, Cout, device=x1.device, dtype=torch.float16)
    check(load().univst_conv_nhwc(ptr(x1), ptr(x2), C1, C2, imgs, Hs, Ws, int(upsample), stride, taps, ptr(w), ptr(bias),
                                  ptr(rowbias), rows_per_rowbias, ptr(residual), ptr(out), Cout, stream_ptr()), "conv_nhwc")
    return out


def conv_nhwc_tapinner(x1, w_ti, bias=None, x2=None, upsample=False, stride=1, rowbias=None, rows_per_rowbias=1, residual=None):
    """3x3 conv with the tap-inner weight layout [Cout, Cin/64, 9, 64]."""
    _f16(x1), _f16(w_ti)
    imgs, Hs, Ws, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cout = w_ti.shape[0]
    He, We = (Hs * 2, Ws * 2) if upsample else (Hs, Ws)
    Ho, Wo = (He - 1) // stride + 1, (We - 1) // stride + 1
    out = torch.empty(imgs, Ho, Wo, Cout, device=x1.device, dtype=torch.float16)
    check(load().univst_conv_nhwc_tapinner(ptr(x1), ptr(x2), C1, C2, imgs, Hs, Ws, int(upsample), stride, ptr(w_ti), ptr(bias),
                                           ptr(rowbias), rows_per_rowbias, ptr(residual), ptr(out), Cout, stream_ptr()), "conv_tapinner")
    return out


def conv3x3_patch(x1, w_t32, bias=None, x2=None, rowbias=None, rows_per_rowbias=1, residual=None, upsample=False):
    """3x3 / stride-1 conv through the LDS-patch kernel, weight layout [Cout, Cin/32, 9, 32]; raises when the problem is
    not eligible for that kernel (it never falls back silently)."""
    _f16(x1), _f16(w_t32)
    imgs, Hs, Ws, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[-1]
    Cout = w_t32.shape[0]
    He, We = (Hs * 2, Ws * 2) if upsample else (Hs, Ws)
    out = torch.empty(imgs, He, We, Cout, device=x1.device, dtype=torch.float16)
    check(load().univst_conv3x3_patch(ptr(x1), ptr(x2), C1, C2, imgs, Hs, Ws, int(upsample), ptr(w_t32), ptr(bias), ptr(rowbias), rows_per_rowbias,
                                      ptr(residual), ptr(out), Cout, stream_ptr()), "conv3x3_patch")
    return out


def groupnorm_fold_linear(x, gamma, beta, groups, eps, rows_per_stat, w, bias=None):
    """-> (W_sets [S, N, C] fp16, bias32 [S, N] fp32): the GroupNorm of x [rows, C] folded into the linear (w [N, C], bias) that consumes it."""
    _f16(x), _f16(w)
    rows, C_ = x.shape
    S, N = rows // rows_per_stat, w.shape[0]
    ws = torch.empty(max(1, load().univst_groupnorm_workspace_bytes(rows, rows_per_stat, groups) // 4), device=x.device, dtype=torch.float32)
    wsets = torch.empty(S, N, C_, device=x.device, dtype=torch.float16)
    b32 = torch.empty(S, N, device=x.device, dtype=torch.float32)
    check(load().univst_groupnorm_fold_linear(ptr(x), C_, rows, rows_per_stat, groups, eps, ptr(gamma), ptr(beta), ptr(w), ptr(bias), N, ptr(wsets), ptr(b32),
                                              ptr(ws), stream_ptr()), "groupnorm_fold_linear")
    return wsets, b32


def linear_sets(x, wsets, bias32, rows_per_set, residual=None, stats_out=None):
    _f16(x), _f16(wsets)
    M, K = x.shape
    N = wsets.shape[1]
    out = torch.empty(M, N, device=x.device, dtype=torch.float16)
    check(load().univst_linear_sets(ptr(x), K, ptr(wsets), ptr(bias32), rows_per_set, ptr(residual), N, ptr(out), N, M, N, K, ptr(stats_out), stream_ptr()), "linear_sets")
    return out


def groupnorm_nhwc(x1, gamma, beta, groups, eps, rows_per_stat, silu=False, x2=None):
    _f16(x1)
    C1 = x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    rows = x1.numel() // C1
    ws = torch.empty(max(1, load().univst_groupnorm_workspace_bytes(rows, rows_per_stat, groups) // 4), device=x1.device,
                     dtype=torch.float32)
    out = torch.empty(*x1.shape[:-1], C1 + C2, device=x1.device, dtype=torch.float16)
    check(load().univst_groupnorm_nhwc(ptr(x1), ptr(x2), C1, C2, rows, rows_per_stat, groups, eps, ptr(gamma), ptr(beta),
                                       int(silu), ptr(out), ptr(ws), stream_ptr()), "groupnorm")
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    _f16(x)
    C_ = x.shape[-1]
    out = torch.empty_like(x)
    check(load().univst_layernorm(ptr(x), ptr(out), ptr(gamma), ptr(beta), x.numel() // C_, C_, eps, stream_ptr()),
          "layernorm")
    return out


def attention(q, k, v, src_idx, heads, ldq=None, ldkv=None, Nq=None, Nkv=None, C_=None, src_cnt=None, src_logw=None, q_prescaled=False):
    """q [BF,Nq,C], k/v [S,Nkv,C] (S source blocks), src_idx int32 [BF,nsrc] -> [BF,Nq,C].
    q_prescaled: q already multiplied by log2(e)/sqrt(head_dim)."""
    BF, Nq_, Cq = q.shape
    C_ = C_ or Cq
    out = torch.empty(BF, Nq_, C_, device=q.device, dtype=torch.float16)
    check(load().univst_attention(ptr(q), ldq or q.stride(1), ptr(k), ptr(v), ldkv or k.stride(1), ptr(out), C_,
                                  ptr(src_idx), ptr(src_cnt), ptr(src_logw), src_idx.shape[1], BF, Nq_, Nkv or k.shape[1], heads, C_ // heads,
                                  int(bool(q_prescaled)), stream_ptr()), "attention")
    return out


def attention_phase(q, k, v, src_idx, src_cnt, heads, out=None, state_out=None, state_in=None, ldq=None, ldkv=None, Nq=None, Nkv=None, C_=None, src_logw=None,
                    q_prescaled=False):
    """one launch of a TWO-PHASE attention (include/univst.h univst_attention_phase): phase 1 returns (out, state) with state [BF, heads, Nq, 2] fp32;
    phase 2 takes them (state_in=, out=) and returns the merged rows in `out`."""
    BF, Nq_, Cq = q.shape[0], (Nq or q.shape[1]), q.shape[2]
    C_ = C_ or Cq
    if out is None:
        out = torch.empty(BF, Nq_, C_, device=q.device, dtype=torch.float16)
    first = state_in is None
    if first and state_out is None:
        state_out = torch.empty(BF, heads, Nq_, 2, device=q.device, dtype=torch.float32)
    check(load().univst_attention_phase(ptr(q), ldq or q.stride(1), ptr(k), ptr(v), ldkv or k.stride(1), ptr(out), C_,
                                        ptr(src_idx), ptr(src_cnt), ptr(src_logw), src_idx.shape[1], BF, Nq_, Nkv or k.shape[1], heads, C_ // heads,
                                        int(bool(q_prescaled)), ptr(state_out) if first else None, None if first else ptr(state_in), stream_ptr()), "attention_phase")
    return (out, state_out) if first else out


_SD3_KEYS = {"to_q": "to_q.weight", "to_q_bias": "to_q.bias", "to_k": "to_k.weight", "to_k_bias": "to_k.bias", "to_v": "to_v.weight",
             "to_v_bias": "to_v.bias", "norm_q": "norm_q.weight", "norm_k": "norm_k.weight", "add_q": "add_q_proj.weight",
             "add_q_bias": "add_q_proj.bias", "add_k": "add_k_proj.weight", "add_k_bias": "add_k_proj.bias", "add_v": "add_v_proj.weight",
             "add_v_bias": "add_v_proj.bias", "norm_added_q": "norm_added_q.weight", "norm_added_k": "norm_added_k.weight",
             "to_out": "to_out.0.weight", "to_out_bias": "to_out.0.bias", "to_add_out": "to_add_out.weight", "to_add_out_bias": "to_add_out.bias"}


class Sd3AttnParams(dict):
    """the fp16 device tensors behind a univst_sd3_attn_weights struct, keyed by its field names.  q | k | v (and the added
    q | k | v) weights and biases are kept as views of ONE [3C, Cin] / [3C] tensor each, so that the library runs one projection
    per stream (csrc/sd3.hip checks that the three pointers are consecutive).  Build once per attention module and pass it in
    place of the state dict (backbones/video_diffusion_sd3/pnp_utils.py caches it, keyed by the parameters' storage and version)."""

    def __init__(self, state, device):
        super().__init__()
        t = {k: state[v].detach().to(device=device, dtype=torch.float16).contiguous() for k, v in _SD3_KEYS.items() if state.get(v) is not None}
        for trio in (("to_q", "to_k", "to_v"), ("add_q", "add_k", "add_v")) if os.environ.get("UNIVST_SD3_FUSED_QKV", "1") != "0" else ():
            for suf in ("", "_bias"):
                names = [n + suf for n in trio]
                if all(n in t for n in names) and len({tuple(t[n].shape) for n in names}) == 1:
                    fused = torch.cat([t[n] for n in names]).contiguous()
                    rows = t[names[0]].shape[0]
                    for i, n in enumerate(names):
                        t[n] = fused[i * rows:(i + 1) * rows]
        self.update(t)


def sd3_joint_attention(params, hidden, enc, heads, clip_length=16, shift=False, idx=-1, eta1=0.0, eta2=0.6, rms_eps=1e-6, fuse=None, comm=None):
    """CrossFrameProcessor / AttentionShiftProcessor of the reference's SD3 plugin on the native kernels.  params: the state dict
    of diffusers' Attention module (to_q.weight, ..., to_add_out.bias; missing entries = absent).  hidden [B, N, Cin],
    enc [B, Nt, Cin] or None -> (img [B, N, Cin], txt [B, Nt, Cin]) or img alone.  clip_length 0: no cross-frame keys.
    fuse (optional): dict(res_img, gate_img[, res_txt, gate_txt]) — the block's gated residuals computed in the out-projections'
    epilogue: the returned tensors are then res + gate[:, None] * attention output.
    comm (optional): pointer of a connected univst_comm — the batch is this rank's clip_length frames of every branch (frame shard)."""
    _f16(hidden)
    B, N, Cin = hidden.shape
    keep = params if isinstance(params, Sd3AttnParams) else Sd3AttnParams(params, hidden.device)
    w = Sd3AttnWeights(**{k: (keep[k].data_ptr() if k in keep else None) for k in Sd3AttnWeights._NAMES})
    inner = keep["to_q"].shape[0]
    out_i = torch.empty_like(hidden)
    Nt = 0
    out_t = None
    if enc is not None:
        _f16(enc)
        Nt = enc.shape[1]
        out_t = torch.empty(B, Nt, Cin if "to_add_out" in keep else inner, device=hidden.device, dtype=torch.float16)
    gr = None
    if fuse is not None:
        gt = fuse.get("gate_txt")
        gr = Sd3GatedResidual(ptr(_f16(fuse["res_img"])), ptr(fuse["gate_img"]), ptr(fuse.get("res_txt")), ptr(gt), _mod_ld(fuse["gate_img"]),
                              0 if gt is None else _mod_ld(gt))
    # pnp_utils.py:183-186 in Python double, like the reference: in fp32 eta1 = 0.3 gives 0.3f * 50 = 15.000001 and idx 15 drops out
    active = bool(shift) and idx >= eta1 * 50 and idx <= eta2 * 50
    beta = ((0.9 - 0.1) / (eta1 * 50 - eta2 * 50) * (idx - eta2 * 50) + 0.1) if active else 0.0
    check(load().univst_sd3_joint_attention(C.byref(w), ptr(hidden), ptr(enc), B, N, Nt, Cin, heads, inner // heads, clip_length, int(active),
                                            float(beta), rms_eps, ptr(out_i), ptr(out_t), C.byref(gr) if gr is not None else None,
                                            comm, stream_ptr()), "sd3_joint_attention")
    return (out_i, out_t) if enc is not None else out_i


def sd3_adain_shift_(qkv, F, N, C_, heads, alpha, beta, gamma):
    """in place on the fused [3*F*N, 3C] buffer (SD3 plugin's AdaIN shift)."""
    _f16(qkv)
    ws = torch.empty(4 * F * 2 * C_ + 2 * F * 2 * heads, device=qkv.device, dtype=torch.float32)
    check(load().univst_sd3_adain_shift(ptr(qkv), qkv.shape[-1], F, N, C_, heads, alpha, beta, gamma, ptr(ws), stream_ptr()), "sd3_adain_shift")
    return qkv


def rmsnorm_heads_(x, heads, weight, eps=1e-6):
    """in place on [rows, heads*d] fp16."""
    _f16(x)
    rows, Cc = x.shape
    check(load().univst_rmsnorm_heads(ptr(x), x.stride(0), rows, heads, Cc // heads, ptr(weight), eps, stream_ptr()), "rmsnorm_heads")
    return x


def _mod_ld(*ts):
    """scale / shift / gate operands are [B, C] views into one [B, k*C] adaLN output: same row stride, unit column stride."""
    ld = ts[0].stride(0)
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float16):
            raise ValueError(f"modulation operand: expected an fp16 CUDA/HIP tensor, got {t.dtype} {t.device}")
        if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) != ld:
            raise ValueError("modulation operands must be [B, C] row views with a common row stride")
    return ld


def adaln_modulate(x, scale, shift, eps=1e-6, scale2=None, shift2=None):
    """x [B, N, C], scale / shift [B, C] (row views allowed) -> LayerNorm(x) * (1 + scale) + shift; with scale2 / shift2 also the
    second modulation of the same normalised rows (AdaLayerNormZeroX) -> (y, y2)."""
    _f16(x)
    B, N, Cc = x.shape
    if not x.is_contiguous():
        raise ValueError("adaln_modulate: x must be contiguous")
    two = scale2 is not None
    ld = _mod_ld(scale, shift, *([scale2, shift2] if two else []))
    out = torch.empty_like(x)
    out2 = torch.empty_like(x) if two else None
    check(load().univst_adaln_modulate(ptr(x), ptr(out), ptr(scale), ptr(shift), ld, B * N, N, Cc, eps, ptr(out2), ptr(scale2), ptr(shift2),
                                       stream_ptr()), "adaln_modulate")
    return (out, out2) if two else out


def gate_residual(x, gate, y, out=None):
    """x + gate[:, None] * y for x, y [B, N, C], gate [B, C] (row view allowed)."""
    _f16(x), _f16(y)
    B, N, Cc = x.shape
    if not (x.is_contiguous() and y.is_contiguous()):
        raise ValueError("gate_residual: x and y must be contiguous")
    out = torch.empty_like(x) if out is None else out
    check(load().univst_gate_residual(ptr(x), ptr(gate), _mod_ld(gate), ptr(y), ptr(out), B * N, N, Cc, stream_ptr()), "gate_residual")
    return out


ACT_SILU, ACT_GELU_TANH = 0, 1


def activation(x, act, out=None):
    _f16(x)
    if not x.is_contiguous():
        raise ValueError("activation: x must be contiguous")
    out = torch.empty_like(x) if out is None else out
    check(load().univst_activation(ptr(x), ptr(out), x.numel(), act, stream_ptr()), "activation")
    return out


def timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000.0):
    """t [B] (any float dtype, device) -> [B, dim] fp16 sinusoidal embedding (diffusers Timesteps)."""
    t = t.to(torch.float32).contiguous()
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=torch.float16)
    check(load().univst_timestep_embedding(ptr(t), ptr(out), t.shape[0], dim, int(flip_sin_to_cos), downscale_freq_shift, max_period, stream_ptr()),
          "timestep_embedding")
    return out


def sd3_patchify(latents, patch):
    """[B, C, H, W] -> [B*(H/p)*(W/p), C*p*p] (k order of the flattened PatchEmbed conv weight)."""
    _f16(latents)
    B, Cc, H, W = latents.shape
    rows = torch.empty(B * (H // patch) * (W // patch), Cc * patch * patch, device=latents.device, dtype=torch.float16)
    check(load().univst_sd3_patchify(ptr(latents.contiguous()), ptr(rows), B, Cc, H, W, patch, stream_ptr()), "sd3_patchify")
    return rows


def sd3_unpatchify(rows, B, Cc, H, W, patch):
    """[B*(H/p)*(W/p), p*p*C] -> [B, C, H, W]."""
    _f16(rows)
    lat = torch.empty(B, Cc, H, W, device=rows.device, dtype=torch.float16)
    check(load().univst_sd3_unpatchify(ptr(rows.contiguous()), ptr(lat), B, Cc, H, W, patch, stream_ptr()), "sd3_unpatchify")
    return lat


def axpbypcz(x, y, z, a, b, c, out=None):
    _f16(x), _f16(y), _f16(z)
    out = torch.empty_like(x) if out is None else out
    check(load().univst_axpbypcz(ptr(x), ptr(y), ptr(z), ptr(out), a, b, c, x.numel(), stream_ptr()), "axpbypcz")
    return out


def attention_adain_shift_(qkv, F, N, C_, alpha, beta, gamma, return_stats=False):
    """in place on the fused [3*F*N, 3C] buffer.  return_stats: also the style branch's per-(frame, channel) statistics the kernel
    used, (mean, unbiased std) each [F, 2C] (K columns then V columns)."""
    _f16(qkv)
    ws = torch.empty(4 * F * 2 * C_, device=qkv.device, dtype=torch.float32)
    check(load().univst_attention_adain_shift(ptr(qkv), qkv.shape[-1], F, N, C_, alpha, beta, gamma, ptr(ws), stream_ptr()),
          "attention_adain_shift")
    if return_stats:
        return qkv, ws[:F * 2 * C_].view(F, 2 * C_), ws[F * 2 * C_:2 * F * 2 * C_].view(F, 2 * C_)
    return qkv


def latent_adain(cnt, sty, out=None):
    _f16(cnt), _f16(sty)
    _, Cl, F, H, W = cnt.shape
    out = torch.empty_like(cnt) if out is None else out
    check(load().univst_latent_adain(ptr(cnt), ptr(sty), ptr(out), Cl, F, H * W, stream_ptr()), "latent_adain")
    return out


def axpby(x, e, cx, ce, out=None):
    _f16(x), _f16(e)
    out = torch.empty_like(x) if out is None else out
    check(load().univst_axpby(ptr(x), ptr(e), ptr(out), cx, ce, x.numel(), stream_ptr()), "axpby")
    return out


def mask_blend(a, b, m, out=None):
    """(1-m)*a + m*b with a,b [1,C,F,h,w], m [F,h,w] fp16 or None."""
    _f16(a), _f16(b)
    out = torch.empty_like(a) if out is None else out
    Cl = a.shape[1]
    check(load().univst_mask_blend(ptr(a), ptr(b), ptr(m), ptr(out), Cl, a.numel() // Cl, stream_ptr()), "mask_blend")
    return out


def mask_resize(mask_u8, h, w):
    """uint8 {0,1} [F,H,W] -> fp16 [F,h,w] (torch bilinear, align_corners=False)."""
    F, H, W = mask_u8.shape
    out = torch.empty(F, h, w, device=mask_u8.device, dtype=torch.float16)
    check(load().univst_mask_resize(ptr(mask_u8), ptr(out), F, H, W, h, w, stream_ptr()), "mask_resize")
    return out


def debug_tr16():
    out = torch.empty(256, device="cuda", dtype=torch.float32)
    check(load().univst_debug_tr16(ptr(out), stream_ptr()), "debug_tr16")
    return out


def unet_query(handle, name: str) -> float:
    """read-out of a UNet handle (include/univst.h ``univst_unet_query``): "emu_wire_us", "arena_high_water"."""
    out = C.c_double(0.0)
    check(load().univst_unet_query(handle, name.encode(), C.byref(out)), f"unet_query({name})")
    return out.value


def delay_us(us: float):
    check(load().univst_debug_delay_us(float(us), stream_ptr()), "debug_delay_us")


PROFILE_CLASSES = ("gemm_big_kernel<0>", "gemm_big_kernel<1>", "gemm_kernel<*,0>", "gemm_kernel<*,1>", "attn_pp40_kernel<true>",
                   "attn_kernel_occ3<96,5,2>", "attn_kernel<other>", "groupnorm", "layernorm", "adain_shift", "attn_kernel<text>",
                   "conv_patch_kernel", "attn2_fused_kernel")


def profile_enable(on: bool):
    check(load().univst_profile_enable(int(on)), "profile_enable")


def profile_symbols():
    """-> {class: [kernel symbols launched since profile_enable(True)]}"""
    out = {}
    for i, name in enumerate(PROFILE_CLASSES):
        buf = C.create_string_buffer(2048)
        check(load().univst_profile_symbols(i, buf, 2048), "profile_symbols")
        out[name] = [s for s in buf.value.decode().split(";") if s]
    return out


def profile_collect():
    """-> {class: dict(ms=, launches=, flops=, bytes=, expanded_bytes=)} for everything launched since profile_enable(True)."""
    n = len(PROFILE_CLASSES)
    ms, cnt, fl, by = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)()
    check(load().univst_profile_collect(ms, cnt, fl, by, n), "profile_collect")
    aux = (C.c_double * n)()
    check(load().univst_profile_collect_aux(aux, n), "profile_collect_aux")
    return {PROFILE_CLASSES[i]: dict(ms=ms[i], launches=cnt[i], flops=fl[i], bytes=by[i], expanded_bytes=aux[i]) for i in range(n)}
