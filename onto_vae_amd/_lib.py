"""ctypes binding of libontovae_hip.so (the C ABI declared in include/ontovae_hip.h).

The library is built in-tree by onto_vae_amd/build.py (hipcc, gfx950).  There is no fallback:
if it is missing, loading raises, and every compute entry point of the package fails loudly.
"""
import ctypes as C
import os

# torch first: it carries its own libamdhip64; loading ours after it makes the dynamic linker reuse
# that runtime (same SONAME) instead of bringing a second HIP runtime into the process, which
# would see no device for streams and memory that torch created.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ONTOVAE_HIP_LIB") or os.path.join(HERE, "libontovae_hip.so")
ABI_VERSION = 8

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)


class Dims(C.Structure):
    _fields_ = [("G", C.c_int32), ("T", C.c_int32), ("K", C.c_int32), ("nu", C.c_int32), ("L", C.c_int32),
                ("C", C.c_int32), ("nnz", C.c_int64), ("nnz_links", C.c_int64), ("nnz_term", C.c_int64)]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("W0", "b0", "gamma", "beta", "Wh", "bh", "dec_vals", "dec_bias", "total")]


class Sell(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_slices", C.c_int32), ("n_chunks", C.c_int32), ("all_single_chunk", C.c_int32),
                ("level_first", c_i32p), ("level_count", c_i32p),
                ("node", C.c_void_p), ("info", C.c_void_p), ("chunk0", C.c_void_p), ("idx", C.c_void_p), ("k", C.c_void_p),
                ("w", C.c_void_p), ("bias", C.c_void_p)]


class GSell(C.Structure):
    _fields_ = [("n_slices", C.c_int32), ("n_quads", C.c_int32), ("gene", C.c_void_p), ("quad0", C.c_void_p), ("col", C.c_void_p),
                ("k", C.c_void_p), ("w", C.c_void_p)]


class Ctx(C.Structure):
    _fields_ = [("dims", Dims), ("lay", Layout),
                ("P", C.c_void_p), ("M", C.c_void_p), ("V", C.c_void_p), ("Gr", C.c_void_p),
                ("bn_rmean", C.c_void_p), ("bn_rvar", C.c_void_p), ("bn_nbt", C.c_void_p),
                ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("cscptr", C.c_void_p), ("csc_row", C.c_void_p),
                ("csc_k", C.c_void_p), ("csc_split", C.c_void_p), ("row_k", C.c_void_p),
                ("level_begin", c_i32p), ("level_end", c_i32p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("max_batch", C.c_int32), ("max_train_batch", C.c_int32),
                ("drop", C.c_float), ("z_drop", C.c_float), ("seed", C.c_uint64), ("flags", C.c_uint32),
                ("sell_fwd", Sell), ("sell_bwd", Sell), ("bf16_ws", C.c_void_p), ("bf16_ws_bytes", C.c_int64),
                ("gsell", GSell), ("gene_tiles", C.c_void_p), ("panel_tiles", C.c_void_p), ("term_tiles", C.c_void_p),
                ("tpanel_tiles", C.c_void_p), ("tpanel_max_entries", C.c_int32)]


VAE_MAXL = 8


class VaeDims(C.Structure):
    _fields_ = [("G", C.c_int32), ("Z", C.c_int32), ("n_enc", C.c_int32), ("n_dec", C.c_int32),
                ("enc", C.c_int32 * VAE_MAXL), ("dec", C.c_int32 * VAE_MAXL)]


class VaeLayout(C.Structure):
    _fields_ = [(n, C.c_int64 * VAE_MAXL) for n in ("enc_W", "enc_b", "enc_gamma", "enc_beta", "dec_W", "dec_b", "dec_gamma", "dec_beta",
                                                     "enc_bn", "dec_bn")] + \
               [(n, C.c_int64) for n in ("Wh", "bh", "Wout", "bout", "total", "bn_total")]


class VaeCtx(C.Structure):
    _fields_ = [("dims", VaeDims), ("P", C.c_void_p), ("M", C.c_void_p), ("V", C.c_void_p), ("Gr", C.c_void_p),
                ("bn_state", C.c_void_p), ("bn_nbt", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("max_batch", C.c_int32), ("drop", C.c_float), ("z_drop", C.c_float), ("seed", C.c_uint64)]


class AdamW(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("decay_mul", "one_m_beta1", "beta2", "one_m_beta2", "step_size", "bc2_sqrt", "eps")]


class OntoVAEHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library once; raise if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OntoVAEHipError(
            f"{LIB_PATH} not found: build it with `python -m onto_vae_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the OntoVAE hot path.")
    L = C.CDLL(LIB_PATH)
    L.ontovae_last_error.restype = C.c_char_p
    L.ontovae_abi_version.restype = C.c_int
    if L.ontovae_abi_version() != ABI_VERSION:
        raise OntoVAEHipError(f"libontovae_hip.so reports ABI {L.ontovae_abi_version()}, expected {ABI_VERSION}: rebuild")
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    sig = {
        "ontovae_plan_create": [i32, C.POINTER(c_u8p), c_i32p, c_i32p, i32, C.POINTER(vp)],
        "ontovae_plan_create_from_links": [i32, c_i32p, c_i32p, i32, i64, c_i32p, c_i32p, c_i32p, C.POINTER(vp)],
        "ontovae_plan_dims": [vp, C.POINTER(Dims)],
        "ontovae_plan_levels": [vp, c_i32p, c_i32p],
        "ontovae_plan_export": [vp, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p],
        "ontovae_plan_entry_coords": [vp, c_i32p, c_i32p, c_i32p],
        "ontovae_plan_csc_split": [vp, c_i32p],
        "ontovae_plan_row_k": [vp, c_i32p],
        "ontovae_plan_sell_sizes": [vp, i32, c_i32p, c_i32p, c_i32p, c_i32p],
        "ontovae_plan_sell_export": [vp, i32, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p],
        "ontovae_plan_tile_info": [vp, i32, c_i32p],
        "ontovae_plan_gsell_sizes": [vp, c_i32p, c_i32p],
        "ontovae_plan_gsell_export": [vp, c_i32p, c_i32p, C.POINTER(C.c_uint16), c_i32p],
        "ontovae_plan_layer_node0": [vp, i32, c_i32p, c_i32p],
        "ontovae_plan_unpack_mask": [vp, i32, c_u8p, i64],
        "ontovae_param_layout": [C.POINTER(Dims), C.POINTER(Layout)],
        "ontovae_forward": [C.POINTER(Ctx), vp, vp, i32, vp, u64, i32, vp, vp, vp, vp],
        "ontovae_train_step": [C.POINTER(Ctx), vp, vp, i32, vp, u64, f32, C.POINTER(AdamW), i32, vp, vp],
        "ontovae_train_step_dp": [C.POINTER(Ctx), i32, vp, vp, i32, i32, vp, u64, f32, vp, vp, vp, vp],
        "ontovae_adamw_apply": [C.POINTER(Ctx), C.POINTER(AdamW), vp],
        "ontovae_adamw_apply_range": [C.POINTER(Ctx), C.POINTER(AdamW), i64, i64, vp],
        "ontovae_dp_bn_bwd": [C.POINTER(Ctx), i32, i32, vp, u64, vp, i32, vp],
        "ontovae_dw0_rows": [C.POINTER(Ctx), vp, vp, i32, i32, i32, vp],
        "ontovae_dw0_global": [C.POINTER(Ctx), vp, vp, vp, i32, i32, vp, i64, C.POINTER(AdamW), i32, vp],
        "ontovae_dp_pack_dy": [C.POINTER(Ctx), i32, vp, i32, vp],
        "ontovae_dp_dh_global": [C.POINTER(Ctx), i32, i32, c_i32p, i32, vp, i64, vp],
        "ontovae_eval_step": [C.POINTER(Ctx), vp, vp, i32, vp, u64, f32, vp, vp],
        "ontovae_pass_data": [C.POINTER(Ctx), vp, vp, i32, vp, vp, u64, vp, i64, vp, i64, vp],
        "ontovae_decoder_forward": [C.POINTER(Ctx), vp, i32, vp, i64, vp, i64, vp],
        "ontovae_encode_pre": [C.POINTER(Ctx), vp, vp, i32, vp, i64, vp],
        "ontovae_pass_data_perturbed": [C.POINTER(Ctx), vp, vp, i32, vp, i64, i32, c_i32p, c_f32p, vp, vp, u64, vp, i64, i32, vp, i64, vp],
        "ontovae_compose_order": [vp, vp, vp, i64, vp],
        "ontovae_gather_rows": [vp, i64, vp, i32, i32, vp, vp],
        "ontovae_vals_to_dense": [vp, vp, vp, i64, i64, vp, i64, vp],
        "ontovae_dense_to_vals": [vp, i64, vp, vp, i64, i64, vp, vp],
        "ontovae_wilcoxon_zsplit": [vp, vp, i64, i32, i32, i32, vp, vp, vp, i64, vp],
        "ontovae_fdr_bh": [vp, i32, i32, vp, vp, i64, vp],
        "ontovae_fdr_bh_static": [vp, i32, i32, vp, vp],
        "ontovae_to_feature_major": [vp, i64, i32, i32, vp, i64, vp],
        "ontovae_vae_layout": [C.POINTER(VaeDims), C.POINTER(VaeLayout)],
        "ontovae_vae_train_step": [C.POINTER(VaeCtx), vp, vp, i32, vp, u64, f32, C.POINTER(AdamW), i32, vp, vp],
        "ontovae_vae_forward": [C.POINTER(VaeCtx), vp, vp, i32, vp, u64, i32, f32, vp, vp, vp, vp, vp, vp],
        "ontovae_vae_adamw_apply": [C.POINTER(VaeCtx), C.POINTER(AdamW), vp],
        "ontovae_vae_decode_forward": [C.POINTER(VaeCtx), vp, vp, vp, i32, vp, u64, i32, f32, vp, vp, vp, vp, vp, vp],
        "ontovae_vae_decode_step": [C.POINTER(VaeCtx), vp, vp, vp, i32, vp, u64, f32, C.POINTER(AdamW), i32, vp, vp, vp],
        "ontovae_decoder_backward": [C.POINTER(Ctx), vp, i32, u64, C.POINTER(AdamW), i32, vp, i64, vp],
        "ontovae_decoder_forward_train": [C.POINTER(Ctx), vp, i32, u64, vp, i64, vp],
        "ontovae_profile_begin": [C.c_uint32, i32],
        "ontovae_profile_read": [i32, c_f32p, i32, c_i32p],
        "ontovae_profile_end": [],
        "ontovae_profile_overhead": [vp, i32, c_f32p, c_f32p],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.ontovae_plan_destroy.argtypes = [vp]
    L.ontovae_plan_destroy.restype = None
    for name in ("ontovae_workspace_bytes", "ontovae_bf16_workspace_bytes"):
        getattr(L, name).argtypes = [C.POINTER(Dims), i32]
        getattr(L, name).restype = C.c_int64
    L.ontovae_workspace_bytes_split.argtypes = [C.POINTER(Dims), i32, i32]
    L.ontovae_workspace_bytes_split.restype = C.c_int64
    L.ontovae_vae_workspace_bytes.argtypes = [C.POINTER(VaeDims), i32]
    L.ontovae_vae_workspace_bytes.restype = C.c_int64
    for name in ("ontovae_wilcoxon_workspace_bytes", "ontovae_fdr_workspace_bytes"):
        getattr(L, name).argtypes = [i32, i32]
        getattr(L, name).restype = C.c_int64
    _lib = L
    return L


EXPORTED = ["ontovae_last_error", "ontovae_abi_version", "ontovae_plan_create", "ontovae_plan_create_from_links",
            "ontovae_plan_destroy", "ontovae_plan_dims", "ontovae_plan_levels", "ontovae_plan_export",
            "ontovae_plan_entry_coords", "ontovae_plan_csc_split", "ontovae_plan_row_k", "ontovae_plan_sell_sizes", "ontovae_plan_sell_export", "ontovae_plan_gsell_sizes", "ontovae_plan_gsell_export", "ontovae_plan_tile_info", "ontovae_plan_layer_node0", "ontovae_plan_unpack_mask", "ontovae_param_layout",
            "ontovae_workspace_bytes", "ontovae_workspace_bytes_split", "ontovae_bf16_workspace_bytes", "ontovae_forward", "ontovae_train_step", "ontovae_train_step_dp",
            "ontovae_adamw_apply", "ontovae_adamw_apply_range", "ontovae_dp_bn_bwd", "ontovae_dw0_rows", "ontovae_dw0_global", "ontovae_dp_pack_dy", "ontovae_dp_dh_global", "ontovae_eval_step", "ontovae_pass_data", "ontovae_decoder_forward",
            "ontovae_encode_pre", "ontovae_pass_data_perturbed",
            "ontovae_compose_order", "ontovae_gather_rows", "ontovae_vals_to_dense", "ontovae_dense_to_vals",
            "ontovae_wilcoxon_workspace_bytes", "ontovae_wilcoxon_zsplit", "ontovae_fdr_workspace_bytes", "ontovae_fdr_bh", "ontovae_fdr_bh_static",
            "ontovae_to_feature_major",
            "ontovae_vae_layout", "ontovae_vae_workspace_bytes", "ontovae_vae_train_step", "ontovae_vae_forward", "ontovae_vae_adamw_apply",
            "ontovae_vae_decode_forward", "ontovae_vae_decode_step", "ontovae_decoder_backward", "ontovae_decoder_forward_train",
            "ontovae_profile_begin", "ontovae_profile_read", "ontovae_profile_end", "ontovae_profile_overhead"]

FLAG_BF16_GEMM = 32
FLAG_PER_BLOCK_WGRAD = 64    # decoder weight-gradient partials per column block everywhere (no LDS accumulators, no level-kernel weight gradient)
FLAG_CB_GENE_FWD = 128       # training forward of the gene layer on 32-column blocks instead of the LDS-panel kernel (tests compare the two)

PHASES = ["enc_gemm", "bn_fwd", "heads", "reparam", "dec_fwd_terms", "dec_fwd_gene", "dec_bwd", "loss", "dec_wgrad",
          "latent_bwd", "da_gemm", "dwh", "bn_bwd", "dw0", "dec_bwd_gene", "sell_refresh"]
PHASE_PAIR_A, PHASE_PAIR_AB = 16, 17     # differential timing of dW0 (include/ontovae_hip.h)


def check(rc, what=""):
    """Turn a negative return code into a Python exception carrying the library's message."""
    if rc == 0:
        return
    msg = lib().ontovae_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    raise OntoVAEHipError(f"{what}: {msg} (rc={rc})")
