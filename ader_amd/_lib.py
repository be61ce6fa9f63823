"""ctypes binding of libader_hip.so (the C ABI declared in include/ader_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a launcher reports an error this
module raises.  (PyTorch is used only for device memory, streams and torch.distributed.)"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADER_HIP_LIB") or os.path.join(_HERE, "libader_hip.so")      # (override: A/B runs of two builds)
XLIB_PATH = os.path.join(_HERE, "libader_xcheck.so")

P, I, U, F, L, Z = c_void_p, c_int, c_uint, c_float, c_long, c_size_t
_DROP = [P]          # const AderDrop* (NULL: no dropout)

_SIGS = {
    "ader_embed_fwd": [P, P, P, P, I, I, I, I] + _DROP + [P, P],
    "ader_sq_accum": [P, P, Z, F, P],
    "ader_ewc_penalty": [P, P, P, P, Z, F, P, P, P],
    "ader_scatter_rows_ordered": [P, P, P, I, P, I, I, F, P, P],
    "ader_ln_fwd": [P, L, P, L, P, P, P, P, P, P, I, I, P],
    "ader_ln_bwd_slabs": [I],
    "ader_ln_bwd": [P, L, P, L, P, P, P, P, L, P, L, P, P, P, I, I, P],
    "ader_gemm_rows": [P, P, P, P, P, P, I, I, I, I, I, I] + _DROP + [P],
    "ader_lbf_sum": [P, I, P, P],
    "ader_gemm_atb_slabs": [I],
    "ader_gemm_atb": [P, P, P, P, P, I, I, P],
    "ader_wprep_elems": [I],
    "ader_wprep": [P, P, I, I, P, P],
    "ader_gemm_x3": [P, P, P, P, P, P, I, I, I, I, I, I] + _DROP + [P],
    "ader_gemm_atb_x3": [P, P, P, P, P, I, I, P],
    "ader_seq_fwd": [P, P],
    "ader_seq_bwd_ffn": [P, P],
    "ader_seq_bwd_qkv": [P, P],
    "ader_gemm_atb_batch_slabs": [P, I],
    "ader_gemm_atb_x3_batch": [P, P, P, P, P, I, P, I, P],
    "ader_mask_dropgrad": [P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_add_rows": [P, P, I, I, I, I, P],
    "ader_attn_last_fwd": [P, P, P, P, P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_attn_last_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_attn_fwd": [P, P, P, P, P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_attn_x3_fwd": [P, P, P, P, P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_attn_x3_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_attn_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I] + _DROP + [P],
    "ader_logits_sub": [I],
    "ader_logits_parts": [I],
    "ader_logits_ranges": [I, I],
    "ader_row_lse": [P, L, I, P, I, P, P],
    "ader_build_rowinfo": [P, I, P, P, I, I, I, F, F, I, P, P, P, P, P],
    "ader_logits_loss_fwd": [P, P, I, I, I, I, P, P, P, P, P, P, L, P, P, P, P, P],
    "ader_logits_bwd_drep": [P, P, I, I, I, I, P, P, P, P, P, P, L, P, P, P, P],
    "ader_logits_bwd_demb": [P, P, I, I, I, I, P, P, P, P, P, P, L, P, P, P],
    "ader_lbf_ranges": [I, I],
    "ader_lbf_shadow_refresh": [P, P, Z, I, P],
    "ader_lbf_fwd": [P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_lbf_prep": [P, P, I, I, I, P],
    "ader_lbf_fwd_kd": [P, P, I, I, I, I, I, I, I, I, P, P, P, L, P, F, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_lbf_ranges_kd": [I, I, I],
    "ader_lbf_readout_ranges": [I, I, I],
    "ader_lx3_readout_ranges": [I, I],
    "ader_lx3_fwd_kd": [P, P, I, I, I, I, I, I, I, I, P, P, P, L, P, F, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_tab_update_sh_kd": [P, P, I, I, I, I, I, I, P, P, P, P, I, P, F, P, P, P, I, P, P, L, P, P, P, P, P, F, F, F, F, P],
    "ader_lbf_merge_parts": [P, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "ader_gather_owned": [P, P, I, I, I, I, P, P],
    "ader_scatter_owned": [P, P, I, I, I, I, I, P, P, P],
    "ader_lbf_fwd_shard": [P, P, I, I, I, I, I, I, P, P, P, P, P],
    "ader_lx3_prep": [P, P, P, I, I, I, P],
    "ader_lx3_fwd_shard": [P, P, P, I, I, I, I, I, I, P, P, P, P, P],
    "ader_lx3_merge_parts": [P, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "ader_lx3_readout_shard": [P, I, I, I, I, I, I, P, L, P, P, P, P, P],
    "ader_lx3_merge_parts_kd": [P, P, I, I, I, I, P, P, P, P, P, P, P],
    "ader_lx3_fwd": [P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_lx3_fwd_img": [P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_lx3_fwd_img_lnf": [P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_lx3_fwd_kd_lnf": [P, P, I, I, I, I, I, I, I, I, P, P, P, L, P, F, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "ader_tab_grad": [P, P, P, I, I, I, I, I, P, P, P, P, P],
    "ader_tab_grad_kd": [P, P, P, I, I, I, I, I, I, P, P, P, P, L, P, P, P, P],
    "ader_host_shuffle": [P, P, L],
    "ader_host_pack_rows": [P, P, L, I, P, P],
    "ader_host_prefix_rows": [P, P, L, I, P, P],
    "ader_host_pack_rows_at": [P, L, P, P, L, I, P, P],
    "ader_x3_rep_image_bytes": [I],
    "ader_x3_update_pair_min_tiles": [I],
    "ader_x3_rep_image": [P, P, I, P, P],
    "ader_tab_update_x3": [P, P, P, I, I, I, I, I, P, P, P, I, P, F, P, P, I, P, P, P, P, P, F, F, F, F, I, I, P, P],
    "ader_tab_update_x3_kd": [P, P, P, I, I, I, I, I, I, P, P, P, I, P, F, P, P, I, P, P, P, L, P, P, P, P, P, F, F, F, F, P],
    "ader_tab_update_x3_kd_range": [P, P, P, I, I, I, I, I, I, P, P, P, I, P, F, P, P, I, P, P, P, L, P, P, P, P, P, F, F, F, F, I, I, P],
    "ader_tab_update_sh": [P, P, I, I, I, I, I, P, P, P, P, I, P, F, P, P, P, I, P, P, P, P, F, F, F, F, I, I, P, P],
    "ader_tab_meta_ints": [I],
    "ader_sparse_lists_scratch_n": [I, I, I],
    "ader_sparse_lists_starts": [I],
    "ader_sparse_lists": [P, I, P, I, I, P, P, P, P, P, P, P, P],
    "ader_sparse_lists_meta": [P, I, P, I, I, P, P, P, P, P, P, P, P, P],
    "ader_tab_tile_meta": [P, P, P, P, P, P, I, P, P],
    "ader_fused_bucket_gran": [],
    "ader_fused_bucket_id0": [],
    "ader_embed_bwd_rows": [P, P, P, I, I, I, I] + _DROP + [P],
    "ader_logits_store": [P, P, I, I, I, I, P, P, L, P],
    "ader_rank_targets": [P, P, I, I, I, I, P, P, P, P, P],
    "ader_adam_step": [P, P, P, P, Z, F, F, F, F, P, Z, I, P],
    "ader_fill": [P, Z, F, P],
    "ader_reduce_slabs": [P, L, I, I, I, I, P, P, P],
    "ader_reduce_slabs_batch": [P, P, P, P, P, P, P, P, I, P],
    "ader_pack_plan": [P, I, I, I, I, I, P, P, P, P, P, P],
    "ader_seq_pack_plan": [P, I, I, I, I, I, I, I, I, P, P],
    "ader_seqp_fwd": [P, P, I, P],
    "ader_seqp_bwd_ffn": [P, P, I, P],
    "ader_seqp_bwd_qkv": [P, P, I, P],
    "ader_attnp_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I] + _DROP + [P, I, P],
    "ader_attnp_last_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I] + _DROP + [P, P],
    "ader_pos_grad_packed": [P, P, P, I, I, I, P],
    "ader_gemm_atb_x3_batch_pk": [P, P, P, P, P, P, P, P, I, P, I, P],
    "ader_feed_step": [P, P, I, I, P, P, I, I, I, P, P, P, P, P],
    "ader_concat_i32": [P, I, P, I, P, P],
    "ader_step_fn_index": [ctypes.c_char_p],
    "ader_step_fn_args": [I],
    "ader_step_plan_create": [P, I, P, I, P, I, P, I, U, P],
    "ader_step_plan_destroy": [P],
    "ader_step_enqueue": [P, P, I, U, P, P],
    "ader_step_plan_peek": [P, P, I, U, P, P],
    "ader_step_plan_failed_op": [P],
    "ader_herding_select": [P, P, P, P, I, L, I, P, P, P, P, P, P],
}
# cross-check kernels of the tests (libader_xcheck.so, built with -DADER_XCHECK): not in the product library
_XSIGS = {
    "ader_tab_update": [P, P, P, I, I, I, I, I, P, P, P, I, P, F, P, P, I, P, P, P, P, P, F, F, F, F, I, I, P, P],
    "ader_tab_update_kd": [P, P, I, I, I, I, I, I, P, P, P, I, P, F, P, P, I, P, P, P, L, P, P, P, P, P, F, F, F, F, P],
    "ader_herding_select_generic": [P, P, P, P, I, L, I, P, P, P, P, P, P],
}
SEQ_MAXL = 4


class AderDrop(ctypes.Structure):
    _fields_ = [("key", c_uint), ("thr", c_uint), ("scale", c_float), ("base", c_uint), ("split", c_uint), ("base2", c_uint)]


class AderSeqBlock(ctypes.Structure):
    """include/ader_hip.h: AderSeqBlock"""
    _PTRS = ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "q_in", "mean1", "std1", "kmask", "qmask", "Q", "K", "V", "P", "x1", "y", "mean2",
             "std2", "h1d", "x2")
    _fields_ = ([("w", c_void_p * 5), ("bias", c_void_p * 5)] + [(k, c_void_p) for k in _PTRS] +
                [("d_attn", AderDrop), ("d_ffn1", AderDrop), ("d_ffn2", AderDrop), ("pruned", c_int), ("pad_", c_int)])


class AderSeqFwd(ctypes.Structure):
    """include/ader_hip.h: AderSeqFwd"""
    _fields_ = ([(k, c_void_p) for k in ("seq", "emb", "pos", "x0", "status", "lnf_g", "lnf_b", "rep", "meanf", "stdf")] +
                [(k, c_int) for k in ("B", "T", "H", "V", "L")] + [("sqrtH", c_float), ("sqrt_dh", c_float), ("pad_", c_int),
                                                                  ("d_emb", AderDrop), ("blk", AderSeqBlock * SEQ_MAXL)])


class AderLnfBwd(ctypes.Structure):
    """include/ader_hip.h: AderLnfBwd"""
    _fields_ = [(k, c_void_p) for k in ("x", "mean", "std", "gamma", "dx", "slab")]


class AderSeqPack(ctypes.Structure):
    """include/ader_hip.h: AderSeqPack (device arrays of the packing plan)"""
    _fields_ = [(k, c_void_p) for k in ("hdr", "tile_rows", "ids", "lpos", "gpos", "info", "srow0", "slen")]


class AderSeqBwdFfn(ctypes.Structure):
    """include/ader_hip.h: AderSeqBwdFfn"""
    _fields_ = ([(k, c_void_p) for k in ("seq", "dx2", "h1d", "x1", "mean2", "std2", "ln2_g", "w2", "w1", "dh2", "da", "dx1", "slab")] +
                [("d_ffn1", AderDrop), ("d_ffn2", AderDrop)] + [(k, c_int) for k in ("B", "T", "H", "pruned")])


class AderSeqBwdQkv(ctypes.Structure):
    """include/ader_hip.h: AderSeqBwdQkv"""
    _fields_ = ([(k, c_void_p) for k in ("seq", "dQ", "dx1", "dK", "dV", "x", "mean1", "std1", "ln1_g", "wq", "wk", "wv", "dx",
                                         "slab")] +
                [("d_emb", AderDrop)] + [(k, c_int) for k in ("B", "T", "H", "pruned", "emb_bwd", "pad_")])


STEP_MAX_ARGS, STEP_MAX_INPUTS = 48, 16


class AderStepOp(ctypes.Structure):
    """include/ader_hip.h: AderStepOp"""
    _fields_ = [(k, c_int) for k in ("kind", "fn", "stream", "other", "n_args", "pad_")] + [("args", ctypes.c_uint64 * STEP_MAX_ARGS)]


class AderStepBlob(ctypes.Structure):
    _fields_ = [("op", c_int), ("arg", c_int), ("src", c_void_p), ("bytes", c_size_t)]


class AderStepPatch(ctypes.Structure):
    _fields_ = [("blob", c_int), ("op", c_int), ("arg", c_int), ("input", c_int), ("offset", c_size_t), ("delta", ctypes.c_int64)]


class AderStepKey(ctypes.Structure):
    _fields_ = [("blob", c_int), ("site", c_int), ("offset", c_size_t)]


_NO_CHECK = {"ader_step_fn_index", "ader_step_fn_args", "ader_step_plan_failed_op", "ader_ln_bwd_slabs", "ader_gemm_atb_batch_slabs", "ader_gemm_atb_slabs", "ader_logits_sub", "ader_logits_parts", "ader_logits_ranges", "ader_lbf_ranges", "ader_lbf_ranges_kd", "ader_lbf_readout_ranges", "ader_lx3_readout_ranges", "ader_wprep_elems", "ader_fused_bucket_gran", "ader_fused_bucket_id0", "ader_tab_meta_ints", "ader_x3_rep_image_bytes", "ader_x3_update_pair_min_tiles", "ader_sparse_lists_scratch_n", "ader_sparse_lists_starts"}


class AderHipError(RuntimeError):
    pass


_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load the HIP library (import torch first so both share one HIP runtime instance)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise AderHipError("libader_hip.so not built: run `python -m ader_amd.build` (no CPU fallback exists)")
    import torch  # noqa: F401  (loads libamdhip64 that libader_hip.so resolves against)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argt in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argt
        fn.restype = c_int
    _lib = lib
    return lib


recorder = None      # ader_amd.engine.plan.Recorder while a train step is being recorded into a native launch plan, else None


_xlib = None


def load_xcheck():
    """The cross-check kernels (tests only): the round-2 fused table update and the generic herding kernel under its own name."""
    global _xlib
    if _xlib is None:
        if not os.path.isfile(XLIB_PATH):
            raise AderHipError("libader_xcheck.so not built: run `python -m ader_amd.build`")
        load()
        lib = ctypes.CDLL(XLIB_PATH)
        for name, argt in _XSIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = argt
            fn.restype = c_int
        _xlib = lib
    return _xlib


def call(name, *args):
    """Invoke a launcher; raises on a non-zero return code."""
    if recorder is not None and name not in _NO_CHECK:
        recorder.launch(name, args)
    rc = getattr(load_xcheck() if name in _XSIGS else load(), name)(*args)
    if name not in _NO_CHECK and rc != 0:
        raise AderHipError("%s failed with code %d" % (name, rc))
    return rc


_fn_index = {}


def step_fn_index(name):
    """Index of a launcher in the native step driver's dispatch table (-1: not a step launcher)."""
    i = _fn_index.get(name)
    if i is None:
        i = _fn_index[name] = load().ader_step_fn_index(name.encode())
    return i


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
