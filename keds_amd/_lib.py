"""ctypes binding of libkeds_hip.so (the C ABI declared in include/keds_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is missing
or a call fails, a RuntimeError is raised.  Torch is used only for device memory,
streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libkeds_hip.so")

# ---- constants mirrored from keds_hip.h ------------------------------------------------------
ABI_VERSION = 8
METRIC_L2, METRIC_IP = 0, 1
EPI_BIAS_BF16, EPI_BIAS_QGELU_BF16, EPI_BIAS_RELU_BF16, EPI_BIAS_RESID_F32, EPI_BIAS_F32, EPI_PATCH_F32 = range(6)
EPI_LN_BIAS_BF16, EPI_LN_QGELU_BF16, EPI_RESID_STATS_F32, EPI_RESID_STATS_F16 = 6, 7, 8, 9
EPI_LN_BIAS_BF16_H, EPI_LN_QGELU_BF16_H = 10, 11
EPI_BIAS_BF16_HEADF32 = 12
EPI_X3_BIAS_F32, EPI_X3_RESID_F32, EPI_X3_QGELU_PAIR = 13, 14, 15
F32_EPI_BIAS, F32_EPI_QGELU, F32_EPI_RESID, F32_EPI_RELU, F32_EPI_PATCH = range(5)
FP8_EPI_BIAS_BF16, FP8_EPI_LN_BIAS_BF16, FP8_EPI_LN_QGELU_MX, FP8_EPI_RESID_STATS_MX, FP8_EPI_RESID_STATS_MX_H = range(5)
PROF_GEMM, PROF_ATTN, PROF_SCAN, PROF_LN, PROF_OTHER = range(5)
SCAN_MAX_K = 128

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class BlockParams(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "qkv_w", "out_w", "fc_w", "proj_w",
                                  "qkv_b", "out_b", "fc_b", "proj_b", "qkv_wf", "fc_wf", "qkv_bc", "fc_bc",
                                  "qkv_q8", "out_q8", "fc_q8", "proj_q8", "qkv_s8", "out_s8", "fc_s8", "proj_s8",
                                  "qkv_bc8", "fc_bc8")] + [("x3_exp", i32 * 4)]      # (f32 == 2: exponents of the split weights)


class TowerParams(C.Structure):
    _fields_ = [("width", i32), ("layers", i32), ("heads", i32), ("seq", i32), ("causal", i32),
                ("blocks", C.POINTER(BlockParams)), ("fp8", i32), ("last_cls_only", i32), ("f32", i32)]


class VitParams(C.Structure):
    _fields_ = [("tower", TowerParams), ("resolution", i32), ("patch", i32), ("kpad", i32), ("embed_dim", i32),
                ("conv_w", vp), ("class_emb", vp), ("pos_emb", vp), ("ln_pre_g", vp), ("ln_pre_b", vp),
                ("ln_post_g", vp), ("ln_post_b", vp), ("proj_t", vp)]


class TextParams(C.Structure):
    _fields_ = [("tower", TowerParams), ("vocab", i32), ("embed_dim", i32), ("token_emb", vp), ("pos_emb", vp),
                ("ln_final_g", vp), ("ln_final_b", vp), ("proj_t", vp)]


class CrossLayerParams(C.Structure):
    _fields_ = [(n, vp) for n in ("wq", "wk", "wv", "wo", "bq", "bk", "bv", "bo")]


class Im2TextParams(C.Structure):
    _fields_ = [("dim_in", i32), ("middle", i32), ("dim_out", i32), ("n_layer", i32), ("w", vp * 4), ("b", vp * 4),
                ("out_w", vp), ("out_b", vp)]


class CrossFormerFused(C.Structure):
    _fields_ = [("wkv", vp), ("bkv", vp), ("wqn", vp * 8), ("bqn", vp * 8)]


class CrossFormerParams(C.Structure):
    _fields_ = [("dim", i32), ("heads", i32), ("layers", i32), ("layer", C.POINTER(CrossLayerParams)),
                ("fused", C.POINTER(CrossFormerFused))]


class KnowledgeParams(C.Structure):
    _fields_ = [("i2t", Im2TextParams), ("fuse", CrossFormerParams), ("cond", CrossFormerParams)]


class Tensor(C.Structure):
    """keds_tensor of include/keds_session.h: one named weight (host or device memory)."""
    _fields_ = [("name", C.c_char_p), ("data", vp), ("dtype", i32), ("ndim", i32), ("shape", i64 * 4)]


DT_F32, DT_BF16, DT_F16, DT_FP8, DT_F32X3 = 0, 1, 2, 3, 4
COMM_ID_BYTES = 128
pp = C.POINTER(vp)      # handle out-parameter

# name -> (restype, argtypes).  Every symbol of include/keds_hip.h and include/keds_session.h is listed; tests
# check the set.
SIGNATURES = {
    # ---- keds_session.h (handles) --------------------------------------------------------------
    "keds_ctx_create": (i32, [i32, pp]),
    "keds_ctx_destroy": (i32, [vp]),
    "keds_vit_create": (i32, [vp, C.POINTER(Tensor), i32, i32, pp]),
    "keds_vit_destroy": (i32, [vp]),
    "keds_vit_info": (i32, [vp] + [C.POINTER(i32)] * 5),
    "keds_vit_forward": (i32, [vp, vp, i32, i32, vp, vp]),
    "keds_text_create": (i32, [vp, C.POINTER(Tensor), i32, i32, pp]),
    "keds_text_destroy": (i32, [vp]),
    "keds_text_info": (i32, [vp] + [C.POINTER(i32)] * 5),
    "keds_text_forward": (i32, [vp, vp, vp, i32, i32, vp, i32, vp, vp]),
    "keds_text_forward_used": (i32, [vp, vp, vp, i32, i32, vp, i32, i32, vp, vp]),
    "keds_text_forward_packed": (i32, [vp, vp, vp, i32, i32, vp, i32, vp, vp]),
    "keds_knowledge_create": (i32, [vp, C.POINTER(Tensor), i32, C.POINTER(Tensor), i32, C.POINTER(Tensor), i32, pp]),
    "keds_knowledge_destroy": (i32, [vp]),
    "keds_knowledge_forward": (i32, [vp, vp, vp, vp, i32, i32, vp, vp]),
    "keds_index_create": (i32, [vp, i32, i32, i32, pp]),
    "keds_index_destroy": (i32, [vp]),
    "keds_index_add": (i32, [vp, vp, i64]),
    "keds_index_ntotal": (i64, [vp]),
    "keds_index_set_base": (i32, [vp, i64]),
    "keds_index_search": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
    "keds_comm_unique_id": (i32, [vp]),
    "keds_comm_init": (i32, [vp, i32, i32, vp]),
    "keds_index_image": (i32, [vp, vp, sz]),
    "keds_index_search_sharded": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
    # ---- keds_hip.h (stateless) -----------------------------------------------------------------
    "keds_abi_version": (i32, []),
    "keds_build_flags": (C.c_char_p, []),
    "keds_last_error": (C.c_char_p, []),
    "keds_numerics_guard_set": (i32, [vp]),
    # ---- training building blocks (keds_hip.h section 9) ------------------------------------------
    "keds_transpose_to_bf16": (i32, [vp, i32, i64, i32, i32, vp, i32, vp]),
    "keds_colsum": (i32, [vp, i32, i64, i32, i32, vp, i32, vp]),
    "keds_dropout_mask": (i32, [vp, i64, C.c_uint64, f32, vp]),
    "keds_dropout_relu_fwd": (i32, [vp, vp, f32, vp, i64, vp]),
    "keds_dropout_relu_bwd": (i32, [vp, i32, vp, vp, f32, vp, i64, vp]),
    "keds_qgelu_fwd": (i32, [vp, vp, i64, vp]),
    "keds_qgelu_bwd": (i32, [vp, vp, vp, i64, vp]),
    "keds_ln_fwd_stats": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, vp]),
    "keds_ln_bwd": (i32, [vp, vp, i64, vp, vp, vp, vp, vp, i32, i32, vp]),
    "keds_attention_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "keds_cross_core_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "keds_cross_core_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "keds_clip_loss_workspace_bytes": (sz, [i32]),
    "keds_clip_loss": (i32, [vp, vp, i32, i32, i32, f32, vp, vp, vp, sz, vp]),
    "keds_l2norm_bwd": (i32, [vp, vp, vp, i32, i32, vp]),
    "keds_adamw_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "keds_rows_scatter": (i32, [vp, vp, vp, i64, i32, i32, i32, vp]),
    "keds_rows_gather": (i32, [vp, i64, vp, vp, i32, i32, vp]),
    "keds_prof_enable": (i32, [i32]),
    "keds_prof_reset": (i32, []),
    "keds_prof_read": (i32, [i32, C.POINTER(C.c_double), C.POINTER(i64)]),
    "keds_prof_read_work": (i32, [i32, C.POINTER(C.c_double)]),
    "keds_scan_debug": (i32, [i32]),
    "keds_merge_stamp_buffer": (i32, [vp]),
    "keds_index_packed_bytes": (sz, [i64, i32]),
    "keds_index_pack": (i32, [vp, i64, i32, i32, vp, vp]),
    "keds_index_pack_append": (i32, [vp, i64, i64, i32, i32, vp, vp]),
    "keds_index_search_workspace_bytes": (sz, [i32, i32]),
    "keds_index_search_workspace_bytes_ex": (sz, [i32, i32, i64, i32]),
    "keds_index_search_packed_ex": (i32, [vp, vp, i64, i32, i32, vp, i32, i32, i32, i64, vp, vp, vp, vp, sz, vp, vp]),
    "keds_index_search_packed": (i32, [vp, vp, i64, i32, i32, vp, i32, i32, i32, i64, vp, vp, vp, vp, sz, vp]),
    "keds_topk_merge_parts": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "keds_exchange_pack": (i32, [vp, vp, vp, i32, i32, i32, i32, i64, vp, vp]),
    "keds_exchange_merge": (i32, [vp, i32, i32, i32, i32, i64, i32, vp, vp, vp, vp]),
    "keds_gather_rows": (i32, [vp, i32, vp, i64, vp, vp]),
    "keds_rank_gallery_workspace_bytes": (sz, [i32, i32]),
    "keds_rank_gallery": (i32, [vp, i32, vp, i32, i32, vp, vp, sz, vp]),
    "keds_cirr_target_rank": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "keds_label_hits": (i32, [vp, i32, i32, vp, vp, C.POINTER(i32), i32, vp, vp, vp]),
    "keds_gemm_bt": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "keds_gemm_set_workspace": (i32, [vp, sz]),
    "keds_gemm_force_small": (i32, [i32]),
    "keds_gemm_x3": (i32, [vp, C.c_int64, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, i32, i32, i32, i32, i32, i32, vp]),
    "keds_split_f16_pair": (i32, [vp, C.c_int64, C.c_int64, i32, vp, C.c_int64, vp, vp]),
    "keds_split_f16_weight": (i32, [vp, C.c_int64, i32, vp, C.c_int64, C.POINTER(i32), vp]),
    "keds_gemm_bt_ex2": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, i32, vp, vp]),
    "keds_fold_layernorm": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "keds_rowstats_cast": (i32, [vp, vp, vp, i32, i32, vp]),
    "keds_rowstats_cast_ex": (i32, [vp, vp, i32, vp, i32, i32, vp]),
    "keds_fold_layernorm_ex": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, vp, vp]),
    "keds_mxfp8_scale_bytes": (sz, [i32, i32]),
    "keds_mxfp8_debug": (i32, [i32]),
    "keds_quantize_mxfp8": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "keds_gemm_mxfp8": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp]),
    "keds_gemm_mxfp8_ex": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "keds_fold_layernorm_mxfp8": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "keds_gemm_bt_ex": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, i32, vp]),
    "keds_layernorm": (i32, [vp, i64, vp, vp, vp, i32, i32, i32, vp]),
    "keds_attention_debug": (i32, [i32]),
    "keds_attention_stamp_buffer": (i32, [vp]),
    "keds_attention_ex": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "keds_attention_mx": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "keds_attention": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "keds_preprocess": (i32, [vp, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp]),
    "keds_preprocess_pil": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, C.POINTER(f32),
                                  C.POINTER(f32), vp, vp, vp]),
    "keds_im2col": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "keds_embed_tokens": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp]),
    "keds_readout_workspace_bytes": (sz, [i32, i32]),
    "keds_readout": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "keds_l2_normalize": (i32, [vp, vp, i32, i32, vp]),
    "keds_mix_normalize": (i32, [vp, vp, f32, f32, vp, vp, vp, i32, i32, vp]),
    "keds_cast_bf16": (i32, [vp, vp, i64, vp]),
    "keds_tower_fill_enable": (i32, [i32]),
    "keds_tower_workspace_bytes": (sz, [i32, i32, i32]),
    "keds_tower_workspace_bytes_ex": (sz, [C.POINTER(TowerParams), i32]),
    "keds_gemm_f32": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, i32, vp]),
    "keds_attention_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "keds_attention_x3": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    "keds_im2col_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "keds_knowledge_f32_workspace_bytes": (sz, [C.POINTER(KnowledgeParams), i32, i32]),
    "keds_knowledge_run_f32": (i32, [C.POINTER(KnowledgeParams), vp, vp, vp, i32, i32, vp, vp, sz, vp]),
    "keds_tokenizer_create": (i32, [C.c_char_p, C.POINTER(vp)]),
    "keds_tokenizer_destroy": (None, [vp]),
    "keds_tokenizer_special": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "keds_tokenize": (i32, [vp, C.POINTER(C.c_char_p), i32, i32, i32, vp]),
    "keds_tower_side_rows": (i32, [i32, i32, i32, i32]),
    "keds_side_lane_enable": (i32, [i32]),
    "keds_tower_forward": (i32, [C.POINTER(TowerParams), vp, i32, vp, sz, vp]),
    "keds_vit_workspace_bytes": (sz, [C.POINTER(VitParams), i32]),
    "keds_vit_run": (i32, [C.POINTER(VitParams), vp, i32, vp, i32, vp, sz, vp]),
    "keds_text_workspace_bytes": (sz, [C.POINTER(TextParams), i32]),
    "keds_text_run": (i32, [C.POINTER(TextParams), vp, vp, vp, i32, i32, i32, vp, i32, vp, sz, vp]),
    "keds_text_run_ex": (i32, [C.POINTER(TextParams), vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    "keds_text_trim_enable": (i32, [i32]),
    "keds_text_trim_mode": (i32, []),
    "keds_text_run_packed": (i32, [C.POINTER(TextParams), vp, vp, vp, i32, i32, vp, i32, i32, i32, vp, i32, vp, sz, vp]),
    "keds_attention_packed": (i32, [vp, vp, i32, i32, vp, i32, i32, vp]),
    "keds_im2text_workspace_bytes": (sz, [C.POINTER(Im2TextParams), i32]),
    "keds_im2text_forward": (i32, [C.POINTER(Im2TextParams), vp, i32, vp, vp, sz, vp]),
    "keds_crossformer_workspace_bytes": (sz, [C.POINTER(CrossFormerParams), i32, i32]),
    "keds_crossformer_forward": (i32, [C.POINTER(CrossFormerParams), vp, vp, vp, i32, i32, vp, vp, sz, vp]),
    "keds_crossformer_fused_bytes": (sz, [C.POINTER(CrossFormerParams)]),
    "keds_crossformer_fuse": (i32, [C.POINTER(CrossFormerParams), vp, sz, C.POINTER(CrossFormerFused), vp]),
    "keds_knowledge_workspace_bytes": (sz, [C.POINTER(KnowledgeParams), i32, i32]),
    "keds_knowledge_run": (i32, [C.POINTER(KnowledgeParams), vp, vp, vp, i32, i32, vp, vp, sz, vp]),
}

_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile libkeds_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libkeds_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def load() -> C.CDLL:
    """Load the HIP library.  Raises RuntimeError if it is missing (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C keds_amd/csrc`).  keds_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.keds_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libkeds_hip.so ABI {got} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def source_digest() -> str:
    """First 16 hex digits of the SHA-256 over the kernel sources (csrc/*.hip, *.h, *.cpp, Makefile and include/*.h, in name
    order).  Measurement files under profiles/ carry the digest of the sources they were taken on; bench.py quotes a
    committed PMC / parity number only while this digest still matches (no .git travels to the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = []
    for pat in ("csrc/*.hip", "csrc/*.h", "csrc/*.cpp", "csrc/Makefile"):
        files += glob.glob(os.path.join(_HERE, pat))
    files += glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h"))
    for f in sorted(files, key=lambda x: os.path.relpath(x, os.path.dirname(_HERE))):
        h.update(os.path.relpath(f, os.path.dirname(_HERE)).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    # a variant build (make EXTRA="-D...": A/B and timing-only scripts) is a different build of the same sources
    try:
        extra = (load().keds_build_flags() or b"").decode().strip()
    except Exception:                                                 # noqa: BLE001  (no library: the sources alone)
        extra = ""
    if extra:
        h.update(b"EXTRA " + extra.encode())
    return h.hexdigest()[:16]


def build_flags() -> str:
    """The EXTRA flags the loaded library was built with ("" for the product build; "-DKEDS_EXPERIMENTS ..." for the A/B tools')."""
    return (load().keds_build_flags() or b"").decode().strip()


def last_error() -> str:
    return (load().keds_last_error() or b"").decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("keds_amd needs an MI355X (ROCm device); there is no CPU fallback")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("keds_amd: tensor is not on the GPU")
    if not t.is_contiguous():
        raise RuntimeError("keds_amd: tensor must be contiguous")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Workspace:
    """Grow-only device scratch buffer (one per owner object)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.zeros(int(nbytes) + 256, dtype=torch.uint8, device=device)
        return self.buf


_gemm_ws = {}


def ensure_gemm_workspace(device=None) -> None:
    """Register an 8 MiB split-K scratch buffer for `device` (idempotent; one registration per device).  Only direct
    `ops.gemm_bt` calls use it (one stream at a time per device): towers and handles split K into their own workspace."""
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _gemm_ws:
        buf = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
        _gemm_ws[key] = buf
        with torch.cuda.device(key):
            check(load().keds_gemm_set_workspace(buf.data_ptr(), buf.numel()), "keds_gemm_set_workspace")


def prof_enable(on, classes=None) -> None:
    """on: False/True, or pass `classes` (iterable of PROF_* ids) to record only those kernel classes."""
    code = 0
    if classes is not None:
        for k in classes:
            code |= 1 << (k + 1)
    elif on:
        code = 1
    check(load().keds_prof_enable(code), "keds_prof_enable")


def prof_reset() -> None:
    check(load().keds_prof_reset(), "keds_prof_reset")


def prof_read_work(klass: int) -> float:
    """Algorithmic flops (PROF_GEMM) / scan-image bytes (PROF_SCAN) of the launches that carried event pairs."""
    u = C.c_double(0)
    check(load().keds_prof_read_work(klass, C.byref(u)), "keds_prof_read_work")
    return u.value


def prof_read(klass: int):
    ms, n = C.c_double(0), i64(0)
    check(load().keds_prof_read(klass, C.byref(ms), C.byref(n)), "keds_prof_read")
    return ms.value, n.value
