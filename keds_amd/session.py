"""Python face of the handle-based C ABI (include/keds_session.h).

This is the binding a host without torch modules would write: weights go in as a list of named
arrays (the reference's state_dict keys), the library packs and owns them, and every forward is
one C call.  The torch-module façade in `model.py` drives the same kernels through the stateless
ABI; both give bit-identical results (tests/test_gpu_session.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, load, ptr, stream


def _tensor_list(sd: Dict[str, "torch.Tensor | np.ndarray"]):
    """state_dict -> (keds_tensor array, keep-alive list).  numpy / CPU tensors are passed as host pointers."""
    arr = (_lib.Tensor * len(sd))()
    keep = []
    for i, (name, v) in enumerate(sd.items()):
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(np.ascontiguousarray(v))
        v = v.detach().contiguous()
        if v.dtype == torch.float32:
            dt = _lib.DT_F32
        elif v.dtype == torch.bfloat16:
            dt = _lib.DT_BF16
        elif v.dtype == torch.float16:
            dt = _lib.DT_F16
        else:
            v, dt = v.float(), _lib.DT_F32
        if v.dim() > 4:
            raise ValueError(f"{name}: more than 4 dimensions")
        nm = name.encode()
        keep += [v, nm]
        arr[i].name = nm
        arr[i].data = v.data_ptr()
        arr[i].dtype = dt
        arr[i].ndim = v.dim()
        for d in range(v.dim()):
            arr[i].shape[d] = v.shape[d]
    return arr, keep


class Context:
    def __init__(self, device: int = 0):
        _lib.require_gpu()
        self.h = C.c_void_p()
        check(load().keds_ctx_create(device, C.byref(self.h)), "keds_ctx_create")
        self.device = torch.device("cuda", device)

    def comm_init(self, rank: int, world: int, unique_id: bytes) -> None:
        buf = C.create_string_buffer(unique_id, _lib.COMM_ID_BYTES)
        check(load().keds_comm_init(self.h, rank, world, buf), "keds_comm_init")

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        check(load().keds_comm_unique_id(buf), "keds_comm_unique_id")
        return buf.raw

    def close(self):
        if self.h:
            load().keds_ctx_destroy(self.h)
            self.h = C.c_void_p()


class _Handle:
    _destroy = ""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.h = C.c_void_p()

    def close(self):
        if self.h:
            getattr(load(), self._destroy)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Vit(_Handle):
    """CLIP.encode_image (src/model/model.py:569-575) from the `visual.*` keys of a CLIP state_dict."""
    _destroy = "keds_vit_destroy"

    def __init__(self, ctx: Context, state_dict, compute: int = _lib.DT_BF16):
        super().__init__(ctx)
        arr, keep = _tensor_list({k: v for k, v in state_dict.items() if k.startswith("visual.")})
        check(load().keds_vit_create(ctx.h, arr, len(arr), compute, C.byref(self.h)), "keds_vit_create")
        info = [C.c_int() for _ in range(5)]
        check(load().keds_vit_info(self.h, *[C.byref(i) for i in info]), "keds_vit_info")
        self.width, self.layers, self.resolution, self.patch, self.embed_dim = [i.value for i in info]

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        dt = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}[image.dtype]
        image = image.contiguous()
        out = torch.empty((image.shape[0], self.embed_dim), dtype=torch.float32, device=image.device)
        check(load().keds_vit_forward(self.h, ptr(image), dt, image.shape[0], ptr(out), stream()), "keds_vit_forward")
        return out


class Text(_Handle):
    """CLIP.encode_text / encode_text_img_retrieval (src/model/model.py:577-590, 808-851)."""
    _destroy = "keds_text_destroy"

    def __init__(self, ctx: Context, state_dict, compute: int = _lib.DT_BF16):
        super().__init__(ctx)
        arr, keep = _tensor_list({k: v for k, v in state_dict.items() if not k.startswith("visual.")})
        check(load().keds_text_create(ctx.h, arr, len(arr), compute, C.byref(self.h)), "keds_text_create")
        info = [C.c_int() for _ in range(5)]
        check(load().keds_text_info(self.h, *[C.byref(i) for i in info]), "keds_text_info")
        self.width, self.layers, self.context, self.vocab, self.embed_dim = [i.value for i in info]

    def forward(self, tokens: torch.Tensor, readout: torch.Tensor, img_tokens: Optional[torch.Tensor] = None,
                insert_idx: int = 0, seq_used: Optional[int] = None) -> torch.Tensor:
        """seq_used: max(readout) + 1 when the caller knows it on the host (None: taken from `readout`, one device-to-host
        copy when that lives on the device; 0: unknown -- every column runs)."""
        tok = tokens.to(torch.int32).contiguous()
        it = None if img_tokens is None else img_tokens.float().contiguous()
        if seq_used is None and not readout.is_cuda and readout.numel():
            # the read-out columns are on the host: the library packs the captions' rows where that pays (keds_text_forward_packed)
            ro_h = readout.to(torch.int32).contiguous()
            out = torch.empty((tok.shape[0], self.embed_dim), dtype=torch.float32, device=tok.device)
            check(load().keds_text_forward_packed(self.h, ptr(tok), ptr(it), 0 if it is None else it.shape[1], insert_idx,
                                                  ro_h.data_ptr(), tok.shape[0], ptr(out), stream()), "keds_text_forward_packed")
            return out
        ro = readout.to(torch.int32).contiguous()
        if seq_used is None:
            seq_used = int(ro.max()) + 1 if ro.numel() else 0
        out = torch.empty((tok.shape[0], self.embed_dim), dtype=torch.float32, device=tok.device)
        check(load().keds_text_forward_used(self.h, ptr(tok), ptr(it), 0 if it is None else it.shape[1], insert_idx, ptr(ro),
                                            int(seq_used), tok.shape[0], ptr(out), stream()), "keds_text_forward")
        return out


class Knowledge(_Handle):
    """img2text + retrieval_fuse + text_condition of one stream (src/eval_utils.py:661-672)."""
    _destroy = "keds_knowledge_destroy"

    def __init__(self, ctx: Context, sd_img2text, sd_retrieval_fuse, sd_text_condition):
        super().__init__(ctx)
        a1, k1 = _tensor_list(sd_img2text)
        a2, k2 = _tensor_list(sd_retrieval_fuse)
        a3, k3 = _tensor_list(sd_text_condition)
        check(load().keds_knowledge_create(ctx.h, a1, len(a1), a2, len(a2), a3, len(a3), C.byref(self.h)),
              "keds_knowledge_create")

    def forward(self, q, nbr_img, nbr_txt) -> torch.Tensor:
        q, nbr_img, nbr_txt = q.float().contiguous(), nbr_img.float().contiguous(), nbr_txt.float().contiguous()
        B, K, d = nbr_img.shape
        out = torch.empty((B, 3, d), dtype=torch.float32, device=q.device)
        check(load().keds_knowledge_forward(self.h, ptr(q), ptr(nbr_img), ptr(nbr_txt), B, K, ptr(out), stream()),
              "keds_knowledge_forward")
        return out


class Index(_Handle):
    """faiss.IndexFlatL2-shaped exact index owned by the library (src/eval_retrieval.py:289-298)."""
    _destroy = "keds_index_destroy"

    def __init__(self, ctx: Context, dim: int, metric: int = _lib.METRIC_L2, row0: int = 0):
        super().__init__(ctx)
        self.dim = dim
        check(load().keds_index_create(ctx.h, dim, metric, _lib.DT_BF16, C.byref(self.h)), "keds_index_create")
        if row0:
            check(load().keds_index_set_base(self.h, row0), "keds_index_set_base")

    @property
    def ntotal(self) -> int:
        return int(load().keds_index_ntotal(self.h))

    def add(self, rows) -> None:
        """rows: numpy float32 [n, dim] (host pointer is handed to the library) or a torch tensor (host or device)."""
        if isinstance(rows, np.ndarray):
            rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32))
        rows = rows.float().contiguous()
        if rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected [n,{self.dim}] rows")
        check(load().keds_index_add(self.h, rows.data_ptr(), rows.shape[0]), "keds_index_add")

    def scan_image(self) -> torch.Tensor:
        """Copy of the bf16 scan image (uint8 device tensor, keds_index_packed_bytes(ntotal, dim) bytes)."""
        n = load().keds_index_packed_bytes(self.ntotal, self.dim)
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        check(load().keds_index_image(self.h, ptr(out), n), "keds_index_image")
        return out

    def search(self, q: torch.Tensor, k: int, gather: bool = False, sharded: bool = False):
        q = q.float().contiguous()
        B = q.shape[0]
        D = torch.empty((B, k), dtype=torch.float32, device=q.device)
        I = torch.empty((B, k), dtype=torch.int64, device=q.device)
        rows = torch.empty((B, k, self.dim), dtype=torch.float32, device=q.device) if gather else None
        if sharded:
            check(load().keds_index_search_sharded(self.h, ptr(q), B, k, ptr(D), ptr(I), ptr(rows), stream()),
                  "keds_index_search_sharded")
            return (D, I, rows) if gather else (D, I)
        check(load().keds_index_search(self.h, ptr(q), B, k, ptr(D), ptr(I), ptr(rows), stream()), "keds_index_search")
        return (D, I, rows) if gather else (D, I)
