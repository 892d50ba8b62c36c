"""Faiss-shaped flat index backed by the gfx950 scan kernel.

Mirrors the object the reference builds at src/eval_retrieval.py:289-298 and calls at
src/eval_utils.py:169,177:

    index = IndexFlatL2(768); index = index_cpu_to_all_gpus(index)
    index.add(bases.numpy());  D, I = index.search(q_numpy, 16)

`add` takes np.float32 [N, d] (or a torch tensor), `search` takes np.float32 [B, d] and
returns (D float32 [B,k] ascending squared L2, I int64 [B,k]) as numpy arrays -- or torch
device tensors when the query is a device tensor, which skips the D2H/H2D hop the reference
pays.  `search_gather` additionally returns the gathered rows (eval_utils.py:171-172).

Row ranges can be sharded over ranks: every rank holds rows
[row0, row0 + n_local), searches them for ALL queries, the [B,k] partials are all-gathered
with torch.distributed and merged keyed on (distance, id), so results are identical for any
number of shards.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import check, load, ptr, stream

ArrayLike = Union[np.ndarray, torch.Tensor]


class FlatIndex:
    def __init__(self, d: int, metric: str = "l2", device: Optional[Union[str, torch.device]] = None,
                 row0: int = 0):
        if metric not in ("l2", "ip"):
            raise ValueError("metric must be 'l2' or 'ip'")
        if d not in (128, 256, 512, 768, 1024):
            raise ValueError(f"dimension {d} unsupported (128, 256, 512, 768, 1024)")
        self.d = d
        self.metric = _lib.METRIC_L2 if metric == "l2" else _lib.METRIC_IP
        self.device = torch.device(device) if device is not None else None
        self.row0 = int(row0)            # global id of local row 0 (sharding)
        self.rows: Optional[torch.Tensor] = None      # fp32 [n, d] on device (re-rank + gather source)
        self.packed: Optional[torch.Tensor] = None    # bf16 scan image
        self._rows_buf: Optional[torch.Tensor] = None     # capacity buffers behind rows / packed (chunked add)
        self._packed_buf: Optional[torch.Tensor] = None
        self._ws = _lib.Workspace()
        self._status: Optional[torch.Tensor] = None   # device int32[2]: queries certified from the candidates / by the exact pass

    # ---- faiss-like surface -------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return 0 if self.rows is None else int(self.rows.shape[0])

    def _dev(self) -> torch.device:
        if self.device is None:
            _lib.require_gpu()
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    def add(self, x: ArrayLike) -> None:
        """Append rows and extend the bf16 scan image.  x: float32 [n, d].  Chunked adds pack only the new stages into
        buffers that grow geometrically (amortised O(rows added), not a re-pack of the whole database per call)."""
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if isinstance(x, np.ndarray) else x
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] rows, got {tuple(t.shape)}")
        if t.shape[0] == 0:
            return
        t = t.to(self._dev(), dtype=torch.float32).contiguous()
        lib = load()
        old_n = self.ntotal
        n = old_n + t.shape[0]
        if old_n == 0:
            self._rows_buf, self._packed_buf = t, torch.empty(lib.keds_index_packed_bytes(n, self.d), dtype=torch.uint8,
                                                              device=self.device)
        else:
            if n > self._rows_buf.shape[0]:
                cap = max(n, 2 * self._rows_buf.shape[0])
                rows_buf = torch.empty((cap, self.d), dtype=torch.float32, device=self.device)
                rows_buf[:old_n] = self.rows
                packed_buf = torch.empty(lib.keds_index_packed_bytes(cap, self.d), dtype=torch.uint8, device=self.device)
                old_bytes = lib.keds_index_packed_bytes(old_n, self.d)
                packed_buf[:old_bytes] = self.packed[:old_bytes]
                self._rows_buf, self._packed_buf = rows_buf, packed_buf
            self._rows_buf[old_n:n] = t
        self.rows = self._rows_buf[:n]
        self.packed = self._packed_buf[:lib.keds_index_packed_bytes(n, self.d)]
        check(lib.keds_index_pack_append(ptr(self.rows), old_n, n, self.d, self.metric, ptr(self.packed), stream()),
              "keds_index_pack_append")

    def reset(self) -> None:
        self.rows = None
        self.packed = None
        self._rows_buf = self._packed_buf = None

    # ---- on-disk form (the database is built once and reloaded by every eval run) ----------------
    MAGIC = "keds-flat-index-v1"

    def save(self, path: str) -> None:
        """One torch file: the fp32 rows (what the reference keeps in cc_*_databases.pt, eval_retrieval.py:281-282) plus
        the packed bf16 scan image, so loading needs no re-pack.  The image layout is tied to the kernel: the file
        carries the ABI version and is re-packed on load if it does not match."""
        if self.rows is None:
            raise RuntimeError("save of an empty index")
        torch.save({"magic": self.MAGIC, "abi": _lib.ABI_VERSION, "d": self.d, "metric": self.metric, "row0": self.row0,
                    "rows": self.rows.cpu().clone(), "packed": self.packed.cpu().clone()}, path)

    @classmethod
    def load(cls, path: str, device=None) -> "FlatIndex":
        """Load an index written by `save`, or a plain float32 [N, d] tensor file such as the reference's
        cc_image_databases.pt / cc_text_databases.pt (then the scan image is built here)."""
        blob = torch.load(path, map_location="cpu")
        if isinstance(blob, torch.Tensor):
            idx = cls(blob.shape[1], "l2", device=device)
            idx.add(blob.float())
            return idx
        if not isinstance(blob, dict) or blob.get("magic") != cls.MAGIC:
            raise RuntimeError(f"{path} is neither a saved FlatIndex nor a [N, d] tensor")
        idx = cls(int(blob["d"]), "l2" if int(blob["metric"]) == _lib.METRIC_L2 else "ip", device=device,
                  row0=int(blob["row0"]))
        dev = idx._dev()
        expect = load().keds_index_packed_bytes(blob["rows"].shape[0], idx.d)
        if int(blob["abi"]) == _lib.ABI_VERSION and blob["packed"].numel() == expect:
            idx.rows = idx._rows_buf = blob["rows"].to(dev, dtype=torch.float32).contiguous()
            idx.packed = idx._packed_buf = blob["packed"].to(dev).contiguous()
        else:
            idx.add(blob["rows"])
        return idx

    # ---- search ---------------------------------------------------------------------------------
    def search_device(self, q: torch.Tensor, k: int, normalize: bool = False,
                      gather: bool = False) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """Device-resident search.  q fp32 [B, d] on the index's device."""
        if self.rows is None:
            raise RuntimeError("search on an empty index")
        if not (1 <= k <= _lib.SCAN_MAX_K):
            raise ValueError(f"k must be in [1, {_lib.SCAN_MAX_K}] for the scan path (got {k})")
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"expected [B, {self.d}] queries, got {tuple(q.shape)}")
        q = q.to(self.device, dtype=torch.float32).contiguous()
        B = q.shape[0]
        lib = load()
        nbytes = lib.keds_index_search_workspace_bytes_ex(B, self.d, self.rows.shape[0], k)
        if nbytes == 0:
            raise ValueError(f"search of {B} queries x k={k} over {self.rows.shape[0]} rows is not supported in one call")
        ws = self._ws.get(nbytes, self.device)
        if self._status is None or self._status.device != self.device:
            self._status = torch.zeros(2, dtype=torch.int32, device=self.device)
        D = torch.empty((B, k), dtype=torch.float32, device=self.device)
        I = torch.empty((B, k), dtype=torch.int64, device=self.device)
        rows = torch.empty((B, k, self.d), dtype=torch.float32, device=self.device) if gather else None
        check(lib.keds_index_search_packed_ex(ptr(self.packed), ptr(self.rows), self.rows.shape[0], self.d, self.metric,
                                              ptr(q), B, 1 if normalize else 0, k, self.row0, ptr(D), ptr(I), ptr(rows),
                                              ptr(ws), ws.numel(), ptr(self._status), stream()), "keds_index_search_packed")
        return D, I, rows

    def certificate_counts(self, reset: bool = False) -> Tuple[int, int]:
        """(queries whose top-k was PROVEN exact from the re-ranked candidates, queries answered by the exact fp32 pass over
        all rows) since the last reset.  Every result is exact either way (IndexFlatL2 semantics); the second number says how
        often the candidate list was too narrow to certify (near-duplicate clusters).  Synchronises."""
        if self._status is None:
            return 0, 0
        a, b = (int(v) for v in self._status.cpu())
        if reset:
            self._status.zero_()
        return a, b

    def search(self, q: ArrayLike, k: int):
        """Faiss call shape: numpy in -> numpy out; device tensor in -> device tensors out."""
        if isinstance(q, np.ndarray):
            D, I, _ = self.search_device(torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)), k)
            return D.cpu().numpy(), I.cpu().numpy()
        D, I, _ = self.search_device(q, k)
        return D, I

    def search_gather(self, q: torch.Tensor, k: int, normalize: bool = False):
        """(D, I, rows [B,k,d]) with the winners' fp32 rows gathered on device."""
        return self.search_device(q, k, normalize=normalize, gather=True)


def IndexFlatL2(d: int) -> FlatIndex:
    """faiss.IndexFlatL2(d) stand-in (src/eval_retrieval.py:291)."""
    return FlatIndex(d, "l2")


def IndexFlatIP(d: int) -> FlatIndex:
    return FlatIndex(d, "ip")


def index_cpu_to_all_gpus(index: FlatIndex) -> FlatIndex:
    """faiss.index_cpu_to_all_gpus stand-in (src/eval_retrieval.py:292): the index already lives on
    this process's GPU; multi-GPU sharding is per process (see ShardedFlatIndex)."""
    return index


# ---------------------------------------------------------------------------------------------------
# sharding over ranks (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm)
# ---------------------------------------------------------------------------------------------------
def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of shard `rank`: contiguous, sizes differ by at most one 32-row stage."""
    stages = (n + 31) // 32
    lo = (stages * rank // world) * 32
    hi = min(n, (stages * (rank + 1) // world) * 32)
    return lo, hi


def merge_partials(D_parts: torch.Tensor, I_parts: torch.Tensor, metric: int = _lib.METRIC_L2):
    """Host-side (torch) merge of [parts, B, k] partial results keyed on (distance, id): the same
    ordering the device merge uses.  Works on CPU tensors, so the N>1 logic is testable with gloo."""
    parts, B, k = D_parts.shape
    D = D_parts.permute(1, 0, 2).reshape(B, parts * k)
    I = I_parts.permute(1, 0, 2).reshape(B, parts * k)
    key = D if metric == _lib.METRIC_L2 else -D
    key = torch.where(I < 0, torch.full_like(key, float("inf")), key)
    # lexicographic (key, id): sort by id first, then stable sort by key
    o1 = torch.sort(I, dim=1, stable=True).indices
    key1 = torch.gather(key, 1, o1)
    o2 = torch.sort(key1, dim=1, stable=True).indices
    order = torch.gather(o1, 1, o2)[:, :k]
    return torch.gather(D, 1, order), torch.gather(I, 1, order)


def exchange_and_merge(D: torch.Tensor, I: torch.Tensor, metric: int = _lib.METRIC_L2, group=None):
    """All-gather every rank's [B,k] partial result and merge keyed on (distance, id).

    One small collective per search (B*k*12 bytes per rank: latency-bound on xGMI).
    Device tensors are merged by the HIP kernel (keds_topk_merge_parts); CPU tensors -- the gloo
    tests of the multi-rank logic -- by `merge_partials`, which a GPU test pins to the kernel."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    Dp = [torch.empty_like(D) for _ in range(world)]
    Ip = [torch.empty_like(I) for _ in range(world)]
    dist.all_gather(Dp, D.contiguous(), group=group)
    dist.all_gather(Ip, I.contiguous(), group=group)
    Dp, Ip = torch.stack(Dp), torch.stack(Ip)
    if D.is_cuda:
        return ops.topk_merge_parts(Dp, Ip, metric)
    return merge_partials(Dp, Ip, metric)


def exchange_merge_gather(D_p: torch.Tensor, I_p: torch.Tensor, rows_p: torch.Tensor, B: int,
                          metric: int = _lib.METRIC_L2, group=None):
    """Sharded search WITH the winners' rows (the knowledge path needs the 16 neighbour rows, which live on the shard
    that found them; SURVEY 8e option B).  Every rank searched its shard for all B*world queries and holds
    D_p / I_p [B*world, k] and rows_p [B*world, k, dim] (rows of its own shard).  Block r of each (the B queries of
    rank r) is sent to rank r only (all-to-all: B*k*dim*4 bytes per peer, 6.3 MB at B=128, k=16), then the owner merges
    its `world` partial lists keyed on (distance, id) and picks each winner's row from the part that supplied it.
    Returns (D [B,k], I [B,k], rows [B,k,dim]) for this rank's own queries.  Works on CPU tensors (gloo tests)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    k, dim = D_p.shape[1], rows_p.shape[2]

    def a2a(t):
        t = t.contiguous()
        out = torch.empty_like(t)
        if t.is_cuda:
            dist.all_to_all_single(out, t, group=group)          # equal splits: block r <-> rank r
        else:                                                     # gloo has no all_to_all_single on every build: use all_gather
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t, group=group)
            rank = dist.get_rank(group)
            n = t.shape[0] // world
            out = torch.cat([p[rank * n:(rank + 1) * n] for p in parts])
        return out

    Dq = a2a(D_p).reshape(world, B, k)                            # part w = what shard w found for MY queries
    Iq = a2a(I_p).reshape(world, B, k)
    Rq = a2a(rows_p).reshape(world, B, k, dim)
    if Dq.is_cuda:
        D, I = ops.topk_merge_parts(Dq, Iq, metric)
    else:
        D, I = merge_partials(Dq, Iq, metric)
    # provenance: ids are unique across shards, so the slot that holds id I[b,j] is the one that supplied it
    flat_i = Iq.permute(1, 0, 2).reshape(B, world * k)            # [B, world*k]
    hit = flat_i[:, None, :] == I[:, :, None]                     # [B, k, world*k]
    src = hit.to(torch.int8).argmax(dim=2)                        # first match (padding ids -1 never win a real slot)
    flat_r = Rq.permute(1, 0, 2, 3).reshape(B, world * k, dim)
    rows = torch.gather(flat_r, 1, src[:, :, None].expand(B, k, dim))
    rows = torch.where((I >= 0)[:, :, None], rows, torch.zeros_like(rows))
    return D, I, rows


def install_host_staged_transport(dist) -> None:
    """DIAGNOSTIC transport for several ranks that share ONE GPU (RCCL refuses two ranks on a device): patches the two
    collectives the packed exchange uses so that device tensors travel device -> host -> the process group (gloo) -> host
    -> device.  Everything around the collectives stays the product's device branch (pack / merge kernels, shard scans).
    Used by tests/exchange2_worker.py and by `KEDS_BENCH_SHARED_GPU=1 python bench.py --gpus N`; never on a real node."""
    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        w = dist.get_world_size(group)
        h = inp.detach().cpu().contiguous()
        parts = [torch.empty_like(h) for _ in range(w)]
        dist.all_gather(parts, h, group=group)
        out.copy_(torch.cat(parts).view(out.shape).to(out.device))

    def all_to_all_single(out, inp, group=None, **kw):
        w, r = dist.get_world_size(group), dist.get_rank(group)
        h = inp.detach().cpu().contiguous()
        parts = [torch.empty_like(h) for _ in range(w)]
        dist.all_gather(parts, h, group=group)
        n = h.shape[0] // w                                   # equal splits: block r of every rank comes to rank r
        out.copy_(torch.cat([p[r * n:(r + 1) * n] for p in parts]).view(out.shape).to(out.device))

    dist.all_gather_into_tensor = all_gather_into_tensor
    dist.all_to_all_single = all_to_all_single


class PackedExchange:
    """The two collectives of one data-parallel sharded search, on preallocated buffers:

      gather_queries(q [B,d])       -> [world*B, d]   one all_gather_into_tensor (393 KB per rank at B = 128, d = 768)
      return_partials(D_p, I_p)     -> (D, I) [B,k]   ONE all-to-all of the packed partial lists (12 B per entry: distance
                                                      bits | id low | id high as int32; rank r receives block r of every
                                                      rank = what each shard found for ITS queries), then the (distance, id)
                                                      merge on the owner.

    Both are KB-sized and latency-bound on xGMI; nothing is allocated per call (the list-form all_gather of round 1
    allocated 2*world tensors per search and moved every rank's lists to every rank).  CPU tensors (gloo tests) take the
    all_gather_into_tensor route for the second step as well, since gloo lacks all_to_all_single on some builds."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._q = self._pin = self._pout = None

    def gather_queries(self, q: torch.Tensor) -> torch.Tensor:
        q = q.contiguous()
        shape = (self.world * q.shape[0], q.shape[1])
        if self._q is None or self._q.shape != shape or self._q.device != q.device or self._q.dtype != q.dtype:
            self._q = torch.empty(shape, dtype=q.dtype, device=q.device)
        self.dist.all_gather_into_tensor(self._q, q, group=self.group)
        return self._q

    def return_partials(self, D_p: torch.Tensor, I_p: torch.Tensor, metric: int = _lib.METRIC_L2):
        """D_p / I_p [world*B, k]: this shard's partial lists for EVERY rank's queries (global ids)."""
        w = self.world
        n, k = D_p.shape
        B = n // w
        shape = (w, B, k, 3)
        if self._pin is None or self._pin.shape != shape or self._pin.device != D_p.device:
            self._pin = torch.empty(shape, dtype=torch.int32, device=D_p.device)
            self._pout = torch.empty(shape, dtype=torch.int32, device=D_p.device)
        pin = self._pin
        if D_p.is_cuda:                                           # pack kernel -> all-to-all -> merge kernel: 3 launches
            ops.exchange_pack(D_p.contiguous(), I_p.contiguous(), None, w, pin, B * k * 3)
            self.dist.all_to_all_single(self._pout, pin, group=self.group)       # block r <-> rank r
            D, I, _ = ops.exchange_merge(self._pout, w, B, k, 0, B * k * 3, metric, False)
            return D, I
        pin[..., 0] = D_p.contiguous().view(torch.int32).view(w, B, k)           # CPU tensors: the gloo tests of this logic
        pin[..., 1:] = I_p.contiguous().view(torch.int32).view(w, B, k, 2)
        every = torch.empty((w * w, B, k, 3), dtype=torch.int32)                  # concatenated along dim 0
        self.dist.all_gather_into_tensor(every, pin, group=self.group)
        got = every.view(w, w, B, k, 3)[:, self.rank].contiguous()
        Dq = got[..., 0].contiguous().view(torch.float32)                          # [world, B, k]: part s = shard s's list
        Iq = got[..., 1:].contiguous().view(torch.int64).view(w, B, k)
        return merge_partials(Dq, Iq, metric)

    def return_partials_rows(self, parts, metric: int = _lib.METRIC_L2):
        """The knowledge path's form (SURVEY 8e option B): `parts` = [(D_p, I_p, rows_p), ...], one triple per DATABASE
        searched with the same gathered queries (image and text database: two), D_p / I_p [world*B, k] and rows_p
        [world*B, k, d] fp32 (this shard's rows of its partial winners).  ALL of it goes out in ONE all-to-all: per list
        entry d + 4 int32 words (distance bits | id low | id high | pad | the row; keds_hip.h, keds_exchange_pack), block r
        to rank r -- so the two databases'
        partials and their rows share one message (6.3 MB per peer and database at B = 128, k = 16, d = 768) instead of
        the six list-form collectives of round 2.  The owner then merges each database's `world` lists keyed on
        (distance, id) and picks every winner's row from the part that supplied it.
        Returns [(D [B,k], I [B,k], rows [B,k,d]), ...] for this rank's own queries."""
        w = self.world
        n, k = parts[0][0].shape
        d = parts[0][2].shape[2]
        B, P, E = n // w, len(parts), d + 4                       # entry: distance | id low | id high | pad | row
        shape = (w, P, B, k, E)
        dev = parts[0][0].device
        if getattr(self, "_rin", None) is None or self._rin.shape != shape or self._rin.device != dev:
            self._rin = torch.zeros(shape, dtype=torch.int32, device=dev)
            self._rout = torch.empty(shape, dtype=torch.int32, device=dev)
        rin = self._rin
        if dev.type == "cuda":                                    # pack kernels -> ONE all-to-all -> merge kernels
            stride = P * B * k * E
            for j, (D_p, I_p, rows_p) in enumerate(parts):
                ops.exchange_pack(D_p.contiguous(), I_p.contiguous(), rows_p.contiguous(), w, rin[0, j], stride)
            self.dist.all_to_all_single(self._rout, rin, group=self.group)       # block r <-> rank r
            return [ops.exchange_merge(self._rout[0, j], w, B, k, d, stride, metric, True) for j in range(P)]
        for j, (D_p, I_p, rows_p) in enumerate(parts):
            rin[:, j, :, :, 0] = D_p.contiguous().view(torch.int32).view(w, B, k)
            rin[:, j, :, :, 1:3] = I_p.contiguous().view(torch.int32).view(w, B, k, 2)
            rin[:, j, :, :, 4:] = rows_p.contiguous().view(torch.int32).view(w, B, k, d)
        every = torch.empty((w * w,) + shape[1:], dtype=torch.int32)
        self.dist.all_gather_into_tensor(every, rin, group=self.group)
        got = every.view((w, w) + shape[1:])[:, self.rank].contiguous()
        out = []
        for j in range(P):
            g = got[:, j]                                                          # [world, B, k, E]
            Dq = g[..., 0].contiguous().view(torch.float32)
            Iq = g[..., 1:3].contiguous().view(torch.int64).view(w, B, k)
            if Dq.is_cuda:
                D, I = ops.topk_merge_parts(Dq, Iq, metric)
            else:
                D, I = merge_partials(Dq, Iq, metric)
            # provenance: ids are unique across shards, so the slot that holds id I[b,j] is the one that supplied it
            flat_i = Iq.permute(1, 0, 2).reshape(B, w * k)
            src = (flat_i[:, None, :] == I[:, :, None]).to(torch.int8).argmax(dim=2)           # [B, k]
            flat_r = g[..., 4:].permute(1, 0, 2, 3).reshape(B, w * k, d)
            rows = torch.gather(flat_r, 1, src[:, :, None].expand(B, k, d)).contiguous().view(torch.float32)
            rows = torch.where((I >= 0)[:, :, None], rows, torch.zeros_like(rows))
            out.append((D, I, rows))
        return out


class ShardedFlatIndex:
    """Row-sharded index: rank r owns rows shard_bounds(n, world, r).

    search(q): every rank passes the SAME queries (all-gather them first if they were produced
    data-parallel); each scans its shard, the [B,k] partials are all-gathered and merged.
    """

    def __init__(self, d: int, metric: str = "l2", group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.d = d
        self.metric_name = metric
        self.local: Optional[FlatIndex] = None
        self.device = device
        self.n_global = 0
        self._xchg: Optional[PackedExchange] = None

    def add_global(self, x: ArrayLike) -> None:
        """Give every rank the full matrix (or a view of it); each keeps only its shard."""
        n = x.shape[0]
        lo, hi = shard_bounds(n, self.world, self.rank)
        self.n_global = n
        self.local = FlatIndex(self.d, self.metric_name, device=self.device, row0=lo)
        self.local.add(x[lo:hi])

    @property
    def ntotal(self) -> int:
        return self.n_global

    def search(self, q: torch.Tensor, k: int, normalize: bool = False):
        D, I, _ = self.local.search_device(q, k, normalize=normalize)
        if self.world == 1:
            return D, I
        return exchange_and_merge(D, I, self.local.metric, self.group)

    def search_own(self, q_local: torch.Tensor, k: int, normalize: bool = False):
        """Data-parallel form without the rows: every rank passes ITS OWN B queries (same B everywhere) and gets (D, I)
        for them -- two packed collectives on preallocated buffers (`PackedExchange`)."""
        if self.world == 1:
            D, I, _ = self.local.search_device(q_local, k, normalize=normalize)
            return D, I
        if self._xchg is None:
            self._xchg = PackedExchange(self.group)
        allq = self._xchg.gather_queries(q_local.to(self.local._dev(), dtype=torch.float32))
        D_p, I_p, _ = self.local.search_device(allq, k, normalize=normalize)
        return self._xchg.return_partials(D_p, I_p, self.local.metric)

    def search_gather(self, q_local: torch.Tensor, k: int, normalize: bool = False):
        """Data-parallel form: every rank passes ITS OWN B queries (same B everywhere) and gets (D, I, rows) for them:
        all-gather of the queries, local scan + gather on the shard, all-to-all of the partials with their rows, merge
        and row selection on the owner (`exchange_merge_gather`)."""
        return ShardedFlatIndex.search_gather_many([self], q_local, k, normalize=normalize)[0]

    @staticmethod
    def search_gather_many(indices, q_local: torch.Tensor, k: int, normalize: bool = False):
        """`search_gather` of the SAME queries against several row-sharded databases (the knowledge path searches the image
        and the text database with one query batch, eval_utils.py:169-183): ONE all-gather of the queries, one local
        scan + gather per database, ONE all-to-all carrying every database's partial lists with their rows
        (`PackedExchange.return_partials_rows`).  Returns [(D, I, rows), ...] in the order of `indices`."""
        first = indices[0]
        if first.world == 1:
            return [ix.local.search_gather(q_local, k, normalize=normalize) for ix in indices]
        if first._xchg is None:
            first._xchg = PackedExchange(first.group)
        x = first._xchg
        allq = x.gather_queries(q_local.to(first.local._dev(), dtype=torch.float32))
        parts = [ix.local.search_gather(allq, k, normalize=normalize) for ix in indices]
        return x.return_partials_rows(parts, first.local.metric)
