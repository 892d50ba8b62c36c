"""CPU (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the facade keeps the
reference's checkpoint contract and fails loudly without a GPU, and the multi-rank search logic
(shard bounds, all-gather, (distance, id) merge) is exact under gloo with world_size 2."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

import keds_amd
from keds_amd import _lib
from keds_amd.index import merge_partials, shard_bounds
from oracle import keds_oracle as O
from tests.conftest import ROOT, golden_path


def _declared_symbols():
    names = set()
    for header in ("keds_hip.h", "keds_session.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(keds_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 60
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes SIGNATURES and include/*.h disagree"
    assert lib.keds_abi_version() == _lib.ABI_VERSION


def test_session_abi_fails_loudly_without_a_gpu():
    """The handle layer (keds_session.h) reports errors instead of computing anywhere else."""
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    if not torch.cuda.is_available():
        assert lib.keds_ctx_create(0, C.byref(h)) != 0 and not h.value
        assert _lib.last_error() != ""
    assert lib.keds_vit_create(None, None, 0, _lib.DT_BF16, C.byref(h)) == -1
    assert lib.keds_index_create(None, 768, 0, _lib.DT_BF16, C.byref(h)) == -1 and "null context" in _lib.last_error()
    assert lib.keds_index_ntotal(None) == -1
    assert lib.keds_index_search_sharded(None, None, 0, 0, None, None, None, None) == -1


def test_size_queries_need_no_gpu():
    lib = _lib.load()
    # 500k x 768: 15625 stages of (32*768*2 + 128) bytes
    # (+ a 128-byte trailer: the rounding bounds of the search certificate)
    assert lib.keds_index_packed_bytes(500000, 768) == 15625 * (32 * 768 * 2 + 128) + 128
    assert lib.keds_index_packed_bytes(33, 128) == 2 * (32 * 128 * 2 + 128) + 128
    assert lib.keds_index_packed_bytes(10, 100) == 0               # unsupported dim
    assert lib.keds_index_search_workspace_bytes(128, 768) > 16 * 1024 * 1024
    # k up to 128 (256 candidates per query) and the chunk lists of the exact fallback grow the workspace
    assert lib.keds_index_search_workspace_bytes_ex(128, 768, 500000, 101) > lib.keds_index_search_workspace_bytes_ex(128, 768, 500000, 16)
    assert lib.keds_index_search_workspace_bytes_ex(128, 768, 500000, 129) == 0
    # rows padded to whole 256-row tiles (32,896 -> 33,024: the ragged last tile can run as a full one on filler rows)
    assert lib.keds_tower_workspace_bytes(1024, 257, 128) == (33024 * 1024 * 2 + 2 * 33024 * 4096 * 2 + 2 * 33024 * 16
                                                              + 6 * (32768 * 1024 + 32768 * 32)        # + MXFP8 operands
                                                              + (8 << 20))                             # + this call's split-K scratch
    # remainder rows of a tower pass run beside the full tiles (side lane): 128 of 32,896 at B = 128, none for a text tower
    assert lib.keds_tower_side_rows(1024, 257, 128, 0) in (0, 128)      # 0 with KEDS_SIDE_STREAM=0 (or KEDS_TOWER_FILL=1)
    assert lib.keds_tower_side_rows(768, 77, 128, 0) == 0


def test_argument_errors_are_reported():
    lib = _lib.load()
    rc = lib.keds_gemm_bt(None, None, None, None, 1, 1, 1, 0, None, 0, None)
    assert rc == -1 and "null" in _lib.last_error()
    rc = lib.keds_index_search_packed(1, 1, 10, 768, 0, 1, 4, 0, 129, 0, 1, 1, None, 1, 0, None)   # k > 128
    assert rc == -1 and "k must be" in _lib.last_error()
    rc = lib.keds_attention(1, 1, 1, 400, 16, 0, None)
    assert rc == -1 and "unsupported" in _lib.last_error()


@pytest.fixture(scope="module")
def ref_keys():
    return json.load(open(golden_path("state_dict_keys.json")))


def test_state_dict_contract_tiny_and_vitl(ref_keys):
    """Keys and shapes equal the reference modules' (SURVEY 8b), so reference .pt files load unchanged."""
    cfg = dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
               context_length=77, vocab_size=512, transformer_width=128, transformer_heads=2, transformer_layers=2)
    m = keds_amd.CLIP(**cfg)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref_keys["clip_tiny"]
    with torch.device("meta"):
        big = keds_amd.CLIP(768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12)
    assert {k: list(v.shape) for k, v in big.state_dict().items()} == ref_keys["clip_vitl14"]
    assert {k: list(v.shape) for k, v in keds_amd.IM2TEXT(768, 512, 768, 2).state_dict().items()} == ref_keys["im2text"]
    xf = keds_amd.CrossFormer(768, 768, 768, num_layers=3)
    assert {k: list(v.shape) for k, v in xf.state_dict().items()} == ref_keys["crossformer"]


def test_build_model_infers_architecture_and_loads():
    tiny = dict(embed_dim=128, image_resolution=56, vision_layers=2, vision_width=128, vision_patch_size=14,
                context_length=77, vocab_size=512, transformer_width=128, transformer_layers=2)
    sd = O.synth_clip_state_dict(**tiny, seed=7)
    m = keds_amd.build_model(dict(sd), fp16=False)
    assert m.visual.input_resolution == 56 and m.transformer.layers == 2 and m.end_id == 511
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    m16 = keds_amd.build_model(dict(sd))                      # reference default: fp16 weights
    assert m16.dtype == torch.float16
    keds_amd.convert_models_to_fp32(m16)
    assert m16.dtype == torch.float32
    # 4-part checkpoint dict with DDP 'module.' prefixes (main.py:330-341, eval_retrieval.py:175-182)
    a, b, c = keds_amd.make_stream_modules(m, middle_dim=128)
    ck = {"epoch": 3, "name": "x",
          "state_dict": {"module." + k: v for k, v in sd.items()},
          "state_dict_img2text": {"module." + k: v for k, v in O.synth_im2text_state_dict(128, 128, 128).items()},
          "state_dict_retrieval_fuse": O.synth_crossformer_state_dict(128, 3, tag="f"),
          "state_dict_text_condition": O.synth_crossformer_state_dict(128, 3, tag="c")}
    keds_amd.load_checkpoint(ck, m, a, b, c)
    assert torch.equal(a.fc_out.weight, ck["state_dict_img2text"]["module.fc_out.weight"])


def test_no_cpu_fallback():
    """The product path must fail loudly without the GPU (no silent eager fallback)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = keds_amd.CLIP(128, 56, 2, 128, 14, 77, 512, 128, 2, 2).eval()
    with pytest.raises(RuntimeError, match="no CPU"):
        m.encode_image(torch.zeros(1, 3, 56, 56))
    with pytest.raises(RuntimeError, match="no CPU"):
        m.encode_text(torch.tensor([[510] + [5] * 10 + [511] + [0] * 65]))
    with pytest.raises(RuntimeError, match="no CPU"):
        keds_amd.IM2TEXT(128, 128, 128).eval()(torch.zeros(2, 128))
    with pytest.raises(RuntimeError, match="no CPU"):
        keds_amd.CrossFormer(128, 128, 128, 3)(torch.zeros(2, 1, 128), torch.zeros(2, 16, 128), torch.zeros(2, 16, 128))
    idx = keds_amd.IndexFlatL2(128)
    with pytest.raises(RuntimeError, match="no CPU"):
        idx.add(np.zeros((4, 128), np.float32))


def test_text_argument_checks_match_reference_errors():
    m = keds_amd.CLIP(128, 56, 2, 128, 14, 77, 512, 128, 2, 2).eval()
    text = torch.zeros(2, 77, dtype=torch.int64)
    text[:, 0], text[:, 1], text[:, 2] = 510, 265, 511
    with pytest.raises(RuntimeError):                              # 4 pseudo tokens: wrong sequence length
        m.encode_text_img_retrieval(text, torch.zeros(2, 4, 128), split_ind=265, repeat=False)
    with pytest.raises(IndexError):                                # split token absent from row 0
        m.encode_text_img_retrieval(text, torch.zeros(2, 3, 128), split_ind=7, repeat=False)
    two = text.clone()
    two[0, 9] = 511
    with pytest.raises(IndexError):                                # two EOTs in a row
        m.encode_text_img_retrieval(two, torch.zeros(2, 3, 128), split_ind=265, repeat=False)
    late = text.clone()
    late[:, 2] = 7
    late[:, 76] = 511
    with pytest.raises(IndexError):                                # read-out beyond the context
        m.encode_text_img_retrieval(late, torch.zeros(2, 3, 128), split_ind=265, repeat=False)


# ---- multi-rank search logic ---------------------------------------------------------------------
def test_shard_bounds_cover_rows_once():
    for n in (1, 31, 32, 33, 1000, 500000, 2000000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and a % 32 == 0


def test_merge_partials_matches_global_search():
    db = O.synth_database(4000, 64, seed=1)
    q = O.synth_database(11, 64, seed=2)
    Dg, Ig = O.flat_l2_search(db, q, 16)
    for world in (2, 3, 8):
        Dp, Ip = [], []
        for r in range(world):
            lo, hi = shard_bounds(4000, world, r)
            d, i = O.flat_l2_search(db[lo:hi], q, 16)
            Dp.append(d)
            Ip.append(i + lo)
        D, I = merge_partials(torch.stack(Dp), torch.stack(Ip))
        assert torch.equal(I, Ig)
        assert torch.allclose(D, Dg)
    # ties across shards resolve to the lower id; -1 fillers sort last
    Dp = torch.tensor([[[0.5, 1.0]], [[0.5, float("inf")]]])
    Ip = torch.tensor([[[7, 9]], [[3, -1]]])
    D, I = merge_partials(Dp, Ip)
    assert I.tolist() == [[3, 7]] and D.tolist() == [[0.5, 0.5]]


def _gloo_worker(rank, world, port, out):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        db = O.synth_database(3000, 64, seed=9)
        q = O.synth_database(7, 64, seed=10)
        lo, hi = shard_bounds(3000, world, rank)
        d, i = O.flat_l2_search(db[lo:hi], q, 10)                # stand-in for the local HIP scan
        from keds_amd.index import exchange_and_merge
        D, I = exchange_and_merge(d, i + lo, _lib.METRIC_L2, group=None)
        Dg, Ig = O.flat_l2_search(db, q, 10)
        ok = bool(torch.equal(I, Ig) and torch.allclose(D, Dg))
        # data-parallel form with the winners' rows: each rank owns 4 of 8 queries; its shard answers all 8 and ships the
        # rows it found with its partial lists (knowledge-path neighbours)
        from keds_amd.index import exchange_merge_gather
        qall = O.synth_database(8, 64, seed=11)
        dp, ip = O.flat_l2_search(db[lo:hi], qall, 16)
        rows_p = db[lo:hi][ip.reshape(-1)].reshape(8, 16, 64)
        D2, I2, R2 = exchange_merge_gather(dp, ip + lo, rows_p, 4, _lib.METRIC_L2, group=None)
        Dg2, Ig2 = O.flat_l2_search(db, qall[rank * 4:(rank + 1) * 4], 16)
        ok = ok and bool(torch.equal(I2, Ig2) and torch.allclose(D2, Dg2) and torch.equal(R2, db[Ig2.reshape(-1)].reshape(4, 16, 64)))
        # the packed form bench.py --gpus N runs: one all-gather of the queries, ONE collective for the partial lists
        from keds_amd.index import PackedExchange
        x = PackedExchange()
        mine_q = qall[rank * 4:(rank + 1) * 4]
        allq = x.gather_queries(mine_q)
        ok = ok and bool(torch.equal(allq, qall))
        for _ in range(2):                                        # second round reuses the buffers
            D3, I3 = x.return_partials(dp, ip + lo, _lib.METRIC_L2)
            ok = ok and bool(torch.equal(I3, Ig2) and torch.allclose(D3, Dg2))
        # the knowledge path's packed form: TWO databases' partial lists and their rows in ONE all-to-all
        db2 = O.synth_database(3000, 64, seed=19)
        dp2, ip2 = O.flat_l2_search(db2[lo:hi], qall, 16)
        rows_p2 = db2[lo:hi][ip2.reshape(-1)].reshape(8, 16, 64)
        Dg3, Ig3 = O.flat_l2_search(db2, mine_q, 16)
        for _ in range(2):
            (D4, I4, R4), (D5, I5, R5) = x.return_partials_rows([(dp, ip + lo, rows_p), (dp2, ip2 + lo, rows_p2)],
                                                                _lib.METRIC_L2)
            ok = ok and bool(torch.equal(I4, Ig2) and torch.equal(D4, D2) and torch.equal(R4, R2))
            ok = ok and bool(torch.equal(I5, Ig3) and torch.allclose(D5, Dg3)
                             and torch.equal(R5, db2[Ig3.reshape(-1)].reshape(4, 16, 64)))
        # evaluation glue: ragged per-rank gallery slices gathered in rank order, then the metric on the full matrices
        from keds_amd.retrieval import all_gather_features
        gal = O.synth_database(101, 64, seed=12)
        cut = 37
        mine = gal[:cut] if rank == 0 else gal[cut:]
        ok = ok and bool(torch.equal(all_gather_features(mine), gal))
        out[rank] = ok
    finally:
        dist.destroy_process_group()


def test_sharded_search_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = 29500 + os.getpid() % 2000
        procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs)
        assert dict(out) == {0: True, 1: True}


def test_sharded_database_build_partitions_rows_without_a_gpu(tmp_path, monkeypatch):
    """Host logic of the per-rank database build (keds_amd.retrieval.extract_feature_database_sharded): every rank asks its
    loaders for exactly its own row range, in batches, and the ranges tile the dataset.  The encoders and the device index
    are stubbed (there is no CPU compute path); the file / manifest handling is the real code."""
    import json
    from keds_amd import retrieval
    asked = []

    class FakeIndex:
        def __init__(self, d, metric="l2", device=None, row0=0):
            self.d, self.row0, self.n = d, row0, 0
        def add(self, x):
            self.n += x.shape[0]
        def save(self, path):
            torch.save({"row0": self.row0, "n": self.n}, path)
    monkeypatch.setattr(retrieval, "FlatIndex", FakeIndex)
    n, world = 1000, 3
    covered = []
    for r in range(world):
        def rows(a, b, r=r):
            asked.append((r, a, b))
            return torch.zeros(b - a, 4)
        retrieval.extract_feature_database_sharded(None, n, rows, rows, str(tmp_path), rank=r, world=world, batch=128,
                                                   encode_image=lambda x: torch.zeros(x.shape[0], 128),
                                                   encode_text=lambda x: torch.zeros(x.shape[0], 128))
        lo, hi = shard_bounds(n, world, r)
        mine = [(a, b) for rr, a, b in asked if rr == r]
        assert mine[0][0] == lo and mine[len(mine) // 2 - 1][1] == hi and all(b - a <= 128 for a, b in mine)
        covered.append((lo, hi))
        meta = torch.load(str(tmp_path / f"cc_text_index.shard{r}-of-{world}.pt"))
        assert meta == {"row0": lo, "n": hi - lo}
    assert covered[0][0] == 0 and covered[-1][1] == n and all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
    man = json.load(open(str(tmp_path / "cc_database_shards.json")))
    assert man["n_rows"] == n and man["world"] == world and man["bounds"] == [list(c) for c in covered]


def _gather_worker(rank, world, port, out):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        from keds_amd.train import gather_own_first
        x = torch.full((3, 4), float(rank)) + torch.arange(3)[:, None] * 0.1
        got = gather_own_first(x)
        order = [rank] + [r for r in range(world) if r != rank]
        want = torch.cat([torch.full((3, 4), float(r)) + torch.arange(3)[:, None] * 0.1 for r in order])
        out[rank] = bool(torch.equal(got, want))
    finally:
        dist.destroy_process_group()


def test_training_negatives_gather_own_rows_first_gloo_world3():
    """trainer.py:78-99: every rank contrasts against the features of all ranks with ITS OWN rows first (the loss targets
    are arange); keds_amd.train.gather_own_first on a 3-rank gloo group."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = 29500 + (os.getpid() + 7) % 2000
        procs = [ctx.Process(target=_gather_worker, args=(r, 3, port, out)) for r in range(3)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs)
        assert dict(out) == {0: True, 1: True, 2: True}


def test_bench_gpus_n_launches_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus N` is the one command the scaling driver runs: outside torch.distributed.run the parent must
    start `python -m torch.distributed.run ... bench.py <same args>` as a child (never exec, never after a GPU call) and
    relay rank 0's JSON line as the last stdout line.  Dry run: the parent prints the argv / env it would start."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["KEDS_BENCH_LAUNCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    argv = plan["launch"]
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-7:] == [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert plan["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the real thing with a stub in place of torch.distributed.run: the child's lines are relayed, the JSON line comes last,
    # and the exit code is the child's
    stub = os.path.join(root, "tests", "_stub_site")
    os.makedirs(os.path.join(stub, "torch", "distributed"), exist_ok=True)
    try:
        open(os.path.join(stub, "torch", "__init__.py"), "w").write("")
        open(os.path.join(stub, "torch", "distributed", "__init__.py"), "w").write("")
        open(os.path.join(stub, "torch", "distributed", "run.py"), "w").write(
            "import sys, os\n"
            "assert '--nproc-per-node=2' in sys.argv and os.environ.get('WORLD_SIZE') is None\n"
            "print('{\"metric\": \"m\", \"value\": 1}')\n"
            "print('RCCL banner after the json line')\n"
            "sys.exit(7)\n")
        code = ("import sys, os; sys.argv = ['bench.py', '--gpus', '2']; import bench; "
                "os.environ['PYTHONPATH'] = %r; sys.exit(bench.self_launch(2, ['--gpus', '2']))" % stub)
        env2 = {k: v for k, v in env.items() if k != "KEDS_BENCH_LAUNCH_DRYRUN"}
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env2, capture_output=True, text=True, timeout=300)
        lines = r.stdout.strip().splitlines()
        assert r.returncode == 7 and lines[-1] == '{"metric": "m", "value": 1}' and "RCCL banner" in lines[0], (r.stdout, r.stderr)
    finally:
        import shutil
        shutil.rmtree(stub, ignore_errors=True)


def test_quad_gemm_kernels_keep_their_accumulators_to_the_generated_statements():
    """The 4-wave GEMM kernels (bf16: gemm.hip, MXFP8: gemm_fp8.hip) keep 256 accumulators in fixed AGPRs that the compiler is
    not told about (gemm_quad_gen.h, gemm_fp8_quad_gen.h): in the gfx950 assembly of every instantiation exactly 256 AGPRs are
    allocated, the compiler touches one only between its read-back and the next tile's first MFMA on it, and there is no scratch
    traffic inside the K-loop (tools/check_quad_asm.py; hipcc -S, no GPU).  The checker itself is tested on a doctored listing:
    a compiler write to a live accumulator must be reported."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_quad_asm.py")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "0 problem(s)" in res.stdout
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import check_quad_asm as cq
    finally:
        sys.path.pop(0)
    good = """
\t.globl\t_Z22gemm_bt_quad_kernel_demo
_Z22gemm_bt_quad_kernel_demo:
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], 0
\t;;#ASMEND
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]
\t;;#ASMEND
%s\t;;#ASMSTART
\tv_accvgpr_read_b32 v8, a0
\tv_accvgpr_read_b32 v9, a1
\tv_accvgpr_read_b32 v10, a2
\tv_accvgpr_read_b32 v11, a3
\t;;#ASMEND
%s\ts_endpgm
\t.amdhsa_accum_offset 256
\t.amdhsa_next_free_vgpr 512
"""
    def problems(text):
        return [e for e in cq.check(text)[1] if "v_accvgpr_read_b32 (expected 256" not in e]
    assert problems(good % ("", "")) == []
    assert problems(good % ("", "\tv_accvgpr_write_b32 a2, v20\n\tv_accvgpr_read_b32 v21, a2\n")) == []      # parked behind the read-back: fine
    bad = problems(good % ("\tv_accvgpr_write_b32 a2, v20\n", ""))
    assert bad and "live accumulator a2" in bad[0]


def test_extract_feature_database_streams_its_inputs_and_reencodes_a_late_tripped_chunk():
    """retrieval.extract_feature_database over a GENERATOR of batches (a DataLoader over 0.5 M images is 300 GB of pixels): at
    most VERIFY_CHUNK input batches are ever held, the numerics guard is synchronised behind every chunk, and a chunk during
    which it tripped late is encoded again from the inputs still held (advisor finding of round 4: the inputs used to be
    materialised with list()).  Host logic only: a stand-in model counts what it is asked to do."""
    import torch
    from keds_amd import retrieval

    alive = {"now": 0, "max": 0, "made": 0}

    class Batch(torch.Tensor):
        pass

    def gen(n):
        import weakref
        for i in range(n):
            t = torch.full((2, 4), float(i))
            alive["now"] += 1
            alive["made"] += 1
            alive["max"] = max(alive["max"], alive["now"])
            weakref.finalize(t, lambda: alive.__setitem__("now", alive["now"] - 1))
            yield t
            del t

    class Model:
        def __init__(self):
            self.calls, self.chunks, self.safe = 0, 0, False

        def encode_image(self, x, normalize=False):
            self.calls += 1
            return x[:, :3] + (0.0 if self.safe else 0.5 * (self.chunks == 1))      # the fast flow of chunk 1 is "wrong"

        encode_text = encode_image

        def numerics_checked(self, fn):
            out = fn()
            if self.chunks == 1 and not self.safe:      # the guard trips late during the second chunk: run it again, safely
                self.safe = True
                out = fn()
            self.chunks += 1
            return out

    m = Model()
    n = 3 * retrieval.VERIFY_CHUNK + 2
    ib, tb = retrieval.extract_feature_database(m, gen(n), gen(n))
    want = torch.cat([torch.full((2, 3), float(i)) for i in range(n)])
    assert torch.equal(ib, want) and torch.equal(tb, want)
    assert alive["made"] == 2 * n and alive["max"] <= retrieval.VERIFY_CHUNK + 1, alive
    assert m.calls == 2 * n + retrieval.VERIFY_CHUNK                     # one chunk encoded twice


def _run_bench_dry(gpus, extra_env=None, timeout=600):
    import subprocess
    env = dict(os.environ)
    env.update({"KEDS_BENCH_CPU_DRYRUN": "1", "OMP_NUM_THREADS": "1"})
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
                           "--db-rows", "4096", "--batch", "4"], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_eight_rank_flow_on_cpu():
    """`python bench.py --gpus 8` cold, the way the driver will start it on an 8-GPU node -- here with KEDS_BENCH_CPU_DRYRUN=1:
    the parent starts eight ranks as a child torch.distributed.run, they run THIS file's whole N = 8 flow (pilot of both search
    placements, packed query all-gather, shard search, packed all-to-all + merge, per-rank report, the self-verification against
    the oracle over all shards) on CPU tensors over gloo with stand-ins for the encoder and the shard scan, and rank 0's JSON
    line is the last line on stdout."""
    res = _run_bench_dry(8)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = res.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["config"]["db_shards"] == 8 and out["config"]["global_batch"] == 32
    assert out["value"] > 0 and out["scaling"] == "weak" and "CPU_DRYRUN" in out["diagnostic"]
    pr = out["per_rank_search"]
    assert all(len(pr[k]) == 8 for k in ("query_allgather_us", "local_search_us", "partials_alltoall_merge_us"))
    assert set(out["search_overlap_pilot"]) == {"serial_ms_per_step", "overlapped_ms_per_step"}
    v = out["verification"]
    assert v["ok"] and v["id_mismatches"] == 0 and v["rows_searched"] == 4096 and v["queries"] == 4
    assert out["stage_ms_per_step"]["attention"] is None            # no events: null, not 0.0
    assert out["recall_parity_measured_in_this_run"] is False


def test_bench_rank_failure_exits_nonzero_with_the_failing_ranks_message():
    """A rank that dies takes the whole job down with a non-zero exit code and its own traceback on stderr (no JSON line is
    printed for a job that did not finish)."""
    res = _run_bench_dry(4, {"KEDS_BENCH_DRYRUN_FAIL_RANK": "3"})
    assert res.returncode != 0
    assert "rank 3 fails on purpose" in res.stderr
    assert '"metric"' not in res.stdout


def test_bench_self_verification_flags_a_wrong_neighbour_list():
    """KEDS_BENCH_INJECT_FAULT=2 swaps two neighbours of one query in the last timed step's result: the line still prints, says
    verification.ok = false with the mismatch count, and the process exits 3 -- a speed for wrong results is not a result."""
    res = _run_bench_dry(2, {"KEDS_BENCH_INJECT_FAULT": "2"})
    assert res.returncode != 0, res.stdout[-1500:]
    out = json.loads([l for l in res.stdout.strip().splitlines() if l.startswith("{") and '"metric"' in l][-1])
    assert out["verification"]["ok"] is False and out["verification"]["id_mismatches"] == 2
