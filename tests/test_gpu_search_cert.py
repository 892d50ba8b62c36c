"""GPU parity of the search's exactness machinery (IndexFlatL2 is EXACT brute force, src/eval_retrieval.py:291):

* the per-query certificate (bf16 candidate selection + rigorous rounding bound vs the exact k-th candidate distance),
* the exact fp32 fallback for queries that fail it (near-duplicate clusters wider than the candidate list),
* k up to 128 (gallery ranking needs the best 101, src/eval_utils.py:1040-1067),
* chunked `.add` (pack only the new stages).

Bar: indices equal to the oracle's exact fp64 search (near-tie swaps below 1e-6 in distance tolerated as in
test_gpu_search._check), distances within 2e-6.
"""
import numpy as np
import pytest
import torch

import keds_amd
from keds_amd import _lib
from oracle import keds_oracle as O
from tests.gpu_util import max_abs, report
from tests.test_gpu_search import _check

pytestmark = pytest.mark.gpu


def test_certificate_holds_on_ordinary_data():
    """iid unit-norm rows: every query's top-16 is certified from the 64 re-ranked candidates.  Clustered rows (512
    clusters of ~117 rows, sigma 0.15) seen from far-away queries are legitimately harder -- the neighbours of a query
    sit inside one cluster at nearly equal distances -- and a minority of queries takes the exact pass; results are exact
    either way."""
    for clustered in (False, True):
        db = O.synth_database(60000, 768, seed=2002, clustered=clustered, n_centroids=512)
        q = O.synth_database(96, 768, seed=3003)
        idx = keds_amd.FlatIndex(768, "l2")
        idx.add(db)
        _check(idx, db, q, 16, f"cert_ordinary_clustered{int(clustered)}")
        ok, fb = idx.certificate_counts(reset=True)
        report("certificate.ordinary", clustered=clustered, certified=ok, fallback=fb)
        assert ok + fb == 96
        assert fb == 0 if not clustered else fb <= 24, "ordinary data must rarely need the exact pass"


@pytest.mark.parametrize("contiguous", [True, False])
def test_near_duplicate_cluster_takes_the_exact_pass(contiguous):
    """300 rows within 1e-4 (squared distance) of each other around the query -- far more than the 64 candidates and
    closer together than bf16 scores can tell apart: the bf16 ranking inside the cluster is noise, so the candidate list
    cannot be certified and the exact pass must answer; its result equals the oracle's."""
    n, dim, m = 40000, 768, 300
    db = O.synth_database(n, dim, seed=2002)
    rs = np.random.RandomState(99)
    c = db[12345].clone()
    dup = O.l2_normalize(c[None, :] + 0.01 / np.sqrt(dim) * torch.from_numpy(rs.standard_normal((m, dim)).astype(np.float32)))
    where = torch.arange(20000, 20000 + m) if contiguous else torch.from_numpy(rs.choice(n, size=m, replace=False))
    db[where] = dup
    q = torch.cat([O.l2_normalize(c[None, :] + 0.002 / np.sqrt(dim) * torch.from_numpy(rs.standard_normal((4, dim)).astype(np.float32))),
                   O.synth_database(28, dim, seed=3003)])
    idx = keds_amd.FlatIndex(dim, "l2")
    idx.add(db)
    D, I = _check(idx, db, q, 16, f"cert_cluster_contiguous{int(contiguous)}")
    ok, fb = idx.certificate_counts(reset=True)
    report("certificate.cluster", contiguous=contiguous, certified=ok, fallback=fb)
    assert ok + fb == 32
    assert fb >= 4, "the four cluster queries cannot be certified from 64 bf16-ranked candidates"
    allowed = torch.cat([where, torch.tensor([12345])])          # (when it was not overwritten, the centre row is a neighbour too)
    assert bool(torch.isin(I[:4], allowed).all()), "cluster queries must return cluster rows"
    # the candidate-only answer really would have been wrong for at least one of them: bf16 cannot rank the cluster
    qq, dd = q[:4].bfloat16().float(), db[where].bfloat16().float()
    s16 = qq @ dd.T - 0.5 * (db[where] ** 2).sum(1)[None, :]
    top64 = where[s16.topk(64, dim=1).indices]
    missed = sum(int((~torch.isin(I[r], top64[r])).sum()) for r in range(4))
    report("certificate.cluster.bf16_top64_misses", missed=missed)


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_exact_pass_alone_equals_oracle(metric):
    """Every certificate forced to fail: the exact chunk / merge kernels answer all queries (k = 16 and k = 101)."""
    lib = _lib.load()
    db = O.synth_database(70001, 256, seed=21) * (1.3 if metric == "ip" else 1.0)
    q = O.synth_database(40, 256, seed=22)
    idx = keds_amd.FlatIndex(256, metric)
    idx.add(db)
    try:
        lib.keds_scan_debug(32)
        for k in (16, 101):
            D, I, _ = idx.search_device(q.cuda(), k)
            Do, Io = (O.flat_l2_search if metric == "l2" else O.flat_ip_search)(db, q, k)
            mism = int((I.cpu() != Io).sum())
            report("exact_pass", metric=metric, k=k, index_mismatches=mism, d_maxabs=max_abs(D, Do))
            assert max_abs(D, Do) <= 2e-6
            assert mism <= 2                       # fp32-vs-fp64 near-tie swaps only
        ok, fb = idx.certificate_counts(reset=True)
        assert ok == 0 and fb == 80
    finally:
        lib.keds_scan_debug(0)


@pytest.mark.parametrize("n,dim,nq,k", [(50000, 256, 33, 17), (50000, 256, 33, 50), (30000, 768, 20, 101),
                                        (1000, 768, 256, 101),          # config 1's gallery: 1 k rows, best 101
                                        (62500, 768, 200, 128)])
def test_k_up_to_128(n, dim, nq, k):
    db = O.synth_database(n, dim, seed=2002)
    q = O.synth_database(nq, dim, seed=3003)
    idx = keds_amd.FlatIndex(dim, "l2")
    idx.add(db)
    # (more neighbours per query = more chances of two exact distances within fp32 rounding of each other: every
    # differing row is still verified to be such a near-tie swap by _check)
    _check(idx, db, q, k, f"search_k{k}", max_swap_rows=max(2, nq // 8))
    ok, fb = idx.certificate_counts(reset=True)
    report("certificate.large_k", n=n, k=k, certified=ok, fallback=fb)
    assert ok + fb == nq


def test_ip_top101_of_a_gallery():
    """Inner-product top-101 (what the recall metrics read of a gallery ranking)."""
    gal = O.synth_database(5000, 768, seed=41)
    q = O.synth_database(64, 768, seed=42)
    idx = keds_amd.FlatIndex(768, "ip")
    idx.add(gal)
    D, I, _ = idx.search_device(q.cuda(), 101)
    Do, Io = O.flat_ip_search(gal, q, 101)
    assert max_abs(D, Do) <= 2e-6
    assert int((I.cpu() != Io).sum()) <= 2


def test_chunked_add_packs_only_new_stages():
    """Adds of ragged sizes (not multiples of the 32-row stage) give byte-for-byte the image of one add."""
    db = O.synth_database(10000, 256, seed=5)
    one = keds_amd.FlatIndex(256, "l2")
    one.add(db)
    many = keds_amd.FlatIndex(256, "l2")
    cuts = [0, 5, 37, 64, 1000, 1001, 4097, 10000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        many.add(db[a:b])
    assert many.ntotal == 10000
    assert torch.equal(many.rows, one.rows)
    assert torch.equal(many.packed.cpu(), one.packed.cpu()), "packed images differ"
    q = O.synth_database(9, 256, seed=6)
    D1, I1, _ = one.search_device(q.cuda(), 16)
    D2, I2, _ = many.search_device(q.cuda(), 16)
    assert torch.equal(I1, I2) and torch.equal(D1, D2)
