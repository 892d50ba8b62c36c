"""GPU parity of the handle-based C ABI (include/keds_session.h): weights handed over as named host arrays
(reference state_dict keys), library-owned packing / workspaces / database.

Bars: identical bits to the torch-hosted façade (same kernels underneath), the usual fp tolerance against the
golden vectors minted from the reference, exact indices against the oracle for the index, and the RCCL
all-gather path exercised with a 1-rank communicator.
"""
import numpy as np
import pytest
import torch

import keds_amd
from keds_amd import _lib, session
from oracle import keds_oracle as O
from tests.conftest import golden_path
from tests.gpu_util import assert_parity, max_abs, min_cosine, rel_l2, report
from tests.test_gpu_model import COS_MIN, REL_MAX, TINY, _streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = session.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def tiny():
    g = dict(np.load(golden_path("clip_tiny.npz")))
    sd = O.synth_clip_state_dict(**TINY, seed=7)
    m = keds_amd.build_model(dict(sd), fp16=False).cuda()
    return g, sd, m


def _close(name, got, want):
    assert_parity(name, got, want)


def test_vit_handle_matches_facade_and_golden(ctx, tiny):
    g, sd, m = tiny
    vit = session.Vit(ctx, {k: v.numpy() for k, v in sd.items()})          # host pointers
    assert (vit.width, vit.layers, vit.resolution, vit.patch, vit.embed_dim) == (128, 2, 56, 14, 128)
    img = torch.from_numpy(g["image"]).cuda()
    out = vit.forward(img)
    _close("session.vit.golden", out, g["encode_image"])
    assert torch.equal(out, m.encode_image(img)), "handle path and facade path must be the same bits"
    # growing the batch re-allocates the handle's workspace
    rs = np.random.RandomState(5)
    big = torch.from_numpy(rs.standard_normal((9, 3, 56, 56)).astype(np.float32))
    _close("session.vit.B9", vit.forward(big.cuda()), O.encode_image(sd, big))
    # fp16 images are converted on the device
    h = vit.forward(img.half())
    _close("session.vit.f16_image", h, g["encode_image"])
    vit.close()


def test_vit_handle_accepts_fp16_and_device_weights(ctx, tiny):
    g, sd, m = tiny
    half = {k: v.half().cuda() for k, v in sd.items() if k.startswith("visual.")}      # convert_weights checkpoints
    vit = session.Vit(ctx, half)
    out = vit.forward(torch.from_numpy(g["image"]).cuda())
    c = min_cosine(out, g["encode_image"])
    report("session.vit.f16_weights", min_cosine=c)
    assert c >= 0.999


def test_vit_create_reports_missing_and_misshapen_weights(ctx, tiny):
    g, sd, m = tiny
    bad = {k: v.numpy() for k, v in sd.items() if k != "visual.ln_post.bias"}
    with pytest.raises(RuntimeError, match="visual.ln_post.bias"):
        session.Vit(ctx, bad)
    bad = {k: v.numpy() for k, v in sd.items()}
    bad["visual.transformer.resblocks.1.mlp.c_fc.weight"] = bad["visual.transformer.resblocks.1.mlp.c_fc.weight"][:-1]
    with pytest.raises(RuntimeError, match="c_fc.weight"):
        session.Vit(ctx, bad)


def test_text_handle_matches_facade_and_golden(ctx, tiny):
    g, sd, m = tiny
    txt = session.Text(ctx, {k: v.numpy() for k, v in sd.items()})
    assert (txt.width, txt.layers, txt.context, txt.vocab, txt.embed_dim) == (128, 2, 77, 512, 128)
    text = torch.from_numpy(g["text"]).cuda()
    eot = (text == 511).int().argmax(dim=1)
    out = txt.forward(text, eot)
    _close("session.text.golden", out, g["encode_text"])
    assert torch.equal(out, m.encode_text(text))
    star = int(g["star"])
    ins = int((text[0] == star).nonzero()[0])
    tok3 = torch.from_numpy(g["tok3"]).cuda()
    out3 = txt.forward(text, eot + 2, tok3, ins)
    _close("session.text.eti3", out3, g["eti3"])
    assert torch.equal(out3, m.encode_text_img_retrieval(text, tok3, split_ind=star, repeat=False))
    tok2 = torch.from_numpy(g["tok2"]).cuda()
    _close("session.text.eti2", txt.forward(text, eot + 1, tok2, ins), g["eti2"])
    with pytest.raises(RuntimeError, match="pseudo tokens"):
        txt.forward(text, eot, torch.zeros(4, 4, 128, device="cuda"), ins)
    # read-out columns on the HOST: keds_text_forward_packed lays the captions' rows out back to back (round 6); ragged captions so
    # that packing really happens -- same bits as the facade (which packs too) and as the rectangular handle call on this model
    rs = np.random.RandomState(4)
    B, L = 24, 77
    rag = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        e = int(rs.randint(6, 60))
        rag[b, :e] = rs.randint(1, 500, size=e)
        rag[b, 0], rag[b, 3], rag[b, e] = 510, star, 511
        rag[b, 1:e][rag[b, 1:e] == star] = star + 1
        rag[b, 3] = star
    rag = torch.from_numpy(rag)
    eot_h = (rag == 511).int().argmax(dim=1)
    tok3b = torch.from_numpy(rs.standard_normal((B, 3, 128)).astype(np.float32) * 0.05).cuda()
    packed = txt.forward(rag.cuda(), eot_h + 2, tok3b, 3)                # host read-out columns -> packed rows
    rect = txt.forward(rag.cuda(), (eot_h + 2).cuda(), tok3b, 3)         # device read-out columns -> the rectangular cut
    assert torch.isfinite(packed).all() and torch.equal(packed, rect)
    assert torch.equal(packed, m.encode_text_img_retrieval(rag.cuda(), tok3b, split_ind=star, repeat=False))


@pytest.mark.parametrize("dim,middle", [(128, 128), (768, 512)])
def test_knowledge_handle_matches_facade(ctx, dim, middle):
    i2t = O.synth_im2text_state_dict(dim, middle, dim, 2, seed=21, tag="i2t")
    fuse = O.synth_crossformer_state_dict(dim, 3, seed=21, tag="fuse")
    cond = O.synth_crossformer_state_dict(dim, 3, seed=21, tag="cond")
    kn = session.Knowledge(ctx, i2t, fuse, cond)
    a = keds_amd.IM2TEXT(dim, middle, dim, 2).eval()
    b = keds_amd.CrossFormer(dim, dim, dim, num_layers=3).eval()
    c = keds_amd.CrossFormer(dim, dim, dim, num_layers=3).eval()
    a.load_state_dict(i2t), b.load_state_dict(fuse), c.load_state_dict(cond)
    facade = keds_amd.KnowledgeStream(a.cuda(), b.cuda(), c.cuda())
    B, K = 9, 16
    q = O.synth_database(B, dim, seed=1).cuda()
    ni = O.synth_database(B * K, dim, seed=2).reshape(B, K, dim).cuda()
    nt = O.synth_database(B * K, dim, seed=3).reshape(B, K, dim).cuda()
    got = kn.forward(q, ni, nt)
    assert torch.equal(got, facade(q, ni, nt))
    want = O.knowledge_tokens(i2t, fuse, cond, q.cpu(), ni.cpu(), nt.cpu())
    _close(f"session.knowledge.d{dim}", got, want)


def test_index_handle_add_search_gather(ctx):
    n, dim = 20000, 768
    db = O.synth_database(n, dim, seed=2002)
    q = O.synth_database(37, dim, seed=3003)
    idx = session.Index(ctx, dim)
    idx.add(db[:12345].numpy())                 # host pointer, two incremental adds
    idx.add(db[12345:].cuda())                  # device pointer
    assert idx.ntotal == n
    D, I, rows = idx.search(q.cuda(), 16, gather=True)
    Do, Io = O.flat_l2_search(db, q, 16)
    assert torch.equal(I.cpu(), Io)
    assert max_abs(D, Do) <= 2e-6
    assert torch.equal(rows.cpu(), db[Io.reshape(-1)].reshape(37, 16, dim))
    ref = keds_amd.FlatIndex(dim)
    ref.add(db)
    D2, I2, _ = ref.search_device(q.cuda(), 16)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    with pytest.raises(RuntimeError):
        session.Index(ctx, 100)
    with pytest.raises(RuntimeError, match="empty"):
        session.Index(ctx, dim).search(q.cuda(), 4)


def test_sharded_search_through_rccl_single_rank():
    """keds_comm_init + keds_index_search_sharded with a 1-rank RCCL communicator: both all-gathers and the merge run;
    with a row offset the ids are global.  (world > 1 needs more GPUs than a test box has; the exchange/merge logic
    for 2 ranks is covered on CPU with gloo in test_host_cpu.)"""
    c = session.Context(0)
    try:
        c.comm_init(0, 1, session.Context.comm_unique_id())
        n, dim = 9000, 256
        db = O.synth_database(n, dim, seed=5)
        q = O.synth_database(20, dim, seed=6).cuda()
        idx = session.Index(c, dim, row0=1000)
        idx.add(db.numpy())
        D, I = idx.search(q, 10, sharded=True)
        D1, I1 = idx.search(q, 10)
        assert torch.equal(D, D1) and torch.equal(I, I1)
        Do, Io = O.flat_l2_search(db, q.cpu(), 10)
        assert torch.equal(I.cpu(), Io + 1000)
        # round 3: k up to 128 and the winners' rows shipped with the partial lists (packed all-to-all + owner merge)
        D3, I3, R3 = idx.search(q, 101, gather=True, sharded=True)
        D4, I4, R4 = idx.search(q, 101, gather=True)
        assert torch.equal(D3, D4) and torch.equal(I3, I4) and torch.equal(R3, R4)
        Do, Io = O.flat_l2_search(db, q.cpu(), 101)
        assert torch.equal(I3.cpu(), Io + 1000)
        assert torch.equal(R3.cpu(), db[Io.reshape(-1)].reshape(20, 101, dim))
        idx.close()
    finally:
        c.close()


def test_exchange_pack_merge_kernels_emulate_three_shards():
    """keds_exchange_pack / keds_exchange_merge (the two kernels around the all-to-all of a sharded search) on one GPU: three
    shards' partial lists for 2 x 3 ranks' queries are packed, the blocks addressed to rank r are laid side by side as the
    all-to-all would deliver them, and the merge must reproduce the single-index search -- ids, distances, rows, ties on
    the smaller id, fewer than k valid entries."""
    from keds_amd import ops
    from keds_amd.index import shard_bounds
    n, dim, B, k, W = 3000, 128, 5, 16, 3
    db = O.synth_database(n, dim, seed=21)
    db[1500] = db[7]                                             # an exact duplicate on another shard: tie -> smaller id
    qall = O.synth_database(W * B, dim, seed=22)
    qall[0] = db[7]
    whole = keds_amd.FlatIndex(dim)
    whole.add(db)
    Dw, Iw, Rw = whole.search_gather(qall.cuda(), k)
    shards = []
    for r in range(W):
        lo, hi = shard_bounds(n, W, r)
        ix = keds_amd.FlatIndex(dim, row0=lo)
        ix.add(db[lo:hi])
        shards.append(ix)
    for with_rows in (True, False):
        E = dim + 4 if with_rows else 3
        sends = []
        for ix in shards:
            Dp, Ip, Rp = ix.search_gather(qall.cuda(), k)
            send = torch.zeros((W, B, k, E), dtype=torch.int32, device="cuda")
            ops.exchange_pack(Dp, Ip, Rp if with_rows else None, W, send, B * k * E)
            sends.append(send)
        for r in range(W):                                       # rank r receives block r of every shard
            recv = torch.stack([sends[s][r] for s in range(W)]).contiguous()
            D, I, R = ops.exchange_merge(recv, W, B, k, dim, B * k * E, _lib.METRIC_L2, with_rows)
            assert torch.equal(I, Iw[r * B:(r + 1) * B]) and torch.equal(D, Dw[r * B:(r + 1) * B])
            if with_rows:
                assert torch.equal(R, Rw[r * B:(r + 1) * B])
    assert Iw[0, 0].item() == 7 and Iw[0, 1].item() == 1500
    # fewer valid entries than k: a 40-row database split three ways, k = 16 per shard list but only 40 rows in all
    small = db[:40].clone()
    parts = []
    for r in range(W):
        lo, hi = shard_bounds(40, W, r)
        if hi > lo:
            ix = keds_amd.FlatIndex(dim, row0=lo)
            ix.add(small[lo:hi])
            parts.append(ix.search_gather(qall[:B].cuda(), 16))
    Wn = len(parts)
    recv = torch.zeros((Wn, B, 16, dim + 4), dtype=torch.int32, device="cuda")
    for j, (Dp, Ip, Rp) in enumerate(parts):
        ops.exchange_pack(Dp, Ip, Rp, 1, recv[j], B * 16 * (dim + 4))
    D, I, R = ops.exchange_merge(recv, Wn, B, 16, dim, B * 16 * (dim + 4), _lib.METRIC_L2, True)
    Do, Io = O.flat_l2_search(small, qall[:B], 16)
    assert torch.equal(I.cpu(), Io) and max_abs(D, Do) <= 2e-6


def test_index_handle_chunked_add_is_byte_identical_to_one_add(ctx):
    """keds_index_add appends in place (geometric capacity, only the new stages packed): a build in ragged chunks must give
    the same scan image and the same results, bit for bit, as one add of all rows -- including chunks that end inside a
    32-row stage and growth steps that move the buffers."""
    n, dim = 7001, 256
    db = O.synth_database(n, dim, seed=31)
    q = O.synth_database(9, dim, seed=32).cuda()
    one = session.Index(ctx, dim)
    one.add(db.numpy())
    many = session.Index(ctx, dim)
    cuts = [0, 1, 31, 32, 33, 500, 513, 2048, 2049, 4097, 7000, 7001]
    for a, b in zip(cuts, cuts[1:]):
        many.add(db[a:b].cuda() if (a % 2) else db[a:b].numpy())
    assert many.ntotal == one.ntotal == n
    for k in (1, 16, 101):
        D1, I1, R1 = one.search(q, k, gather=True)
        D2, I2, R2 = many.search(q, k, gather=True)
        assert torch.equal(D1, D2) and torch.equal(I1, I2) and torch.equal(R1, R2)
    ref = keds_amd.FlatIndex(dim)
    ref.add(db)
    img1, img2 = one.scan_image(), many.scan_image()
    assert torch.equal(img1, img2) and torch.equal(img1, ref.packed)
    one.close(); many.close()


def test_two_threads_two_handles_one_device_share_the_side_lane(tiny):
    """keds_session.h allows different handles on different threads.  The towers' side lane (remainder-row chain on a second
    stream) has ONE fork / join event pair per device: record + wait must be atomic per caller, or one thread's wait binds
    to the other's record and its remainder rows read unfinished data (round-2 advisor finding).  Two threads, two ViT
    handles, each on its own stream, many passes: every output must equal the single-threaded one."""
    import threading
    g, sd, m = tiny
    c = session.Context(0)
    vits = [session.Vit(c, {k: v.numpy() for k, v in sd.items()}) for _ in range(2)]
    rs = np.random.RandomState(5)
    imgs = [torch.from_numpy(rs.standard_normal((37, 3, 56, 56)).astype(np.float32)).cuda() for _ in range(2)]
    want = [vits[i].forward(imgs[i]).clone() for i in range(2)]
    torch.cuda.synchronize()
    bad = [0, 0]

    def work(i):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(60):
                y = vits[i].forward(imgs[i])
                st.synchronize()
                if not torch.equal(y, want[i]):
                    bad[i] += 1
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert bad == [0, 0]
    for v in vits:
        v.close()
    c.close()


def test_handle_lifetimes_and_batch_sequences(tiny):
    """Handles survive growing / shrinking batches, several handles share a context, and destroying the context before its
    handles does not crash (handles keep their own device id)."""
    g, sd, m = tiny
    c = session.Context(0)
    a = session.Vit(c, {k: v.numpy() for k, v in sd.items()})
    b = session.Vit(c, {k: v.cuda() for k, v in sd.items()})
    rs = np.random.RandomState(2)
    for B in (1, 9, 2, 33, 1):
        img = torch.from_numpy(rs.standard_normal((B, 3, 56, 56)).astype(np.float32)).cuda()
        ya, yb = a.forward(img), b.forward(img)
        assert torch.equal(ya, yb) and torch.equal(ya, m.encode_image(img))
    idx = session.Index(c, 128)
    db = O.synth_database(700, 128, seed=3)
    for lo, hi in ((0, 1), (1, 33), (33, 700)):
        idx.add(db[lo:hi].numpy())
    D, I = idx.search(db[:5].cuda(), 3)
    assert I[:, 0].tolist() == [0, 1, 2, 3, 4]
    c.close()                                                       # context first ...
    a.close(); b.close(); idx.close()                               # ... handles afterwards
