"""One of TWO processes that share cuda:0 and run the DEVICE side of the row-sharded search against each other
(tests/test_gpu_exchange2.py; started by tests/conftest.py before the pytest process touches the GPU).

Each process owns one shard of two synthetic databases on the device and its own 8 queries.  The torch.distributed transport
is a gloo group whose `all_gather_into_tensor` / `all_to_all_single` are patched to stage device tensors through host memory
(two ranks cannot share one device under RCCL), so everything AROUND the collectives is the product's CUDA branch, moving
data between two real processes: `PackedExchange.gather_queries` -> shard scan (`search_device` / `search_gather`) ->
`ops.exchange_pack` -> transport -> `ops.exchange_merge` (keds_amd/index.py, the branch bench.py --gpus N and
`ShardedFlatIndex.search_gather_many` run).  Results must equal a single index over all rows, bit for bit.

    python tests/exchange2_worker.py RANK WORLD PORT_OR_INIT_URL OUT_DIR"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    init = port if "://" in port else f"tcp://127.0.0.1:{port}"     # a file:// rendezvous in the job's directory, or a TCP port
    res = {"rank": rank, "ok": False}
    try:
        import torch
        import torch.distributed as dist
        import datetime
        dist.init_process_group("gloo", init_method=init, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=240))      # a peer that died must not hang the suite
        import keds_amd
        from keds_amd import _lib
        from keds_amd.index import PackedExchange, ShardedFlatIndex, install_host_staged_transport, shard_bounds
        install_host_staged_transport(dist)
        from oracle import keds_oracle as O
        _lib.load()
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        N, D_, B = 40000, 768, 8
        checks = {}
        dbs = [O.synth_database(N, D_, seed=s).to(dev) for s in (2002, 2003)]
        qall = O.synth_database(world * B, D_, seed=3003).to(dev)
        mine = qall[rank * B:(rank + 1) * B].contiguous()
        lo, hi = shard_bounds(N, world, rank)
        full = []
        shards = []
        for db in dbs:
            f = keds_amd.FlatIndex(D_, "l2", device=dev)
            f.add(db)
            full.append(f)
            s = ShardedFlatIndex(D_, "l2", device=dev)
            s.add_global(db)
            shards.append(s)
        assert shards[0].world == world and shards[0].local.row0 == lo and shards[0].local.ntotal == hi - lo
        # (1) bench.py --gpus N: query all-gather, shard scan, pack -> all-to-all -> merge on the owner; buffers reused
        x = PackedExchange()
        for rnd in range(2):
            allq = x.gather_queries(mine)
            checks[f"gather_queries.{rnd}"] = bool(torch.equal(allq, qall))
            D_p, I_p, _ = shards[0].local.search_device(allq, 10)
            assert D_p.is_cuda
            Dm, Im = x.return_partials(D_p, I_p, shards[0].local.metric)
            Dg, Ig, _ = full[0].search_device(mine, 10)
            checks[f"return_partials.{rnd}"] = bool(torch.equal(Im, Ig) and torch.equal(Dm, Dg))
        # (2) search_own / search_gather on the sharded index
        Do, Io = shards[1].search_own(mine, 10)
        Dg, Ig, _ = full[1].search_device(mine, 10)
        checks["search_own"] = bool(torch.equal(Io, Ig) and torch.equal(Do, Dg))
        # (3) the knowledge path: two databases, top-16 with rows, ONE all-gather + ONE all-to-all
        for rnd in range(2):
            got = ShardedFlatIndex.search_gather_many(shards, mine, 16, normalize=True)
            for j, (Dk, Ik, Rk) in enumerate(got):
                Dg, Ig, Rg = full[j].search_gather(mine, 16, normalize=True)
                checks[f"search_gather_many.{rnd}.db{j}"] = bool(Dk.is_cuda and torch.equal(Ik, Ig) and torch.equal(Dk, Dg)
                                                                 and torch.equal(Rk, Rg))
        # (4) every winner the owner received from the OTHER process really lives there
        Dk, Ik, _ = got[0]
        other = ((Ik < lo) | (Ik >= hi)).sum().item()
        checks["winners_from_the_other_shard"] = other > 0
        torch.cuda.synchronize()
        res.update(ok=all(checks.values()), checks=checks, rows_from_other_shard=int(other), shard=[lo, hi])
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                                   # noqa: BLE001
        res["error"] = traceback.format_exc()[-3000:]
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
