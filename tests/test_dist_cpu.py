"""world_size-2 gloo test of the N>1 path: page sharding + result gather."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ocrs_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = D.shard_pages(7, rank, world)
    local = {str(p): ["page %d line %d €" % (p, i) for i in range(p % 3 + 1)] for p in pages}
    allr = D.gather_results(local)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, pages, allr))


def test_shard_and_gather_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] == [0, 2, 4, 6] and results[1][1] == [1, 3, 5]
    for rank, pages, allr in results:
        assert len(allr) == 2
        merged = {}
        for part in allr:
            merged.update(part)
        assert sorted(int(k) for k in merged) == list(range(7))
        assert merged["5"] == ["page 5 line 0 €", "page 5 line 1 €", "page 5 line 2 €"]


def test_single_process_gather_is_identity():
    sys.path.insert(0, ROOT)
    from ocrs_amd import dist as D
    assert D.gather_results({"0": ["a"]}) == [{"0": ["a"]}]
    assert D.shard_pages(5, 0, 1) == [0, 1, 2, 3, 4]
