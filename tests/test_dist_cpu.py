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


def _run_bench_selftest(n, extra=()):
    """`python bench.py --gpus N` with no launcher must itself become N ranks (torch.distributed.run on 127.0.0.1);
    --dist-selftest runs the same rank plumbing (rendezvous, page sharding, result gather, MAX/SUM reductions)
    with fake results and the gloo backend, so it runs without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OCRS_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-selftest"] + list(extra),
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout   # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_two_ranks_by_itself():
    d = _run_bench_selftest(2)
    assert d["n_gpus"] == 2 and d["complete"] and d["elapsed_is_max"] and d["gathered_pages"] == d["pages"] == 16


def test_bench_spawns_four_ranks_and_shards_a_stream_round_robin():
    d = _run_bench_selftest(4, ["--stream-pages", "30"])   # configs[4] sharding: page i -> rank i mod 4, ragged tail
    assert d["n_gpus"] == 4 and d["complete"] and d["pages"] == 30 and d["gathered_pages"] == 30
    assert int(d["layout_threads"]) >= 1 and int(d["omp_threads"]) >= 1
