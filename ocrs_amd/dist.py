"""Page-sharded multi-GPU operation: pages are independent, so
each rank (one process per GPU) runs the whole pipeline on its own shard with
no collective on the compute path.  The only exchange is the final gather of
the variable-length results to rank 0 — `torch.distributed` all_gather
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests)."""
import json

import numpy as np
import torch
import torch.distributed as dist


def shard_pages(n_pages, rank, world):
    """Round-robin: page i -> rank i mod world."""
    return list(range(rank, n_pages, world))


def gather_results(local_results, device=None):
    """local_results: JSON-serialisable per-rank payload (e.g. {page_id: [line strings]}).
    Returns the list of all ranks' payloads on every rank (rank order)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_results]
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    raw = np.frombuffer(json.dumps(local_results, ensure_ascii=False).encode("utf-8"), dtype=np.uint8)
    n = torch.tensor([raw.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=device)
    buf[: raw.size] = torch.from_numpy(raw.copy()).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = []
    for s, b in zip(sizes, bufs):
        k = int(s.item())
        out.append(json.loads(bytes(b[:k].cpu().numpy()).decode("utf-8")) if k else None)
    return out
