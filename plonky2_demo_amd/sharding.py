"""Multi-GPU plumbing for batches of independent proofs: one process per GPU, circuit data replicated,
proof i goes to rank i mod world, no data-path collective; the only collective is one all_gather of fixed-size
per-proof records (Merkle caps: 3 x 16 x 4 u64 = 1536 B) at the end.  Backend-agnostic (`nccl` = RCCL over xGMI on
the GPU box, `gloo` in the CPU tests)."""
import numpy as np

CAP_WORDS = 3 * 16 * 4     # wires, Z/partial-products, quotient caps of one proof


def proofs_for_rank(num_proofs, rank, world):
    """Indices of the proofs rank `rank` proves: round-robin, so ragged batches differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, num_proofs, world))


def gather_caps(local_caps, num_proofs, device=None):
    """all_gather the caps of the locally proved proofs; returns [num_proofs][3][16][4] uint64 in proof order.

    local_caps: array [len(proofs_for_rank)][3][16][4].  Ranks may hold different counts (ragged): records are padded
    to ceil(num_proofs / world) before the collective."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (num_proofs + world - 1) // world
    mine = proofs_for_rank(num_proofs, rank, world)
    buf = np.zeros((per, CAP_WORDS), dtype=np.int64)
    lc = np.ascontiguousarray(local_caps, dtype=np.uint64).reshape(len(mine), CAP_WORDS)
    buf[: len(mine)] = lc.view(np.int64)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    res = np.zeros((num_proofs, CAP_WORDS), dtype=np.uint64)
    for r in range(world):
        idx = proofs_for_rank(num_proofs, r, world)
        res[idx] = out[r].cpu().numpy().view(np.uint64)[: len(idx)]
    return res.reshape(num_proofs, 3, 16, 4)
