"""CPU test of the N > 1 path: world_size 2 over gloo -- proof sharding and the Merkle-cap all_gather."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_proofs, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from plonky2_demo_amd import sharding
    mine = sharding.proofs_for_rank(num_proofs, rank, world)
    # a fake "cap" per proof that encodes the proof index, standing in for gl_proof_caps
    local = np.zeros((len(mine), 3, 16, 4), dtype=np.uint64)
    for k, i in enumerate(mine):
        local[k] = (np.arange(192, dtype=np.uint64).reshape(3, 16, 4) + np.uint64(1000003) * np.uint64(i + 1)) | np.uint64(1 << 63)
    allc = sharding.gather_caps(local, num_proofs)
    q.put((rank, mine, allc))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_sharding():
    from plonky2_demo_amd import sharding
    assert sharding.proofs_for_rank(7, 0, 2) == [0, 2, 4, 6] and sharding.proofs_for_rank(7, 1, 2) == [1, 3, 5]
    assert sharding.proofs_for_rank(512, 3, 8) == list(range(3, 512, 8))
    assert sorted(sum((sharding.proofs_for_rank(13, r, 4) for r in range(4)), [])) == list(range(13))
    assert sharding.proofs_for_rank(0, 0, 1) == []


def test_cap_gather_world2_gloo():
    world, num_proofs = 2, 7          # ragged: rank 0 proves 4, rank 1 proves 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_proofs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([(np.arange(192, dtype=np.uint64).reshape(3, 16, 4) + np.uint64(1000003) * np.uint64(i + 1)) | np.uint64(1 << 63)
                       for i in range(num_proofs)])
    for rank, mine, allc in res:
        assert (allc == expect).all()          # every rank ends with all caps, in proof order, bit-exact (top bit survives int64 transport)
