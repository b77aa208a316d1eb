"""GPU: the block-sharded path (dark-archon_amd/archon_shard.py: block b -> rank b mod G, one gather of BWT||baseId per
round) with the REAL HIP forward on every rank -- two ranks sharing this box's one GPU over gloo; the oracle only checks.
(tests/test_shard_gloo.py runs the same module on CPU with the oracle standing in for the device.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, sizes, q):
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import archon_shard
    import archon_synth as S
    import oracle_binding
    import pyarchon
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = [torch.from_numpy(S.gen_shape(shape, n, block=i)) for i, (shape, n) in enumerate(sizes)]

    def forward_fn(x_t):          # the product path: device-resident forward through the C ABI
        x_d = x_t.cuda()
        bwt_d = torch.empty(x_d.numel(), dtype=torch.uint8, device="cuda")
        base_d = torch.zeros(1, dtype=torch.int32, device="cuda")
        pyarchon.forward_dev(x_d, None, bwt_d, base_d)
        return bwt_d.cpu(), int(base_d.item())

    res = archon_shard.run_sharded(dist, rank, world, blocks, forward_fn)
    if rank == 0:
        orc = oracle_binding.Oracle()
        ok = True
        for i, (bwt, base) in enumerate(res):
            _, b0, base0 = orc.forward(blocks[i].numpy())
            ok = ok and base == base0 and (bwt == b0).all()
        q.put((len(res), ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_two_ranks_on_the_gpu():
    sizes = [("random", 1 << 20), ("dna", 700001), ("text", 300000), ("ab", 250000), ("motif", 123457)]   # short last round
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
        assert p.exitcode == 0
    count, ok = q.get(timeout=10)
    assert count == len(sizes) and ok
