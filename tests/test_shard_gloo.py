"""CPU, world_size 2, gloo: the block-sharded path (block b -> rank b mod G, one gather)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, sizes, q):
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import archon_shard
    import archon_synth as S
    import oracle_binding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_binding.Oracle()
    blocks = [torch.from_numpy(S.gen_shape(shape, n, block=i)) for i, (shape, n) in enumerate(sizes)]

    def forward_fn(x_t):   # the CPU oracle stands in for the HIP path in this host-logic test
        _, bwt, base = orc.forward(x_t.numpy())
        return torch.from_numpy(bwt), base

    res = archon_shard.run_sharded(dist, rank, world, blocks, forward_fn)
    if rank == 0:
        ok = True
        for i, (bwt, base) in enumerate(res):
            _, b0, base0 = orc.forward(blocks[i].numpy())
            ok = ok and base == base0 and (bwt == b0).all()
        q.put((len(res), ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [
    [("random", 5000), ("dna", 5000)],
    [("text", 4000), ("a", 3000), ("motif", 2500)],          # short last round
    [("random", 1000)] * 5,
])
def test_sharded_two_ranks(sizes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    count, ok = q.get(timeout=10)
    assert count == len(sizes) and ok


def test_block_assignment():
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard
    assert archon_shard.blocks_of(0, 8, 8) == [0]
    assert archon_shard.blocks_of(3, 4, 10) == [3, 7]
    owners = [archon_shard.block_owner(b, 8) for b in range(16)]
    assert owners == list(range(8)) * 2
    p = archon_shard.pack_payload(torch.tensor([1, 2, 3], dtype=torch.uint8), 0x01020304)
    assert p.tolist() == [1, 2, 3, 4, 3, 2, 1]
    bwt, base = archon_shard.unpack_payload(p)
    assert bwt.tolist() == [1, 2, 3] and base == 0x01020304


def _pipe_worker(rank, world, port, rotate, q, via_host=True, threaded=True, batch=1, drain_every=1):
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nbytes, steps = 1000, 5 if batch == 1 else 11
    pipe = archon_shard.GatherPipe(dist, rank, world, nbytes, torch.device("cpu"), via_host=via_host, rotate=rotate, threaded=threaded, batch=batch)
    ok = True
    if batch != 1:
        ok = pipe.batch == world and pipe.nbuf == 2 * world and all((o.data_ptr() - pipe.send[0].data_ptr()) % 256 == 0 for o in pipe.outs[:world])
    elif not via_host and (rotate or rank == 0):
        # the branch the RCCL run takes: a root's own slot of the gathered list IS its payload buffer
        ok = all(pipe.lists[k][rank] is pipe.outs[k] for k in range(2))
    for k in range(steps):
        buf = pipe.next_buffer()
        buf[:] = (17 * k + 3 * rank) % 251               # the payload of (step k, rank)
        pipe.submit()
        if (k + 1) % drain_every and k != steps - 1:
            continue                                      # (batch mode: also without a drain between the steps of a batch)
        pipe.drain()                                      # (the bench overlaps; here every step is checked at once)
        root = pipe.root_of(k)
        ok = ok and root == ((k % world) if rotate else 0) and pipe.last_root() == root
        own, got = pipe.last()
        if rank == root:
            for r in range(world):
                ok = ok and bool((got[r] == (17 * k + 3 * r) % 251).all())
    pipe.close()
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rotate", [False, True])
def test_gather_pipe_roots(rotate):
    """bench.py's exchange step: one gather per step, on rank 0 or -- rotate -- on rank k mod N at step k"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000) + (7 if rotate else 0)
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, rotate, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


@pytest.mark.parametrize("threaded", [True, False])
@pytest.mark.parametrize("rotate", [False, True])
def test_gather_pipe_aliased_root_slot(rotate, threaded):
    """the branch bench.py takes over RCCL (no staging through the host) -- the root's own slot of the gathered
    list is its payload tensor itself -- with two ranks, every slot compared, issued from the helper thread and from the
    calling thread"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1000) + (7 if rotate else 0) + (13 if threaded else 0)
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, rotate, q, False, threaded)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


@pytest.mark.parametrize("drain_every", [1, 3, 100])
@pytest.mark.parametrize("threaded", [True, False])
@pytest.mark.parametrize("via_host", [True, False])
def test_gather_pipe_batched_exchange(via_host, threaded, drain_every):
    """bench.py's exchange at N > 1: the `world` rotated gathers of a batch as ONE all_to_all_single (every xGMI link of every rank
    carries one payload at once, instead of one link per rank and step).  Every step's payloads are compared on the step's root
    (rank k mod world) -- with a drain behind every step (a batch that is not full travels as it stands, and again when it completes),
    behind every third, and only at the end (11 steps at world 2: five full batches and a half)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 1000) + (7 if via_host else 0) + (13 if threaded else 0) + drain_every % 29
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, True, q, via_host, threaded, 2, drain_every)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


@pytest.mark.parametrize("drain_every", [1, 100])
def test_gather_pipe_batched_exchange_four_ranks(drain_every):
    """the same at world 4 (the slot arithmetic with more than two ranks: step k in slot k mod 4 of batch buffer (k div 4) mod 2, its
    payloads on rank k mod 4): 11 steps = two full batches and three quarters of one, every step checked on its root, and only the last"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 1000) + drain_every % 29
    procs = [ctx.Process(target=_pipe_worker, args=(r, 4, port, True, q, False, True, 4, drain_every)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(4))
    assert res == {0: True, 1: True, 2: True, 3: True}


def test_gather_pipe_batched_exchange_one_rank():
    """bench.py --gather-batch -1 under torch.distributed.run with ONE rank: the batched exchange as the collective's self-test (the
    only form of it a one-GPU box can run over RCCL)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34500 + (os.getpid() % 1000)
    p = ctx.Process(target=_pipe_worker, args=(0, 1, port, True, q, False, True, -1, 3))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert q.get(timeout=10) == (0, True)


def test_gather_pipe_batched_fails_fast():
    """the fail-fast rule in batch mode: the exchange of the first full batch fails in the helper thread -- the owner's next call raises,
    nothing further is issued, a drain raises too, the helper is stopped all the same"""
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard

    class Boom(RuntimeError):
        pass

    class FakeDist:
        calls = 0

        def all_to_all_single(self, *a, **k):
            FakeDist.calls += 1
            raise Boom("exchange failed")

    pipe = archon_shard.GatherPipe(FakeDist(), 0, 2, 16, torch.device("cpu"), via_host=True, rotate=True, batch=2)
    for _ in range(2):
        pipe.next_buffer()
        pipe.submit()                   # the second one completes the batch: handed to the helper, which fails
    with pytest.raises(Boom):
        for _ in range(3):              # (buffers 2 and 3 are free; buffer 0 waits for the failed exchange)
            pipe.next_buffer()
            pipe.submit()
    with pytest.raises(Boom):
        pipe.submit()
    assert FakeDist.calls == 1
    with pytest.raises(Boom):
        pipe.close()
    assert pipe.queue is None


def test_gather_pipe_batch_needs_rotating_roots():
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard

    class FakeDist:
        pass

    with pytest.raises(ValueError):
        archon_shard.GatherPipe(FakeDist(), 0, 2, 16, torch.device("cpu"), via_host=True, rotate=False, batch=2)
    with pytest.raises(ValueError):
        archon_shard.GatherPipe(FakeDist(), 0, 4, 16, torch.device("cpu"), via_host=True, rotate=True, batch=2)
    # one rank, or no process group: the per-step path (a one-rank gather moves nothing: the root's slot IS its payload buffer)
    assert archon_shard.GatherPipe(None, 0, 1, 16, torch.device("cpu"), batch=1).batch == 1
    assert archon_shard.GatherPipe(FakeDist(), 0, 1, 16, torch.device("cpu"), rotate=True, batch=1, threaded=False).batch == 1


def test_gather_pipe_completes_for_the_host(monkeypatch):
    """an exchange still running when its buffer is asked for again: the pipe waits on the work AND, over RCCL (device buffers, not
    staged through the host), for torch's stream on the host -- the sort that reuses the buffer runs on the library's own stream,
    which work.wait() does not order; a finished exchange is not waited for at all"""
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard

    class Work:
        def __init__(self, done):
            self.done, self.waits = done, 0

        def is_completed(self):
            return self.done

        def wait(self):
            self.waits += 1

    class Stream:
        syncs = 0

        def synchronize(self):
            Stream.syncs += 1

    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: Stream())

    class Dev:            # (what GatherPipe reads of a device; no GPU here)
        type = "cuda"

    for via_host, want in ((False, 1), (True, 0)):
        Stream.syncs = 0
        pipe = archon_shard.GatherPipe(None, 0, 1, 16, torch.device("cpu"))
        pipe.device, pipe.via_host = Dev(), via_host
        running, finished = Work(False), Work(True)
        pipe.pending[0], pipe.pending[1] = running, finished
        pipe._wait(0)
        pipe._wait(1)
        assert running.waits == 1 and finished.waits == 0 and Stream.syncs == want
        assert pipe.pending[0] is None and pipe.pending[1] is None


def test_gather_pipe_fails_fast():
    """the first exception of the helper thread ends the pipe -- no later gather is issued, and the owner's next
    call raises instead of leaving its peers in a collective nobody matches"""
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard

    class Boom(RuntimeError):
        pass

    class FakeDist:
        calls = 0

        def gather(self, *a, **k):
            FakeDist.calls += 1
            raise Boom("gather failed")

    pipe = archon_shard.GatherPipe(FakeDist(), 0, 2, 16, torch.device("cpu"), via_host=True)
    pipe.next_buffer()
    pipe.submit()                       # handed to the helper, which fails
    with pytest.raises(Boom):
        pipe.next_buffer()              # buffer 1: nothing pending, but the pipe is dead
        pipe.submit()
        pipe.next_buffer()
    with pytest.raises(Boom):
        pipe.submit()                   # raises at once, hands nothing over
    assert FakeDist.calls == 1
    with pytest.raises(Boom):
        pipe.close()
    assert pipe.queue is None           # the helper was stopped all the same


def _feeders_worker(rank, world, port, q, batch=1):
    """bench.py --in-flight 2 at world 2: two feeder threads per rank take the steps in turn; the SECOND feeder is always done first, and the
    gathers must still be issued in step order on both ranks (a rank that swapped two would pair step k of one rank with step k+1 of the
    other, or hang)"""
    import threading
    import time
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nbytes, steps, flight = 1000, 12, 2
    pipe = archon_shard.GatherPipe(dist, rank, world, nbytes, torch.device("cpu"), via_host=False, rotate=True, nbuf=2 * flight, batch=batch)
    seen = {}
    lock = threading.Lock()

    def feeder(t):
        for k in range(t, steps, flight):
            buf = pipe.buffer_of(k)
            # the gather that used this buffer last (step k - 4) is complete: its result may be read before the buffer is overwritten
            if k >= pipe.nbuf and rank == pipe.root_of(k - pipe.nbuf):
                with lock:
                    seen[k - pipe.nbuf] = [int(pipe.lists[k % pipe.nbuf][r][0]) for r in range(world)]
            time.sleep(0.03 if t == 0 else 0.0)          # feeder 1 always arrives first
            buf[:] = (17 * k + 3 * rank) % 251
            pipe.submit_step(k)

    ts = [threading.Thread(target=feeder, args=(t,)) for t in range(flight)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    ok = not any(t.is_alive() for t in ts) and pipe.step_no == steps
    pipe.drain()
    for k in range(steps - pipe.nbuf, steps):
        if rank == pipe.root_of(k):
            seen[k] = [int(pipe.lists[k % pipe.nbuf][r][0]) for r in range(world)]
    for k, got in seen.items():
        ok = ok and got == [(17 * k + 3 * r) % 251 for r in range(world)]
    ok = ok and len(seen) == steps // world
    pipe.close()
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [1, 2])
def test_gather_pipe_two_feeders_keep_step_order(batch):
    """batch = 2 (= world): the two rotated gathers of a batch travel as one all_to_all_single; a buffer is handed out again only when
    the exchange of the batch that used it last is complete, whichever feeder completed that batch"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() % 1000) + 17 * batch
    procs = [ctx.Process(target=_feeders_worker, args=(r, 2, port, q, batch)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}


def test_gather_pipe_abort_wakes_the_other_feeder():
    """a feeder that fails outside the pipe (its transform raised) must not leave the feeder of the NEXT step waiting for it"""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
    import archon_shard
    pipe = archon_shard.GatherPipe(None, 0, 1, 16, torch.device("cpu"), nbuf=4)
    out = []

    def second():
        try:
            pipe.submit_step(1)          # step 0 is never handed over
        except RuntimeError as e:
            out.append(str(e))

    t = threading.Thread(target=second)
    t.start()
    pipe.abort(RuntimeError("transform failed"))
    t.join(10)
    assert not t.is_alive() and out == ["transform failed"]
    with pytest.raises(RuntimeError):
        pipe.submit()
