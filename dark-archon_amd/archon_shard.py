"""Block sharding over the GPUs of one node.

A block is a self-contained BWT (one block per file in a7/a4/a6; a sequence of
independent blocks each with its own index in x1-x3, bwt/final/x3/archon.c:122-125),
so the path shards with no cross-block merge: block b -> rank b mod G, every rank
runs the whole single-GPU pipeline on its blocks, and ONE collective gathers the
per-block payloads BWT||baseId (N+4 bytes) on rank 0.  The collective is issued
through torch.distributed: backend "nccl" is RCCL over xGMI on the GPU node, "gloo"
in the CPU tests.
"""
import struct

import torch


def block_owner(block, world):
    return block % world


def blocks_of(rank, world, num_blocks):
    return [b for b in range(num_blocks) if block_owner(b, world) == rank]


def pack_payload(bwt_t, base_id):
    """BWT || baseId (uint32 LE) -- the a7 file layout (archon.cpp:895,898) as one tensor."""
    tail = torch.tensor(list(struct.pack("<I", int(base_id))), dtype=torch.uint8, device=bwt_t.device)
    return torch.cat([bwt_t.reshape(-1), tail])


def unpack_payload(payload_t):
    p = payload_t.detach().cpu().numpy()
    return p[:-4], struct.unpack("<I", p[-4:].tobytes())[0]


def gather_payloads(dist, payload_t, rank, world, dst=0):
    """One gather of equal-size payloads to `dst`; returns the list there, None elsewhere."""
    if world == 1:
        return [payload_t]
    out = [torch.empty_like(payload_t) for _ in range(world)] if rank == dst else None
    dist.gather(payload_t, out, dst=dst)
    return out


class GatherPipe:
    """The path's one exchange step as bench.py runs it: per step every rank contributes one fixed-size payload
    (BWT || baseId of its block) to ONE gather on rank `dst`; the gather is asynchronous and double-buffered, so the
    gather of step k (RCCL's own stream) overlaps the sort of step k+1.  `via_host` stages through host memory
    (gloo rehearsal on fewer GPUs than ranks).

    The collective is ISSUED from a helper thread: `submit()` only hands the step over, so the host time of the call into
    torch.distributed (0.1-0.3 ms of Python and C++ per gather) runs beside the next block's kernels instead of in
    front of them -- the forward call that follows releases the GIL for its whole duration.  One helper thread per
    pipe issues the gathers strictly in step order, on every rank alike.

    Several blocks in flight per rank (bench.py --in-flight F): F feeder threads take the steps in turn, `nbuf` = 2 F payload
    buffers (step k uses buffer k mod nbuf), each feeder asks for the buffer of ITS step (`buffer_of`) and hands the step over with
    `submit_step`, which waits until every earlier step has been handed over -- the collectives still leave every rank in step order.

    `batch` = world (rotating roots only): the gathers of `world` consecutive steps -- one per root -- are issued as ONE collective
    (`all_to_all_single`: chunk j of a rank's send buffer is its payload of the step whose root is rank j; chunk r of the
    receive buffer on rank j is rank r's payload of that step).  xGMI is point to point: a gather moves a rank's N + 4 bytes
    over the ONE link to that step's root (268 MB at 60-77 GB/s per direction = 3.5-4.5 ms, longer than the 2.6 ms sort of the
    step) while its other six links idle; `world` rotated gathers in one collective use every link of every rank at once, so the
    same 3.5-4.5 ms carry `world` steps.  Blocks land exactly where the per-step gathers put them (step k whole on rank k mod
    world).  Two batches of buffers: batch b is exchanged while batch b + 1 is sorted.  `drain()` exchanges a batch that is not
    full yet as it stands (every rank drains at the same step); its steps travel again when the batch completes."""

    def __init__(self, dist, rank, world, payload_bytes, device, via_host=False, dst=0, rotate=False, threaded=True, nbuf=2, batch=1):
        """rotate: the gather of step k lands on rank (dst + k) mod world instead of always on `dst` -- every rank takes its
        turn as the root, so no GPU has to take in world - 1 payloads per step (7 x 256 MiB next to its own sort): the
        outputs of step k (blocks k * world .. k * world + world - 1, one contiguous stretch of the container) then sit
        on rank k mod world, which writes that stretch."""
        self.dist, self.rank, self.world, self.dst, self.via_host, self.rotate = dist, rank, world, dst, via_host, rotate
        import threading
        # batch < 0: the batched exchange whatever the world size (one rank: the collective's self-test under RCCL, a local copy)
        self.batched = dist is not None and (int(batch) < 0 or (world > 1 and int(batch) > 1))
        self.batch = world if self.batched else 1
        gdev = torch.device("cpu") if via_host else device
        self.cv = threading.Condition()
        self.covered = 0                     # batch mode: steps 0 .. covered-1 are inside an exchange that has been handed to the helper
        if self.batched:
            if (int(batch) > 0 and int(batch) != world) or not rotate or dst != 0:
                raise ValueError("GatherPipe: batch must equal the world size, with rotating roots from rank 0")
            nbuf = 2 * world
            # a slot = one payload, on a 256-byte boundary (the kernels store the BWT 16 bytes at a time)
            self.stride = stride = (payload_bytes + 255) & ~255
            self.send = [torch.empty(world * stride, dtype=torch.uint8, device=device) for _ in range(2)]
            self.recv = [torch.empty(world * stride, dtype=torch.uint8, device=gdev) for _ in range(2)]
        self.nbuf = nbuf
        if self.batched:
            # step k: buffer k mod 2W = slot k mod W of batch buffer (k div W) mod 2; its root (rank k mod W) finds the payloads of all ranks
            # in the receive buffer of the same parity
            self.outs = [self.send[k // world][(k % world) * stride:(k % world) * stride + payload_bytes] for k in range(nbuf)]
            self.lists = [[self.recv[k // world][r * stride:r * stride + payload_bytes] for r in range(world)] if k % world == rank else None
                          for k in range(nbuf)]
        else:
            self.outs = [torch.empty(payload_bytes, dtype=torch.uint8, device=device) for _ in range(nbuf)]
            self.lists = [None] * nbuf
        if not self.batched and dist is not None and (rotate or rank == dst):
            # A root's own payload is where it belongs already: its slot of the gathered list IS its payload buffer (torch's
            # gather copies the root's input into that slot with copy_, which does nothing when both are one tensor), so a
            # gather moves the world - 1 foreign payloads and nothing else -- 2 x 268 MB of HBM traffic less on the root per
            # step it is root, all of a one-rank gather.
            self.lists = [[self.outs[k] if (r == rank and not via_host) else torch.empty(payload_bytes, dtype=torch.uint8, device=gdev)
                           for r in range(world)] for k in range(nbuf)]
        self.pending = [None] * nbuf         # per buffer: the work handle, once the helper has issued the gather
        self.issued = [None] * nbuf          # per buffer: threading.Event set when the helper has issued it (or failed)
        self.error = None
        self.step_no = 0
        self.device = device
        self.queue = None
        if dist is not None and threaded:
            import queue
            self.queue = queue.Queue()
            self.thread = threading.Thread(target=self._issuer, daemon=True)
            self.thread.start()

    def root_of(self, step):
        return (self.dst + step) % self.world if self.rotate else self.dst

    def _issue(self, k, step):
        src = self.outs[k].cpu() if self.via_host else self.outs[k]
        root = self.root_of(step)
        self.pending[k] = self.dist.gather(src, self.lists[k] if self.rank == root else None, dst=root, async_op=True)

    def _exchange(self, g):
        """batch mode: the `world` rotated gathers of one batch as one collective (equal splits: one slot per peer)"""
        src = self.send[g].cpu() if self.via_host else self.send[g]
        self.pending[g] = self.dist.all_to_all_single(self.recv[g], src, async_op=True)

    def _issuer(self):
        if self.device is not None and getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.set_device(self.device)
        while True:
            job = self.queue.get()
            if job is None:
                return
            k, step, ev = job
            if self.error is None:
                try:
                    if step is None:
                        self._exchange(k)
                    else:
                        self._issue(k, step)
                except Exception as e:      # noqa: BLE001
                    # The first failure ends the pipe: no later gather is issued (a rank that went on issuing collectives its
                    # peers never match would block them until the NCCL timeout), every job still queued has its event set,
                    # and the owner thread learns at its very next call (submit / next_buffer / drain all raise self.error).
                    self.error = e
            ev.set()

    def _complete(self, work):
        """the collective behind `work` has finished when this returns -- for the HOST.  Over RCCL `work.wait()` only makes torch's
        current stream wait; the sort that reuses the buffer runs on the library's own (non-blocking) stream, which that does not
        order, so the host waits for the stream too (gloo's wait blocks the host by itself)."""
        work.wait()
        if not self.via_host and getattr(self.device, "type", "cpu") == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def _wait_batch(self, need):
        """batch mode: step `need` (the last user of a buffer; < 0: none) has been exchanged -- first handed over (by whichever
        feeder completed its batch, or by a drain), then issued by the helper, then complete.  Several feeders may wait for one
        batch: nothing is reset here, and over RCCL each caller's own stream waits for the collective."""
        if need >= 0:
            with self.cv:
                self.cv.wait_for(lambda: self.covered > need or self.error is not None)
        if self.error is not None:
            raise self.error
        if need < 0:
            return
        g = (need // self.batch) % 2
        ev = self.issued[g]
        if ev is not None:
            ev.wait()
        if self.error is not None:
            raise self.error
        work = self.pending[g]
        if work is not None and not work.is_completed():
            self._complete(work)

    def _wait(self, k):
        if self.issued[k] is not None:
            self.issued[k].wait()
            self.issued[k] = None
        if self.error is not None:
            raise self.error
        if self.pending[k] is not None:
            # (a gather that has completed -- the usual case: it was issued two steps ago -- needs no wait on the stream)
            if not self.pending[k].is_completed():
                self._complete(self.pending[k])
            self.pending[k] = None

    def next_buffer(self):
        """payload buffer of the coming step (waits until its previous gather has completed)"""
        k = self.step_no % self.nbuf
        if self.batched:
            self._wait_batch(self.step_no - self.nbuf)
        else:
            self._wait(k)
        return self.outs[k]

    def buffer_of(self, step):
        """payload buffer of step `step` (several feeders: each asks for its own steps, in increasing order; waits until the gather
        that last used the buffer -- step - nbuf -- has completed)"""
        k = step % self.nbuf
        if self.batched:
            self._wait_batch(step - self.nbuf)
        else:
            self._wait(k)
        return self.outs[k]

    def submit_step(self, step):
        """submit() for several feeders: blocks until every earlier step has been handed over, so that the gathers are issued in step
        order whatever order the feeders finish in; raises the pipe's error if another feeder (or the helper) has failed meanwhile"""
        with self.cv:
            self.cv.wait_for(lambda: self.step_no == step or self.error is not None)
            if self.error is not None:
                raise self.error
            self.submit()
            self.cv.notify_all()

    def abort(self, err):
        """a feeder failed outside the pipe: wake the others (their next submit_step raises) -- nothing further is issued"""
        with self.cv:
            if self.error is None:
                self.error = err
            self.cv.notify_all()

    def submit(self):
        """the payload of the current step is complete in its buffer (the producer has synchronised): gather it"""
        if self.error is not None:
            raise self.error
        if self.batched:
            with self.cv:
                self.step_no += 1
                if self.step_no % self.batch == 0:
                    self._hand_over_batch()
            return
        k = self.step_no % self.nbuf
        step = self.step_no
        self.step_no += 1
        if self.dist is None:
            return
        if self.queue is None:
            self._issue(k, step)
            return
        import threading
        ev = threading.Event()
        self.issued[k] = ev
        self.queue.put((k, step, ev))

    def _hand_over_batch(self):
        """(under self.cv) the batch that holds step step_no - 1 goes to the helper -- or is issued here -- as it stands"""
        g = ((self.step_no - 1) // self.batch) % 2
        if self.queue is None:
            try:
                self._exchange(g)
            except Exception as e:      # noqa: BLE001
                self.error = e
                self.cv.notify_all()
                raise
        else:
            import threading
            ev = threading.Event()
            self.issued[g] = ev
            self.queue.put((g, None, ev))
        self.covered = self.step_no
        self.cv.notify_all()

    def drain(self):
        if self.batched:
            with self.cv:
                if self.error is None and self.covered < self.step_no:
                    self._hand_over_batch()
            self._wait_batch(self.step_no - 1 - self.batch)       # the batch before the last one (if any) ...
            self._wait_batch(self.step_no - 1)                    # ... and the last one
            if self.error is not None:
                raise self.error
            return
        for k in range(self.nbuf):
            self._wait(k)

    def close(self):
        try:
            self.drain()
        finally:
            self._stop()

    def _stop(self):
        if self.queue is not None:
            self.queue.put(None)
            self.thread.join(10)
            self.queue = None

    def last(self):
        """(own payload buffer, gathered list -- valid on last_root() only) of the most recent step"""
        k = (self.step_no - 1) % self.nbuf
        return self.outs[k], self.lists[k]

    def last_root(self):
        return self.root_of(self.step_no - 1)


def run_sharded(dist, rank, world, blocks, forward_fn, device="cpu"):
    """Encode `blocks` (list of uint8 tensors, identical on every rank) block-sharded.

    forward_fn(x_t) -> (bwt_t, base_id).  Rounds of `world` blocks: round r handles blocks
    r*world .. r*world+world-1, rank k encodes block r*world+k (a short last round pads
    with an empty payload).  Returns on rank 0 the list of (bwt ndarray, base_id) in block
    order, elsewhere None.
    """
    results = [] if rank == 0 else None
    nb = len(blocks)
    size = max((int(b.numel()) for b in blocks), default=0)
    for r0 in range(0, nb, world):
        b = r0 + rank
        payload = torch.zeros(size + 8, dtype=torch.uint8, device=device)   # [len:u32][BWT||baseId]...
        if b < nb:
            bwt_t, base = forward_fn(blocks[b].to(device))
            p = pack_payload(bwt_t, base)
            payload[:4] = torch.tensor(list(struct.pack("<I", int(p.numel()))), dtype=torch.uint8, device=device)
            payload[4:4 + p.numel()] = p
        got = gather_payloads(dist, payload, rank, world)
        if rank == 0:
            for k in range(world):
                if r0 + k < nb:
                    raw = got[k].detach().cpu().numpy()
                    ln = struct.unpack("<I", raw[:4].tobytes())[0]
                    results.append(unpack_payload(torch.from_numpy(raw[4:4 + ln].copy())))
    return results
