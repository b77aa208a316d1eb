#!/usr/bin/env python3
"""bench.py -- forward-BWT throughput of the MI355X path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: every
rank runs the whole forward pipeline (two-byte count -> two LSB radix passes ->
in-LDS bucket sorts -> tie resolution; prefix doubling where ties remain; SA and
BWT emitted) on its own 256 MiB block of uniform random bytes already resident
in HBM (BASELINE.json configs[1], SURVEY.md 8(d) cfg 2;
block b is seeded SEED_BASE+2+b), and for N>1 the per-block outputs BWT||baseId
are gathered to rank 0 over RCCL (the path's one exchange step).
The root rotates with the step (step k lands whole on rank k mod N), and the N gathers of N consecutive steps
-- one per root -- travel as ONE collective (all_to_all_single; --gather-batch 1 = one gather per step): xGMI is
point to point, a gather uses one link per rank (268 MB at 60-77 GB/s per direction: 3.5-4.5 ms, longer than the sort of
the step), N rotated gathers at once use every link of every rank.  The exchange of a batch is asynchronous (RCCL's own
stream) and overlaps the sorts of the next batch; the
timed region ends only after every exchange has completed.  Blocks are independent, per-GPU work is
fixed: "scaling": "weak".  Every rank keeps TWO blocks in flight (--in-flight,
config.blocks_in_flight): two feeder threads, each bound to its own compute context of the library
(include/archon_hip.h: "two threads feeding one GPU"), take the K steps in turn -- every step is
still one whole forward pass over one block, and the timed region is still exactly K of them; for
N>1 a feeder hands its step to the gather pipe only when every earlier step has been handed over
(the collectives leave every rank in step order).  The same K steps then run once more one block
at a time ("one_block_at_a_time"), and the per-kernel times of `roofline` are taken there (a
kernel's elapsed time beside another block's kernels is not its own).  --in-flight 1 is the run
of the earlier rounds.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the dominant kernel = whichever of the three streaming kernels (LSB pass A, LSB
                pass B, in-LDS bucket sort) ran longest: algorithmic bytes per launch (5 / 9 / 13 B
                x N, SURVEY.md 8(d) and DESIGN.md 4) / its mean launch time, taken live from HIP
                events recorded around each launch on the stream the library launches on
                ("kernels" lists all three);
                "traffic" = HBM bytes per launch from the committed rocprofv3 PMC passes
                (the newest profiles/rNN_final/pmc_traffic.json, named in "traffic_source"),
                null when no matching measurement is committed
  cpu_baseline  the reference a7 (oracle/_ref/a7ref, built from /root/reference by
                oracle/Makefile; kind "reference") or, when that binary is absent, the
                repo's CPU oracle (kind "port"): single thread, taskset-pinned to one core,
                the WHOLE block of the metric (one stated run, about half a minute;
                --cpu-sample-mib bounds it), rank 0 at N=1 only.
and, at N = 1 behind the timed region (never part of `value`; --no-shapes skips them):
  inverse       the graded block decoded again (A8 + A9): ms, MB/s, hops/s, fraction of the
                B_inv = 17 roofline; output compared with the input
  shapes        dna, a, ab, motif, text, prose at the same block size: one warm-up + best
                of 3 each, gated by the reference's digests (tests/golden/golden_full.json)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_FWD_CFG2 = 57.0              # algorithmic bytes per input byte, config 2 (D = 6)
B_RADIX_PASS = 9.0             # per item per later LSB pass (read idx 4 + digit 1, write idx 4)
B_FIRST_PASS = 5.0             # first LSB pass (read 1 key byte, write 4)
B_LOCAL_SORT = 13.0            # in-LDS bucket sort: read one 8-byte record, write SA 4 + BWT 1


def cpu_baseline(block_bytes, sample_mib):
    """Single-thread CPU figure: the reference a7 on the WHOLE block of the metric (the block, one
    core, taskset-pinned, clock() around compute as bwt/a7/src/main.cpp:39-41 does), one stated run -- about a minute on
    a 256 MiB block; --cpu-sample-mib bounds it to the first so many MiB."""
    import archon_synth
    ref = os.path.join(ROOT, "oracle", "_ref", "a7ref")
    sample_n = min(block_bytes, sample_mib << 20)
    x = archon_synth.gen_random(sample_n)
    whole = sample_n == block_bytes
    sample = ("the whole rank-0 block (%d MiB of uniform random bytes)" if whole else "first %d MiB of the rank-0 block (uniform random bytes)") % (sample_n >> 20)
    sample += ", SA+BWT, clock() around compute, one run"
    # one core of those this process may use (the last one: away from the cores the runtime's helper threads favour)
    try:
        core = sorted(os.sched_getaffinity(0))[-1]
    except Exception:
        core = None
    pin = ["taskset", "-c", str(core)] if core is not None and os.path.exists("/usr/bin/taskset") else []
    if os.path.exists(ref):
        tag = "/tmp/bench_a7ref_%d" % os.getpid()
        x.tofile(tag + ".in")
        try:
            r = subprocess.run(pin + [ref, "e", tag + ".in", tag + ".bwt"], capture_output=True, text=True, timeout=900)
            kv = dict(t.split("=") for t in r.stdout.split())
            if r.returncode == 0 and int(kv.get("validate", "0")) == 1:
                secs = float(kv["sa_time"])
                out = {"value": round(sample_n / 1e6 / secs, 3), "unit": "MB/s", "cores": 1, "kind": "reference",
                       "sample": sample + "; reference a7 -O3 (oracle/_ref/a7ref)", "seconds": round(secs, 2),
                       "pinned": bool(pin), "core": core, "runs": 1, "host_cores": os.cpu_count()}
                return out
        except Exception:
            pass
        finally:
            for ext in (".in", ".bwt"):
                if os.path.exists(tag + ext):
                    os.remove(tag + ext)
    # port: the repo's own CPU restatement (oracle/), checker used as a timed baseline only
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    orc = oracle_binding.Oracle()
    if pin:
        try:
            os.sched_setaffinity(0, {core})
        except Exception:
            pin = []
    sample_n = min(sample_n, 64 << 20)
    x = x[:sample_n]
    t0 = orc.L.oracle_clock_seconds()
    orc.forward(x)
    secs = orc.L.oracle_clock_seconds() - t0
    return {"value": round(sample_n / 1e6 / secs, 3), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": "first %d MiB of the rank-0 block; oracle/archon_oracle.c, one run" % (sample_n >> 20), "seconds": round(secs, 2),
            "pinned": bool(pin), "core": core, "runs": 1, "host_cores": os.cpu_count()}


def reference_digest(shape, block, n):
    """SHA-256 digests the reference itself produced for this block (tests/golden/golden_full.json, generated by
    tests/golden/make_golden_full.py from oracle/_ref/a7ref[_nt] in the development container), or None."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "golden_full.json")) as f:
            for c in json.load(f)["cases"]:
                if c["shape"] == shape and c["block"] == block and c["n"] == n:
                    return c
    except Exception:
        pass
    return None


def sha256_of(*parts):
    import hashlib
    h = hashlib.sha256()
    for p in parts:
        b = memoryview(p).cast("B")
        for o in range(0, len(b), 1 << 26):
            h.update(b[o:o + (1 << 26)])
    return h.hexdigest()


def pmc_traffic(path, n, shape, kname):
    """HBM bytes per launch of the dominant kernel, from the committed PMC summary of the newest round that holds one
    (profiles/rNN_final/pmc_traffic.json: same command, separate --pmc passes, gfx950 FETCH_SIZE correction applied as
    documented there), with the file it came from."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_final", "pmc_traffic.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
            ent = t.get("path%d_%s_%d" % (path, shape, n), {}).get(kname.split(" ")[0])
            if ent:
                return ent["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


PATH_NAMES = {0: "7 LSB passes + refinement rounds", 1: "streaming (hist16, 2 LSB passes, in-LDS bucket sort)",
              2: "closed form (clean periodic block)"}
EXTRA_SHAPES = ("dna", "a", "ab", "motif", "text", "prose")


def extra_shapes(n, dev, names):
    """The other named shapes of BASELINE.json (configs[2..4]) at the block size of the metric, AFTER the timed region of the
    graded config: one warm-up + best of 3 per shape, each gated by the reference's own digests (golden_full.json)."""
    import torch
    import archon_synth
    import pyarchon
    out = {}
    x_t = torch.empty(n, dtype=torch.uint8, device=dev)
    sa_t = torch.empty(n, dtype=torch.int32, device=dev)
    o_t = torch.empty(n + 4, dtype=torch.uint8, device=dev)
    ok = True
    for name in names:
        x_t.copy_(torch.from_numpy(archon_synth.gen_shape(name, n)))
        best, best_dev, st = None, None, None
        for r in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pyarchon.forward_dev(x_t, sa_t, o_t[:n], o_t[n:].view(torch.int32))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = pyarchon.stats(dev.index or 0)
            if r and (best is None or dt < best):
                best, best_dev = dt, st["ms_total"]
        ref = reference_digest(name, 0, n)
        sha = None
        if ref is not None:
            sha = sha256_of(o_t.cpu().numpy()) == ref["sha256_bwt_base"] and sha256_of(sa_t.cpu().numpy()) == ref["sha256_P"]
            ok = ok and sha
        out[name] = {"ms": round(best * 1e3, 3), "device_ms": round(best_dev, 3), "MB_s": round(n / 1e6 / best, 1),
                     "frac_57B_model": round(B_FWD_CFG2 * n / (best_dev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "path": PATH_NAMES.get(st["path"], str(st["path"])),
                     "rounds": st["text_rounds"] + st["break_rounds"] + st["doubling_rounds"], "radix_passes": st["radix_passes"],
                     "period": st["period"], "kernel_launches": st["kernel_launches"],
                     "sa_sha256_matches_reference": sha}
    return out, ok


def inverse_leg(bwt_t, base, x_t):
    """The inverse of the graded block (A8 + A9), after the timed region: one warm-up + best of 3, output compared with x."""
    import torch
    import pyarchon
    n = x_t.numel()
    out_t = torch.empty(n, dtype=torch.uint8, device=x_t.device)
    best, st = None, None
    for r in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pyarchon.inverse_dev(bwt_t, base, out_t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r and (best is None or dt < best):
            best, st = dt, pyarchon.stats(x_t.device.index or 0)
    same = bool(torch.equal(out_t, x_t))
    # the walk's tail is latency (a few long chains): a second block in flight (two feeder threads with a context each) uses the
    # chip meanwhile -- eight blocks, best of two runs, per block
    outs = [torch.empty(n, dtype=torch.uint8, device=x_t.device) for _ in range(2)]
    feeders = Feeders(2, x_t.device.index or 0)
    body = lambda t, _k: pyarchon.inverse_dev(bwt_t, base, outs[t])
    feeders.run(2, body)
    torch.cuda.synchronize()
    best2 = None
    for _ in range(2):
        t0 = time.perf_counter()
        feeders.run(8, body)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        best2 = dt if best2 is None or dt < best2 else best2
    feeders.close()
    same = same and all(bool(torch.equal(o, x_t)) for o in outs)
    return {"ms": round(best * 1e3, 3), "MB_s": round(n / 1e6 / best, 1), "hops_per_s": round(n / best, 0),
            "frac_B17": round(17.0 * n / best / 1e9 / HBM_PEAK_GBS, 4), "ms_lf_build": round(st["ms_lf_build"], 3),
            "ms_lf_walk": round(st["ms_lf_walk"], 3), "chains": st["walk_chains"], "kernel_launches": st["kernel_launches"],
            "two_blocks_in_flight": {"ms_per_block": round(best2 * 1e3, 3), "MB_s": round(n / 1e6 / best2, 1),
                                     "frac_B17": round(17.0 * n / best2 / 1e9 / HBM_PEAK_GBS, 4)},
            "equals_input": same}, same


class Feeders:
    """F host threads, thread t bound to compute context t of the library on `dev_index` (include/archon_hip.h: a context = arena, stream,
    mailbox; "two threads feeding one GPU") and launching on a stream of its own; run(count, body) deals steps 0..count-1 to them in turn
    (step k -> thread k mod F, body(t, k)) and returns when all are done."""

    def __init__(self, flight, dev_index, on_error=None):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.flight, self.dev_index, self.on_error = flight, dev_index, on_error
        self.tls = threading.local()
        self.lock, self.bound = threading.Lock(), 0
        self.pool = ThreadPoolExecutor(flight)

    def _feed(self, t, count, body):
        import torch
        import pyarchon
        if not hasattr(self.tls, "stream"):
            torch.cuda.set_device(self.dev_index)
            # a context per THREAD, in the order the pool's threads first get work (the pool may hand part t to any of its threads,
            # and to the same one twice when a part is empty): never two threads on one context, which would take turns at its mutex
            with self.lock:
                ctx, self.bound = self.bound, self.bound + 1
            pyarchon.bind_context(ctx, self.dev_index)
            self.tls.stream = torch.cuda.Stream(device=torch.device("cuda", self.dev_index))
        try:
            with torch.cuda.stream(self.tls.stream):
                for k in range(t, count, self.flight):
                    body(t, k)
        except Exception as e:      # noqa: BLE001
            if self.on_error is not None:
                self.on_error(e)        # (the other feeders may be waiting for this one's step)
            raise

    def run(self, count, body):
        for f in [self.pool.submit(self._feed, t, count, body) for t in range(self.flight)]:
            f.result()

    def close(self):
        self.pool.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)       # (the first steps behind a cold start run 4 % slower: clocks, first touch of the arena)
    ap.add_argument("--block-mib", type=int, default=256, help="block size per GPU (BASELINE config: 256)")
    ap.add_argument("--shape", default="random", help="random|dna|text|a|ab|motif (graded config: random)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=1 << 20, help="bound the CPU baseline to the first so many MiB of the block (default: the whole block)")
    ap.add_argument("--no-shapes", action="store_true", help="skip the other named shapes and the inverse behind the timed region (N=1 runs them by default)")
    ap.add_argument("--no-sa", action="store_true", help="emit BWT only (the metric is quoted WITH the SA)")
    ap.add_argument("--in-flight", type=int, default=0, help="blocks in flight per GPU: F host threads, each bound to its own compute context "
                    "of the library (include/archon_hip.h: 'two threads feeding one GPU'), take the steps in turn (step k -> thread k mod F). "
                    "0 = default: 2 (per rank, whatever N: the gathers of N > 1 are still issued in step order)")
    ap.add_argument("--gather-root", default="rotate", help="rotate (step k gathers on rank k mod N: no GPU takes in N-1 payloads "
                    "every step) | 0 (always rank 0)")
    ap.add_argument("--gather-batch", type=int, default=0, help="0 = default: at N > 1 with rotating roots the N gathers of N consecutive steps (one per "
                    "root) travel as ONE collective (all_to_all_single: every xGMI link of every rank carries one payload at once) | 1: one "
                    "gather per step (one link per rank and step; A/B runs) | -1: the batched exchange even with one rank (the collective's self-test)")
    ap.add_argument("--gather-threaded", type=int, default=1, help="1: the gather is issued from the pipe's helper thread (default) | 0: from "
                    "the calling thread (A/B runs)")
    ap.add_argument("--pass-ranges", type=int, default=-1, help="library option pass_ranges (default: one per CU at N=1, 224 at N>1: 32 CUs left to RCCL's kernels)")
    ap.add_argument("--pass-b-buckets", type=int, default=-1, help="library option pass_b_buckets (default: the library's own, 1; 0 = pass B by ranges: A/B runs)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the measured configuration) | gloo (rehearsal of the N>1 "
                    "control flow on fewer GPUs than ranks: ranks share devices, the gather is staged through host memory)")
    args = ap.parse_args()

    import torch
    import archon_synth
    import pyarchon

    pyarchon.lib()   # fail loudly if the HIP extension is missing
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            local_rank %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank)

    import archon_shard
    n = args.block_mib << 20
    # block sharding (dark-archon_amd/archon_shard.py): one step = `world` blocks, block b on rank b mod world
    my_block = archon_shard.blocks_of(rank, world, world)[0]
    x = archon_synth.gen_shape(args.shape, n, block=my_block)
    x_t = torch.from_numpy(x).to(dev)
    sa_t = None if args.no_sa else torch.empty(n, dtype=torch.int32, device=dev)
    # BWT || baseId (LE), double-buffered: the gather of step k runs on RCCL's stream while step k+1 sorts
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    in_flight = max(1, min(args.in_flight if args.in_flight > 0 else 2, 4, args.steps))
    gather_batch = -1 if args.gather_batch < 0 else (world if (world > 1 and args.gather_root == "rotate" and args.gather_batch != 1) else 1)
    pipe = archon_shard.GatherPipe(dist, rank, world, n + 4, dev, via_host=(args.backend != "nccl"), rotate=(args.gather_root == "rotate"),
                                   threaded=bool(args.gather_threaded), nbuf=2 * in_flight, batch=gather_batch)
    pass_ranges = 0
    if world > 1:
        # RCCL's send/recv kernels hold CUs while the exchange of a batch overlaps the sorts of the next one, and a pass workgroup
        # needs a whole CU (all its LDS): with one range per CU a few held CUs mean a second round of workgroups, i.e. a pass twice
        # as long.  224 ranges leave 32 CUs to RCCL (and to the other block in flight); with two blocks in flight they cost nothing
        # when no exchange runs (profiles/r05_final/in_flight_sweep.txt: 103.1 GB/s against 102.3 at 256).  Measured with one rank
        # over RCCL and a 268 MB self-exchange beside every step (profiles/r05_final/exchange/exchange_ranges.txt): 224 ranges with
        # pass B by buckets 92.9 GB/s, 256 ranges 90.8, 224 ranges with pass B by ranges 85.7, 1024 ranges with pass B by ranges
        # (what this script asked for before) 82.9 -- and without an exchange 100.0 / 90.4 / 89.0 for 256 by buckets / 224 by
        # ranges / 1024 by ranges.  Product options of the library (include/archon_hip.h, archon_hip_set_option), per device.
        pass_ranges = args.pass_ranges if args.pass_ranges >= 0 else 224
        pyarchon.set_option("pass_ranges", pass_ranges, local_rank)
    elif args.pass_ranges > 0:
        pass_ranges = args.pass_ranges
        pyarchon.set_option("pass_ranges", pass_ranges, local_rank)
    if args.pass_b_buckets >= 0:
        pyarchon.set_option("pass_b_buckets", args.pass_b_buckets, local_rank)
    # (no archon_hip_reserve: the arenas grow in the warm-up step, by what the block really needs -- a block the streaming stage settles
    #  never allocates the general stage's lists and tables; pipeline.arena_bytes_per_input_byte says what the timed steps used)

    pass_ms, pass_cnt, stage = [], [], []

    host_trace = []

    def step():
        t_a = time.perf_counter()
        out_t = pipe.next_buffer()
        # baseId lands behind the BWT: BWT || baseId (LE); n is a multiple of 4, the view is aligned
        bwt_v, base_v = out_t[:n], out_t[n:].view(torch.int32)
        t_b = time.perf_counter()
        pyarchon.forward_dev(x_t, sa_t, bwt_v, base_v)
        t_c = time.perf_counter()
        stage.append(pyarchon.stats_raw(local_rank))      # (the structure as it is: turned into dicts behind the timed region)
        pipe.submit()
        host_trace.append((t_b - t_a, t_c - t_b, time.perf_counter() - t_c))

    def fence():
        pipe.drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    alone = None
    if in_flight > 1:
        # F blocks in flight: F feeder threads, thread t bound to compute context t of the library (its own arena, stream and mailbox), with
        # its own SA and BWT||baseId buffers, takes steps t, t+F, t+2F, ...  Every step is still one whole forward pass over one block; what the
        # second block buys is the chip's idle corners -- the tails of the first block's kernels (one workgroup per CU, the slowest CU sets the
        # kernel's time) and its host round trip.  The same K steps run once more one block at a time behind the timed region: the per-kernel
        # times of `roofline` come from THAT region (a kernel's elapsed time beside another block's kernels is not its own).
        # The payload buffers are the gather pipe's (2 F of them: step k uses buffer k mod 2F); a feeder hands its step over with
        # submit_step, which keeps the gathers of all ranks in step order whatever order the feeders finish in.
        pyarchon.bind_context(0, local_rank)
        sa_f = [sa_t] + [None if args.no_sa else torch.empty(n, dtype=torch.int32, device=dev) for _ in range(in_flight - 1)]
        stage_f = [[] for _ in range(in_flight)]
        feeders = Feeders(in_flight, local_rank, on_error=pipe.abort)

        def run_steps(count):
            base = pipe.step_no

            def one_step(t, k):
                out_t = pipe.buffer_of(base + k)
                pyarchon.forward_dev(x_t, sa_f[t], out_t[:n], out_t[n:].view(torch.int32))
                stage_f[t].append(pyarchon.stats_raw(local_rank))
                pipe.submit_step(base + k)

            feeders.run(count, one_step)

        # (each thread warms its own context: its arena grows in its first step)
        run_steps(args.warmup * in_flight)
        for lst in stage_f:
            del lst[:]
        fence()
        t0 = time.perf_counter()
        run_steps(args.steps)
        fence()
        dt = time.perf_counter() - t0
        stage_flight = [s.asdict() for lst in stage_f for s in lst]
        feeders.close()
        # every feeder's last block against the reference's digests (the one-at-a-time region's goes through the ordinary gate below)
        flight_ok = True
        ref_f = reference_digest(args.shape, my_block, n)
        pipe.drain()
        for back in range(min(in_flight, args.steps)):
            if ref_f is not None:
                flight_ok = flight_ok and sha256_of(pipe.outs[(pipe.step_no - 1 - back) % pipe.nbuf].cpu().numpy()) == ref_f["sha256_bwt_base"]
        for t in range(1, in_flight):
            if ref_f is not None and sa_f[t] is not None and stage_f[t]:
                flight_ok = flight_ok and sha256_of(sa_f[t].cpu().numpy()) == ref_f["sha256_P"]
        # ... and the same K steps one block at a time (the calling thread: context 0 again), for the kernels' own times
        for _ in range(2):
            step()
        del stage[:], host_trace[:]
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_alone = time.perf_counter() - t1
        alone = {"ms_per_step": round(dt_alone * 1e3 / args.steps, 3), "MB_s": round(float(n) * args.steps / 1e6 / dt_alone, 1), "steps": args.steps}
    else:
        flight_ok = True
        for _ in range(args.warmup):
            step()
        del pass_ms[:], pass_cnt[:], stage[:], host_trace[:]
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
    stage[:] = [s.asdict() for s in stage]
    pass_ms[:] = [s["ms_radix_pass_sum"] for s in stage]
    pass_cnt[:] = [s["radix_pass_timed"] for s in stage]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness gates on the timed output (after the timed region):
    #  (1) the rank's SA and BWT||baseId hashed against what THE REFERENCE produced for the same block
    #      (golden_full.json; every rank checks its own block), (2) LF-consistency of the SA on the device
    ok = flight_ok
    sha_ok = None
    ref = reference_digest(args.shape, my_block, n)
    own_last, gathered_last = pipe.last()
    if ref is not None:
        sha_ok = sha256_of(own_last.cpu().numpy()) == ref["sha256_bwt_base"]
        if sa_t is not None:
            sha_ok = sha_ok and sha256_of(sa_t.cpu().numpy()) == ref["sha256_P"]
        ok = ok and sha_ok
    lf_ok = None
    if sa_t is not None:
        lf_ok = pyarchon.validate_dev(x_t, sa_t)
        ok = lf_ok and ok
    # the exchange step: rank 0 decodes the block it received from the LAST rank (inverse BWT on its own GPU)
    # and compares with that rank's input, regenerated from the seed -- after the timed region
    gathered_ok = None
    last_root = pipe.last_root()
    if dist is not None and rank == last_root:
        got = gathered_last[world - 1].to(dev)
        base_r = int(got[n:].view(torch.int32).item())
        back = torch.empty(n, dtype=torch.uint8, device=dev)
        pyarchon.inverse_dev(got[:n], base_r, back)
        want = torch.from_numpy(archon_synth.gen_shape(args.shape, n, block=archon_shard.blocks_of(world - 1, world, world)[0])).to(dev)
        gathered_ok = bool(torch.equal(back, want))
        ok = ok and gathered_ok
    if dist is not None:
        # (the gathered block is decoded on the rank that was the root of the LAST step: its verdict travels with the flags)
        flag = torch.tensor([1 if ok else 0, 0 if sha_ok is False else 1, 0 if sha_ok is None else 1, 0 if lf_ok is False else 1,
                             0 if gathered_ok is False else 1], device=gdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag[0].item())
        lf_ok = None if lf_ok is None else bool(flag[3].item())
        gathered_ok = bool(flag[4].item())
        sha_ok = bool(flag[1].item()) if flag[2].item() else None      # true only if EVERY rank had a digest and matched it

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        total_bytes = float(n) * world * args.steps
        value = total_bytes / 1e6 / dt
        last = stage[-1]
        kernels = None
        if last["path"] == 1:
            # streaming first stage: three kernels carry the step; the dominant one is whichever ran longest (measured
            # live, HIP events around each launch).  Algorithmic bytes per item: pass A 5 and pass B 9 are
            # SURVEY.md 8(d)'s LSB-pass figures; the in-LDS bucket sort must read one 8-byte record and write 4 (SA) + 1
            # (BWT) bytes per item = 13 (in 8(d)'s pass-by-pass model it stands for passes 3..6 and sa_to_bwt, 42 B --
            # NOT used here: the fraction below prices only bytes the kernel itself has to move).
            spec = [("bs::k_pass_text (LSB pass A: text -> 8-byte records, 256-way by x[s-2])", "ms_pass_text", B_FIRST_PASS),
                    ("bs::k_pass_rec (LSB pass B: records 256-way by x[s-1])", "ms_pass_rec", B_RADIX_PASS),
                    ("bs::k_local_sort (in-LDS sort of a 16-bit bucket, SA + BWT out)", "ms_local_sort", B_LOCAL_SORT)]
            kernels = []
            for nm, key, bpi in spec:
                t = float(np.mean([s[key] for s in stage]))
                gbs = bpi * n / (t * 1e-3) / 1e9 if t > 0 else 0.0
                kernels.append({"kernel": nm, "launch_ms": round(t, 4), "algorithmic_bytes_per_item": bpi,
                                "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)})
            dom = max(kernels, key=lambda k: k["launch_ms"])
            t_pass_ms, kname, bpi_dom, per_step = dom["launch_ms"], dom["kernel"], dom["algorithmic_bytes_per_item"], 1
        else:
            npass = max(1, sum(pass_cnt))
            t_pass_ms = sum(pass_ms) / npass
            kname = "rs::k_scatter (one LSB radix pass over %d (key,index) pairs)" % n
            per_step = last["radix_pass_timed"]
            bpi_dom = B_RADIX_PASS
        achieved = bpi_dom * n / (t_pass_ms * 1e-3) / 1e9 if t_pass_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(last["path"], n, args.shape, kname)
        dev_ms = float(np.mean([s["ms_total"] for s in stage]))
        line = {
            "metric": "forward-BWT MB/s on 256 MB block (SA bit-exact vs a7 order)",
            "value": round(value, 1),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: %d MiB uniform-random bytes (splitmix64, seed 0x20261003+2+rank), one BWT block per GPU, SA+BWT+baseId emitted"
                            % args.block_mib if args.shape == "random" else "%d MiB '%s' block per GPU" % (args.block_mib, args.shape),
                "block_bytes": n,
                "blocks_per_step": world,
                "parallelism": "block-sharded x%d, one RCCL gather of BWT||baseId per step (async, overlapped with the next steps)" % world if world > 1 else "single GPU",
                "sa_emitted": sa_t is not None,
                "sa_lf_consistent": lf_ok,
                "sa_sha256_matches_reference": sha_ok,     # null: no reference digest committed for this shape / size
                "gates_passed": ok,                        # reference digests + LF-consistency (+ gathered round trip)
                "gathered_block_round_trip": gathered_ok,
                "pass_ranges": pass_ranges or 256,
                "blocks_in_flight": in_flight,     # F feeder threads, one compute context each; a step is still one whole block
                "backend": (dist.get_backend() if dist is not None else None),
                "dist_world_size": (dist.get_world_size() if dist is not None else 1),
                "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if dist is not None and args.backend == "nccl" else None),
                "exchange": (("torch.distributed.all_to_all_single (RCCL grouped send/recv): the %d gathers of %d consecutive steps, root k mod N at step k, as one "
                              "collective; %d bytes per rank and step" % (world, world, n + 4)) if pipe.batched else
                             ("torch.distributed.gather (RCCL grouped send/recv) of %d bytes per rank per step, root %s" % (n + 4, "k mod N at step k" if args.gather_root == "rotate" else "0"))) if dist is not None else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launch_ms": round(t_pass_ms, 4),
                "launches_per_step": per_step,
                "algorithmic_bytes_per_launch": bpi_dom * n,
                "kernels": kernels,
                "measured_over": ("the %d steps run one block at a time behind the timed region (HIP events around each launch); with %d blocks "
                                  "in flight a kernel's elapsed time includes the other block's kernels" % (args.steps, in_flight)) if in_flight > 1
                                 else "the timed region (HIP events around each launch)",
            },
            "pipeline": {
                "device_ms_per_block": round(dev_ms, 3),           # one block alone on the chip
                "algorithmic_bytes_per_input_byte": B_FWD_CFG2,
                "frac_of_hbm_roofline": round(B_FWD_CFG2 * n / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dev_ms > 0 else None,
                "path": PATH_NAMES.get(last["path"], str(last["path"])),
                "ms_hist": round(last["ms_hist"], 3), "ms_sort": round(last["ms_sort"], 3),
                "ms_pass_text": round(last["ms_pass_text"], 3), "ms_pass_rec": round(last["ms_pass_rec"], 3),
                "ms_local_sort": round(last["ms_local_sort"], 3), "ms_resolve": round(last["ms_resolve"], 3),
                "ms_doubling": round(last["ms_doubling"], 3), "ms_bwt": round(last["ms_bwt"], 3),
                # host side of a step, mean microseconds: waiting for the payload buffer's previous gather, the forward call
                # (device pipeline + its one host round trip), statistics + handing the gather to the helper thread
                "host_us_buffer_forward_submit": [round(1e6 * float(np.mean([h[i] for h in host_trace])), 1) for i in range(3)],
                "arena_bytes_per_input_byte": round(last["arena_bytes"] / n, 2),
                "tie_groups_after_5_bytes": last["tie_groups"],
                "radix_passes": last["radix_passes"], "doubling_rounds": last["doubling_rounds"],
                "unresolved_initial": last["unresolved_initial"], "kernel_launches": last["kernel_launches"],
            },
        }
        if alone is not None:
            # the timed region's own fraction: 57 B x N / (wall time per block) / 8 TB/s; and the one-block-at-a-time region
            line["pipeline"]["frac_of_hbm_roofline_in_flight"] = round(B_FWD_CFG2 * n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            line["pipeline"]["device_ms_per_block_in_flight"] = round(float(np.mean([s["ms_total"] for s in stage_flight])), 3)
            line["one_block_at_a_time"] = alone
        if world == 1 and dist is None and not args.no_shapes:
            # behind the timed region: the other named shapes at this block size and the inverse of the graded block, each gated
            inv, inv_ok = inverse_leg(own_last[:n], int(own_last[n:].view(torch.int32).item()), x_t)
            line["inverse"] = inv
            shapes, shapes_ok = extra_shapes(n, dev, [sh for sh in EXTRA_SHAPES if sh != args.shape])
            line["shapes"] = shapes
            ok = ok and inv_ok and shapes_ok
            line["config"]["gates_passed"] = ok
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_sample_mib)
            # informational, NOT measured on this host and never part of cpu_baseline / vs_baseline: a7 slows down with the block
            # size; what it took on the WHOLE block of the metric when the reference digests were made (development container)
            full = reference_digest("random", 0, n)
            if full is not None and full.get("reference_sa_time_s"):
                line["reference_elsewhere"] = {"what": "reference a7 on the whole %d MiB block, 1 core" % (n >> 20), "unit": "MB/s",
                                               "value": round(n / 1e6 / full["reference_sa_time_s"], 3), "seconds": round(full["reference_sa_time_s"], 1),
                                               "host": "development container, recorded in tests/golden/golden_full.json -- not this machine"}
        print(json.dumps(line), flush=True)
    pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        print("bench.py: correctness gate FAILED (see config.gates_passed / sa_sha256_matches_reference)", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
