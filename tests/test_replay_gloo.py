"""World-size-2 gloo tests (CPU) of the multi-GPU replay plumbing in leg-kilo_amd/replay.py: shard
partition, map-blob transport (ring broadcast and scatter + all-gather), result all-gather.  The engine
is a stub that records what it is given: the collectives, not the kernels, are under test here."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lk_pkg  # noqa: E402  (spawned workers import this module without conftest.py)

lk_pkg.load()

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from legkilo_amd import replay


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 128, 1024, 1031):
        for w in (1, 2, 3, 8):
            spans = [replay.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert replay.shard_range(1024, 3, 8) == (384, 512)  # config 5: 128 scans per GPU


class StubEngine:
    def __init__(self, blob=None):
        self.blob = blob
        self.imported = None

    def map_export(self):
        return self.blob

    def map_import(self, b):
        self.imported = np.array(b, copy=True)


class StubReplayEngine:
    """Answers batch_replay_ragged with poses that encode what it was asked to replay."""

    def __init__(self):
        self.calls = []

    def batch_replay_ragged(self, scans, t_begins, xs, Ps):
        from legkilo_amd import abi

        self.calls.append(len(scans))
        out = np.zeros(len(scans), dtype=abi.pose_dtype())
        for i, (sc, tb, x) in enumerate(zip(scans, t_begins, xs)):
            out["pos"][i] = (tb, float(len(sc)), x[9])
            out["n_buckets"][i] = len(sc) % 7
        return out


def _worker_run(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 11
    scans = [np.zeros(10 + 3 * i) for i in range(n)]
    tbs = [0.1 * i for i in range(n)]
    xs = [np.full(36, float(i)) for i in range(n)]
    Ps = [np.eye(30)] * n
    eng = StubReplayEngine()
    rows = replay.replay_recorded_run(eng, dist, rank, world, torch.device("cpu"), scans, tbs, xs, Ps, max_batch=4)
    ok = rows.shape == (n, 18) and np.allclose(rows[:, 0], tbs) and np.array_equal(rows[:, 1], [10.0 + 3 * i for i in range(n)]) \
        and np.array_equal(rows[:, 2], np.arange(float(n))) and np.array_equal(rows[:, 16], [(10 + 3 * i) % 7 for i in range(n)])
    a, b = replay.shard_range(n, rank, world)
    ok_calls = sum(eng.calls) == b - a and max(eng.calls) <= 4
    q.put((rank, bool(ok), bool(ok_calls)))
    dist.destroy_process_group()


def test_replay_recorded_run_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, True), (1, True, True)], res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, algo, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    rng = np.random.default_rng(42)
    blob = rng.integers(0, 256, size=100003, dtype=np.uint8)  # odd size: exercises the padding path
    eng = StubEngine(blob if rank == 0 else None)
    n, secs = replay.broadcast_map(eng, dist, rank, world, dev, src=0, algo=algo)
    ok_blob = n == blob.size and (rank == 0 or np.array_equal(eng.imported, blob))
    start, stop = replay.shard_range(11, rank, world)
    local = np.array([[float(i), float(i) * 0.5, float(rank)] for i in range(start, stop)])
    allr = replay.gather_results(dist, local, world, dev)
    ok_gather = allr.shape == (11, 3) and np.array_equal(allr[:, 0], np.arange(11.0))
    q.put((rank, bool(ok_blob), bool(ok_gather)))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["broadcast", "scatter_allgather"])
def test_map_transport_and_gather_world2(algo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, algo, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, True), (1, True, True)], res
