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
    # raw pose records (what bench.py gathers inside its timed region): every rank gets every rank's bytes, in rank order
    mine = torch.full((3, 136 * 4), rank + 1, dtype=torch.uint8)
    allb = replay.gather_pose_bytes(dist, mine, world, dev)
    ok_gather = ok_gather and tuple(allb.shape) == (world, 3 * 136 * 4) and all(int(allb[r].min()) == int(allb[r].max()) == r + 1 for r in range(world))
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


class OracleReplayEngine:
    """TEST-ONLY engine: answers batch_replay_ragged by running the oracle's bucket loop scan by scan on a frozen map (the
    product has no CPU engine).  Lets the CPU suite check that the SHARDED run (world 2, gloo) gathers exactly the rows of the
    unsharded one (world 1) for the same scans - shard boundaries, ragged batches of max_batch, gather order."""

    def __init__(self):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import oracle_binding as ob
        import scenes

        self.sc = scenes.Scene()
        self.o = ob.Oracle(self.sc.cfg(), imu_mode_only=True)
        t0 = 1.0
        x0 = scenes.init_filter(self.o, self.sc, t0)
        scenes.first_frame(self.o, self.sc, t0, x0)
        self.o.set_map_insert(False)
        self.scenes = scenes

    def inputs(self, n):
        from legkilo_amd import synth

        scans, tbs, xs, Ps = [], [], [], []
        for k in range(n):
            tb = 1.0 + 0.1 * k
            scans.append(self.scenes.vlp_scan_input(self.sc, tb, k)[: 600 + 50 * k])
            tbs.append(tb)
            xs.append(synth.initial_state(self.sc.traj, tb, self.sc.P, np.random.default_rng(k), 0.02, 0.5))
            Ps.append(1e-4 * np.eye(30))
        return scans, tbs, xs, Ps

    def batch_replay_ragged(self, scans, t_begins, xs, Ps):
        from legkilo_amd import abi

        out = np.zeros(len(scans), dtype=abi.pose_dtype())
        for i, (sc, tb, x, P) in enumerate(zip(scans, t_begins, xs, Ps)):
            self.o.set_state(x, P)
            self.o.set_times(tb, tb)
            pose, _ = self.o.process_scan(sc, tb)
            out["pos"][i], out["vel"][i], out["rot"][i] = pose.pos, pose.vel, pose.rot
            out["n_effect"][i], out["n_buckets"][i], out["n_updates"][i] = pose.n_effect, pose.n_buckets, pose.n_updates
        return out


def _worker_same_rows(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleReplayEngine()
    scans, tbs, xs, Ps = eng.inputs(8)
    rows = replay.replay_recorded_run(eng, dist, rank, world, torch.device("cpu"), scans, tbs, xs, Ps, max_batch=3)
    q.put((rank, rows))
    dist.destroy_process_group()


class StubStateEngine:
    """batch_get_states of a rank whose slots hold recognisable records."""

    def __init__(self, rank):
        self.rank = rank

    def batch_get_states(self, first_slot, n):
        x = np.array([[1000.0 * self.rank + first_slot + i + 0.001 * k for k in range(36)] for i in range(n)])
        P = np.array([np.full(900, 10.0 * self.rank + first_slot + i) for i in range(n)])
        return x, P.reshape(n, 30, 30)


def _worker_states(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, P = replay.gather_state_records(dist, StubStateEngine(rank), 4, 3, world, torch.device("cpu"))
    q.put((rank, x.numpy().copy(), P.numpy().copy()))
    dist.destroy_process_group()


def test_gather_state_records_world2():
    """The per-scan result record with covariance (state 36 + P 900), all-gathered in rank order on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_states, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (x, P) for r, x, P in (q.get(timeout=180) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        x, P = res[r]
        assert x.shape == (6, 36) and P.shape == (6, 900)
        for owner in (0, 1):
            xe, Pe = StubStateEngine(owner).batch_get_states(4, 3)
            assert np.array_equal(x[3 * owner:3 * owner + 3], xe) and np.array_equal(P[3 * owner:3 * owner + 3], Pe.reshape(3, 900))
    x1, P1 = replay.gather_state_records(None, StubStateEngine(0), 4, 3, 1, torch.device("cpu"))
    assert x1.shape == (3, 36) and P1.shape == (3, 900)


def test_world1_and_world2_gather_identical_rows():
    """The same 8 scans replayed unsharded (N = 1) and sharded over 2 ranks (gloo): every rank ends up with the SAME 8 rows,
    bit for bit, in scan order - the property the multi-GPU bench relies on."""
    eng = OracleReplayEngine()
    scans, tbs, xs, Ps = eng.inputs(8)
    solo = replay.replay_recorded_run(eng, None, 0, 1, torch.device("cpu"), scans, tbs, xs, Ps, max_batch=3)
    assert solo.shape == (8, 18) and (solo[:, 15] > 100).all()     # every scan matched points
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_same_rows, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert np.array_equal(res[r], solo), (r, np.abs(res[r] - solo).max())


class OracleOverlayEngine(OracleReplayEngine):
    """TEST-ONLY engine for replay.replay_batch_overlay: answers batch_replay_overlay the way lk_batch_replay_overlay_dev is specified -
    every scan through KILO::process WITH insert on a PRIVATE copy of the shared map (the oracle re-imports the base blob per scan)."""

    def __init__(self):
        super().__init__()
        self.blob = self.o.map_export()
        self.priors = None
        self.batches = []

    def overlay_inputs(self, n, n_pts=480, n_b=4):
        from legkilo_amd import synth

        scans, xs, Ps = [], [], []
        for k in range(n):
            tb = 1.0 + 0.1 * k
            sc = np.array(self.scenes.vlp_scan_input(self.sc, tb, k)[:n_pts], copy=True)
            assert len(sc) == n_pts
            sc["curvature"] = (0.025 * (np.arange(n_pts) // (n_pts // n_b))).astype(np.float32)   # one bucket table for the whole batch
            scans.append(sc)
            xs.append(synth.initial_state(self.sc.traj, tb, self.sc.P, np.random.default_rng(100 + k), 0.02, 0.5))
            Ps.append((1e-4 * np.eye(30)).reshape(900))
        off, dt = synth.buckets_of(scans[0])
        return scans, off, dt, np.stack(xs), np.stack(Ps)

    def batch_set_priors(self, x36, P900):
        self.priors = (np.array(x36, copy=True).reshape(-1, 36), np.array(P900, copy=True).reshape(-1, 900))

    def batch_replay_overlay(self, scans, t_begin, bucket_off, bucket_dt):
        from legkilo_amd import abi

        assert self.priors is not None and len(self.priors[0]) == len(scans)
        self.batches.append(len(scans))
        out = np.zeros(len(scans), dtype=abi.pose_dtype())
        self.states = []
        for i, sc in enumerate(scans):
            self.o.map_import(self.blob)
            self.o.set_map_insert(True)
            self.o.set_state(self.priors[0][i], self.priors[1][i].reshape(30, 30))
            self.o.set_times(t_begin, t_begin)
            pose, _ = self.o.process_scan(sc, t_begin)
            out["pos"][i], out["vel"][i], out["rot"][i] = pose.pos, pose.vel, pose.rot
            out["n_effect"][i], out["n_buckets"][i], out["n_updates"][i] = pose.n_effect, pose.n_buckets, pose.n_updates
            self.states.append(self.o.get_state())
        return out

    def batch_get_states(self, first_slot, n):
        x = np.stack([np.asarray(s[0]).reshape(36) for s in self.states[first_slot:first_slot + n]])
        P = np.stack([np.asarray(s[1]).reshape(30, 30) for s in self.states[first_slot:first_slot + n]])
        return x, P


def _worker_overlay(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleOverlayEngine()
    scans, off, dt, xs, Ps = eng.overlay_inputs(7)
    rows, xa, Pa = replay.replay_batch_overlay(eng, dist, rank, world, torch.device("cpu"), scans, off, dt, xs, Ps, max_batch=3, want_states=True)
    a, b = replay.shard_range(7, rank, world)
    q.put((rank, rows, xa, Pa, sum(eng.batches) == b - a and max(eng.batches) <= 3))
    dist.destroy_process_group()


def test_overlay_replay_world2_equals_world1():
    """Config 5 WITH the map insert, sharded (replay.replay_batch_overlay): 7 equally shaped scans replayed unsharded and over 2 gloo
    ranks, batches of at most 3 - every rank ends up with the same 7 result rows and state records, bit for bit, in scan order; the
    insert matters (the same scans against the frozen map match a different number of points)."""
    eng = OracleOverlayEngine()
    scans, off, dt, xs, Ps = eng.overlay_inputs(7)
    solo, xs1, Ps1 = replay.replay_batch_overlay(eng, None, 0, 1, torch.device("cpu"), scans, off, dt, xs, Ps, max_batch=3, want_states=True)
    assert solo.shape == (7, 18) and (solo[:, 15] > 50).all() and (solo[:, 16] == len(dt)).all()
    assert xs1.shape == (7, 36) and Ps1.shape == (7, 900)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlay, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (rows, xa, Pa, ok) for r, rows, xa, Pa, ok in (q.get(timeout=240) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        rows, xa, Pa, ok = res[r]
        assert ok
        assert np.array_equal(rows, solo), (r, np.abs(rows - solo).max())
        assert np.array_equal(xa, xs1) and np.array_equal(Pa, Ps1)
