"""Many-scan batch replay across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path has no exchange step inside it: scans are independent replay units against one
shared, frozen voxel map.  Communication is therefore exactly
  (1) once per map snapshot: the map blob from the rank that built it to every other rank;
  (2) once per batch: an all-gather of the per-scan results (a few hundred bytes per scan).
The same holds for the replay WITH the map insert (replay_batch_overlay): every scan's copy-on-write overlay is private to it.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring broadcast is bound by ONE link, so for a
large blob `scatter_allgather` sends a distinct 1/W slice to every peer (all links of the root carry
different data) and then all-gathers the slices.

`engine` is anything with the LegKiloHip map methods (map_export / map_import for host blobs,
map_export_dev_size / map_export_dev / map_import_dev for HBM-resident blobs).
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous block partition of n_total replay units: rank r owns [start, stop)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _bcast_tensor(dist, t, src, algo, rank, world):
    import torch

    if algo == "broadcast" or world == 1:
        dist.broadcast(t, src)
        return t
    # scatter + all-gather: pad to a multiple of world, root sends slice i to rank i, then all-gather
    n = t.numel()
    per = (n + world - 1) // world
    padded = t if n == per * world else torch.cat([t, t.new_zeros(per * world - n)])
    mine = torch.empty(per, dtype=t.dtype, device=t.device)
    chunks = list(padded.split(per)) if rank == src else None
    dist.scatter(mine, chunks, src=src)
    out = torch.empty(per * world, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, mine)
    t.copy_(out[:n])
    return t


def broadcast_map(engine, dist, rank, world, device, src=0, algo="broadcast"):
    """Ship the voxel map of rank `src` to every rank.  Returns (bytes, seconds spent in the collective)."""
    import time

    import torch

    on_gpu = device.type == "cuda"
    if on_gpu:
        size = engine.map_export_dev_size() if rank == src else 0
    else:
        host_blob = engine.map_export() if rank == src else None
        size = int(host_blob.size) if rank == src else 0
    nbytes = torch.tensor([size], dtype=torch.int64, device=device)
    if world > 1:
        dist.broadcast(nbytes, src)
    n = int(nbytes.item())
    blob = torch.empty(n, dtype=torch.uint8, device=device)
    if rank == src:
        if on_gpu:
            engine.map_export_dev(blob.data_ptr(), n)
        else:
            blob.copy_(torch.from_numpy(np.asarray(host_blob)))
    if on_gpu:
        torch.cuda.synchronize()
    secs = 0.0
    if world > 1:
        dist.barrier()
        t0 = time.perf_counter()
        _bcast_tensor(dist, blob, src, algo, rank, world)
        if on_gpu:
            torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        if rank != src:
            if on_gpu:
                engine.map_import_dev(blob.data_ptr(), n)
            else:
                engine.map_import(blob.numpy())
    return n, secs


def gather_results(dist, local, world, device):
    """All-gather per-scan result rows (float64 [n_local, k]); every rank gets [n_total, k] in rank order.
    Shards may differ in length by one (shard_range), so rows are padded to the longest shard."""
    import torch

    local = np.ascontiguousarray(local, dtype=np.float64)
    if world == 1:
        return local
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = np.zeros((m, local.shape[1]))
    pad[: local.shape[0]] = local
    t = torch.from_numpy(pad).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)], axis=0)


def gather_pose_bytes(dist, local_bytes, world, device):
    """All-gather the raw lk_pose records of a replay (a uint8 tensor, e.g. the pinned ring the asynchronous batch entry
    writes its poses to; the same number of bytes on every rank) and leave the result ON THE DEVICE: [world, n_bytes] uint8.
    Nothing is converted or copied back here - that is the caller's business outside any timed region."""
    import torch

    t = local_bytes.reshape(-1).to(device, non_blocking=True)
    if world == 1:
        return t.reshape(1, -1)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out)


def gather_state_records(dist, engine, first_slot, n_local, world, device):
    """SURVEY 8(e)'s per-scan result record - state (36) and covariance (900: ESKF::cov(), getRotCov / getPosCov / getVelCov are its
    blocks (0,0), (3,3), (6,6), eskf.h:46-109) - of the n_local scans a rank replayed in filter slots first_slot .. first_slot +
    n_local, all-gathered in rank order: returns (x [n_total, 36], P [n_total, 900]).  Every rank must pass the same n_local
    (bench: scans_per_gpu).  On a GPU the records never touch the host: one gather kernel (lk_batch_get_states_dev) fills two
    device tensors and RCCL all-gathers them device to device; on the CPU (gloo tests) the engine's batch_get_states arrays are
    all-gathered."""
    import torch

    if device.type == "cuda":
        x = torch.empty((n_local, 36), dtype=torch.float64, device=device)
        P = torch.empty((n_local, 900), dtype=torch.float64, device=device)
        engine.batch_get_states_dev(first_slot, n_local, x.data_ptr(), P.data_ptr())
        engine.synchronize()
    else:
        xh, Ph = engine.batch_get_states(first_slot, n_local)
        x = torch.from_numpy(np.ascontiguousarray(xh, dtype=np.float64).reshape(n_local, 36))
        P = torch.from_numpy(np.ascontiguousarray(Ph, dtype=np.float64).reshape(n_local, 900))
    if world == 1:
        return x, P
    xo = torch.empty((world * n_local, 36), dtype=torch.float64, device=device)
    Po = torch.empty((world * n_local, 900), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(xo, x)
    dist.all_gather_into_tensor(Po, P)
    return xo, Po


def pose_rows(poses):
    """lk_pose records (a ctypes array, or any buffer / numpy array of abi.pose_dtype()) -> float64 rows
    [pos(3), vel(3), rot(9), n_effect, n_buckets, n_updates]; vectorised (a 1024-scan batch per step)."""
    from . import abi

    a = poses if isinstance(poses, np.ndarray) and poses.dtype == abi.pose_dtype() else np.frombuffer(poses, dtype=abi.pose_dtype())
    return np.concatenate([a["pos"], a["vel"], a["rot"], a["n_effect"].astype(np.float64)[:, None],
                           a["n_buckets"].astype(np.float64)[:, None], a["n_updates"].astype(np.float64)[:, None]], axis=1)


def replay_recorded_run(engine, dist, rank, world, device, scans, t_begins, xs, Ps, max_batch):
    """Batch replay of a recorded run against the engine's (already distributed, frozen) map: scans i in [0, n) are
    block-partitioned over the ranks (shard_range), every rank replays its block in ragged batches of at most
    `max_batch` scans (= the engine's filter slots; lk_batch_replay_ragged_dev: every scan keeps its own size, time
    buckets and start time), and the per-scan result rows (pose_rows) are all-gathered in scan order.
    `scans` / `t_begins` / `xs` / `Ps` are the full lists on every rank (only the rank's block is touched)."""
    start, stop = shard_range(len(scans), rank, world)
    rows = []
    for a in range(start, stop, max_batch):
        b = min(a + max_batch, stop)
        poses = engine.batch_replay_ragged(scans[a:b], t_begins[a:b], xs[a:b], Ps[a:b])
        rows.append(pose_rows(poses))
    local = np.concatenate(rows, axis=0) if rows else np.zeros((0, 18))
    return gather_results(dist, local, world, device) if world > 1 else local


def replay_batch_overlay(engine, dist, rank, world, device, scans, bucket_off, bucket_dt, xs, Ps, max_batch, t_begin=0.0, want_states=False):
    """Config 5 WITH the map insert, sharded: scans i in [0, n) - equally shaped (same size, same bucket table: lk_batch_replay_overlay_dev)
    - are block-partitioned over the ranks; every rank replays its block in batches of at most `max_batch` scans (= the engine's filter
    slots), each scan on its own copy-on-write overlay of the engine's (already distributed) map (KILO.cc:216-233 after every bucket);
    the per-scan result rows (pose_rows) are all-gathered in scan order.  An overlay is private to its scan, so there is no collective
    on the data path: exactly the frozen-map replay's communication (map blob once, results once).
    `scans` is the full list of lk_point arrays on every rank (only the rank's block is uploaded) or, on a GPU, a (device pointer,
    n_pts) pair addressing the RANK'S OWN block, already resident.  `xs` / `Ps` are the priors of ALL scans ([n, 36] / [n, 900]).
    want_states: also return the all-gathered (x [n, 36], P [n, 900]) records (every rank's block must then be equally long)."""
    n = len(xs)
    start, stop = shard_range(n, rank, world)
    resident = isinstance(scans, tuple)
    n_pts = scans[1] if resident else len(scans[0])
    rows, xl, Pl = [], [], []
    for a in range(start, stop, max_batch):
        b = min(a + max_batch, stop)
        engine.batch_set_priors(np.ascontiguousarray(xs[a:b]), np.ascontiguousarray(Ps[a:b]))
        if resident:
            poses = engine.batch_replay_overlay_dev(scans[0] + (a - start) * n_pts * 16, b - a, n_pts, t_begin, bucket_off, bucket_dt)
        else:
            poses = engine.batch_replay_overlay(scans[a:b], t_begin, bucket_off, bucket_dt)
        rows.append(pose_rows(poses))
        if want_states:
            x_, P_ = engine.batch_get_states(0, b - a)
            xl.append(np.asarray(x_).reshape(b - a, 36)), Pl.append(np.asarray(P_).reshape(b - a, 900))
    local = np.concatenate(rows, axis=0) if rows else np.zeros((0, 18))
    out = gather_results(dist, local, world, device) if world > 1 else local
    if not want_states:
        return out
    xa = np.concatenate(xl, axis=0) if xl else np.zeros((0, 36))
    Pa = np.concatenate(Pl, axis=0) if Pl else np.zeros((0, 900))
    if world > 1:
        xa, Pa = gather_results(dist, xa, world, device), gather_results(dist, Pa, world, device)
    return out, xa, Pa
