#!/usr/bin/env python
"""The torch.distributed side of the multi-GPU replay (leg-kilo_amd/replay.py) on the REAL backend - "nccl" = RCCL - with the one GPU a
gpurun box has: world size 1.  The collectives are degenerate, but everything around them is what an 8-GPU run executes:
init_process_group(nccl, device_id), the device-resident map blob through dist.broadcast AND through scatter + all_gather_into_tensor,
lk_map_import_dev of what arrived, a sharded ragged replay, the all-gather of pose records and of state + covariance records.
    python tools/rccl_world1_selftest.py      (run with MASTER_ADDR=127.0.0.1; writes one JSON line)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lk_pkg  # noqa: E402

lk_pkg.load()
import scenes  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from legkilo_amd import binding, replay, synth  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
sc = scenes.Scene()
S = 8
src = binding.LegKiloHip(sc.cfg(n_slots=S))
dst = binding.LegKiloHip(sc.cfg(n_slots=S))
t0 = 1.0
x0 = scenes.init_filter(src, sc, t0)
scenes.first_frame(src, sc, t0, x0)
scenes.replay_vlp(src, sc, t0, 3)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rccl": torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else None}
for algo in ("broadcast", "scatter_allgather"):
    t1 = time.perf_counter()
    nbytes, secs = replay.broadcast_map(src, dist, 0, 1, dev, src=0, algo=algo)
    # world 1 skips the import on the source rank: push the same bytes through the collective by hand and import them elsewhere
    blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.map_export_dev(blob.data_ptr(), nbytes)
    replay._bcast_tensor(dist, blob, 0, algo, 0, 1)
    torch.cuda.synchronize()
    dst.map_import_dev(blob.data_ptr(), nbytes)
    out[f"map_{algo}_bytes"] = int(nbytes)
    out[f"map_{algo}_s"] = round(time.perf_counter() - t1, 4)
assert scenes.maps_identical(src.map_export(), dst.map_export()) > 0
dst.init_process_cov_q()
scans = [scenes.vlp_scan_input(sc, t0 + 0.1 * (3 + k), 3 + k) for k in range(S)]
tbs = [t0 + 0.1 * (3 + k) for k in range(S)]
xs = [synth.initial_state(sc.traj, tb, sc.P) for tb in tbs]
Ps = [1e-4 * np.eye(30)] * S
rows = replay.replay_recorded_run(dst, dist, 0, 1, dev, scans, tbs, xs, Ps, max_batch=S)
poses = torch.from_numpy(np.ascontiguousarray(rows)).view(torch.uint8)
gathered = replay.gather_pose_bytes(dist, poses, 1, dev)
x_all, P_all = replay.gather_state_records(dist, dst, 0, S, 1, dev)
out.update({"scans": S, "rows": list(rows.shape), "matched_points": int(rows[:, 15].sum()), "gathered_pose_bytes": int(gathered.numel()),
            "state_records": list(x_all.shape), "cov_records": list(P_all.shape), "cov_finite": bool(torch.isfinite(P_all).all().item())})
dist.barrier()
dist.destroy_process_group()
print(json.dumps(out))
