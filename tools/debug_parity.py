"""Debug: HIP batch replay vs oracle on (a) the HIP-built map imported into the oracle, (b) the oracle's own map imported into HIP."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import lk_pkg; lk_pkg.load()
from legkilo_amd import abi, binding, config, synth
import oracle_binding as ob
import bench as B
import torch

P = config.LEG_FUSION
traj = synth.Trajectory()
B._init_worker()
t0 = 5.0
warm_t = [t0 + 3.0 * k for k in range(3)]
first = B._gen(("first", (t0,)))
warm = [B._gen(("dense", (tb, 5, 2002 + k, 3003 + k))) for k, tb in enumerate(warm_t)]
S = 6
scans = [B._gen(("dense", (B.scan_time(5.0, u), 5, 5005 + u, 1000003 + u))) for u in range(S)]
cfg = config.make_config(P, n_slots=2 * S, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
g = binding.LegKiloHip(cfg)
B.build_map(g, traj, P, first, warm, warm_t)
blob = g.map_export()
o = ob.Oracle(cfg, imu_mode_only=True)
o.init_process_cov_q(); o.set_acc_norm(9.81)
o.map_import(blob)
print("re-export equal:", np.array_equal(np.asarray(o.map_export()), np.asarray(blob)), len(blob))
o.set_map_insert(False)
off, dt = synth.buckets_of(scans[0])
xs = np.stack([synth.initial_state(traj, B.scan_time(5.0, s), P, np.random.default_rng(9009 + s), 0.02, 0.5) for s in range(S)])
Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
dev = torch.device("cuda")
d_batch = torch.from_numpy(np.stack([np.ascontiguousarray(sc).view(np.uint8).reshape(-1) for sc in scans])).to(dev)
d_x, d_P = torch.from_numpy(xs).to(dev), torch.from_numpy(Ps).to(dev)
g.batch_set_priors_dev(d_x.data_ptr(), d_P.data_ptr(), S)
poses = np.frombuffer(g.batch_replay_dev(d_batch.data_ptr(), S, 100000, 0.0, off, dt), dtype=abi.pose_dtype())
ring = torch.empty(S * abi.pose_dtype().itemsize, dtype=torch.uint8).pin_memory()
g.batch_replay_async_dev(d_batch.data_ptr(), 0, S, 100000, 0.0, off, dt, d_x36=d_x.data_ptr(), d_P900=d_P.data_ptr(), host_out_ptr=ring.data_ptr())
g.synchronize()
pa = ring.numpy().view(abi.pose_dtype())
for s in range(S):
    o.set_state(xs[s], Ps[s]); o.set_times(0.0, 0.0)
    po, _ = o.process_scan(scans[s], 0.0, with_sort=True)
    print(s, "oracle", po.n_buckets, po.n_updates, po.n_effect, "| sync", poses[s]["n_buckets"], poses[s]["n_updates"], poses[s]["n_effect"],
          "| async", pa[s]["n_buckets"], pa[s]["n_updates"], pa[s]["n_effect"],
          "dpos sync %.3g async %.3g" % (np.abs(np.array(po.pos) - poses[s]["pos"]).max(), np.abs(np.array(po.pos) - pa[s]["pos"]).max()))
# residual rows on both, state = prior of scan 0
o.set_state(xs[0], Ps[0]); g.set_state(xs[0], Ps[0].reshape(30, 30))
xb = np.stack([scans[0]["x"], scans[0]["y"], scans[0]["z"]], 1)[:20000].astype(np.float32)
ho, zo, Ro, vo = o.residuals(xb)
hg, zg, Rg, vg = g.residuals(xb)
print("residual valid: oracle", vo.sum(), "hip", vg.sum(), "diff", int((vo != vg).sum()))
both = (vo & vg).astype(bool)
print("max |z| delta", np.abs(np.abs(zo[both]) - np.abs(zg[both])).max(), "R rel", np.abs(Ro[both] / Rg[both] - 1).max())
# (b) oracle builds its own map
o2 = ob.Oracle(cfg, imu_mode_only=True)
B.build_map(o2, traj, P, first, warm, warm_t)
o2.set_map_insert(False)
o2.set_state(xs[0], Ps[0])
h2, z2, R2, v2 = o2.residuals(xb)
print("own-map oracle valid", v2.sum(), "vs imported-map oracle diff", int((v2 != vo).sum()), "vs hip diff", int((v2 != vg).sum()))
import scenes
try:
    print(scenes.compare_maps(o2.map_export(), blob, rtol=1e-5, ptol=1e-6))
except AssertionError as e:
    print("map compare:", str(e)[:300])
