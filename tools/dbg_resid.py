import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import lk_pkg; lk_pkg.load()
from legkilo_amd import synth, binding as hip_lib
sys.path.insert(0, "oracle")
import oracle_binding as oracle_lib; oracle_lib.build()
import scenes
scene = scenes.Scene()
o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
t0 = 5.0
x0 = scenes.init_filter(o, scene, t0)
scenes.first_frame(o, scene, t0, x0, dense=100000)
for k in range(2):
    tb = t0 + 0.1 * k
    p_ = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=2002 + k, seed_noise=3003 + k)
    o.process_scan(p_, tb)
g2 = hip_lib.LegKiloHip(scene.cfg())
g2.map_import(o.map_export())
ts = t0 + 0.3
pts = synth.dense_scan(scene.world, scenes.Frozen(scene.traj, ts), ts, scene.P, n=100000, n_buckets=1, seed_scan=99)
xs = synth.initial_state(scene.traj, ts, scene.P)
_, Ps = o.get_state()
for obj in (o, g2):
    obj.set_state(xs, Ps)
xb = scenes.xyz_of(pts)
ho, zo, Ro, vo = o.residuals(xb)
for rep in range(3):
    hg, zg, Rg, vg = g2.residuals(xb)
    bad = np.nonzero(vo != vg)[0]
    print("rep", rep, "mismatch", len(bad), "oracle valid", int(vo.sum()), "gpu valid", int(vg.sum()))
    for b in bad[:20]:
        print("  i", b, "blk", b // 512, "wave", (b % 512) // 64, "lane", b % 64, "vo", vo[b], "vg", vg[b], "zg", zg[b], "zo", zo[b])
cm = scenes.canon_map(o.map_export())
P = scene.P


R = xs[:9].reshape(3, 3); pos = xs[9:12]
cfg = scene.cfg()
ER = np.array(cfg.ext_R[:]).reshape(3, 3); ET = np.array(cfg.ext_T[:])
vs = 0.5
for b in bad[:10]:
    pi = ER @ xb[b].astype(float) + ET
    pw = R @ pi + pos
    loc = (pw / vs).astype(np.float32)
    loc = np.where(loc < 0, (loc.astype(float) - 1.0).astype(np.float32), loc)
    key = tuple(int(v) for v in loc.astype(np.int64))
    near = list(key)
    for j in range(3):
        vc = (0.5 + key[j]) * vs
        if float(loc[j]) > vc + vs / 4: near[j] += 1
        elif float(loc[j]) < vc - vs / 4: near[j] -= 1
    print("  i", b, "pw", pw, "key", key, "home in map", key in cm, "near", tuple(near), "near in map", tuple(near) in cm,
          "home is_plane", cm[key]["is_plane"] if key in cm else None)
