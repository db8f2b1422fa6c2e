import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import lk_pkg; lk_pkg.load()
from legkilo_amd import abi, binding, config, synth
import oracle_binding as ob
np.set_printoptions(linewidth=200, precision=4)
root = ROOT
exe = "/tmp/lk_host_example_gpu"
r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "leg-kilo_amd", "host"),
                    os.path.join(root, "leg-kilo_amd", "host", "example_kilo_path.cc"), "-o", exe, "-L", os.path.join(root, "leg-kilo_amd"),
                    "-llegkilo_hip", "-Wl,-rpath," + os.path.join(root, "leg-kilo_amd")], capture_output=True, text=True)
print(r.stderr[-500:])
dump = "/tmp/d.bin"
r = subprocess.run([exe, dump], capture_output=True, text=True); print(r.stdout)
raw = open(dump, "rb").read()
n, nb, n_succ, n_poses = np.frombuffer(raw, dtype=np.uint32, count=4)
o_ = 16
world = np.frombuffer(raw, dtype=np.float32, count=3 * n, offset=o_).reshape(n, 3); o_ += 12 * n
body = np.frombuffer(raw, dtype=np.float32, count=3 * n, offset=o_).reshape(n, 3); o_ += 12 * n
x_hip = np.frombuffer(raw, dtype=np.float64, count=36, offset=o_)
x0 = np.zeros(36); x0[:9] = np.eye(3).reshape(9); x0[9:12] = [0, 0, 0.5]; x0[21:24] = [0, 0, -9.81]
res = {}
for name, mk in (("oracle", lambda: ob.Oracle(config.make_config(config.LEG_FUSION), imu_mode_only=True)), ("hip", lambda: binding.LegKiloHip(config.make_config(config.LEG_FUSION)))):
    o = mk()
    o.set_state(x0, 1e-6 * np.eye(30)); o.init_process_cov_q(); o.set_times(0.0, 0.0)
    o.map_build(world, body)
    _, _, ne = o.update_points(0.01, body[:nb])
    x, _ = o.get_state()
    res[name] = x
    print(name, ne, x)
print("mirror ", n_succ, x_hip)
print("hip-oracle", np.abs(res["hip"] - res["oracle"]).max(), "mirror-hip", np.abs(res["hip"] - x_hip).max())
