"""The two steps in front of the path: pcl::VoxelGrid centroid filter + time sort.
CPU: the numpy oracle against a plain-loop restatement and its defining properties.
GPU: lk_preprocess_scan bit-exact against the oracle; raw scan -> pose equals oracle-preprocessed -> pose."""
import numpy as np
import pytest

import preprocess_oracle as po
import scenes
from legkilo_amd import synth


def raw_scan(scene, t, k=0):
    raw = synth.vlp16_scan(scene.world, scene.traj, t, scene.P, seed_noise=3003 + k)
    return synth.preprocess_velodyne(raw, scene.P["filter_num"], scene.P["blind"])


@pytest.fixture(scope="module")
def scene():
    return scenes.Scene(max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 15, max_scan_points=1 << 17)


def test_oracle_vectorised_equals_loops(scene):
    pre = raw_scan(scene, 1.0)[:1500]
    a, b = po.voxel_grid_centroid(pre, 0.3), po.voxel_grid_centroid_loops(pre, 0.3)
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f
    # the host-side definition the product documents (synth.py) is the same function
    c = synth.voxel_grid_centroid(pre, 0.3)
    for f in a.dtype.names:
        assert np.array_equal(a[f], c[f]), f


def test_oracle_properties(scene):
    pre = raw_scan(scene, 2.0)
    for leaf in (0.3, 0.5, 1.0):
        ds = po.voxel_grid_centroid(pre, leaf)
        assert 0 < len(ds) <= len(pre)
        inv = np.float32(1.0) / np.float32(leaf)
        cell = lambda p: np.floor(np.stack([p["x"], p["y"], p["z"]], 1).astype(np.float32) * inv).astype(np.int64)  # noqa: E731
        # every centroid lies in a distinct cell, and the set of cells is exactly the set of occupied cells
        cc = {tuple(r) for r in cell(ds)}
        assert len(cc) >= 0.999 * len(ds)  # a float32 centroid can round onto a cell face
        assert {tuple(r) for r in cell(pre)} >= set() and len({tuple(r) for r in cell(pre)}) == len(ds)
        # mass conservation of the time stamps: sum of (centroid * count) == sum of inputs (float32 rounding apart)
        srt = po.preprocess(pre, leaf)
        assert np.all(np.diff(srt["curvature"]) >= 0)
        assert sorted(srt["x"].tolist()) == sorted(ds["x"].tolist())
    # a leaf larger than the cloud: cells are the (at most 8) octants around the grid origin, and the
    # count-weighted mean of the centroids is the mean of the cloud
    one = po.voxel_grid_centroid(pre, 1000.0)
    assert 1 <= len(one) <= 8
    oct_id = (pre["x"] >= 0).astype(int) + 2 * (pre["y"] >= 0) + 4 * (pre["z"] >= 0)
    assert len(one) == len(np.unique(oct_id))
    cnt = np.array([np.sum(oct_id == u) for u in np.unique(oct_id)])
    assert abs((one["x"].astype(np.float64) * cnt).sum() / cnt.sum() - pre["x"].astype(np.float64).mean()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", [0.3, 0.5])
def test_gpu_preprocess_bit_exact(scene, hip_lib, leaf):
    g = hip_lib.LegKiloHip(scene.cfg())
    for k, pre in enumerate([raw_scan(scene, 1.0), raw_scan(scene, 7.3, 5)]):
        want = po.preprocess(pre, leaf)
        got = g.preprocess_scan(pre, leaf)
        assert len(got) == len(want), (len(got), len(want))
        for f in want.dtype.names:
            assert np.array_equal(got[f], want[f]), (k, f, int((got[f] != want[f]).sum()))
    # 100k-point raw cloud (the size the north star quotes), many points per cell
    big = synth.dense_scan(scene.world, scene.traj, 3.0, scene.P, n=100000, n_buckets=51, seed_scan=4242)
    want = po.preprocess(big, leaf)
    got = g.preprocess_scan(big, leaf)
    assert len(got) == len(want)
    for f in want.dtype.names:
        assert np.array_equal(got[f], want[f]), (f, int((got[f] != want[f]).sum()))
    g.close()


@pytest.mark.gpu
def test_gpu_raw_scan_to_pose(scene, hip_lib, oracle_lib):
    """raw cloud -> (device) voxel grid + sort -> bucket loop, against oracle preprocessing + oracle path."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    t0 = 1.0
    for obj in (o, g):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0)
    leaf = scene.P["voxel_grid_resolution"]
    for k in range(4):
        tb = t0 + 0.1 * k
        pre = raw_scan(scene, tb, k)
        imus = synth.imu_stream(scene.traj, tb, tb + 0.1, seed=3003 + k)
        po_, _ = o.process_scan(po.preprocess(pre, leaf), tb, imus=imus)
        pg, nd = g.process_raw_scan(pre, leaf, tb, imus=imus)
        assert nd == len(po.preprocess(pre, leaf))
        assert (po_.n_buckets, po_.n_updates, po_.n_effect) == (pg.n_buckets, pg.n_updates, pg.n_effect), k
        xo, _ = o.get_state()
        xg, _ = g.get_state()
        assert np.abs(xo - xg).max() < 1e-7, (k, np.abs(xo - xg).max())
    g.close()
    o.close()


# ----------------------------------------------------------------------------- sensor decode
def raw_message(scene, t, lidar_type):
    """A synthetic sensor_msgs::PointCloud2 payload in the Velodyne / Ouster / Hesai point layout."""
    pts = synth.vlp16_scan(scene.world, scene.traj, t, scene.P)
    dt = {1: po.VELODYNE_DTYPE, 2: po.OUSTER_DTYPE, 3: po.HESAI_DTYPE}[lidar_type]
    raw = np.zeros(len(pts), dtype=dt)
    raw["x"], raw["y"], raw["z"] = pts["x"], pts["y"], pts["z"]
    raw["intensity"] = 10.0
    if lidar_type == 1:
        raw["time"] = pts["curvature"]
        scale = 1.0
    elif lidar_type == 2:
        raw["t"] = np.round(pts["curvature"].astype(np.float64) * 1e9).astype(np.uint32)
        scale = 1e-9
    else:
        raw["timestamp"] = 1.7e9 + pts["curvature"].astype(np.float64)
        scale = 1.0
    tname = {1: "time", 2: "t", 3: "timestamp"}[lidar_type]
    layout = dict(point_step=dt.itemsize, off_x=dt.fields["x"][1], off_y=dt.fields["y"][1], off_z=dt.fields["z"][1],
                  off_time=dt.fields[tname][1], lidar_type=lidar_type)
    return raw, layout, scale


def test_decode_oracle_matches_vectorised_host_version(scene):
    raw, layout, scale = raw_message(scene, 1.0, 1)
    got, tb, te = po.decode(raw[:3000], 1, scale, 3, 1.5, header_stamp=100.0)
    host = synth.preprocess_velodyne(synth.vlp16_scan(scene.world, scene.traj, 1.0, scene.P)[:3000], 3, 1.5)
    assert len(got) == len(host) > 500
    for f in got.dtype.names:
        assert np.array_equal(got[f], host[f]), f
    assert abs(tb - (100.0 + float(raw["time"][0]))) < 1e-9 and te > tb
    # 2 ms bins (lidar_processing.cc:48)
    assert np.allclose(got["curvature"] * 500.0, np.round(got["curvature"] * 500.0), atol=1e-4)


@pytest.mark.parametrize("lidar_type", [1, 2])
def test_decode_vectorised_equals_the_loop(scene, lidar_type):
    """po.decode_vec (array form, used for the 65 536-point Ouster scans of the config-4 run) == po.decode (the plain loop that is
    pinned against the reference's handlers), field for field, incl. the begin / end stamps."""
    raw, _, scale = raw_message(scene, 1.5, lidar_type)
    raw = raw[:5000]
    a, ab, ae = po.decode(raw, lidar_type, scale, 3, 1.5, header_stamp=7.0)
    b, bb, be = po.decode_vec(raw, lidar_type, scale, 3, 1.5, header_stamp=7.0)
    assert len(a) == len(b) > 800 and (ab, ae) == (bb, be)
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.gpu
@pytest.mark.parametrize("lidar_type", [1, 2, 3])
def test_gpu_decode_bit_exact(scene, hip_lib, lidar_type):
    g = hip_lib.LegKiloHip(scene.cfg())
    raw, layout, scale = raw_message(scene, 2.0, lidar_type)
    raw = raw[:6000]  # the oracle is a plain Python loop
    want, tb, te = po.decode(raw, lidar_type, scale, 3, 1.5, header_stamp=50.0)
    got, gb, ge = g.decode_scan(raw.tobytes(), len(raw), layout, scale, 3, 1.5, header_stamp=50.0)
    assert len(got) == len(want) > 1000
    for f in want.dtype.names:
        assert np.array_equal(got[f], want[f]), (f, int((got[f] != want[f]).sum()))
    assert gb == tb and ge == te
    # decode -> voxel grid -> time sort, all on the device, equals the oracle chain
    full, layout, scale = raw_message(scene, 2.0, lidar_type)
    dec, _, _ = g.decode_scan(full.tobytes(), len(full), layout, scale, 3, 1.5)
    ds = g.preprocess_scan(dec, 0.3)
    assert np.all(np.diff(ds["curvature"]) >= 0) and 1000 < len(ds) < len(dec)
    want_ds = po.preprocess(dec, 0.3)
    for f in want_ds.dtype.names:
        assert np.array_equal(ds[f], want_ds[f]), f
    g.close()
