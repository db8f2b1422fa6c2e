"""Checkpoint / resume of one handle (the reference has none; 8f rank 4: the map blob is the format).

A checkpoint = filter state (x, P, Q, the two time stamps, acc_norm_ of KILO.cc:349 read back from the handle) + the voxel-map blob of lk_map_export + the map's
last_slide_position (voxel_map.h:201).  Restoring it into a
fresh handle with the same configuration continues bit-identically: node / block ids may differ after the
re-import, results do not depend on them.
"""
import numpy as np


def save(path, handle):
    x, P = handle.get_state()
    tp, tu = handle.get_times()
    np.savez_compressed(path, x=x, P=P, Q=handle.get_Q(), times=np.array([tp, tu]), acc_norm=handle.get_acc_norm(),
                        last_slide_position=handle.get_last_slide_position(), blob=np.asarray(handle.map_export(), dtype=np.uint8))


def restore(path, handle):
    c = np.load(path)
    handle.map_import(c["blob"])
    handle.set_state(c["x"], c["P"])
    handle.set_Q(c["Q"])
    handle.set_times(float(c["times"][0]), float(c["times"][1]))
    handle.set_acc_norm(float(c["acc_norm"]))
    if "last_slide_position" in c:
        handle.set_last_slide_position(c["last_slide_position"])
    return c
