"""legkilo_amd — MI355X-native hot path of Leg-KILO (per-time-bucket LiDAR ESKF update).

Layout
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI shim (include/legkilo_hip.h)
  host/      C++ mirror of the reference class surface (ESKF / VoxelMapManager / KILO path)
  abi.py     ctypes declarations of the C-ABI PODs
  binding.py ctypes wrapper over liblegkilo_hip.so (fails loudly when the library is missing)
  config.py  parameter sets keyed like legkilo/config/*.yaml
  synth.py   synthetic world / scan / IMU / kinematic stream generator
  replay.py  one-process-per-GPU batch replay (torch.distributed, RCCL map broadcast)

The product path never imports anything from oracle/.
"""
__all__ = ["abi", "binding", "config", "synth", "replay", "tum", "checkpoint"]
