"""nifty_amd -- MI355X-native implementation of the nifty.cl MGVI/geoVI hot path.

Host code is Python on PyTorch-ROCm tensors; all numerics on device Fields run in hand-written
HIP kernels (libniftyk, C ABI in include/niftyk.h) -- there is no CPU or eager-PyTorch fallback
for device data.
"""
__version__ = "0.1.0"
