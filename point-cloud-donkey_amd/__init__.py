"""point-cloud-donkey_amd — MI355X-native implicit_shape_model recognition hot path.

The product is csrc/ -> libismhip.so behind include/ismhip.h (C ABI) plus the C++ host mirror of the reference's
plugin interface (host/). This Python package is the thin harness used by tests/ and bench.py:
  capi      ctypes marshalling of the C ABI (torch tensors = device memory only)
  pipeline  batch recognition driver built on capi (train a codebook the reference's way, detect a batch)
  synthetic seeded ModelNet-like object generator (real datasets are not available offline)
The directory name contains a hyphen; load it with __graft_entry__.load_package() (module name point_cloud_donkey_amd).
"""
from . import capi  # noqa: F401
from . import synthetic  # noqa: F401
from . import pipeline  # noqa: F401
from . import shard  # noqa: F401
