"""d3fields_amd -- MI355X-native (gfx950) field query of d3fields.

    from d3fields_amd import Fusion            # drop-in for the reference's fusion.Fusion query API
    from d3fields_amd import corr_utils        # drop-in for the reference's utils/corr_utils.py

The compute lives in libd3fields_hip.so (hand-written HIP, C ABI in include/d3fields_hip.h);
this package is the host-side mirror of the reference's Python interface.
"""
from .fusion import Fusion, create_init_grid, fps, instance2onehot, onehot2instance  # noqa: F401
from . import corr_utils  # noqa: F401
from . import pcd_utils  # noqa: F401
from . import rigid  # noqa: F401
from . import sharding  # noqa: F401

__version__ = "0.1.0"
