"""MI355X-native P3D saliency forward/backward path (drop-in for the reference's p3d.py graph
functions and the session calls around them).  All arithmetic runs in libp3dhip.so (HIP, gfx950)."""
from ._lib import LIB_PATH, P3dError, lib        # noqa: F401
from .session import P3DSession                  # noqa: F401
from . import ops                                # noqa: F401
from . import p3d, p3d_gn                        # noqa: F401  (eager mirrors of the reference's graph functions)
from . import metrics, dataflow                  # noqa: F401  (utils/metrics.py and dataflow.py mapf on the GPU)
