"""pyimcom_amd -- the IMCOM postage-stamp path on MI355X (see README.md / DESIGN.md)."""

import os
import sys

# The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue wait for
# each other.  The path itself keeps to few streams -- the caller's, ONE side stream for the pixel selection and the uploads, and a
# second queue of the library's that exists only while a call uses it (the Eigen kernel's sub-batch or overlap stream) -- so it fits
# the default beside a host application of one or two streams of its own.  A host with more streams should give the runtime more
# queues (export GPU_MAX_HW_QUEUES=8 before the first HIP call: with one stream more than queues the Eigen path at batch 32 lost
# 30 %, 7.2 -> 9.5 ms per cfg-3 stamp).  A plug-in does not change its host's runtime configuration behind its back: the variable is
# set here ONLY when this import comes before anything has touched the device (then it is this package's process to configure, e.g.
# bench.py, the tests, the farm's ranks); if the runtime is already up and the variable is unset, the situation is reported once.
if "GPU_MAX_HW_QUEUES" not in os.environ:
    _torch = sys.modules.get("torch")
    _up = bool(_torch is not None and getattr(_torch, "cuda", None) is not None and _torch.cuda.is_initialized())
    if _up:
        import warnings

        warnings.warn("pyimcom_amd: the HIP runtime was initialised before this import with GPU_MAX_HW_QUEUES unset (4 hardware queues): "
                      "if the application runs more than two streams of its own beside this package, export GPU_MAX_HW_QUEUES=8 before the "
                      "first HIP call (streams that share a hardware queue serialise)", RuntimeWarning, stacklevel=2)
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
