"""pyimcom_amd -- the IMCOM postage-stamp path on MI355X (see README.md / DESIGN.md)."""

import os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue wait
# for each other.  This package runs up to eight at a time (main, selection, upload, the Eigen kernel's second queue and sub-batch
# streams, the caller's own): with four queues the Eigen path at batch 32 lost 30 % (7.2 -> 9.5 ms per cfg-3 stamp) and the Block seam
# 20 % as soon as one more stream existed in the process.  Read by the runtime when it initialises, i.e. at the first HIP call: import
# this package (or export the variable) before touching the device.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
