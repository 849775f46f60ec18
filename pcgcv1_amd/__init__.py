"""pcgcv1_amd — the PCGCv1 hyperprior hot path on MI355X (see DESIGN.md / INTEGRATION.md)."""
import os

# The host pipelines drive about eight HIP streams per process (two pipeline streams, each with an entropy and an upload
# stream, the z stream, the encode-ahead stream).  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) in creation order: with 4, a decoder pipeline's entropy stream lands on the queue of its own synthesis stream
# and the next slice's hyper decoder / CDF rows wait for the previous slice's synthesis (tools/timeline2.py: 371 -> 354 ms
# per 1 640-cube cloud with 8 or more; per 205-cube cloud 49.2 ms with 4, 48.5 with 8, 48.0 with 16).  Read by the HIP runtime when it initialises, so this
# only takes effect if the package is imported before the first HIP call; bench.py and tests/conftest.py set it themselves.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# dmabuf IPC between the processes of one node (RCCL, shared device tensors): the host driver of this pool supports nothing else
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
