"""Device-side construction of the per-image training blobs (f-3), mirror of /root/reference/lib/roi_data."""
from .minibatch import get_minibatch, get_minibatch_blob_names  # noqa: F401
