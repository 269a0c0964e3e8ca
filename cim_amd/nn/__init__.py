"""Mirror of /root/reference/lib/nn (only what the training hot path uses: DataParallel)."""
from .parallel import DataParallel

__all__ = ["DataParallel"]
