from .data_parallel import DataParallel

__all__ = ["DataParallel"]
