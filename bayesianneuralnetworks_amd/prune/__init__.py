from .prune import PruneNormal

__all__ = ['PruneNormal']
