from .utils import _item_or_list, _single, _pair, _triple, apply_wb, traverse

__all__ = ['_item_or_list', '_single', '_pair', '_triple', 'apply_wb', 'traverse']
