"""``from learn2learn.algorithms.maml import MAML as MAML_BASE`` (reference core_functions/maml.py:8) -> ``exploring_meta_amd.algorithms.maml``."""
from ..core_functions.maml import MAML

__all__ = ['MAML']
