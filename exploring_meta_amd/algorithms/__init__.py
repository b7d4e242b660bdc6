"""Import path of the reference's drivers: ``from learn2learn.algorithms import MAML`` (vision/maml_vision.py:9, anil_vision.py:9,
rl/maml_ppo.py:11, rl/anil_trpo.py:12, misc_scripts/eval_rl.py:9) becomes ``from exploring_meta_amd.algorithms import MAML``; the class is
core_functions.maml.MAML (``MAML(model, lr, first_order=False)``, ``.clone()``, ``.adapt(loss)``, ``.module``)."""
from ..core_functions.maml import MAML
from . import maml

__all__ = ['MAML', 'maml']
