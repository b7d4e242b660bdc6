"""Mirror of the reference's ``core_functions`` package for the vision MAML/ANIL hot path."""
from .vision import fast_adapt, accuracy, evaluate, meta_batch_adapt
from .vision_models import OmniglotCNN, MiniImagenetCNN, ConvBase, ConvBlock
from .maml import MAML
from ..utils.data_pre import prepare_batch
