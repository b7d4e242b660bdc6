"""Mirror of the reference's ``core_functions`` package for the vision MAML/ANIL hot path."""
from .vision import fast_adapt, accuracy, evaluate, meta_batch_adapt
from .vision_models import OmniglotCNN, MiniImagenetCNN, ConvBase, ConvBlock
from .maml import MAML
from ..utils.data_pre import prepare_batch
from .policies import DiagNormalPolicy, DiagNormalPolicyANIL
from .rl import (fast_adapt_trpo, meta_optimize_trpo, meta_surrogate_loss, trpo_update, trpo_a2c_loss, fast_adapt_vpg, fast_adapt_ppo,
                 evaluate_vpg, evaluate_ppo, evaluate_trpo,
                 compute_advantages, set_device, LinearValue, Particles2DRunner, Particles2DEnv, EnvRunner, get_ep_successes)
