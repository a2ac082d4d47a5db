from .utils import *  # noqa: F401,F403
from .utils import Denormalize, denormalize, fix_bn, mkdir, set_bn_momentum  # noqa: F401
from .scheduler import PolyLR  # noqa: F401
from .loss import CrossEntropyLoss, FocalLoss, calculate_class_weights, create_loss  # noqa: F401
