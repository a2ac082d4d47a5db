"""Host-side pieces of the hot path that the reference keeps in utils/: the criterion (utils/loss.py) and the input
transforms (utils/ext_transforms.py).  The reference's plotting / scheduler / denormalisation helpers are not on the
training step and are not rebuilt."""
from .loss import CrossEntropyLoss, FocalLoss, calculate_class_weights, create_loss  # noqa: F401
