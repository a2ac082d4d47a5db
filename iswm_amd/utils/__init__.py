from .loss import CrossEntropyLoss, FocalLoss, calculate_class_weights, create_loss  # noqa: F401
