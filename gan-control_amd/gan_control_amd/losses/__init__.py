"""Attribute / contrastive (same - not-same) losses of the controllable generator step."""
from .loss_model import LossModelClass, CRITERIA, PREDICTORS  # noqa: F401
