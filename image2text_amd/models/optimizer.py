"""``models.optimizer`` of the reference (models/optimizer.py): home of ``SNRAdam``, imported by trainer.py:10.
The implementation is the fused arena optimizer in ``training/optim.py`` (one HIP launch per step)."""
from ..training.optim import SNRAdam

__all__ = ['SNRAdam']
