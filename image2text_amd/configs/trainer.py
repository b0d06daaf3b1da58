"""Trainer-side configuration schema (mirror of reference configs/trainer.py:6-41; plain data)."""
from typing import List, Optional, Tuple

from pydantic import BaseModel

from .models import VisionEncoderDecoderConfig


class TrainerWrapperConfig(BaseModel):
    """Loss options of ModelTrainerWrapper.  The default ``trainer: {}`` (CE only) is the hot path."""
    moco_momentum: Optional[float] = None
    moco_alpha: Optional[float] = None
    training_temperature: float = 1.0
    weight_fn: str = 'constant'
    mask_fraction: float = 0.0
    random_mask_fraction: float = 0.0
    eos_token_weight: Optional[float] = None
    add_contrastive_loss: bool = False
    training_contrastive_temperature: float = 1.0


class OptimizerConfig(BaseModel):
    lr: float
    weight_decay: float = 0.0
    betas: Tuple[float, float] = (0.9, 0.999)
    target_modules: Optional[List[str]] = None


class TrainingConfig(BaseModel):
    model: VisionEncoderDecoderConfig
    disable_flash: bool = False
    ignore_index: int = -100
    batch_size: int
    dataloader_buffer_size: int = 5
    shuffle: bool = True
    gradient_accumulation_steps: int = 1
    epochs: int = 1
    num_steps: Optional[int] = None
    num_val_steps: Optional[int] = None
    precision: str = 'no'
    tokenizer_str: str
    reset_moco_after_k_epochs: Optional[List[int]] = None
    trainer: TrainerWrapperConfig
    optimizers: List[OptimizerConfig]
    use_snr_optim: bool = False
