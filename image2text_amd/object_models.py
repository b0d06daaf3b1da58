"""Output record of VisionEncoderDecoder.forward (same field names as the reference's object_models.py:4-5)."""
from typing import NamedTuple

import torch


class VisionEncoderDecoderModelOutput(NamedTuple):
    encoder_output: torch.Tensor
    logits: torch.Tensor
    hidden_state: torch.Tensor
