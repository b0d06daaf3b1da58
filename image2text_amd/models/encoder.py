"""Vision encoders of the plugin surface (reference models/encoder.py:25-195).

``Encoder.from_config`` keeps the reference's factory contract.  Only the from-scratch dense ViT is executed by the
HIP hot path; a ``PretrainedViTConfig`` (torchvision backbone + network fetch) is refused loudly.
"""
import abc
import math
from typing import Union

import torch
import torch.nn as nn

from ..configs.models import PretrainedViTConfig, VisionTransformerEncoderConfig
from .layers import ConvMLP, LayerNorm, LayerNormND, TransformerBlock


class Encoder(nn.Module, abc.ABC):
    """Base class: ``forward(images) -> (B, num_outputs, output_embed_dim)``."""

    def __init__(self, config):
        super().__init__()
        self.config = config

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        raise ValueError('Not implemented in base class!')

    @classmethod
    def from_config(cls, config: Union[VisionTransformerEncoderConfig, PretrainedViTConfig]):
        if isinstance(config, VisionTransformerEncoderConfig):
            return VisionTransformerEncoder(config)
        if isinstance(config, PretrainedViTConfig):
            raise NotImplementedError('PretrainedViT (torchvision ViT-B/16 SWAG weights + heads) is outside the HIP hot '
                                      'path: it needs a network fetch and third-party arithmetic (SURVEY.md 8(f) next #3)')
        raise ValueError('Unknown config')

    @property
    def num_outputs(self):
        raise ValueError('Not implemented in base class')

    @property
    def output_embed_dim(self):
        raise ValueError('Not implemented in base class')


class VisionTransformerEncoder(Encoder):
    """From-scratch ViT: conv feature extractor -> flat 'patches' -> projector -> LayerNormND (twice, shared weights,
    position embedding in between) -> CLS tokens prepended -> non-causal blocks -> ln_f on the CLS rows
    (reference encoder.py:130-178).  Holds the parameters; ``forward`` runs the HIP path."""

    def __init__(self, config: VisionTransformerEncoderConfig):
        super().__init__(config)
        self.n_patches = n = config.num_patches
        assert config.input.width % n == 0
        assert config.input.height % n == 0
        self.patch_size = (config.input.width // n, config.input.height // n)
        ac = config.transformer_config.attn_config
        self.feature_extractor = ConvMLP(config.input.n_channels, config.n_channels, config.feature_extractor_kernel_size,
                                         config.feature_extractor_gate_sizes)
        self.input_d = config.n_channels * self.patch_size[0] * self.patch_size[1]
        self.out_dim = ac.n_embd
        if self.input_d % 8 or (config.input.width * config.input.height) % 8:
            raise NotImplementedError('patch size must give a flat patch length that is a multiple of 8')
        self.projector = nn.Linear(self.input_d, self.out_dim, bias=ac.bias)
        self.ln_input = LayerNormND((n ** 2, self.out_dim), ac.bias)
        self.transformer = nn.ModuleDict(dict(
            wpe=nn.Embedding(n ** 2, self.out_dim),
            drop=nn.Dropout(ac.dropout),
            h=nn.ModuleList([TransformerBlock(config.transformer_config, seed=depth) for depth in range(config.n_layer)]),
            ln_f=LayerNorm(self.out_dim, bias=ac.bias),
        ))
        self.cls_token = nn.Parameter(torch.randn(1, config.n_cls, self.out_dim) / math.sqrt(self.out_dim))
        self.n_cls = config.n_cls
        self.enable_gradient_checkpointing = config.enable_gradient_checkpointing
        self._standalone = None

    def forward(self, images: torch.Tensor):
        """Standalone use (not through VisionEncoderDecoder): wraps itself in a one-module hot path."""
        from .vision_encoder_decoder import run_encoder_standalone
        return run_encoder_standalone(self, images)

    @property
    def num_outputs(self):
        return self.n_cls

    @property
    def output_embed_dim(self):
        return self.out_dim
