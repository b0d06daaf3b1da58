"""Vision encoders of the plugin surface (reference models/encoder.py:25-195).

``Encoder.from_config`` keeps the reference's factory contract: the from-scratch ViT (``VisionTransformerEncoder``) and the
torchvision ViT-B/16 backbone with its three heads (``PretrainedViT``) both run on the HIP hot path; the modules here own
parameters only.
"""
import abc
import math
from typing import Union

import torch
import torch.nn as nn

import os
from collections import OrderedDict

from ..configs.models import PretrainedViTConfig, VisionTransformerEncoderConfig
from .layers import (AdvancedPositionalBiasMLP, CompositeCosineVectorEmbedding, ConvMLP, LayerNorm, LayerNormND, PeerLookup,
                     TransformerBlock, _Container)


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
            if config.lora_spec is not None:        # reference encoder.py:43-45 -> peft LoraModel over the torchvision module
                raise NotImplementedError('LoRA adapters on the PretrainedViT backbone (vision_encoder_config.lora_spec; commented out '
                                          'in every shipped yaml) are outside the HIP hot path')
            return PretrainedViT(config)
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


# torchvision.models.vit_b_16 (vision_transformer.py): the architecture constants of the checkpoint the reference loads
VIT_B16 = dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072)
SWAG_LINEAR_FILE = 'vit_b_16_lc_swag-4e70ced5.pth'      # ViT_B_16_Weights.IMAGENET1K_SWAG_LINEAR_V1 in torch hub's checkpoint dir


class _ViTBlock(_Container):
    """torchvision EncoderBlock: ln_1, self_attention (packed in_proj), ln_2, mlp = [Linear, GELU, Dropout, Linear, Dropout]
    (state-dict keys ``mlp.0`` / ``mlp.3``); LayerNorm eps 1e-6."""

    def __init__(self, d: int, heads: int, mlp_dim: int):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d, eps=1e-6)
        self.self_attention = nn.MultiheadAttention(d, heads, dropout=0.0, batch_first=True)
        self.ln_2 = nn.LayerNorm(d, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(d, mlp_dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(mlp_dim, d), nn.Dropout(0.0))


class _ViTEncoder(_Container):
    def __init__(self, seq: int, spec):
        super().__init__()
        d = spec['hidden_dim']
        self.pos_embedding = nn.Parameter(torch.empty(1, seq, d).normal_(std=0.02))
        self.layers = nn.Sequential(OrderedDict((f'encoder_layer_{i}', _ViTBlock(d, spec['num_heads'], spec['mlp_dim']))
                                                for i in range(spec['num_layers'])))
        self.ln = nn.LayerNorm(d, eps=1e-6)


class TorchvisionViT(_Container):
    """The parameter tree of torchvision's ``VisionTransformer`` with ``heads = Identity`` (reference encoder.py:60-61), key for key
    (conv_proj, class_token, encoder.pos_embedding, encoder.layers.encoder_layer_N.*, encoder.ln).  torchvision itself is never
    imported: a checkpoint in its format is read with ``torch.load``."""

    def __init__(self, spec):
        super().__init__()
        self.spec = dict(spec)
        d, p = spec['hidden_dim'], spec['patch_size']
        if spec['image_size'] % p or d % spec['num_heads'] or d // spec['num_heads'] != 64 or (3 * p * p) % 8:
            raise NotImplementedError(f'ViT backbone spec {spec}: the HIP attention kernels of this path take 64-wide heads')
        self.conv_proj = nn.Conv2d(3, d, kernel_size=p, stride=p)
        self.class_token = nn.Parameter(torch.zeros(1, 1, d))
        self.encoder = _ViTEncoder((spec['image_size'] // p) ** 2 + 1, spec)
        self.heads = nn.Identity()


def _load_backbone_weights(model: TorchvisionViT):
    """The reference downloads IMAGENET1K_SWAG_LINEAR_V1 inside its constructor (encoder.py:59-60).  Here: ``I2T_VIT_B16_CHECKPOINT`` =
    a torchvision-format state-dict file, or the literal ``random`` (keep the random initialisation: tests, benchmarks); unset = the
    file torchvision would have cached.  A missing file is an error -- a silently random backbone would train to nothing."""
    src = os.environ.get('I2T_VIT_B16_CHECKPOINT')
    if src == 'random':
        return
    path = src or os.path.join(torch.hub.get_dir(), 'checkpoints', SWAG_LINEAR_FILE)
    if not os.path.exists(path):
        raise FileNotFoundError(f'PretrainedViT: no ViT-B/16 checkpoint at {path}.  Put torchvision\'s {SWAG_LINEAR_FILE} there, or set '
                                'I2T_VIT_B16_CHECKPOINT to a torchvision-format state dict (or to "random" for a randomly initialised backbone)')
    sd = torch.load(path, map_location='cpu', weights_only=True)
    sd = {k: v for k, v in sd.items() if not k.startswith('heads.')}        # heads = Identity (encoder.py:61)
    model.load_state_dict(sd, strict=True)


class PretrainedViT(Encoder):
    """torchvision ViT-B/16 backbone (class-token feature, 768 wide) + one of three heads that turn it into ``n_cls`` encoder
    outputs (reference encoder.py:56-127): a private MLP per slot between two L2 normalisations (default), a PEER product-key
    lookup behind a learned (768, 768, n_cls) expansion (``peer_config``), or LSH cosine-bucket embeddings (``lsh_config``;
    forces the backbone frozen).  ``backbone_spec`` exists for tests (fewer layers); checkpoints need ``VIT_B16``."""
    backbone_spec = VIT_B16

    def __init__(self, config: PretrainedViTConfig):
        super().__init__(config)
        self.out_dim = config.n_embd_out_vit
        self.n_cls = config.n_cls
        self.use_peer = config.peer_config is not None
        self.use_lsh = not self.use_peer and config.lsh_config is not None
        d = self.backbone_spec['hidden_dim']
        self.proj = AdvancedPositionalBiasMLP(context_width=config.n_cls, in_features=d, out_features=config.n_embd_out_vit,
                                              gate_sizes=config.gate_sizes, add_residual_connection=True) \
            if not (self.use_lsh or self.use_peer) else nn.Identity()
        self.model = TorchvisionViT(self.backbone_spec)
        _load_backbone_weights(self.model)
        self.refine = config.refine_base_model if not self.use_lsh else False
        if self.use_peer:
            pc = config.peer_config
            self.peer = PeerLookup(d, config.n_embd_out_vit, pc.num_units_sqrt ** 2, pc.topk, pc.nhead, pc.query_dim)
            self.peer_proj_wt = nn.Parameter(torch.randn((d, d, self.n_cls)) / math.sqrt(d), requires_grad=True)
        else:
            self.peer = nn.Identity()
            self.peer_proj_wt = nn.Parameter(torch.zeros((1,)), requires_grad=False)
        if self.use_lsh:
            lc = config.lsh_config
            self.lsh_emb = nn.ModuleList([CompositeCosineVectorEmbedding(d, config.n_embd_out_vit, lc.num_bins, lc.num_proj, lc.learnable)
                                          for _ in range(self.n_cls)])
        else:
            self.lsh_emb = nn.ModuleList([nn.Identity()])
        self._check_shapes(config)

    def _check_shapes(self, config):
        widths = [config.n_embd_out_vit] + list(config.gate_sizes or [])
        if not (self.use_peer or self.use_lsh) and any(w % 32 for w in widths):
            raise NotImplementedError(f'PretrainedViT slot MLP widths {widths} must be multiples of 32 (grouped MFMA GEMM tiles)')
        if self.use_peer:
            pc = config.peer_config
            if pc.topk > 16 or pc.topk * pc.topk > 256 or pc.num_units_sqrt > 1024 or config.n_embd_out_vit % 8 or self.peer.query_dim % 8:
                raise NotImplementedError(f'peer_config {pc} is outside what csrc/vit.hip::peer_lookup covers (topk <= 16, '
                                          'num_units_sqrt <= 1024, widths multiples of 8)')
        if self.use_lsh and config.n_embd_out_vit % 4:
            raise NotImplementedError('n_embd_out_vit must be a multiple of 4 for the LSH embedding kernels')

    def forward(self, images: torch.Tensor):
        from .vision_encoder_decoder import run_encoder_standalone
        return run_encoder_standalone(self, images)

    @property
    def num_outputs(self):
        return self.n_cls

    @property
    def output_embed_dim(self):
        return self.out_dim
