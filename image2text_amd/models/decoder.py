"""Text decoders of the plugin surface (reference models/decoder.py:32-282).

``Decoder.from_config(config, loose, space_for_prompt)`` keeps the reference's factory contract; the from-scratch
nanoGPT decoder (``pretrained_model: null``) is what the HIP hot path runs.  GPT-2 weight import and the Hugging Face
decoder family need network fetches and are refused loudly.
"""
import abc
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ..configs.models import HuggingfaceDecoderConfig, TransformerDecoderConfig
from .layers import AdvancedPositionalBiasMLP, LayerNorm, TransformerBlock, init_gpt_weights_
from .utils import mutate_transformer_config


class Decoder(nn.Module, abc.ABC):
    def __init__(self):
        super().__init__()

    @classmethod
    def from_config(cls, config: Union[TransformerDecoderConfig, HuggingfaceDecoderConfig], loose=False, space_for_prompt=0):
        if isinstance(config, TransformerDecoderConfig):
            if config.pretrained_model is not None:
                raise NotImplementedError(f'pretrained_model={config.pretrained_model.value}: importing GPT-2 weights needs '
                                          'GPT2LMHeadModel.from_pretrained (network); out of the HIP hot-path scope')
            return TransformerDecoder(config, space_for_prompt)
        if isinstance(config, HuggingfaceDecoderConfig):
            raise NotImplementedError('HuggingfaceDecoder family (AutoModelForCausalLM.from_pretrained, 4-bit, LoRA) is '
                                      'outside the HIP hot path (SURVEY.md 8(f) next #3)')
        raise ValueError('Unknown config type!!!')

    def forward(self, idx: Optional[torch.LongTensor] = None, inputs_embeds: Optional[torch.FloatTensor] = None,
                cross_attn_embeds: Optional[torch.FloatTensor] = None, attn_msk: Optional[torch.Tensor] = None) -> \
            Tuple[torch.FloatTensor, torch.FloatTensor]:
        raise ValueError('not implemented in the base class')

    def tie_weights(self):
        pass

    def get_inputs_embeds(self, idx: torch.LongTensor):
        raise ValueError('not implemented in the base class')

    @property
    def block_size(self):
        raise ValueError('not implemented in the base class')

    @property
    def n_embd(self):
        raise ValueError('not implemented in the base class')


class TransformerDecoder(Decoder):
    """wte + wpe -> causal blocks (cross-attention on even depths when skip_alternate_cross_attn) -> ln_f -> lm_head
    tied to wte (reference decoder.py:161-256).  Holds the parameters; arithmetic is the HIP path."""

    def __init__(self, config: TransformerDecoderConfig, space_for_prompt: int):
        super().__init__()
        if config.use_advanced_pos_emb:
            gs = tuple(config.advanced_pos_emb_gate_sizes or ())
            d_ = config.transformer_config.attn_config.n_embd
            if any(g % 32 for g in gs) or d_ % 32:
                raise NotImplementedError('advanced_pos_emb_gate_sizes / n_embd must be multiples of 32 (MFMA k-step of the grouped GEMM)')
        self.config = config
        self.use_advanced_pos_emb = config.use_advanced_pos_emb
        self.enable_gradient_checkpointing = config.enable_gradient_checkpointing
        self.skip_alternate_cross_attn = config.skip_alternate_cross_attn
        d = config.transformer_config.attn_config.n_embd
        self.transformer = nn.ModuleDict(dict(
            wte=nn.Embedding(config.vocab_size, d),
            wpe=AdvancedPositionalBiasMLP(config.block_size, d, d, config.advanced_pos_emb_gate_sizes, add_residual_connection=True)
            if config.use_advanced_pos_emb else nn.Embedding(config.block_size, d),
            drop=nn.Dropout(config.transformer_config.attn_config.dropout),
            h=nn.ModuleList([
                TransformerBlock(mutate_transformer_config(config.transformer_config, depth, config.skip_alternate_cross_attn),
                                 depth, space_for_prompt)
                for depth in range(config.n_layer)]),
            ln_f=LayerNorm(d, bias=config.transformer_config.attn_config.bias),
        ))
        self.lm_head = nn.Linear(d, config.vocab_size, bias=False)
        self.tie_weights()
        init_gpt_weights_(self, config.n_layer)

    def tie_weights(self):
        self.transformer.wte.weight = self.lm_head.weight

    def forward(self, idx=None, inputs_embeds=None, cross_attn_embeds=None, attn_msk=None):
        """Standalone decoder call (reference decoder.py:214-256).  ``attn_msk`` must be None or the soft-prompt /
        zero masks VisionEncoderDecoder builds: arbitrary additive masks are not supported by the HIP kernels."""
        from .vision_encoder_decoder import run_decoder_standalone
        return run_decoder_standalone(self, idx, inputs_embeds, cross_attn_embeds, attn_msk)

    def get_inputs_embeds(self, idx: torch.LongTensor):
        return self.transformer.wte(idx)

    @property
    def block_size(self):
        return self.config.block_size

    @property
    def n_embd(self):
        return self.config.transformer_config.attn_config.n_embd
