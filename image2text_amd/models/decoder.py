"""Text decoders of the plugin surface (reference models/decoder.py:32-282).

``Decoder.from_config(config, loose, space_for_prompt)`` keeps the reference's factory contract; the from-scratch
nanoGPT decoder is what the HIP hot path runs -- freshly initialised (``pretrained_model: null``) or carrying imported GPT-2
weights (``pretrained_model: gpt2 | gpt2-medium | gpt2-large | gpt2-xl``: the same module, the weights are read from the local
Hugging Face cache through ``GPT2LMHeadModel.from_pretrained``; there is no network in the build image, so an absent cache
surfaces as transformers' own error).  The Hugging Face decoder family (other architectures, 4-bit, LoRA) is refused loudly.
"""
import abc
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ..configs.models import HuggingfaceDecoderConfig, MLPConfig, ModelType, TransformerDecoderConfig
from .layers import AdvancedPositionalBiasMLP, LayerNorm, TransformerBlock, init_gpt_weights_
from .utils import mutate_transformer_config


class Decoder(nn.Module, abc.ABC):
    def __init__(self):
        super().__init__()

    @classmethod
    def from_config(cls, config: Union[TransformerDecoderConfig, HuggingfaceDecoderConfig], loose=False, space_for_prompt=0):
        if isinstance(config, TransformerDecoderConfig):
            if config.pretrained_model is not None:
                return cls._from_pretrained_gpt2(config, loose, space_for_prompt)
            return TransformerDecoder(config, space_for_prompt)
        if isinstance(config, HuggingfaceDecoderConfig):
            raise NotImplementedError('HuggingfaceDecoder family (AutoModelForCausalLM.from_pretrained, 4-bit, LoRA) is '
                                      'outside the HIP hot path (SURVEY.md 8(f) next #3)')
        raise ValueError('Unknown config type!!!')

    @staticmethod
    def _from_pretrained_gpt2(config: TransformerDecoderConfig, loose: bool, space_for_prompt: int):
        """GPT-2 weights into the nanoGPT decoder (reference decoder.py:45-117): the shapes must be GPT-2's unless ``loose``; the
        OpenAI checkpoints store Conv1D weights, i.e. the four projection matrices are transposed on the way in; keys the
        checkpoint does not have (cross-attention, ln_3) keep their fresh initialisation."""
        if config.lora_spec is not None:
            raise NotImplementedError('LoRA adapters (peft) are outside the HIP hot path (SURVEY.md 8(f) next #3)')
        model_type = config.pretrained_model
        want = {ModelType.GPT2: dict(n_layer=12, n_head=12, n_embd=768), ModelType.GPT2_MEDIUM: dict(n_layer=24, n_head=16, n_embd=1024),
                ModelType.GPT2_LARGE: dict(n_layer=36, n_head=20, n_embd=1280), ModelType.GPT2_XL: dict(n_layer=48, n_head=25, n_embd=1600)}[model_type]
        tc = config.transformer_config
        if not loose:
            msg = 'provided configs do not match the pretrained model'
            assert config.n_layer == want['n_layer'], msg
            assert tc.attn_config.n_embd == want['n_embd'] and tc.attn_config.n_head == want['n_head'] and tc.attn_config.bias is True, msg
            assert config.block_size == 1024 and not tc.is_sparse_attn and tc.is_causal is True, msg
            assert isinstance(tc.rotator_config, MLPConfig) and tc.rotator_config.ff_mult == 4, msg
        assert config.vocab_size >= 50257 or loose, 'vocab should not shrink'
        model = TransformerDecoder(config, space_for_prompt)
        from transformers import GPT2LMHeadModel
        model_hf = GPT2LMHeadModel.from_pretrained(model_type.value)
        if config.vocab_size > model_hf.config.vocab_size:
            model_hf.resize_token_embeddings(config.vocab_size)
        sd, sd_hf = model.state_dict(), model_hf.state_dict()
        transposed = ('attn.c_attn.weight', 'attn.c_proj.weight', 'mlp.c_fc.weight', 'mlp.c_proj.weight')
        with torch.no_grad():
            for k, v in sd_hf.items():
                if k.endswith('.attn.masked_bias') or k.endswith('.attn.bias'):          # buffers of the HF attention, not weights
                    continue
                src = v.t() if k.endswith(transposed) else v
                if k not in sd:
                    if not loose:
                        raise ValueError(f'{k} is not present in state dict!!!')
                elif sd[k].shape != src.shape:
                    if not loose:
                        raise ValueError(f'{k} is not the same shape in state dict!!!')
                else:
                    sd[k].copy_(src)
        model.tie_weights()
        return model

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
