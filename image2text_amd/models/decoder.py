"""Text decoders of the plugin surface (reference models/decoder.py:32-282).

``Decoder.from_config(config, loose, space_for_prompt)`` keeps the reference's factory contract; the from-scratch
nanoGPT decoder is what the HIP hot path runs -- freshly initialised (``pretrained_model: null``) or carrying imported GPT-2
weights (``pretrained_model: gpt2 | gpt2-medium | gpt2-large | gpt2-xl``: the same module, the weights are read from the local
Hugging Face cache through ``GPT2LMHeadModel.from_pretrained``; there is no network in the build image, so an absent cache
surfaces as transformers' own error).  ``HuggingfaceDecoderConfig`` with ``model_str: gpt2*`` (reference decoder.py:285-382,
``GPT2HuggingfaceDecoder``) is the same arithmetic again -- Hugging Face's GPT-2 block, cross-attention included, IS the nanoGPT
block -- so it runs on the HIP path too and keeps Hugging Face's parameter names and Conv1D layout in its state dict.
``model_str: meta-llama/Llama-2*`` / ``*Qwen*`` (decoder.py:404-440) run their own block kind (engine_llama.py: RMSNorm, rotary
embedding, grouped K/V heads, SwiGLU) and keep the transformers module itself as the parameter container.  Falcon, 4-bit loading
and LoRA are refused loudly.
"""
import abc
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from ..configs.models import (HuggingfaceDecoderConfig, MLPConfig, ModelType, SelfAttentionConfig, SelfAttentionType,
                              TransformerConfig, TransformerDecoderConfig)
from .layers import AdvancedPositionalBiasMLP, LayerNorm, TransformerBlock, init_gpt_weights_
from .utils import mutate_transformer_config


def _freeze_like_prepare_for_kbit_training(decoder: nn.Module):
    """``prepare_for_kbit_training: True`` on a model that is NOT loaded in 4 bits (reference decoder.py:325-327 -> peft's
    prepare_model_for_kbit_training; e.g. training_configs/local/llama2-7b.yaml): what remains of that call is "freeze the base model's
    layers" -- every parameter's requires_grad goes off (the fp32 upcast of half-precision parameters is a no-op here: the arena's
    master copy is fp32; gradient checkpointing has no counterpart on a hand-written backward).  The hot path then skips every
    weight-gradient GEMM of the decoder and only carries the input gradient back to the soft prompt, i.e. to the encoder."""
    for p in decoder.parameters():
        p.requires_grad = False


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
            if config.model_str.startswith('gpt2'):                    # reference decoder.py:120-121 (+ get_lora_model, :133-134)
                return GPT2HuggingfaceDecoder(config, space_for_prompt)
            if config.model_str.startswith('meta-llama/Llama-2'):      # reference decoder.py:124-125
                return Llama2HuggingfaceDecoder(config)
            if 'Qwen' in config.model_str:                             # reference decoder.py:126-127
                return Qwen2HuggingfaceDecoder(config)
            if config.model_str.startswith('tiiuae/falcon'):           # reference decoder.py:122-123
                return FalconHuggingfaceDecoder(config)
            raise NotImplementedError(f'HuggingfaceDecoder {config.model_str!r}: GPT-2, Falcon, Llama-2 and Qwen2 checkpoints run on the HIP '
                                      'hot path; free-form AutoModelForCausalLM architectures do not (SURVEY.md 8(f) next #3)')
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


class GPT2HuggingfaceDecoder(TransformerDecoder):
    """``HuggingfaceDecoderConfig(model_str='gpt2*')`` (reference decoder.py:285-382): the checkpoint is loaded through
    ``AutoModelForCausalLM.from_pretrained`` exactly as the reference does (cross-attention layers requested through
    ``add_cross_attention`` and freshly initialised by transformers, embeddings resized to ``vocab_size + extra_tokens``), its weights
    move into the hot path's own decoder (Hugging Face's GPT-2 block -- ln_1/attn, ln_cross_attn/crossattention, ln_2/mlp, tanh
    GELU, eps 1e-5 -- is the nanoGPT block with other names) and the transformers module is dropped.

    The state dict keeps the reference's names and layout (``backbone.transformer.h.N.attn.c_attn.weight`` as Conv1D ``[in, out]``,
    ``crossattention.q_attn`` / ``c_attn`` / ``c_proj``, ``ln_cross_attn``, ``backbone.lm_head.weight``), so checkpoints travel in both
    directions.  As in the reference the attention mask is ignored (always causal, decoder.py:349-350), ``block_size`` is the
    checkpoint's 1024 positions and the decoder's dropout rate is the checkpoint's (``resid_pdrop`` = ``embd_pdrop`` = ``attn_pdrop``, as in every GPT-2 release) at
    transformers' sites: embeddings, attention probabilities, after every ``c_proj`` (cross-attention's included)."""

    _CONV1D = ('attn.c_attn.weight', 'attn.c_proj.weight', 'mlp.c_fc.weight', 'mlp.c_proj.weight')

    def __init__(self, config: HuggingfaceDecoderConfig, space_for_prompt: int = 0):
        assert config.model_str.startswith('gpt2')
        if config.load_in_4bit:
            raise NotImplementedError('4-bit loading (bitsandbytes) is outside the HIP hot path')
        from transformers import AutoConfig, AutoModelForCausalLM
        kwargs = {}
        if config.use_cross_attn:
            hf_config = AutoConfig.from_pretrained(config.model_str)
            if not hasattr(hf_config, 'add_cross_attention'):
                raise ValueError("Don't know how to use cross attention with this model. Suggest you try a different config!!!")
            hf_config.add_cross_attention = True
            kwargs['config'] = hf_config
        hf = AutoModelForCausalLM.from_pretrained(config.model_str, **kwargs)
        hc = hf.config
        hf.resize_token_embeddings(config.vocab_size + config.extra_tokens)
        problems = [msg for bad, msg in (
            (hc.model_type != 'gpt2', f'model_type {hc.model_type!r}'),
            (hc.activation_function not in ('gelu_new', 'gelu_pytorch_tanh'), f'activation {hc.activation_function!r}'),
            (abs(hc.layer_norm_epsilon - 1e-5) > 1e-12, f'layer_norm_epsilon {hc.layer_norm_epsilon}'),
            (not hc.scale_attn_weights or hc.scale_attn_by_inverse_layer_idx, 'attention scaling other than 1/sqrt(head_dim)'),
            (abs(hc.embd_pdrop - hc.resid_pdrop) > 1e-12 or abs(hc.attn_pdrop - hc.resid_pdrop) > 1e-12,
             'embd_pdrop / attn_pdrop != resid_pdrop (one rate drives the embedding, residual and attention-probability sites)'),
        ) if bad]
        if problems:
            raise NotImplementedError('GPT-2 checkpoint outside the HIP hot path: ' + '; '.join(problems))
        n_inner = hc.n_inner if hc.n_inner is not None else 4 * hc.n_embd
        hot = TransformerDecoderConfig(
            vocab_size=config.vocab_size + config.extra_tokens, n_layer=hc.n_layer, block_size=hc.n_positions,
            enable_gradient_checkpointing=config.enable_gradient_checkpointing,
            transformer_config=TransformerConfig(
                rotator_config=MLPConfig(ff_mult=n_inner / hc.n_embd), is_causal=True, is_cross_attn=bool(config.use_cross_attn),
                # dropout sites of transformers' GPT-2: embeddings, attention probabilities (self and cross), after every c_proj;
                # attn_dropout = 0: the per-token q / k / v multipliers of the reference's own attention (layers.py:454-461) do not exist here
                attn_config=SelfAttentionConfig(n_head=hc.n_head, n_embd=hc.n_embd, bias=True, dropout=hc.resid_pdrop,
                                                attn_dropout=0.0, attn_type=SelfAttentionType.MULTI_HEAD)),
            skip_alternate_cross_attn=False)
        super().__init__(hot, space_for_prompt)
        self.hot_config = hot               # what the HIP engine reads in place of VisionEncoderDecoderConfig.decoder_config
        self.config = config
        self.hf_config = hc
        self.use_cross_attn = config.use_cross_attn
        self._register_state_dict_hook(self._to_hf_keys)
        self._register_load_state_dict_pre_hook(self._from_hf_keys)
        self.lora = None
        self.load_state_dict({'backbone.' + k: v for k, v in hf.state_dict().items()
                              if not (k.endswith('.attn.masked_bias') or k.endswith('.attn.bias')
                                      or k.endswith('.crossattention.masked_bias') or k.endswith('.crossattention.bias'))})
        self.tie_weights()
        if config.prepare_for_kbit_training:
            _freeze_like_prepare_for_kbit_training(self)
        if config.lora_spec is not None:
            self._apply_lora(config.lora_spec)
        from .utils import register_decoder_name_aliases
        register_decoder_name_aliases(self.reference_parameter_names())        # fnmatch patterns over the reference's parameter names

    # -- LoRA (reference models/utils.py:46-65 -> peft LoraModel over the transformers module) -----------------------------------
    _LORA_SITES = {'attn.c_attn': 'attn_c_attn', 'crossattention.c_attn': 'xattn_c_attn', 'mlp.c_fc': 'mlp_c_fc', 'mlp.c_proj': 'mlp_c_proj'}
    _HF_LINEARS = ('attn.c_attn', 'attn.c_proj', 'crossattention.c_attn', 'crossattention.q_attn', 'crossattention.c_proj', 'mlp.c_fc',
                   'mlp.c_proj')

    def _apply_lora(self, spec):
        """What ``get_lora_model(backbone, CAUSAL_LM, lora_spec)`` does to the transformers module, on this module's own parameters:
        every Conv1D whose name ends in one of ``target_modules`` (peft's suffix rule; default ``c_attn``) gets
        ``y += lora_B(lora_A(dropout(x))) * lora_alpha / r`` with lora_A [r, in] ~ kaiming_uniform(a = sqrt 5) and lora_B [out, r] = 0;
        every other parameter of the decoder is frozen; ``force_enable_update_modules`` (fnmatch patterns over the LoraModel's
        parameter names, ``model.transformer...``) switch named ones back on."""
        import fnmatch
        import math
        targets = list(spec.target_modules) if spec.target_modules else ['c_attn']
        L, d = self.hot_config.n_layer, self.n_embd
        ff = int(self.hot_config.transformer_config.rotator_config.ff_mult * d)
        have = [m for m in self._HF_LINEARS if self.use_cross_attn or not m.startswith('crossattention.')]
        hit = [m for m in have if any(f'transformer.h.0.{m}' == t or f'transformer.h.0.{m}'.endswith('.' + t) for t in targets)]
        other = [t for t in targets if not any(f'transformer.h.0.{m}'.endswith('.' + t) or f'transformer.h.0.{m}' == t for m in self._HF_LINEARS)]
        bad = [m for m in hit if m not in self._LORA_SITES] + other
        if bad or not hit:
            raise NotImplementedError(f'LoRA target_modules {targets}: the HIP hot path adapts {sorted(self._LORA_SITES)} '
                                      f'(unsupported here: {bad or "no module matched"})')
        if not 0 < spec.r <= 128:
            raise NotImplementedError('LoRA rank must be in 1..128 (the adapters run as GEMMs with the rank padded to 128)')
        shapes = {'attn.c_attn': (d, 3 * d), 'crossattention.c_attn': (d, 2 * d), 'mlp.c_fc': (d, ff), 'mlp.c_proj': (ff, d)}
        self.lora_params = nn.ParameterDict()
        for l in range(L):
            for m in hit:
                fin, fout = shapes[m]
                A = torch.empty(spec.r, fin)
                nn.init.kaiming_uniform_(A, a=math.sqrt(5))
                self.lora_params[f'h{l}_{self._LORA_SITES[m]}_A'] = nn.Parameter(A)
                self.lora_params[f'h{l}_{self._LORA_SITES[m]}_B'] = nn.Parameter(torch.zeros(fout, spec.r))
        self.lora = SimpleNamespace(r=spec.r, scale=spec.lora_alpha / spec.r, p=float(spec.lora_dropout),
                                    sites=tuple(self._LORA_SITES[m] for m in hit), hf_sites=tuple(hit))
        # trainable set: LoRA matrices + whatever force_enable_update_modules names (in the LoraModel's own parameter names)
        pats = spec.force_enable_update_modules
        for name, p in self.named_parameters():
            if name.startswith('lora_params.'):
                p.requires_grad = True
                continue
            peft_names = self._peft_names(name)
            on = [pats is not None and (len(pats) == 0 or any(fnmatch.fnmatch(n, pat) for pat in pats)) for n in peft_names]
            if any(on) != all(on):
                raise NotImplementedError(f'force_enable_update_modules splits {name} (transformers parameters {peft_names} share one '
                                          'matrix on the HIP hot path)')
            p.requires_grad = all(on)
        if pats is not None and len(pats) == 0:          # PatternMatcher: an empty list matches everything (models/utils.py:22-23)
            for p in self.parameters():
                p.requires_grad = True

    def reference_parameter_names(self):
        """{name under ``named_parameters()`` here: [the reference's name(s) for the same numbers]}.  State-dict keys already ARE the
        reference's (hooks below); parameter NAMES cannot be (nn.Module walks its own attribute tree), so fnmatch patterns written
        against the reference's names -- optimizer ``target_modules``, the checkpoint matchers of training/utils.py -- are translated
        through this table (``training/utils.py::parameter_name_matches``)."""
        out = {}
        for name, _ in self.named_parameters():
            if name.startswith('lora_params.'):
                l, rest = name[len('lora_params.h'):].split('_', 1)
                site, ab = rest.rsplit('_', 1)
                hf_mod = {v: k for k, v in self._LORA_SITES.items()}[site]
                out[name] = [f'backbone.model.transformer.h.{l}.{hf_mod}.lora_{ab}.default.weight']
            else:
                out[name] = ['backbone.' + self._decorate(k) for k, _, _ in self._hf_entries(name)]
        return out

    def _peft_names(self, internal: str):
        """the LoraModel parameter name(s) of one parameter of this module: ``model.<transformers name>``, ``base_layer`` inside adapted
        modules (``lm_head.weight`` is tied to wte and never listed by named_parameters)"""
        return [self._decorate(k) for k, _, _ in self._hf_entries(internal)]

    # -- state dict in Hugging Face's names and layout -----------------------------------------------------------------------
    def _hf_entries(self, name: str):
        """internal parameter name -> [(transformers key, row slice | None, transposed)]: Conv1D stores [in, out]; the fused
        cross-attention in_proj is transformers' q_attn (rows 0..d) and c_attn (rows d..3d)"""
        d = self.n_embd
        if '.cross_attn.' in name:
            blk, leaf = name.split('.cross_attn.')
            base = blk + '.crossattention.'
            if leaf == 'in_proj_weight':
                return [(base + 'q_attn.weight', slice(0, d), True), (base + 'c_attn.weight', slice(d, 3 * d), True)]
            if leaf == 'in_proj_bias':
                return [(base + 'q_attn.bias', slice(0, d), False), (base + 'c_attn.bias', slice(d, 3 * d), False)]
            w = leaf.split('.')[1]
            return [(base + 'c_proj.' + w, None, w == 'weight')]
        return [(name.replace('.ln_3.', '.ln_cross_attn.'), None, name.endswith(self._CONV1D))]

    def _decorate(self, hf_key: str) -> str:
        """transformers key -> key under ``backbone.``: unchanged without LoRA; with LoRA the backbone is peft's LoraModel, i.e.
        ``model.<key>`` and ``<module>.base_layer.<weight|bias>`` for the adapted modules"""
        if self.lora is None:
            return hf_key
        for m in self.lora.hf_sites:
            if f'.{m}.' in hf_key:
                hf_key = hf_key.replace(f'.{m}.', f'.{m}.base_layer.')
        return 'model.' + hf_key

    @staticmethod
    def _to_hf_keys(module, sd, prefix, local_metadata):
        sites = {v: k for k, v in module._LORA_SITES.items()}
        for k in [k for k in sd if k.startswith(prefix)]:
            v, name = sd.pop(k), k[len(prefix):]
            if name.startswith('lora_params.'):                       # h{l}_{site}_{A|B} -> <module>.lora_{A|B}.default.weight
                l, rest = name[len('lora_params.h'):].split('_', 1)
                site, ab = rest.rsplit('_', 1)
                sd[f'{prefix}backbone.model.transformer.h.{l}.{sites[site]}.lora_{ab}.default.weight'] = v
                continue
            for hk, rows, tr in module._hf_entries(name):
                t = v if rows is None else v[rows]
                sd[prefix + 'backbone.' + module._decorate(hk)] = t.t() if tr else t
        return sd

    def _from_hf_keys(self, sd, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        bb = prefix + 'backbone.'
        cross = {}
        for k in [k for k in sd if k.startswith(bb)]:
            v, name = sd.pop(k), k[len(bb):]
            if name.startswith('model.'):                             # a LoraModel's keys (or a checkpoint saved from one)
                name = name[len('model.'):]
            name = name.replace('.base_layer.', '.')
            if '.lora_A.' in name or '.lora_B.' in name:
                mod, rest = name.split('.lora_')
                l = mod.split('.')[2]
                site = self._LORA_SITES.get(mod.split('.', 3)[3])
                sd[f'{prefix}lora_params.h{l}_{site}_{rest[0]}'] = v
            elif name.endswith(self._CONV1D):
                sd[prefix + name] = v.t()
            elif '.crossattention.' in name:
                cross.setdefault(name.split('.crossattention.')[0], {})[name.split('.crossattention.')[1]] = v
            else:
                sd[prefix + name.replace('.ln_cross_attn.', '.ln_3.')] = v
        for blk, parts in cross.items():
            base = prefix + blk + '.cross_attn.'
            if 'q_attn.weight' in parts and 'c_attn.weight' in parts:
                sd[base + 'in_proj_weight'] = torch.cat([parts['q_attn.weight'].t(), parts['c_attn.weight'].t()], 0)
            if 'q_attn.bias' in parts and 'c_attn.bias' in parts:
                sd[base + 'in_proj_bias'] = torch.cat([parts['q_attn.bias'], parts['c_attn.bias']], 0)
            if 'c_proj.weight' in parts:
                sd[base + 'out_proj.weight'] = parts['c_proj.weight'].t()
            if 'c_proj.bias' in parts:
                sd[base + 'out_proj.bias'] = parts['c_proj.bias']

    @property
    def block_size(self):
        return self.hf_config.n_positions               # 1024 for every GPT-2 checkpoint (reference decoder.py:377-378 hard-codes it)

    @property
    def n_embd(self):
        return self.hf_config.n_embd


class _LlamaFamilyHuggingfaceDecoder(Decoder):
    """Llama-2 / Qwen2 checkpoints behind the reference's ``HuggingfaceDecoder`` surface (decoder.py:285-361, 404-440).

    ``self.backbone`` IS the transformers module ``AutoModelForCausalLM.from_pretrained`` returned (embeddings resized by
    ``extra_tokens``): it owns the parameters -- state-dict keys, ``tie_weights`` and ``get_inputs_embeds`` are the reference's by
    construction -- but is never called; the arithmetic is engine_llama.LlamaBlocks over the flat arena its parameters become views
    of.  These decoders have no cross-attention (``use_cross_attn`` raises the reference's ValueError) and are always causal."""

    def __init__(self, config: HuggingfaceDecoderConfig):
        super().__init__()
        import os
        if config.load_in_4bit and os.environ.get('I2T_4BIT_AS_FP8') != '1':
            raise NotImplementedError('4-bit loading (bitsandbytes NF4) is outside the HIP hot path: bitsandbytes is not in this image.  '
                                      'I2T_4BIT_AS_FP8=1 runs such a config with the quantised base held as e4m3 fp8 operands instead '
                                      '(frozen block linears on the fp8 MFMA path, DESIGN 4h) -- a different 8-bit format, not NF4')
        if config.use_cross_attn:
            raise ValueError("Don't know how to use cross attention with this model. Suggest you try a different config!!!")
        from transformers import AutoModelForCausalLM
        self.config = config
        self.use_cross_attn = False
        hf = AutoModelForCausalLM.from_pretrained(config.model_str)
        self.hf_config = hf.config
        hf.resize_token_embeddings(config.vocab_size + config.extra_tokens)
        spec = self._inspect(hf, config)              # raises NotImplementedError for checkpoints outside the hot path
        if config.enable_gradient_checkpointing:
            pass            # the hot path keeps what its hand-written backward needs; nothing to switch on (decoder.py:322-323)
        self.backbone = hf
        # load_in_4bit under I2T_4BIT_AS_FP8=1: what bitsandbytes would quantise -- every nn.Linear of the blocks (not the head) -- is frozen
        # (4-bit parameters never train) and the engine runs those GEMMs on e4m3 operands
        self.fp8_request = bool(config.load_in_4bit)
        if config.load_in_4bit:
            blocks = hf.get_submodule(self._BLOCKS)
            for mod in blocks.modules():
                if isinstance(mod, nn.Linear):
                    for p in mod.parameters():
                        p.requires_grad = False
        if config.prepare_for_kbit_training:
            _freeze_like_prepare_for_kbit_training(self)
        self.lora = None
        self._register_state_dict_hook(self._to_peft_keys)
        self._register_load_state_dict_pre_hook(self._from_peft_keys)
        if config.lora_spec is not None:
            self._apply_lora(config.lora_spec)
            from .utils import register_decoder_name_aliases
            register_decoder_name_aliases(self.reference_parameter_names())
        self.llama_spec = spec

    # transformers attribute paths under ``backbone`` (the Falcon subclass has its own)
    _BLOCKS, _ROTARY = 'model.layers', 'model.rotary_emb'

    def _inspect(self, hf, config):
        hc = hf.config
        hd = getattr(hc, 'head_dim', None) or hc.hidden_size // hc.num_attention_heads
        attn0 = hf.model.layers[0].self_attn
        problems = [msg for bad, msg in (
            (hc.model_type not in ('llama', 'qwen2'), f'model_type {hc.model_type!r}'),
            (hc.hidden_act != 'silu', f'activation {hc.hidden_act!r}'),
            (hd not in (16, 32, 64, 128), f'head_dim {hd}'),
            (hc.num_attention_heads % hc.num_key_value_heads != 0, 'query heads not a multiple of key/value heads'),
            (attn0.o_proj.bias is not None, 'o_proj bias'),
            (hf.model.layers[0].mlp.gate_proj.bias is not None, 'mlp bias'),
            ((attn0.q_proj.bias is None) != (attn0.k_proj.bias is None) or (attn0.q_proj.bias is None) != (attn0.v_proj.bias is None),
             'q / k / v biases present on some projections only'),
            (float(getattr(hc, 'attention_dropout', 0.0) or 0.0) != 0.0, 'attention_dropout > 0'),
            (bool(getattr(hc, 'use_sliding_window', False)), 'sliding-window attention'),
            (getattr(hc, 'pretraining_tp', 1) not in (None, 1), 'pretraining_tp > 1'),
            (hc.hidden_size % 8 != 0 or hc.intermediate_size % 8 != 0, 'hidden / intermediate size not a multiple of 8'),
        ) if bad]
        if problems:
            raise NotImplementedError(f'{hc.model_type} checkpoint outside the HIP hot path: ' + '; '.join(problems))
        return SimpleNamespace(
            arch='llama', d=hc.hidden_size, H=hc.num_attention_heads, Hkv=hc.num_key_value_heads, hd=hd, L=hc.num_hidden_layers,
            ff=hc.intermediate_size, V=config.vocab_size + config.extra_tokens, eps=float(hc.rms_norm_eps), block=self.block_size,
            qkv_bias=attn0.q_proj.bias is not None, tied=hf.lm_head.weight is hf.model.embed_tokens.weight,
            wte='backbone.model.embed_tokens.weight', norm_f='backbone.model.norm')

    # -- LoRA (reference models/utils.py:46-65 -> peft LoraModel over the transformers module; decoder.py:404-440 reads the embedding
    #    through it: backbone.model.model.embed_tokens) ---------------------------------------------------------------------------
    # hot-path site (ONE GEMM) -> the transformers linears it fuses, in row order
    _LORA_SITES = {'qkv': ('self_attn.q_proj', 'self_attn.k_proj', 'self_attn.v_proj'), 'o': ('self_attn.o_proj',),
                   'gu': ('mlp.gate_proj', 'mlp.up_proj'), 'dn': ('mlp.down_proj',)}
    _LORA_TAGS = {'self_attn.q_proj': 'q', 'self_attn.k_proj': 'k', 'self_attn.v_proj': 'v', 'self_attn.o_proj': 'o',
                  'mlp.gate_proj': 'gate', 'mlp.up_proj': 'up', 'mlp.down_proj': 'down'}

    def _apply_lora(self, spec):
        """``get_lora_model(backbone, CAUSAL_LM, lora_spec)`` on this module's own parameters: every linear of the blocks whose name ends
        in one of ``target_modules`` (peft's suffix rule; peft's default for llama / qwen2: q_proj, v_proj) gets
        ``y += lora_B(lora_A(dropout(x))) * lora_alpha / r``, lora_A [r, in] ~ kaiming_uniform(a = sqrt 5), lora_B [out, r] = 0; every
        other parameter is frozen, ``force_enable_update_modules`` (fnmatch over the LoraModel's names, ``model.model.layers...``)
        switches named ones back on.  The adapters of linears the hot path fuses into one GEMM (q | k | v, gate | up) keep ONE stacked
        lora_A parameter ``h{l}_{site}_A`` [n_adapted * r, in] and one lora_B per linear; the state dict speaks peft's per-module keys.
        (One input-dropout mask per fused site: q / k / v adapters see the same masked x where peft draws three masks of the same
        distribution -- each adapter's own statistics are peft's.)"""
        import fnmatch
        import math
        hc = self.hf_config
        targets = list(spec.target_modules) if spec.target_modules else list(self._LORA_DEFAULT_TARGETS)
        linears = [m for ms in self._LORA_SITES.values() for m in ms]
        match = (lambda key, t: key == t or key.endswith('.' + t))
        hit = [m for m in linears if any(match(f'{self._BLOCKS}.0.{m}', t) for t in targets)]
        other = [t for t in targets if not any(match(f'{self._BLOCKS}.0.{m}', t) for m in linears)]
        if other or not hit:
            raise NotImplementedError(f'LoRA target_modules {targets}: the HIP hot path adapts the block linears {linears} '
                                      f'(unsupported here: {other or "no module matched"})')
        shapes = self._lora_shapes()
        sites = {site: [m for m in ms if m in hit] for site, ms in self._LORA_SITES.items()}
        sites = {site: ms for site, ms in sites.items() if ms}
        if not 0 < spec.r or max(len(ms) for ms in sites.values()) * spec.r > 128:
            raise NotImplementedError('LoRA rank: the adapters of one fused projection share a 128-column panel (3 r <= 128 with q, k and v adapted)')
        self.lora_params = nn.ParameterDict()
        for l in range(hc.num_hidden_layers):
            for site, ms in sites.items():
                As = []
                for m in ms:
                    A = torch.empty(spec.r, shapes[m][0])
                    nn.init.kaiming_uniform_(A, a=math.sqrt(5))
                    As.append(A)
                self.lora_params[f'h{l}_{site}_A'] = nn.Parameter(torch.cat(As, 0))
                for m in ms:
                    self.lora_params[f'h{l}_{self._LORA_TAGS[m]}_B'] = nn.Parameter(torch.zeros(shapes[m][1], spec.r))
        self.lora = SimpleNamespace(r=spec.r, scale=spec.lora_alpha / spec.r, p=float(spec.lora_dropout), sites=tuple(sites),
                                    hf_sites=tuple(hit), members=sites)
        pats = spec.force_enable_update_modules
        every = pats is not None and len(pats) == 0      # PatternMatcher: an empty list matches everything (models/utils.py:22-23)
        for name, p in self.named_parameters():
            if name.startswith('lora_params.'):
                p.requires_grad = True
            else:
                p.requires_grad = every or (pats is not None and any(fnmatch.fnmatch(self._peft_name(name), pat) for pat in pats))

    _LORA_DEFAULT_TARGETS = ('q_proj', 'v_proj')           # peft's TRANSFORMERS_MODELS_TO_LORA_TARGET_MODULES_MAPPING for llama / qwen2

    def _lora_shapes(self):
        """{linear: (in features, out features)}"""
        hc = self.hf_config
        hd = getattr(hc, 'head_dim', None) or hc.hidden_size // hc.num_attention_heads
        d, ff = hc.hidden_size, hc.intermediate_size
        return {'self_attn.q_proj': (d, hc.num_attention_heads * hd), 'self_attn.k_proj': (d, hc.num_key_value_heads * hd),
                'self_attn.v_proj': (d, hc.num_key_value_heads * hd), 'self_attn.o_proj': (hc.num_attention_heads * hd, d),
                'mlp.gate_proj': (d, ff), 'mlp.up_proj': (d, ff), 'mlp.down_proj': (ff, d)}

    def _peft_name(self, internal: str) -> str:
        """own parameter name ``backbone.<transformers name>`` -> the LoraModel's: ``model.<name>``, ``base_layer`` inside adapted linears"""
        key = internal[len('backbone.'):]
        for m in (self.lora.hf_sites if self.lora is not None else ()):
            if f'.{m}.' in key:
                key = key.replace(f'.{m}.', f'.{m}.base_layer.')
        return 'model.' + key

    def _lora_rows(self, name: str):
        """``lora_params.h{l}_{x}_{A|B}`` -> [(peft key under ``backbone.``, row slice | None)]"""
        l, rest = name[len('lora_params.h'):].split('_', 1)
        x, ab = rest.rsplit('_', 1)
        base = f'model.{self._BLOCKS}.{l}.'
        if ab == 'B':
            m = {v: k for k, v in self._LORA_TAGS.items()}[x]
            return [(f'{base}{m}.lora_B.default.weight', None)]
        r = self.lora.r
        return [(f'{base}{m}.lora_A.default.weight', slice(i * r, (i + 1) * r)) for i, m in enumerate(self.lora.members[x])]

    def reference_parameter_names(self):
        """{own parameter name: [the reference's name(s) for the same numbers]} (see GPT2HuggingfaceDecoder.reference_parameter_names)"""
        out = {}
        for name, _ in self.named_parameters():
            if name.startswith('lora_params.'):
                out[name] = ['backbone.' + k for k, _ in self._lora_rows(name)]
            else:
                out[name] = ['backbone.' + self._peft_name(name)]
        return out

    @staticmethod
    def _to_peft_keys(module, sd, prefix, local_metadata):
        if module.lora is None:
            return sd
        for k in [k for k in sd if k.startswith(prefix)]:
            v, name = sd.pop(k), k[len(prefix):]
            if name.startswith('lora_params.'):
                for key, rows in module._lora_rows(name):
                    sd[f'{prefix}backbone.{key}'] = v if rows is None else v[rows]
            else:
                sd[f'{prefix}backbone.{module._peft_name(name)}'] = v
        return sd

    def _from_peft_keys(self, sd, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self.lora is None:
            return
        bb = prefix + 'backbone.'
        stacks = {}
        top = self._BLOCKS.split('.')[0]                 # 'model' (Llama / Qwen2) | 'transformer' (Falcon)
        depth = self._BLOCKS.count('.') + 1
        for k in [k for k in sd if k.startswith((f'{bb}model.{top}.', bb + 'model.lm_head.'))]:      # a LoraModel's keys; plain transformers keys pass
            v, name = sd.pop(k), k[len(bb) + len('model.'):]
            if '.lora_A.' in name or '.lora_B.' in name:
                mod, rest = name.split('.lora_')
                l, m = mod.split('.')[depth], mod.split('.', depth + 1)[depth + 1]
                if rest[0] == 'B':
                    sd[f'{prefix}lora_params.h{l}_{self._LORA_TAGS[m]}_B'] = v
                else:
                    site = next(s for s, ms in self.lora.members.items() if m in ms)
                    stacks.setdefault((l, site), {})[m] = v
            else:
                sd[bb + name.replace('.base_layer.', '.')] = v
        for (l, site), parts in stacks.items():
            if all(m in parts for m in self.lora.members[site]):
                sd[f'{prefix}lora_params.h{l}_{site}_A'] = torch.cat([parts[m] for m in self.lora.members[site]], 0)

    def rope_table(self, n_positions: int) -> torch.Tensor:
        """fp32 [n_positions, head_dim] = [cos(p f_i) | sin(p f_i)], i < head_dim / 2: the values the checkpoint's own rotary module
        produces (whatever its rope type and scaling), in the layout of i2t_rope"""
        rot = self.backbone.get_submodule(self._ROTARY)
        dev = rot.inv_freq.device
        with torch.no_grad():
            cos, sin = rot(torch.zeros(1, 1, dtype=torch.float32, device=dev), torch.arange(n_positions, device=dev)[None])
        half = cos.shape[-1] // 2
        return torch.cat((cos[0, :, :half], sin[0, :, :half]), dim=-1).float()

    def tie_weights(self):
        self.backbone.tie_weights()

    def forward(self, idx=None, inputs_embeds=None, cross_attn_embeds=None, attn_msk=None):
        from .vision_encoder_decoder import run_decoder_standalone
        return run_decoder_standalone(self, idx, inputs_embeds, cross_attn_embeds, attn_msk)

    def get_inputs_embeds(self, idx: torch.LongTensor):
        return self.backbone.model.embed_tokens(idx)

    @property
    def n_embd(self):
        return self.hf_config.hidden_size


class FalconHuggingfaceDecoder(_LlamaFamilyHuggingfaceDecoder):
    """``tiiuae/falcon*`` checkpoints (reference decoder.py:383-400) of the falcon-7b architecture: parallel attention + MLP behind ONE
    LayerNorm, multi-query attention (71 query heads on one key/value head), rotary embedding, exact GELU, no biases.  Same container
    as the Llama-2 / Qwen2 plugins (the transformers module owns the parameters, engine_llama runs the arithmetic).

    What the reference does with it: the id-driven ``forward`` (decoder.py:332-361) works -- that is what the parity test matches
    against transformers' FalconForCausalLM; its ``get_inputs_embeds`` reads ``backbone.model.embed_tokens`` (decoder.py:389-392), an
    attribute FalconForCausalLM does not have (``transformer.word_embeddings``), so a soft prompt -- the mode gpu/falcon-7b.yaml asks
    for -- raises AttributeError there.  Here it returns the word embeddings, which is what that code path can only have meant."""
    _BLOCKS, _ROTARY = 'transformer.h', 'transformer.rotary_emb'
    _LORA_SITES = {'qkv': ('self_attention.query_key_value',), 'o': ('self_attention.dense',), 'gu': ('mlp.dense_h_to_4h',),
                   'dn': ('mlp.dense_4h_to_h',)}
    _LORA_TAGS = {'self_attention.query_key_value': 'qkv', 'self_attention.dense': 'o', 'mlp.dense_h_to_4h': 'fc', 'mlp.dense_4h_to_h': 'proj'}
    _LORA_DEFAULT_TARGETS = ('query_key_value',)           # peft's default for falcon

    def __init__(self, config: HuggingfaceDecoderConfig):
        assert config.model_str.startswith('tiiuae/falcon')
        assert config.vocab_size >= 65024
        super().__init__(config)

    def _inspect(self, hf, config):
        hc = hf.config
        hd = hc.hidden_size // hc.num_attention_heads
        problems = [msg for bad, msg in (
            (hc.model_type != 'falcon', f'model_type {hc.model_type!r}'),
            (hc.new_decoder_architecture, 'new_decoder_architecture (falcon-40b / 180b blocks)'),
            (not hc.parallel_attn, 'sequential attention + MLP (falcon-rw blocks)'),
            (not hc.multi_query, 'multi_query off'),
            (hc.alibi, 'alibi positions'),
            (hc.bias, 'linear biases'),
            (getattr(hc, 'activation', 'gelu') != 'gelu', f'activation {getattr(hc, "activation", None)!r}'),
            (float(hc.hidden_dropout) != 0.0 or float(hc.attention_dropout) != 0.0, 'hidden / attention dropout > 0'),
            (hd not in (16, 32, 64, 128), f'head_dim {hd}'),
            (hc.hidden_size % 8 != 0, 'hidden size not a multiple of 8'),
        ) if bad]
        if problems:
            raise NotImplementedError('Falcon checkpoint outside the HIP hot path: ' + '; '.join(problems))
        return SimpleNamespace(
            arch='falcon', d=hc.hidden_size, H=hc.num_attention_heads, Hkv=1, hd=hd, L=hc.num_hidden_layers,
            ff=getattr(hc, 'ffn_hidden_size', None) or 4 * hc.hidden_size, V=config.vocab_size + config.extra_tokens,
            eps=float(hc.layer_norm_epsilon), block=self.block_size, qkv_bias=False,
            tied=hf.lm_head.weight is hf.transformer.word_embeddings.weight,
            wte='backbone.transformer.word_embeddings.weight', norm_f='backbone.transformer.ln_f')

    def _lora_shapes(self):
        hc = self.hf_config
        d, hd = hc.hidden_size, hc.hidden_size // hc.num_attention_heads
        ff = getattr(hc, 'ffn_hidden_size', None) or 4 * d
        return {'self_attention.query_key_value': (d, d + 2 * hd), 'self_attention.dense': (d, d), 'mlp.dense_h_to_4h': (d, ff),
                'mlp.dense_4h_to_h': (ff, d)}

    def get_inputs_embeds(self, idx: torch.LongTensor):
        return self.backbone.transformer.word_embeddings(idx)

    @property
    def block_size(self):
        return 2048                                        # reference decoder.py:394-396


class Llama2HuggingfaceDecoder(_LlamaFamilyHuggingfaceDecoder):
    def __init__(self, config: HuggingfaceDecoderConfig):
        assert config.model_str.startswith('meta-llama/Llama-2')
        assert config.vocab_size >= 32000
        super().__init__(config)

    @property
    def block_size(self):
        return 4096                                        # reference decoder.py:416-417


class Qwen2HuggingfaceDecoder(_LlamaFamilyHuggingfaceDecoder):
    def __init__(self, config: HuggingfaceDecoderConfig):
        assert 'Qwen' in config.model_str
        assert config.vocab_size >= 151936
        super().__init__(config)

    @property
    def block_size(self):
        return self.hf_config.max_position_embeddings      # reference decoder.py:434-435
