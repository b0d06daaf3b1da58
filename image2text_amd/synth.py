"""Named hot-path configurations, synthetic batches and a machine-independent weight initialiser.

Nothing here computes the model; it only describes workloads:

* ``nano224_config``  -- the benchmark configuration of SURVEY.md 8(d): reference
  ``training_configs/local/nano.yaml:76-97`` decoder with ``pretrained_model`` removed + the commented
  from-scratch encoder block ``nano.yaml:46-75`` made dense (224x224, 14x14 patches, multi_head, MLP ff 4).
* ``tiny_config``     -- same topology at toy sizes (2+2 layers, 64/128 wide) for the committed golden fixtures.
* ``synthetic_batch`` -- Flickr30K-shaped inputs: normalised-image-like ``randn`` and captions with EOS at a random
  length and ``ignore_index`` after it, i.e. what reference ``training/utils.py:16-20`` (normalize_label) emits.
* ``det_init_``       -- fills every parameter from a per-name seeded CPU generator so that the container that
  generated the goldens and the GPU box hold bit-identical weights without shipping them.
"""
import zlib
from types import SimpleNamespace
from typing import Optional

import torch

from .configs.models import (
    ImageInputSpec,
    MLPConfig,
    MoEConfig,
    SelfAttentionConfig,
    SelfAttentionType,
    TransformerConfig,
    TransformerDecoderConfig,
    VisionEncoderDecoderConfig,
    VisionTransformerEncoderConfig,
)

GPT2_VOCAB = 50257
GPT2_EOS = 50256


def _model_config(*, img: int, num_patches: int, conv_gates, conv_out: int, kernel: int,
                  enc_layers: int, enc_d: int, enc_heads: int, n_cls: int, enc_bias: bool,
                  dec_layers: int, dec_d: int, dec_heads: int, block_size: int, vocab: int,
                  dropout: float, use_cross_attn: bool = True, use_soft_prompting: bool = True,
                  no_repeat_n_grams=(2, 3, 4, 5)) -> VisionEncoderDecoderConfig:
    enc_tf = TransformerConfig(
        rotator_config=MLPConfig(ff_mult=4),
        is_causal=False,
        is_cross_attn=False,
        attn_config=SelfAttentionConfig(attn_dropout=dropout, bias=enc_bias, dropout=dropout, n_head=enc_heads,
                                        n_embd=enc_d, attn_type=SelfAttentionType.MULTI_HEAD),
    )
    enc = VisionTransformerEncoderConfig(
        transformer_config=enc_tf,
        enable_gradient_checkpointing=False,
        input=ImageInputSpec(n_channels=3, width=img, height=img),
        n_layer=enc_layers,
        n_cls=n_cls,
        num_patches=num_patches,
        n_channels=conv_out,
        feature_extractor_gate_sizes=tuple(conv_gates),
        feature_extractor_kernel_size=(kernel, kernel),
    )
    dec_tf = TransformerConfig(
        rotator_config=MLPConfig(ff_mult=4),
        is_causal=True,
        is_cross_attn=True,
        attn_config=SelfAttentionConfig(attn_dropout=dropout, bias=True, dropout=dropout, n_head=dec_heads,
                                        n_embd=dec_d, attn_type=SelfAttentionType.MULTI_HEAD),
    )
    dec = TransformerDecoderConfig(
        transformer_config=dec_tf,
        n_layer=dec_layers,
        block_size=block_size,
        vocab_size=vocab,
        enable_gradient_checkpointing=False,
    )
    return VisionEncoderDecoderConfig(
        vision_encoder_config=enc,
        decoder_config=dec,
        use_cross_attn=use_cross_attn,
        use_soft_prompting=use_soft_prompting,
        no_repeat_n_grams=tuple(no_repeat_n_grams),
    )


def nano224_config(dropout: float = 0.0) -> VisionEncoderDecoderConfig:
    """SURVEY.md 8(d) "nano-224": 6x512 ViT (8 heads, 196 flat patches of 8192, 64 CLS) + 12x768 nanoGPT decoder."""
    return _model_config(img=224, num_patches=14, conv_gates=(8, 16), conv_out=32, kernel=6,
                         enc_layers=6, enc_d=512, enc_heads=8, n_cls=64, enc_bias=False,
                         dec_layers=12, dec_d=768, dec_heads=12, block_size=256, vocab=GPT2_VOCAB,
                         dropout=dropout)


def tiny_config(dropout: float = 0.0, **overrides) -> VisionEncoderDecoderConfig:
    """Same topology as nano-224 at fixture size: 32x32 images, 16 flat patches of 512, 8 CLS, vocab 384."""
    kw = dict(img=32, num_patches=4, conv_gates=(4, 8), conv_out=8, kernel=6,
              enc_layers=2, enc_d=64, enc_heads=1, n_cls=8, enc_bias=False,
              dec_layers=2, dec_d=128, dec_heads=2, block_size=48, vocab=384,
              dropout=dropout)
    kw.update(overrides)
    return _model_config(**kw)


def _family_config(*, img: int, num_patches: int, conv_gates, conv_out: int, kernel: int, n_cls: int, d: int, heads: int,
                   enc_layers: int, dec_layers: int, block_size: int, vocab: int, dropout: float, experts: int, proj: int, gate_sizes,
                   enc_ff: float, dec_ff: float, enc_top_k: int, dec_top_k: int, sparsity: float, sparse: bool = True,
                   attn_type=SelfAttentionType.MULTI_QUERY, moe: bool = True, use_cross_attn: bool = True,
                   use_soft_prompting: bool = True, skip_alternate_cross_attn: bool = True,
                   advanced_pos_emb_gate_sizes=None) -> VisionEncoderDecoderConfig:
    """The nano-mini topology (reference training_configs/gpu/nano-mini.yaml:17-78): multi-query attention, MoE rotators,
    sparse token subsets with max_block_size = patches + n_cls (encoder) / block_size + n_cls (decoder)."""
    P2 = num_patches ** 2

    def rot(ff, top_k):
        if not moe:
            return MLPConfig(ff_mult=ff)
        return MoEConfig(num_experts=experts, proj_features=proj, gate_sizes=tuple(gate_sizes) if gate_sizes else None,
                         ff_mult_factor=ff, top_k=top_k)

    def tf(bias, ff, top_k, causal, cross, max_block):
        return TransformerConfig(
            rotator_config=rot(ff, top_k), is_causal=causal, is_cross_attn=cross, max_block_size=max_block,
            is_sparse_attn=sparse, sparsity_factor=sparsity,
            attn_config=SelfAttentionConfig(attn_dropout=dropout, bias=bias, dropout=dropout, n_head=heads, n_embd=d, attn_type=attn_type))

    enc = VisionTransformerEncoderConfig(
        transformer_config=tf(False, enc_ff, enc_top_k, False, False, P2 + n_cls),
        enable_gradient_checkpointing=False, input=ImageInputSpec(n_channels=3, width=img, height=img), n_layer=enc_layers,
        n_cls=n_cls, num_patches=num_patches, n_channels=conv_out, feature_extractor_gate_sizes=tuple(conv_gates),
        feature_extractor_kernel_size=(kernel, kernel))
    dec = TransformerDecoderConfig(
        transformer_config=tf(True, dec_ff, dec_top_k, True, True, block_size + n_cls),
        n_layer=dec_layers, block_size=block_size, vocab_size=vocab, enable_gradient_checkpointing=False,
        skip_alternate_cross_attn=skip_alternate_cross_attn, use_advanced_pos_emb=advanced_pos_emb_gate_sizes is not None,
        advanced_pos_emb_gate_sizes=tuple(advanced_pos_emb_gate_sizes) if advanced_pos_emb_gate_sizes else None)
    return VisionEncoderDecoderConfig(vision_encoder_config=enc, decoder_config=dec, use_cross_attn=use_cross_attn,
                                      use_soft_prompting=use_soft_prompting, no_repeat_n_grams=(2, 3, 4, 5))


def nano_mini_config(dropout: float = 0.0) -> VisionEncoderDecoderConfig:
    """reference training_configs/gpu/nano-mini.yaml: 12x1024 ViT over 128x128 images (256 flat patches of 2048 + 64 CLS) and a
    12x1024 decoder (block 256, vocab 50258); 8 query heads of 128 on one shared K/V head, 4 experts of rank 16 (top-2 in the
    encoder, top-1 in the decoder), half of the positions attended per layer."""
    return _family_config(img=128, num_patches=16, conv_gates=(8, 16), conv_out=32, kernel=6, n_cls=64, d=1024, heads=8,
                          enc_layers=12, dec_layers=12, block_size=256, vocab=50258, dropout=dropout, experts=4, proj=16,
                          gate_sizes=(32,), enc_ff=2, dec_ff=4, enc_top_k=2, dec_top_k=1, sparsity=0.5)


def mini_config(dropout: float = 0.0, **overrides) -> VisionEncoderDecoderConfig:
    """nano-mini at fixture size: 32x32 images, 16 flat patches of 512 + 8 CLS, 2+2 layers of width 256 (2 heads of 128),
    block 40, vocab 384."""
    kw = dict(img=32, num_patches=4, conv_gates=(4, 8), conv_out=8, kernel=6, n_cls=8, d=256, heads=2, enc_layers=2, dec_layers=2,
              block_size=40, vocab=384, dropout=dropout, experts=4, proj=16, gate_sizes=(32,), enc_ff=2, dec_ff=4, enc_top_k=2,
              dec_top_k=1, sparsity=0.5)
    kw.update(overrides)
    return _family_config(**kw)


def reference_unit_test_config() -> VisionEncoderDecoderConfig:
    """The configuration of the reference's only unit test (models/vision_encoder_decoder_test.py:21-88): multi-query attention with
    16-wide heads, MoE rotators without a gate hidden layer (top-2 of 4 experts of rank 8, ff 2.5), 4x4 convolutions, 32x32 = 1024
    flat patches of 512 + 24 CLS, and a NON-causal decoder with cross-attention and a soft prompt."""
    def tf(causal, cross):
        return TransformerConfig(
            rotator_config=MoEConfig(num_experts=4, proj_features=8, gate_sizes=None, ff_mult_factor=2.5, top_k=2),
            attn_config=SelfAttentionConfig(attn_type=SelfAttentionType.MULTI_QUERY, n_embd=64, n_head=4),
            is_causal=causal, is_cross_attn=cross)
    enc = VisionTransformerEncoderConfig(
        transformer_config=tf(False, False), enable_gradient_checkpointing=True, input=ImageInputSpec(n_channels=3, width=128, height=128),
        n_layer=2, n_cls=24, num_patches=32, n_channels=32, feature_extractor_gate_sizes=(8, 16), feature_extractor_kernel_size=(4, 4))
    dec = TransformerDecoderConfig(transformer_config=tf(False, True), n_layer=2, block_size=256, vocab_size=1024)
    return VisionEncoderDecoderConfig(vision_encoder_config=enc, decoder_config=dec, use_cross_attn=True, use_soft_prompting=True)


def fake_tokenizer(vocab_size: int, eos: Optional[int] = None):
    """The four attributes ModelTrainerWrapper reads from a tokenizer (reference training/wrapper.py:88,157,173,188)."""
    eos = vocab_size - 1 if eos is None else eos
    return SimpleNamespace(eos_token_id=eos, bos_token_id=eos, mask_token_id=None, vocab_size=vocab_size)


def synthetic_batch(batch: int, img: int, caption_len: int, vocab: int, seed: int = 1, eos: Optional[int] = None,
                    ignore_index: int = -100, min_len: int = 8):
    """images (B,3,img,img) fp32 ~ N(0,1); labels (B,caption_len) int64 = ids, EOS at a random length, ignore after."""
    eos = vocab - 1 if eos is None else eos
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, 3, img, img, generator=g)
    min_len = min(min_len, caption_len - 1)
    lens = torch.randint(min_len, caption_len, (batch,), generator=g)
    ids = torch.randint(0, eos, (batch, caption_len), generator=g)
    pos = torch.arange(caption_len).unsqueeze(0)
    labels = torch.where(pos < lens.unsqueeze(1), ids, torch.full_like(ids, eos))
    labels = torch.where(pos <= lens.unsqueeze(1), labels, torch.full_like(ids, ignore_index))
    return images, labels


def _reference_style(name: str, p: torch.Tensor, x: torch.Tensor, n_layer_dec: int) -> torch.Tensor:
    """The reference's own initial DISTRIBUTIONS (same moments, Gaussian draws): decoder = nanoGPT init
    (decoder.py:192-212: N(0, 0.02) weights, zero biases, c_proj N(0, 0.02/sqrt(2L)), xavier in_proj); encoder = torch
    defaults (Linear/Conv kaiming-uniform(a=sqrt 5) => std 1/sqrt(3 fan_in), Embedding N(0,1), CLS randn/sqrt(d))."""
    leaf = name.rsplit('.', 1)[-1]
    is_ln = ('.ln_' in name or name.startswith('ln_') or 'ln_input' in name)
    if is_ln:
        return torch.ones_like(x) if leaf == 'weight' else torch.zeros_like(x)
    if name.startswith('decoder.'):
        if leaf in ('bias', 'in_proj_bias'):
            return torch.zeros_like(x)
        if leaf == 'in_proj_weight':
            return x * (2.0 / (p.shape[0] + p.shape[1])) ** 0.5
        if name.endswith('c_proj.weight'):
            return x * 0.02 / (2 * n_layer_dec) ** 0.5
        return 0.02 * x
    if 'cls_token' in name:
        return x / (p.shape[-1] ** 0.5)
    if '.wpe.' in name:
        return x
    fan_in = p[0].numel() if p.dim() >= 2 else None
    if fan_in is None:                                      # conv / linear biases: U(+-1/sqrt(fan_in)) scale
        return 0.05 * x
    return x / (3.0 * fan_in) ** 0.5


@torch.no_grad()
def det_init_(module: torch.nn.Module, seed: int = 0, style: str = 'stress') -> torch.nn.Module:
    """Overwrite every parameter from ``Generator(seed ^ crc32(name))`` on the CPU, then copy to the param's device.

    ``style='stress'`` (default): scales chosen so that nothing is hidden by a trivial value and every residual
    branch contributes O(1): linear maps N(0, 0.7^2/fan_in), LayerNorm gains 1 + 0.1 N, biases 0.02 N, CLS N(0, 1/d).
    ``style='reference'``: the reference's own initial distributions (see ``_reference_style``).
    Tied parameters are visited once (``named_parameters`` de-duplicates).
    """
    n_layer_dec = sum(1 for n, _ in module.named_parameters() if n.startswith('decoder.') and n.endswith('ln_1.weight'))
    for name, p in module.named_parameters():
        g = torch.Generator().manual_seed((seed * 1000003) ^ zlib.crc32(name.encode()))
        x = torch.randn(p.shape, generator=g, dtype=torch.float32)
        leaf = name.rsplit('.', 1)[-1]
        if style == 'reference':
            p.copy_(_reference_style(name, p, x, max(n_layer_dec, 1)).to(p.device, p.dtype))
            continue
        if p.dim() >= 2 and ('ln_' in name or 'ln_input' in name) and leaf == 'weight':
            x = 1.0 + 0.1 * x                      # LayerNormND gain (2-D)
        elif p.dim() == 1 and leaf == 'weight':
            x = 1.0 + 0.1 * x                      # LayerNorm gain
        elif leaf in ('bias', 'in_proj_bias'):
            x = 0.02 * x
        elif 'cls_token' in name:
            x = x / (p.shape[-1] ** 0.5)
        elif p.dim() == 4:                         # conv kernels: keep activations O(1)
            fan_in = p.shape[1] * p.shape[2] * p.shape[3]
            x = x / (fan_in ** 0.5)
        elif ('.wte.' in name or '.wpe.' in name or name.startswith('lm_head') or '.lm_head.' in name) and '.wpe.models.' not in name:
            x = 0.02 * x                           # embeddings / tied head: reference decoder.py:206-212 scale
        elif p.dim() == 2:
            x = 0.7 * x / (p.shape[1] ** 0.5)      # linear maps: O(1) activations, non-flat attention softmax
        else:
            x = 0.02 * x
        p.copy_(x.to(p.device, p.dtype))
    return module


@torch.no_grad()
def sharpen_gates_(module: torch.nn.Module, gain: float = 4.0) -> torch.nn.Module:
    """Scale the last Linear of every MoE expert gate by gain * sqrt(in_features).  MoELinear divides its gate logits by
    sqrt(in_features) (reference layers.py:335), so freshly initialised gates are uniform to three decimals and the top-k choice
    is decided by noise; with this the logits differ by O(1) and the fixtures exercise a decisive, input-dependent routing.
    Works on the reference's modules and on ours (same attribute names)."""
    for m in module.modules():
        if hasattr(m, 'expert_gates') and hasattr(m, '_in_features'):
            last = m.expert_gates.model[-1]
            last.weight.mul_(gain * m._in_features ** 0.5)
            if last.bias is not None:
                last.bias.mul_(gain * m._in_features ** 0.5)
    return module
