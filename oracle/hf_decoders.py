"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

The Hugging Face decoders the reference wraps (models/decoder.py:285-440) live in a third-party dependency, ``transformers``
(unpinned in the reference's requirements.txt:7; 5.15.0 in the build image).  This file restates, as pure fp32 functions over a
state dict in transformers' own key names, the published forward of the three architectures the HIP hot path serves:

  * GPT-2 with optional cross-attention  (transformers models/gpt2/modeling_gpt2.py: GPT2Block / GPT2Attention / GPT2MLP) -- reached
    from the reference through GPT2HuggingfaceDecoder (decoder.py:364-382);
  * Llama-2 and Qwen2  (models/llama/modeling_llama.py, models/qwen2/modeling_qwen2.py: RMSNorm, rotate_half rotary embedding,
    grouped-query attention, SwiGLU) -- Llama2HuggingfaceDecoder / Qwen2HuggingfaceDecoder (decoder.py:404-440);
  * peft's LoRA layer around a Conv1D  (peft tuners/lora/layer.py: result + lora_B(lora_A(dropout(x))) * scaling) --
    models/utils.py:46-65.  peft is NOT in the image: this part is a restatement of its published formula only.

and the reference's composition of encoder output and decoder (models/vision_encoder_decoder.py:84-134 with a HuggingfaceDecoder:
soft prompt = encoder outputs concatenated in front of the token embeddings, attention_mask=None -> one causal sequence).

Parity status: PINNED for GPT-2 / Llama-2 / Qwen2 -- ``tests/test_hf_oracle.py`` checks every function against the transformers
modules themselves (randomly initialised configurations, logits and hidden states to <= 2e-5, gradients by autograd through both);
the LoRA restatement is UNPINNED (no peft to run) and says so wherever it is used.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _causal(T: int) -> torch.Tensor:
    return torch.zeros(T, T).masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float('-inf'))


# --------------------------------------------------------------------------------------------------------------
# GPT-2 (modeling_gpt2.py).  Conv1D stores [in, out]: y = x @ W + b
# --------------------------------------------------------------------------------------------------------------
def _conv1d(sd: SD, p: str, x, lora: Optional[dict] = None):
    y = x @ sd[f'{p}.weight'] + sd[f'{p}.bias']
    if lora is not None and p in lora:            # peft: base(x) + lora_B(lora_A(dropout(x))) * lora_alpha / r   (UNPINNED restatement)
        A, B, scale, mask = lora[p]
        xd = x if mask is None else x * mask.view(x.shape)
        y = y + (xd @ A.t() @ B.t()) * scale
    return y


def _gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def _mha(q, k, v, n_head: int, add_mask):
    B, Tq, d = q.shape
    hd = d // n_head
    q, k, v = (t.view(B, -1, n_head, hd).transpose(1, 2) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if add_mask is not None:
        s = s + add_mask
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Tq, d)


def gpt2_decoder(sd: SD, n_layer: int, n_head: int, inputs_embeds, encoder_hidden_states=None, eps: float = 1e-5, lora: Optional[dict] = None):
    """GPT2LMHeadModel.forward(inputs_embeds=..., encoder_hidden_states=...) in eval mode -> (logits, last hidden state after ln_f).
    ``sd``: GPT2LMHeadModel.state_dict().  ``lora``: {module path: (A, B, scale, input mask | None)} for adapted Conv1D modules."""
    B, T, d = inputs_embeds.shape
    x = inputs_embeds + sd['transformer.wpe.weight'][:T]
    mask = _causal(T)
    for l in range(n_layer):
        p = f'transformer.h.{l}'
        h = F.layer_norm(x, (d,), sd[f'{p}.ln_1.weight'], sd[f'{p}.ln_1.bias'], eps)
        q, k, v = _conv1d(sd, f'{p}.attn.c_attn', h, lora).split(d, dim=-1)
        x = x + _conv1d(sd, f'{p}.attn.c_proj', _mha(q, k, v, n_head, mask), lora)
        if encoder_hidden_states is not None and f'{p}.crossattention.q_attn.weight' in sd:
            h = F.layer_norm(x, (d,), sd[f'{p}.ln_cross_attn.weight'], sd[f'{p}.ln_cross_attn.bias'], eps)
            q = _conv1d(sd, f'{p}.crossattention.q_attn', h, lora)
            k, v = _conv1d(sd, f'{p}.crossattention.c_attn', encoder_hidden_states, lora).split(d, dim=-1)
            x = x + _conv1d(sd, f'{p}.crossattention.c_proj', _mha(q, k, v, n_head, None), lora)
        h = F.layer_norm(x, (d,), sd[f'{p}.ln_2.weight'], sd[f'{p}.ln_2.bias'], eps)
        x = x + _conv1d(sd, f'{p}.mlp.c_proj', _gelu_new(_conv1d(sd, f'{p}.mlp.c_fc', h, lora)), lora)
    x = F.layer_norm(x, (d,), sd['transformer.ln_f.weight'], sd['transformer.ln_f.bias'], eps)
    return x @ sd['lm_head.weight'].t(), x


# --------------------------------------------------------------------------------------------------------------
# Llama-2 / Qwen2 (modeling_llama.py / modeling_qwen2.py)
# --------------------------------------------------------------------------------------------------------------
def rms_norm(x, w, eps: float):
    """LlamaRMSNorm: w * x * rsqrt(mean(x^2) + eps), statistics in fp32"""
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def rotary_tables(T: int, head_dim: int, theta: float = 10000.0):
    """LlamaRotaryEmbedding (rope_type 'default'): cos / sin of position x inv_freq, the frequency vector duplicated over both halves"""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = torch.outer(torch.arange(T, dtype=torch.float32), inv)
    emb = torch.cat((ang, ang), dim=-1)
    return emb.cos(), emb.sin()


def apply_rotary(x, cos, sin):
    """apply_rotary_pos_emb: x cos + rotate_half(x) sin, rotate_half([x1 | x2]) = [-x2 | x1]   (x: [B, heads, T, head_dim])"""
    h = x.shape[-1] // 2
    return x * cos + torch.cat((-x[..., h:], x[..., :h]), dim=-1) * sin


def llama_decoder(sd: SD, n_layer: int, n_head: int, n_kv_head: int, eps: float, inputs_embeds, theta: float = 10000.0):
    """LlamaForCausalLM / Qwen2ForCausalLM .forward(inputs_embeds=...) -> (logits, last hidden state after the final norm).  ``sd``: the
    model's state_dict() (q / k / v biases are used when present: Qwen2)."""
    B, T, d = inputs_embeds.shape
    hd = sd['model.layers.0.self_attn.q_proj.weight'].shape[0] // n_head
    cos, sin = rotary_tables(T, hd, theta)
    mask = _causal(T)
    x = inputs_embeds
    lin = lambda t, p: F.linear(t, sd[f'{p}.weight'], sd.get(f'{p}.bias'))
    for l in range(n_layer):
        p = f'model.layers.{l}'
        h = rms_norm(x, sd[f'{p}.input_layernorm.weight'], eps)
        q = lin(h, f'{p}.self_attn.q_proj').view(B, T, n_head, hd).transpose(1, 2)
        k = lin(h, f'{p}.self_attn.k_proj').view(B, T, n_kv_head, hd).transpose(1, 2)
        v = lin(h, f'{p}.self_attn.v_proj').view(B, T, n_kv_head, hd).transpose(1, 2)
        q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
        rep = n_head // n_kv_head                                      # repeat_kv: query head h uses K/V head h // rep
        k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        s = q @ k.transpose(-1, -2) / math.sqrt(hd) + mask
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, n_head * hd)
        x = x + lin(a, f'{p}.self_attn.o_proj')
        h = rms_norm(x, sd[f'{p}.post_attention_layernorm.weight'], eps)
        x = x + lin(F.silu(lin(h, f'{p}.mlp.gate_proj')) * lin(h, f'{p}.mlp.up_proj'), f'{p}.mlp.down_proj')
    x = rms_norm(x, sd['model.norm.weight'], eps)
    return x @ sd['lm_head.weight'].t(), x


# --------------------------------------------------------------------------------------------------------------
# the reference's glue with a HuggingfaceDecoder (models/vision_encoder_decoder.py:84-134, models/decoder.py:332-361)
# --------------------------------------------------------------------------------------------------------------
def soft_prompt_forward(decoder_fn, wte, encoder_output, ids, n_positions: int, use_cross_attn: bool):
    """inputs_embeds = [encoder_output | wte[ids]] cropped to the decoder's positions (v_e_d.py:84-88); the mask built there is NOT
    handed to transformers (decoder.py:349-350): one causal sequence.  Returns (text logits, hidden state of every row)."""
    n_p = encoder_output.shape[1]
    emb = torch.cat((encoder_output, wte[ids]), dim=-2)[..., :n_positions, :]
    logits, hidden = decoder_fn(emb, encoder_output if use_cross_attn else None)
    return logits[..., n_p:, :], hidden
