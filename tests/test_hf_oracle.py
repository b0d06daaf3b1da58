"""The oracle's restatement of the Hugging Face decoders (oracle/hf_decoders.py) pinned against transformers itself: randomly
initialised GPT-2 (with and without cross-attention), Llama-2 and Qwen2 configurations -- logits, hidden states and every parameter's
gradient.  CPU only."""
import pytest
import torch

from oracle import hf_decoders as hfo


def _perturb(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias') or 'norm' in n or '.ln_' in n:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    return model


@pytest.mark.parametrize('cross', [False, True])
def test_gpt2_restatement_matches_transformers(cross):
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(1)
    hf = _perturb(GPT2LMHeadModel(GPT2Config(n_layer=2, n_head=2, n_embd=64, n_positions=40, vocab_size=97, add_cross_attention=cross,
                                             resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)).eval(), 2)
    g = torch.Generator().manual_seed(3)
    emb = (torch.randn(2, 23, 64, generator=g) * 0.3).requires_grad_(True)
    mem = (torch.randn(2, 5, 64, generator=g) * 0.5).requires_grad_(True) if cross else None
    ref = hf(inputs_embeds=emb, encoder_hidden_states=mem, output_hidden_states=True)
    sd = {k: v for k, v in hf.named_parameters()}
    sd['lm_head.weight'] = sd['transformer.wte.weight']
    logits, hidden = hfo.gpt2_decoder(sd, 2, 2, emb, mem)
    assert float((logits - ref.logits).abs().max()) < 2e-5 and float((hidden - ref.hidden_states[-1]).abs().max()) < 2e-5
    w = torch.randn(ref.logits.shape, generator=g)
    ins = [emb] + ([mem] if cross else []) + list(hf.parameters())
    g_ref = torch.autograd.grad((ref.logits * w).sum(), ins, allow_unused=True)
    g_got = torch.autograd.grad((logits * w).sum(), ins, allow_unused=True)
    for a, b in zip(g_got, g_ref):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize('kind', ['llama', 'qwen2'])
def test_llama_restatement_matches_transformers(kind):
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(4)
    if kind == 'llama':
        cfg = LlamaConfig(hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          vocab_size=111, max_position_embeddings=64, rms_norm_eps=1e-5)
        hf = LlamaForCausalLM(cfg)
    else:
        cfg = Qwen2Config(hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                          vocab_size=111, max_position_embeddings=64, tie_word_embeddings=True)
        hf = Qwen2ForCausalLM(cfg)
    hf = _perturb(hf.eval(), 5)
    g = torch.Generator().manual_seed(6)
    emb = (torch.randn(2, 37, 128, generator=g) * 0.3).requires_grad_(True)
    ref = hf(inputs_embeds=emb, output_hidden_states=True)
    sd = {k: v for k, v in hf.named_parameters()}
    sd.setdefault('lm_head.weight', sd['model.embed_tokens.weight'])
    theta = float(cfg.rope_parameters['rope_theta'])
    logits, hidden = hfo.llama_decoder(sd, 2, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.rms_norm_eps, emb, theta)
    assert float((logits - ref.logits).abs().max()) < 2e-5 and float((hidden - ref.hidden_states[-1]).abs().max()) < 2e-5
    w = torch.randn(ref.logits.shape, generator=g)
    ins = [emb] + list(hf.parameters())
    for a, b in zip(torch.autograd.grad((logits * w).sum(), ins, allow_unused=True), torch.autograd.grad((ref.logits * w).sum(), ins, allow_unused=True)):
        assert (a is None) == (b is None)          # (embed_tokens is not reached through inputs_embeds unless it is the tied head)
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))
    # the rotary table against the checkpoint's own module (what the hot path's i2t_rope table is built from)
    cos, sin = hf.model.rotary_emb(torch.zeros(1, 1), torch.arange(37)[None])
    c2, s2 = hfo.rotary_tables(37, cfg.hidden_size // cfg.num_attention_heads, theta)
    assert float((cos[0] - c2).abs().max()) < 1e-6 and float((sin[0] - s2).abs().max()) < 1e-6


def test_soft_prompt_composition_is_one_causal_sequence():
    """v_e_d.py:84-134 with a HuggingfaceDecoder: text logits depend on the prompt rows (no mask reaches transformers), prompt rows do not
    depend on the text"""
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(7)
    hf = GPT2LMHeadModel(GPT2Config(n_layer=1, n_head=2, n_embd=32, n_positions=16, vocab_size=50, resid_pdrop=0.0, embd_pdrop=0.0,
                                    attn_pdrop=0.0)).eval()
    sd = dict(hf.named_parameters())
    sd['lm_head.weight'] = sd['transformer.wte.weight']
    fn = lambda emb, mem: hfo.gpt2_decoder(sd, 1, 2, emb, mem)
    enc = torch.randn(1, 4, 32)
    ids = torch.randint(0, 50, (1, 30))
    with torch.no_grad():
        l0, h0 = hfo.soft_prompt_forward(fn, sd['transformer.wte.weight'], enc, ids, 16, False)
        l1, h1 = hfo.soft_prompt_forward(fn, sd['transformer.wte.weight'], enc + torch.randn(1, 4, 32), ids, 16, False)
        ids2 = ids.clone(); ids2[0, 5] = (ids2[0, 5] + 1) % 50
        l2, h2 = hfo.soft_prompt_forward(fn, sd['transformer.wte.weight'], enc, ids2, 16, False)
    assert l0.shape == (1, 12, 50) and h0.shape == (1, 16, 32)                 # cropped to the 16 positions, prompt rows sliced off the logits
    assert float((l0 - l1).abs().max()) > 1e-3                                 # the text sees the prompt
    assert float((h0[:, :4] - h2[:, :4]).abs().max()) == 0.0 and float((l0[:, :5] - l2[:, :5]).abs().max()) == 0.0      # causal
