#!/usr/bin/env python3
"""GPU-box diagnostic (test infrastructure: imports the CPU oracle): where does the bf16 error of the nano-224 forward come
from?  Stage-by-stage comparison of the HIP path's saved activations with the fp32 oracle, plus three substitution
experiments (decoder fed the oracle's encoder output; lm_head fed the oracle's hidden state; lm_head with a hi+lo split of the
hidden state), for both weight initialisations the parity tests use.

    python tools/diag_precision.py [reference|stress]      -> gpurun_out/diag_precision_<style>.json + a table on stdout
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder  # noqa: E402
from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch  # noqa: E402
from oracle import reference_model as orc  # noqa: E402


def stats(name, got, ref, out):
    got, ref = got.detach().float().cpu().reshape(-1), ref.detach().float().cpu().reshape(-1)
    e = (got - ref).abs()
    out[name] = dict(max=float(e.max()), rms=float(e.pow(2).mean().sqrt()), ref_absmax=float(ref.abs().max()),
                     ref_rms=float(ref.pow(2).mean().sqrt()))
    o = out[name]
    print(f'{name:44s} max {o["max"]:.5f} rms {o["rms"]:.6f}  rel-rms {o["rms"] / max(o["ref_rms"], 1e-30):.5f}  '
          f'(ref absmax {o["ref_absmax"]:.3f} rms {o["ref_rms"]:.4f})')


def oracle_stages(sd, cfg, images, ids):
    """The oracle's forward, stage by stage (same functions as orc.forward's text segment)."""
    st = {}
    e = orc._sub(sd, 'encoder.0.')
    ec = cfg.vision_encoder_config
    ac = ec.transformer_config.attn_config
    P2 = ec.num_patches ** 2
    x = orc.conv_stack(e, 'feature_extractor', images)
    st['enc.conv'] = x
    n = x.size(0)
    x = F.linear(x.reshape(n, P2, -1), e['projector.weight'], e.get('projector.bias'))
    st['enc.projector'] = x
    x = orc.layer_norm(x, e['ln_input.weight'], e.get('ln_input.bias'))
    x = orc.layer_norm(x + e['transformer.wpe.weight'][:P2].unsqueeze(0), e['ln_input.weight'], e.get('ln_input.bias'))
    x = torch.cat((e['cls_token'].expand(n, -1, -1), x), dim=1)
    st['enc.x0'] = x
    for i in range(ec.n_layer):
        x = orc.transformer_block(e, f'transformer.h.{i}', x, ac.n_head, False, None, None)
        st[f'enc.h{i}'] = x
    y = orc.layer_norm(x[:, :ec.n_cls].contiguous(), e['transformer.ln_f.weight'], e.get('transformer.ln_f.bias'))
    st['enc.ln_f'] = y
    mem = F.linear(y, sd['encoder.1.weight'])
    st['encoder_output'] = mem
    d = orc._sub(sd, 'decoder.')
    dc = cfg.decoder_config
    dac = dc.transformer_config.attn_config
    t = ids.size(1)
    x = d['transformer.wte.weight'][ids] + d['transformer.wpe.weight'][ec.n_cls:ec.n_cls + t]
    st['dec.x0'] = x
    for l in range(dc.n_layer):
        m = mem if (l % 2 == 0 or not dc.skip_alternate_cross_attn) else None
        x = orc.transformer_block(d, f'transformer.h.{l}', x, dac.n_head, True, m, None)
        st[f'dec.h{l}'] = x
    h = orc.layer_norm(x, d['transformer.ln_f.weight'], d.get('transformer.ln_f.bias'))
    st['hidden_text'] = h
    st['logits'] = F.linear(h, d['transformer.wte.weight'])
    return st


def main():
    style = sys.argv[1] if len(sys.argv) > 1 else 'reference'
    cfg = nano224_config()
    model = det_init_(VisionEncoderDecoder(cfg), seed=0, style=style)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    ids, _ = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id)
    with torch.no_grad():
        ref = oracle_stages(sd, cfg, images, ids)
    model = model.cuda().eval()
    eng = model._engine
    out = {}
    B, T, ncls = 2, ids.size(1), cfg.vision_encoder_config.n_cls
    with torch.no_grad():
        eng.prepare(False)
        enc_out, ectx = eng.encode(images.cuda(), True)
        stats('enc.conv (bf16 NCHW pre-activation)', ectx.acts[-1], ref['enc.conv'], out)
        stats('enc.projector', ectx.proj, ref['enc.projector'], out)
        stats('enc.x0 (2x LayerNormND + wpe + CLS)', ectx.saves[0].x, ref['enc.x0'], out)
        for l in range(1, len(ectx.saves)):
            stats(f'enc.h{l - 1}', ectx.saves[l].x, ref[f'enc.h{l - 1}'], out)
        last = len(ectx.saves) - 1
        stats(f'enc.h{last} (CLS rows)', ectx.cls, ref[f'enc.h{last}'][:, :ncls], out)
        stats('encoder_output (ln_f + bridge)', enc_out, ref['encoder_output'], out)
        mem = eng._mem_bf16(enc_out)
        hid, hb, dctx = eng.decode_segment(B, T, mem, ncls, True, ids=ids.cuda(), pos_offset=ncls)
        stats('dec.x0', dctx.saves[0].x, ref['dec.x0'], out)
        for l in range(1, len(dctx.saves)):
            stats(f'dec.h{l - 1}', dctx.saves[l].x, ref[f'dec.h{l - 1}'], out)
        stats(f'dec.h{len(dctx.saves) - 1}', dctx.xl, ref[f'dec.h{len(dctx.saves) - 1}'], out)
        stats('hidden_text (ln_f)', hid, ref['hidden_text'], out)
        logits = eng.logits_f32(hb, B * T)
        stats('logits', logits, ref['logits'], out)
        # substitution experiments
        mem_o = eng._mem_bf16(ref['encoder_output'].cuda())
        hid2, hb2, _ = eng.decode_segment(B, T, mem_o, ncls, False, ids=ids.cuda(), pos_offset=ncls)
        stats('hidden_text | oracle encoder_output', hid2, ref['hidden_text'], out)
        stats('logits      | oracle encoder_output', eng.logits_f32(hb2, B * T), ref['logits'], out)
        h_o = ref['hidden_text'].reshape(B * T, -1).cuda()
        stats('logits      | oracle hidden, bf16 head', eng.logits_f32(h_o.to(torch.bfloat16).contiguous(), B * T), ref['logits'], out)
        hi = hid.to(torch.bfloat16)
        lo = (hid - hi.float()).to(torch.bfloat16)
        lg = eng.logits_f32(hi.contiguous(), B * T) + eng.logits_f32(lo.contiguous(), B * T)
        stats('logits      | own hidden, hi+lo split A', lg, ref['logits'], out)
        w = eng.arena.P('decoder.transformer.wte.weight')
        stats('logits      | own hidden, fp32 head (torch)', hid @ w.t(), ref['logits'], out)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open(f'gpurun_out/diag_precision_{style}.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
