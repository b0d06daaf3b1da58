#!/usr/bin/env python3
"""GPU-box diagnostic: where does the bf16 error of the nano-224 forward come from?  Compares stage outputs of the HIP
path with the fp32 CPU oracle (test infrastructure) and isolates encoder / decoder / lm_head contributions."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder  # noqa: E402
from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch  # noqa: E402
from oracle import reference_model as orc  # noqa: E402


def stats(name, got, ref, out):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    e = (got - ref).abs()
    out[name] = dict(max=float(e.max()), rms=float(e.pow(2).mean().sqrt()), ref_absmax=float(ref.abs().max()),
                     ref_rms=float(ref.pow(2).mean().sqrt()))
    print(f'{name:40s} max {out[name]["max"]:.5f} rms {out[name]["rms"]:.6f}  (ref absmax {out[name]["ref_absmax"]:.3f} rms {out[name]["ref_rms"]:.3f})')


def main():
    cfg = nano224_config()
    model = det_init_(VisionEncoderDecoder(cfg), seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    ids, msk = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id)
    with torch.no_grad():
        enc_ref, logits_ref, hid_ref = orc.forward(sd, cfg, images, ids, msk)
    model = model.cuda().eval()
    out = {}
    with torch.no_grad():
        o = model(images=images.cuda(), ids=ids.cuda())
        stats('encoder_output', o.encoder_output, enc_ref, out)
        stats('hidden_text', o.hidden_state[:, 64:], hid_ref[:, 64:], out)
        stats('hidden_prompt', o.hidden_state[:, :64], hid_ref[:, :64], out)
        stats('logits', o.logits, logits_ref, out)
        # decoder alone, fed the oracle's encoder output
        o2 = model(images=None, ids=ids.cuda(), encoder_output=enc_ref.cuda())
        stats('hidden_text | oracle enc', o2.hidden_state[:, 64:], hid_ref[:, 64:], out)
        stats('logits | oracle enc', o2.logits, logits_ref, out)
        # lm_head alone, fed the oracle's hidden state
        eng = model._engine
        hb = hid_ref[:, 64:].reshape(128, -1).cuda().to(torch.bfloat16).contiguous()
        lg = eng.logits_f32(hb, 128).view(2, 64, -1)
        stats('logits | oracle hidden (bf16 head)', lg, logits_ref, out)
        # encoder stages
        conv_ref = orc.conv_stack(orc._sub(sd, 'encoder.0.'), 'feature_extractor', images)
        eng.prepare(False)
        _, _ = eng.encode(images.cuda(), False)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/diag_precision.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
