#!/usr/bin/env python3
"""Round-3 golden fixtures, produced by RUNNING THE REFERENCE (container-only tooling; same import recipe as
tools/gen_goldens.py, whose fixtures this script leaves untouched).

    python tools/gen_goldens_r3.py [beam] [vit_heads]      # default: all

  tiny_beam.npz       reference models/generation_utils.py::BeamSearchTokenGenerator on the trained tiny model
                      (tests/golden/tiny_weights.npz, images of tiny_decode.npz), four runs:
                        det         temperature 0 / consolidation temperature 0 (its deterministic setting), an EOS id no caption uses
                        det_eos     the same with top_k 5, an EOS id that the captions really emit and length_boost 1.5
                                    (exercises the "ended beam pads with EOS for free" rule and the early exit)
                        smp, smp_eos  the sampling setting (temperature 2.5 / consolidation temperature 6; smp_eos 1 / 1, with an EOS
                                    and length_boost 2) with every torch.multinomial draw RECORDED, so that a re-run which
                                    replays the draws must reproduce the beams exactly (torch's Philox stream itself
                                    cannot be matched across devices): ids, cumulative scores and the draws
  vit_heads.npz       reference models/encoder.py::PretrainedViT heads (per-slot MLP + normalize, PEER lookup, LSH cosine
                      embeddings) on recorded 768-wide backbone features: the torchvision backbone is absent from the image,
                      so a stand-in module that returns the recorded features takes its place (the heads are the reference's
                      own code, the backbone is not exercised here)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
sys.path.insert(0, REPO)
from gen_goldens import OUT, REF, install_stubs, to_ref_config      # noqa: E402


def gen_beam():
    from image2text_amd.synth import tiny_config
    from models.generation_utils import BeamSearchTokenGenerator as RefBeam
    from models.vision_encoder_decoder import VisionEncoderDecoder as RefVED
    cfg = tiny_config(dropout=0.0)
    V = cfg.decoder_config.vocab_size
    model = RefVED(to_ref_config(cfg)).eval()
    with np.load(os.path.join(OUT, 'tiny_weights.npz')) as z:
        model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
    with np.load(os.path.join(OUT, 'tiny_decode.npz')) as z:
        images = torch.from_numpy(z['images'])
        greedy = z['ids']
    B = images.shape[0]
    prompt = torch.full((B, 1), V - 1, dtype=torch.long)              # BOS of the fake tokenizer
    eos = int(greedy[0, 6])                                           # a token the first caption emits mid-way
    print('greedy rows:', greedy[:, :10].tolist(), 'eos stand-in:', eos)
    rare = 5      # (the reference's loop test needs SOME eos id: `decoded_ids == None` is a bool; an id the captions do not use)
    out = {'images': images.numpy(), 'prompt': prompt.numpy(), 'eos': np.int64(eos), 'rare': np.int64(rare)}
    runs = {
        'det': dict(beam_width=3, temperature=0.0, top_k=None, max_new_tokens=12, no_repeat_n_grams=(2, 3, 4), beam_expansion_factor=4,
                    eos_token_id=rare, consolidation_temperature=0.0, length_boost=1.0),
        'det_eos': dict(beam_width=3, temperature=0.0, top_k=5, max_new_tokens=14, no_repeat_n_grams=(2, 3), beam_expansion_factor=4,
                        eos_token_id=eos, consolidation_temperature=0.0, length_boost=1.5),
        'smp': dict(beam_width=3, temperature=2.5, top_k=None, max_new_tokens=10, no_repeat_n_grams=(2, 3, 4), beam_expansion_factor=4,
                    eos_token_id=rare, consolidation_temperature=6.0, length_boost=1.0),
        'smp_eos': dict(beam_width=4, temperature=1.0, top_k=None, max_new_tokens=14, no_repeat_n_grams=(2, 3), beam_expansion_factor=3,
                        eos_token_id=eos, consolidation_temperature=1.0, length_boost=2.0),
    }
    real_multinomial = torch.multinomial
    for tag, kw in runs.items():
        draws = []

        def rec(p, num_samples, *a, **k):
            r = real_multinomial(p, num_samples, *a, **k)
            draws.append(r.clone())
            return r
        torch.manual_seed(100 + len(tag))
        torch.multinomial = rec
        try:
            with torch.no_grad():
                ids, scores = RefBeam(model, **kw)(images, prompt)
        finally:
            torch.multinomial = real_multinomial
        out[f'{tag}.ids'], out[f'{tag}.scores'] = ids.numpy(), scores.numpy()
        for i, dr in enumerate(draws):
            out[f'{tag}.draw.{i}'] = dr.numpy()
        out[f'{tag}.n_draws'] = np.int64(len(draws))
        print(tag, 'ids', tuple(ids.shape), 'draws', len(draws), 'scores', scores[0].tolist())
        print('   beams of caption 0:', ids[0].tolist())
    np.savez_compressed(os.path.join(OUT, 'tiny_beam.npz'), **out)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {'beam', 'vit_heads'}
    if 'beam' in which:
        gen_beam()
    if 'vit_heads' in which:
        from gen_goldens_vit import gen_vit_heads
        gen_vit_heads()


if __name__ == '__main__':
    main()
