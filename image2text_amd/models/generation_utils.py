"""Stochastic beam search over the captioning model (reference models/generation_utils.py:10-148).

Same constructor, ``__call__(inputs, decoded_ids) -> (ids (B, beam_width, L), cumulative log scores (B, beam_width))``
and per-step semantics as the reference's ``BeamSearchTokenGenerator``; the logits come from the HIP hot path
(``model(images=None, ids=..., encoder_output=...)``: the encoder runs once, every step re-runs the decoder over the
grown prefix as the reference does), the candidate bookkeeping is a handful of small torch index ops on the device.
Beams are kept batch-major, (B, W, L), instead of the reference's beam-major (W, B, L) with transposes around every gather.

Per step, for every live beam (reference :56-93): last-position logits -> no-repeat-n-gram ban -> optional top-k crop ->
``beam_expansion_factor`` candidate tokens (temperature <= 0: the top ones of the raw scores; else sampled without
replacement from softmax(scores / temperature)) with their log-probabilities.  A beam that already ended in EOS keeps
emitting EOS at log-score 0 whenever continuing would cost more than ``log(length_boost)``; other candidates get the
length boost added.  Consolidation (reference :95-148): of the W x E candidates of a caption keep W, the best by
cumulative score (``consolidation_temperature <= 0``, sorted) or sampled from softmax(cumulative / temperature).
The loop ends at ``max_new_tokens`` or when every beam contains an EOS anywhere -- with ``prompt = BOS = EOS`` that is
true before the first step (reference behaviour), so pass ``eos_token_id=None`` or a distinct BOS.
"""
import math
from typing import Optional

import torch
from transformers import LogitsProcessorList, NoRepeatNGramLogitsProcessor


class BeamSearchTokenGenerator:
    def __init__(self,
                 model,
                 beam_width: int = 3,
                 temperature: float = 1.0,
                 top_k: Optional[int] = None,
                 max_new_tokens=64,
                 no_repeat_n_grams=(2, 3, 4),
                 beam_expansion_factor: int = 4,
                 eos_token_id: Optional[int] = None,
                 consolidation_temperature: float = 1.0,
                 length_boost: float = 1.0):
        self.model = model
        self.beam_width = beam_width
        self.beam_expansion_factor = beam_expansion_factor
        self.max_new_tokens = max_new_tokens
        self.temperature = temperature
        self.consolidation_temperature = consolidation_temperature
        self.top_k = top_k
        self.eos_token_id = eos_token_id
        self.length_boost = math.log(length_boost)
        self.processor = LogitsProcessorList([NoRepeatNGramLogitsProcessor(ngram_size=n) for n in no_repeat_n_grams])

    @torch.no_grad()
    def __call__(self, inputs, decoded_ids):
        self.model.eval()
        W = self.beam_width
        mem = self.model.encoder(inputs)                                        # (B, n_cls, d), once
        B = mem.size(0)
        mem = mem.unsqueeze(1).expand(B, W, *mem.shape[1:]).reshape(B * W, *mem.shape[1:]).contiguous()
        provided = decoded_ids.size(-1) - 1
        beams = decoded_ids.to(mem.device).unsqueeze(1).expand(B, W, -1).contiguous()      # (B, W, L)
        scores = torch.zeros(B, W, device=mem.device)
        while beams.size(-1) < self.max_new_tokens + provided and not self._all_ended(beams):
            cand_ids, cand_lp = self.decode_next(mem, beams)                    # (B, W, E) each
            beams, scores = self.consolidate_candidates(beams, scores, cand_ids, cand_lp)
        return beams, scores

    def _all_ended(self, beams) -> bool:
        if self.eos_token_id is None:
            return False
        return bool((beams == self.eos_token_id).any(dim=-1).all())

    def decode_next(self, mem, beams):
        B, W, L = beams.shape
        E = self.beam_expansion_factor
        flat = beams.reshape(B * W, L)
        ended = (flat[:, -1:] == self.eos_token_id) if self.eos_token_id is not None else torch.zeros_like(flat[:, -1:], dtype=torch.bool)
        logits = self.model(images=None, ids=flat, encoder_output=mem).logits[:, -1, :].float()
        logits = self.processor(flat, logits)
        if self.top_k is not None:
            kth = torch.topk(logits, min(self.top_k, logits.size(-1)), dim=-1).values[:, -1:]
            logits = logits.masked_fill(logits < kth, -float('inf'))
        if self.temperature <= 0:
            logp = logits.log_softmax(dim=-1)
            nxt = logits.topk(k=E, dim=-1, sorted=False).indices
        else:
            logp = (logits / self.temperature).log_softmax(dim=-1)
            nxt = torch.multinomial(logp.exp(), num_samples=E)
        lp = logp.gather(-1, nxt)
        if self.eos_token_id is not None:
            stay = ended & (lp + self.length_boost < 0)                         # an ended beam pads with EOS for free
            nxt = torch.where(stay, torch.full_like(nxt, self.eos_token_id), nxt)
            lp = torch.where(stay, torch.zeros_like(lp), lp + self.length_boost)
        return nxt.view(B, W, E), lp.view(B, W, E)

    def identify(self, scores, cand_lp):
        """Which (beam, candidate) pairs survive: two (B, W) index tensors."""
        B, W, E = cand_lp.shape
        total = (scores.unsqueeze(-1) + cand_lp).reshape(B, W * E)
        if self.consolidation_temperature <= 0:
            pick = total.topk(k=W, dim=-1, sorted=True).indices
        else:
            pick = torch.multinomial((total / self.consolidation_temperature).softmax(dim=-1), num_samples=W)
        return pick // E, pick % E

    def consolidate_candidates(self, beams, scores, cand_ids, cand_lp):
        beam_idx, cand_idx = self.identify(scores, cand_lp)
        L = beams.size(-1)
        kept = beams.gather(1, beam_idx.unsqueeze(-1).expand(-1, -1, L))
        flat_idx = beam_idx * cand_ids.size(-1) + cand_idx
        new_ids = cand_ids.flatten(1).gather(1, flat_idx).unsqueeze(-1)
        new_lp = cand_lp.flatten(1).gather(1, flat_idx)
        return torch.cat((kept, new_ids), dim=-1), scores.gather(1, beam_idx) + new_lp
