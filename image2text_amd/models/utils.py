"""Small host-side helpers of the plugin surface (reference models/utils.py:18-43)."""
import fnmatch
from copy import deepcopy
from typing import List, Optional

import torch
import torch.nn as nn

from ..configs.models import TransformerConfig


# A decoder plugin whose parameter NAMES differ from the reference's (GPT2HuggingfaceDecoder keeps the hot path's own layout and
# translates only its state dict) registers {own name under 'decoder.': [reference names]} here; patterns are tried against both.
_DECODER_NAME_ALIASES = {}          # own name -> set of reference names (union over every plugin instance built in this process)


def register_decoder_name_aliases(table):
    for name, refs in table.items():
        _DECODER_NAME_ALIASES.setdefault(name, set()).update(refs)


def reference_names(candidate: str) -> List[str]:
    """every reference-side name the parameter the hot path calls ``candidate`` may carry (``[candidate]`` when there is no alias)"""
    i = candidate.find('decoder.')
    refs = _DECODER_NAME_ALIASES.get(candidate[i + 8:]) if i >= 0 else None
    return [candidate] if not refs else sorted(candidate[:i + 8] + r for r in refs)


def state_dict_keys_of_parameters(model: nn.Module):
    """{parameter name under ``model.named_parameters()``: [its state-dict key(s)]} -- the same string except under a module that
    publishes ``reference_parameter_names()`` (GPT2HuggingfaceDecoder: Hugging Face / peft keys, a fused matrix may have two)."""
    table = {n: [n] for n, _ in model.named_parameters()}
    for mod_name, mod in model.named_modules():
        fn = getattr(mod, 'reference_parameter_names', None)
        if fn is not None:
            pre = mod_name + '.' if mod_name else ''
            for n, refs in fn().items():
                if pre + n in table:
                    table[pre + n] = [pre + r for r in refs]
    return table


class PatternMatcher:
    """fnmatch over parameter names; an empty/None pattern list matches everything (reference utils.py:18-28).  A pattern written
    against the reference's name of a parameter (``*.crossattention.*``, ``*lora_A*``) also selects it under the hot path's name."""

    def __init__(self, patterns: Optional[List[str]]):
        self.patterns = patterns

    def match(self, candidate: str) -> bool:
        if not self.patterns:
            return True
        names = {candidate, *reference_names(candidate)}
        return any(fnmatch.fnmatch(n, pat) for n in names for pat in self.patterns)


def update_state_dict_from_partial_checkpoint(model: nn.Module, chkpt_path: str, map_location=None) -> nn.Module:
    """Overlay a (possibly partial) checkpoint keyed by state-dict names (reference utils.py:31-36)."""
    merged = model.state_dict()
    try:                                          # the same opener the checkpoint was written with (training/utils.save_checkpoint):
        from smart_open import open as _open      # local paths and URLs; absent from the image -> plain files
    except ImportError:
        _open = open
    with _open(chkpt_path, 'rb') as fh:
        merged.update(torch.load(fh, map_location=map_location))
    model.load_state_dict(merged)
    return model


def mutate_transformer_config(config: TransformerConfig, depth: int, skip_alternate_cross_attn: bool) -> TransformerConfig:
    """Odd-depth decoder blocks lose their cross-attention when skip_alternate_cross_attn (reference utils.py:39-43)."""
    if config.is_cross_attn and skip_alternate_cross_attn and depth % 2:
        config = deepcopy(config)
        config.is_cross_attn = False
    return config
