"""Small host-side helpers of the plugin surface (reference models/utils.py:18-43)."""
import fnmatch
from copy import deepcopy
from typing import List, Optional

import torch
import torch.nn as nn

from ..configs.models import TransformerConfig


class PatternMatcher:
    """fnmatch over parameter names; an empty/None pattern list matches everything (reference utils.py:18-28)."""

    def __init__(self, patterns: Optional[List[str]]):
        self.patterns = patterns

    def match(self, candidate: str) -> bool:
        if not self.patterns:
            return True
        return any(fnmatch.fnmatch(candidate, pat) for pat in self.patterns)


def update_state_dict_from_partial_checkpoint(model: nn.Module, chkpt_path: str, map_location=None) -> nn.Module:
    """Overlay a (possibly partial) checkpoint keyed by state-dict names (reference utils.py:31-36)."""
    merged = model.state_dict()
    with open(chkpt_path, 'rb') as fh:
        merged.update(torch.load(fh, map_location=map_location))
    model.load_state_dict(merged)
    return model


def mutate_transformer_config(config: TransformerConfig, depth: int, skip_alternate_cross_attn: bool) -> TransformerConfig:
    """Odd-depth decoder blocks lose their cross-attention when skip_alternate_cross_attn (reference utils.py:39-43)."""
    if config.is_cross_attn and skip_alternate_cross_attn and depth % 2:
        config = deepcopy(config)
        config.is_cross_attn = False
    return config
