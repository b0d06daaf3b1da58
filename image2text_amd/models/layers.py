"""Parameter containers for the transformer / conv blocks (reference models/layers.py).

The module tree, attribute names, parameter shapes and initial distributions match the reference so that
``state_dict()`` keys and checkpoints are interchangeable.  The arithmetic does NOT live here: it is
``engine.HotPath`` launching HIP kernels; these classes only own parameters (and refuse unsupported variants).
"""
import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..configs.models import MLPConfig, MoEConfig, SelfAttentionConfig, SelfAttentionType, TransformerConfig

_STANDALONE = ('{} is a parameter container in image2text_amd: its arithmetic runs inside the fused HIP path '
               '(VisionEncoderDecoder / Encoder / Decoder forward), not as a standalone torch module')


class _Container(nn.Module):
    def forward(self, *args, **kwargs):
        raise NotImplementedError(_STANDALONE.format(type(self).__name__))


class LayerNorm(_Container):
    """weight (+ optional bias) of a row LayerNorm, eps 1e-5 (reference layers.py:349-358)."""

    def __init__(self, ndim: int, bias: bool):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ndim))
        self.bias = nn.Parameter(torch.zeros(ndim)) if bias else None


class LayerNormND(_Container):
    """weight (+ optional bias) of a LayerNorm over the trailing ``shape`` dims jointly (reference layers.py:361-370)."""

    def __init__(self, shape: Tuple[int, ...], bias: bool):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(*shape))
        self.bias = nn.Parameter(torch.zeros(*shape)) if bias else None


class ConvMLP(_Container):
    """Conv2d('same') [GELU(tanh) Conv2d('same')]* -- Sequential slots 0,2,4,... hold the convs (layers.py:258-282)."""

    def __init__(self, in_features: int, out_features: int, kernel_size: Tuple[int, int],
                 gate_sizes: Optional[Tuple[int, ...]] = None):
        super().__init__()
        if kernel_size[0] != kernel_size[1] or kernel_size[0] not in (4, 6):
            raise NotImplementedError(f'feature_extractor_kernel_size {kernel_size}: the HIP conv kernels cover 4x4 and 6x6')
        blocks, prev = [], in_features
        for width in (gate_sizes or []):
            blocks += [nn.Conv2d(prev, width, kernel_size, padding='same'), nn.GELU(approximate='tanh')]
            prev = width
        blocks.append(nn.Conv2d(prev, out_features, kernel_size, padding='same'))
        self.model = nn.Sequential(*blocks)


class MultiHeadAttention(_Container):
    """c_attn (d -> 3d) and c_proj (d -> d) of the fused self-attention (reference layers.py:433-445)."""

    def __init__(self, config: SelfAttentionConfig):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.config = config
        self.c_attn = nn.Linear(config.n_embd, 3 * config.n_embd, bias=config.bias)
        self.c_proj = nn.Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.n_head, self.n_embd, self.dropout = config.n_head, config.n_embd, config.dropout


class MultiQueryAttention(_Container):
    """q_proj (d -> d), kv_proj (d -> 2 d/h: ONE key/value head shared by all query heads) and out_proj (d -> d)
    (reference layers.py:391-403)."""

    def __init__(self, config: SelfAttentionConfig):
        super().__init__()
        assert config.n_embd % config.n_head == 0
        self.config = config
        self.q_proj = nn.Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.kv_proj = nn.Linear(config.n_embd, 2 * config.n_embd // config.n_head, bias=config.bias)
        self.out_proj = nn.Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.n_head, self.n_embd, self.dropout = config.n_head, config.n_embd, config.dropout


class SelfAttention:
    @classmethod
    def from_config(cls, config: SelfAttentionConfig):
        if config.attn_type == SelfAttentionType.MULTI_HEAD:
            return MultiHeadAttention(config)
        if config.attn_type == SelfAttentionType.MULTI_QUERY:
            return MultiQueryAttention(config)
        raise ValueError('unknown self attn implementation!')


class MLP(_Container):
    """Linear [GELU(tanh) Linear]* -- Sequential slots 0,2,4,... hold the Linears (reference layers.py:222-255): the expert gate of
    MoELinear (no residual connector) and the per-position MLPs of AdvancedPositionalBiasMLP (identity residual connector)."""

    def __init__(self, in_features: int, out_features: int, gate_sizes: Optional[Tuple[int, ...]] = None, bias: bool = True,
                 add_residual_connection: bool = False):
        super().__init__()
        blocks, prev = [], in_features
        for width in (gate_sizes or []):
            blocks += [nn.Linear(prev, width, bias=bias), nn.GELU(approximate='tanh')]
            prev = width
        blocks.append(nn.Linear(prev, out_features, bias=bias))
        self.add_residual_connection = add_residual_connection
        self.model = nn.Sequential(*blocks)
        # a Linear when the widths differ (reference layers.py:246-250): only the PretrainedViT slot heads use that form
        self.residual_connector = nn.Linear(in_features, out_features) if (add_residual_connection and in_features != out_features) \
            else nn.Identity()


class AdvancedPositionalBiasMLP(_Container):
    """One MLP (with identity residual) per position: x_p = MLP_p(e_p) + e_p (reference layers.py:617-638); the decoder's ``wpe``
    when ``use_advanced_pos_emb`` is set."""

    def __init__(self, context_width: int, in_features: int, out_features: int, gate_sizes: Optional[Tuple[int, ...]] = None,
                 add_residual_connection: bool = True):
        super().__init__()
        self.models = nn.ModuleList([MLP(in_features, out_features, gate_sizes, bias=True, add_residual_connection=add_residual_connection)
                                     for _ in range(context_width)])


class PeerLookupQueryUnit(_Container):
    """One half of the product key: Linear(query_dim -> sqrt(num_units)), top-k of its scores (reference layers.py:20-34)."""

    def __init__(self, num_embed: int, emb_dim: int, topk: int):
        super().__init__()
        self.linear = nn.Linear(emb_dim, num_embed, bias=False)
        self.topk = topk


class PeerLookup(_Container):
    """Parameters of the product-key expert lookup (reference layers.py:37-109; arithmetic: engine_vit / csrc/vit.hip)."""

    def __init__(self, in_features: int, out_features: int, num_units: int, topk: int, nhead: int = 1, query_dim: Optional[int] = None):
        super().__init__()
        self.query_dim = query_dim or (in_features // 2)
        self.residual = nn.Linear(in_features, out_features, bias=False)
        self.query_linear = nn.Linear(in_features, self.query_dim * nhead, bias=False)
        self.key_linear = nn.Linear(in_features, in_features * nhead, bias=False)
        self.nhead, self.topk = nhead, topk
        self.num_query_units = int(math.sqrt(num_units))
        if self.num_query_units * self.num_query_units != num_units:
            raise ValueError(f"num_units must be a perfect square but {num_units} was not")
        self.query_left = PeerLookupQueryUnit(self.num_query_units, self.query_dim, topk)
        self.query_right = PeerLookupQueryUnit(self.num_query_units, self.query_dim, topk)
        self.emb_in = nn.Embedding(num_units, in_features)
        self.emb_out = nn.Embedding(num_units, out_features)


class CosineVectorEmbedding(_Container):
    """Fixed random cosine projections -> bucket ids -> EmbeddingBag(mean) (reference layers.py:112-143).  The three buffers are
    persistent (they are part of the state dict: a checkpoint carries its own projections)."""

    def __init__(self, inp_dim: int, emb_dim: int, n_proj: int = 16, num_bins: int = 20):
        super().__init__()
        self.register_buffer('projection_mat', torch.nn.functional.normalize(torch.randn((inp_dim, n_proj)), p=2.0, dim=0), persistent=True)
        resolution = 2.0 / num_bins
        self.register_buffer('grid', torch.linspace(-1, 1, num_bins + 1)[:-1] + 0.5 * resolution, persistent=True)
        self.register_buffer('pos_offset', ((num_bins + 1) * torch.arange(0, n_proj, dtype=torch.long)).long().reshape(-1, 1, 1), persistent=True)
        self.emb = nn.EmbeddingBag((num_bins + 1) * n_proj, emb_dim)
        self.emb_dim, self.n_proj, self.num_bins = emb_dim, n_proj, num_bins


class CompositeCosineVectorEmbedding(_Container):
    """Sum of CosineVectorEmbeddings at several bin resolutions (reference layers.py:190-219)."""

    def __init__(self, inp_dim: int, emb_dim: int, num_bins: Tuple[int, ...], n_proj: int, learnable: bool):
        super().__init__()
        if learnable:
            raise NotImplementedError('LearnableCosineVectorEmbedding (lsh_config.learnable: True; no shipped yaml uses it) is outside '
                                      'the HIP hot path')
        self.emb = nn.ModuleList([CosineVectorEmbedding(inp_dim=inp_dim, emb_dim=emb_dim, n_proj=n_proj, num_bins=k) for k in num_bins])


class _MoEUnit(_Container):
    """One low-rank expert: l1 (in -> proj), GELU(tanh), l2 (proj -> out), both always biased (reference layers.py:285-298)."""

    def __init__(self, in_features: int, out_features: int, proj_features: int):
        super().__init__()
        self.l1 = nn.Linear(in_features, proj_features)
        self.l2 = nn.Linear(proj_features, out_features)
        self.activation = nn.GELU(approximate='tanh')


class MoELinear(_Container):
    """softmax(gate(x) / sqrt(in)) -> top-k experts, y = sum_k w_k expert_k(x); the weights are NOT renormalised
    (reference layers.py:301-346)."""

    def __init__(self, in_features: int, out_features: int, proj_features: int, num_experts: int, bias: bool = True, top_k: int = 1,
                 gate_sizes: Optional[Tuple[int, ...]] = None):
        super().__init__()
        self._in_features, self._out_features = in_features, out_features
        self.expert_gates = MLP(in_features, num_experts, gate_sizes=gate_sizes, bias=bias)
        self.experts = nn.ModuleList([_MoEUnit(in_features, out_features, proj_features) for _ in range(num_experts)])
        self.top_k = top_k


class _MoEMLP(_Container):
    """c_fc / c_proj as MoELinear with GELU(tanh) between and dropout after (reference layers.py:489-518)."""

    def __init__(self, n_embd: int, bias: bool, dropout: float, config: MoEConfig):
        super().__init__()
        hidden = int(config.ff_mult_factor * n_embd)
        kw = dict(proj_features=config.proj_features, num_experts=config.num_experts, bias=bias, top_k=config.top_k,
                  gate_sizes=config.gate_sizes)
        self.c_fc = MoELinear(n_embd, hidden, **kw)
        self.c_proj = MoELinear(hidden, n_embd, **kw)


class _MLP(_Container):
    """c_fc (d -> ff d), c_proj (ff d -> d) of the GELU-MLP (reference layers.py:473-479)."""

    def __init__(self, n_embd: int, bias: bool, dropout: float, config: MLPConfig):
        super().__init__()
        hidden = int(config.ff_mult * n_embd)
        self.c_fc = nn.Linear(n_embd, hidden, bias=bias)
        self.c_proj = nn.Linear(hidden, n_embd, bias=bias)


class _CrossAttentionParams(_Container):
    """Parameters of nn.MultiheadAttention(batch_first) as the reference instantiates it (layers.py:537-542):
    packed in_proj (3d, d) + bias (always present) and out_proj; same names and the same default initialisation."""

    def __init__(self, embed_dim: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class TransformerBlock(_Container):
    """Pre-LN block: x += attn(ln_1 x); x += cross(ln_3 x, enc); x += mlp(ln_2 x); normalize_gradients(x).  With
    ``is_sparse_attn`` the block runs on a fixed random subset of the positions and every other position takes
    x + null_connector(x) (reference layers.py:521-614)."""

    def __init__(self, config: TransformerConfig, seed: Optional[int] = None, n_cls: int = 0):
        super().__init__()
        why = config.hot_path_unsupported_reason()
        if why is not None:
            raise NotImplementedError(f'TransformerBlock variant outside the HIP hot path: {why}')
        ac = config.attn_config
        self.is_causal = config.is_causal
        self.ln_1 = LayerNorm(ac.n_embd, bias=ac.bias)
        self.attn = SelfAttention.from_config(ac)
        self.ln_2 = LayerNorm(ac.n_embd, bias=ac.bias)
        if isinstance(config.rotator_config, MLPConfig):
            self.mlp = _MLP(ac.n_embd, ac.bias, ac.dropout, config.rotator_config)
        elif isinstance(config.rotator_config, MoEConfig):
            self.mlp = _MoEMLP(ac.n_embd, ac.bias, ac.dropout, config.rotator_config)
        else:
            raise ValueError('Unknown rotator config')
        self.is_cross_attn = config.is_cross_attn
        self.cross_attn = _CrossAttentionParams(ac.n_embd) if config.is_cross_attn else nn.Identity()
        self.ln_3 = LayerNorm(ac.n_embd, bias=ac.bias) if config.is_cross_attn else nn.Identity()
        self.is_sparse = config.is_sparse_attn
        if self.is_sparse:
            # the kept positions: the first n_cls always, then a seeded permutation of the rest (layers.py:545-559); sorted --
            # causality inside the subset depends on it
            n_non_zeros = int(config.sparsity_factor * config.max_block_size)
            gen = np.random.Generator(np.random.PCG64(seed=seed)) if seed is not None else np.random.default_rng()
            full_mask = torch.cat((torch.arange(0, n_cls),
                                   torch.tensor(gen.permutation(config.max_block_size - n_cls) + n_cls, dtype=torch.long)), dim=0)
            self.register_buffer('input_mask_idx', full_mask[:n_non_zeros].sort().values, persistent=True)
            self.register_buffer('input_mask_not_idx', full_mask[n_non_zeros:].sort().values, persistent=True)
            self.null_connector = nn.Linear(ac.n_embd, ac.n_embd, bias=ac.bias)
        else:
            self.null_connector = nn.Identity()


def init_gpt_weights_(module: nn.Module, n_layer: int):
    """nanoGPT initialisation of the decoder (reference decoder.py:192-212): N(0, 0.02) for Linear / Embedding
    weights, zero Linear biases, N(0, 0.02 / sqrt(2 n_layer)) for the residual projections ``*.c_proj.weight``."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Embedding):
            nn.init.normal_(m.weight, mean=0.0, std=0.02)
    for name, p in module.named_parameters():
        if name.endswith('c_proj.weight'):
            nn.init.normal_(p, mean=0.0, std=0.02 / math.sqrt(2 * n_layer))
