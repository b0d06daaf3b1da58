"""Model-side configuration schema.

Mirror of the reference's YAML/pydantic surface (reference: configs/models.py:9-135) so that every shipped
``training_configs/*.yaml`` parses into these classes unchanged.  The schema is plain data: class and field
names and their defaults are the on-disk format and therefore identical to the reference; everything else
(validation helpers, ``hot_path_supported``) is ours.

The from-scratch families (dense nano-224 and the multi-query / MoE / sparse nano-mini family) are executed by the
HIP hot path; the pretrained families parse fine but ``Encoder.from_config`` /
``Decoder.from_config`` refuse them loudly.
"""
from enum import Enum
from typing import List, Optional, Tuple, Union

from pydantic import BaseModel


class LoraSpec(BaseModel):
    """LoRA adapter request (reference configs/models.py:9-14).  Executed by ``engine_lora.LoraAdapters`` on the Hugging Face decoder plugins
    (GPT-2, Llama-2 / Qwen2, Falcon); refused by name on the GPT-2-imported nanoGPT decoder and on the encoder."""
    r: int = 16
    lora_alpha: int = 64
    lora_dropout: float = 0.1
    target_modules: Optional[List[str]] = None
    force_enable_update_modules: Optional[List[str]] = None


class MLPConfig(BaseModel):
    """Dense GELU-MLP rotator: d -> ff_mult*d -> d (reference configs/models.py:17-18)."""
    ff_mult: float


class MoEConfig(BaseModel):
    """Mixture-of-experts rotator (reference configs/models.py:21-26)."""
    num_experts: int
    proj_features: int
    ff_mult_factor: float
    gate_sizes: Optional[Tuple[int, ...]] = None
    top_k: int = 1


class SelfAttentionType(Enum):
    MULTI_HEAD = 'multi_head'
    MULTI_QUERY = 'multi_query'


class SelfAttentionConfig(BaseModel):
    """reference configs/models.py:34-40"""
    attn_dropout: float = 0.1
    bias: bool = True
    dropout: float = 0.1
    n_head: int = 12
    n_embd: int = 768
    attn_type: SelfAttentionType


class TransformerConfig(BaseModel):
    """reference configs/models.py:43-50"""
    rotator_config: Union[MoEConfig, MLPConfig]
    is_causal: bool = False
    is_cross_attn: bool = False
    max_block_size: Optional[int] = None
    is_sparse_attn: bool = False
    sparsity_factor: float = 0.5
    attn_config: SelfAttentionConfig

    def hot_path_unsupported_reason(self) -> Optional[str]:
        """None when the HIP path can run this block, else a human-readable reason."""
        ac = self.attn_config
        if ac.n_embd % ac.n_head != 0:
            return 'n_embd not divisible by n_head'
        hd = ac.n_embd // ac.n_head
        if hd not in (16, 32, 64, 128):
            return 'head_dim must be 16, 32, 64 or 128'
        if self.is_sparse_attn and self.max_block_size is None:
            return 'need to specify max_block_size for sparse attention'       # reference layers.py:547
        if isinstance(self.rotator_config, MoEConfig):
            rc = self.rotator_config
            if rc.top_k > rc.num_experts:
                return 'top_k > num_experts'
            if rc.num_experts * (rc.proj_features + 1) > 256:
                return 'num_experts * (proj_features + 1) > 256 (the fused expert GEMM holds every expert in one K panel)'
        return None

    @property
    def is_family(self) -> bool:
        """True for blocks that run on the generalised (nano-mini family) block path instead of the dense one."""
        return (self.attn_config.attn_type != SelfAttentionType.MULTI_HEAD or self.is_sparse_attn
                or not isinstance(self.rotator_config, MLPConfig) or self.attn_config.n_embd != 64 * self.attn_config.n_head)


class ImageInputSpec(BaseModel):
    n_channels: int = 3
    width: int
    height: int


class LshConfig(BaseModel):
    num_bins: Tuple[int, ...]
    num_proj: int
    learnable: bool


class PeerConfig(BaseModel):
    num_units_sqrt: int
    topk: int
    nhead: int
    query_dim: Optional[int] = None


class EncoderConfig(BaseModel):
    n_cls: int
    lora_spec: Optional[LoraSpec] = None


class VisionTransformerEncoderConfig(EncoderConfig):
    """From-scratch ViT (reference configs/models.py:80-88, models/encoder.py:130-195)."""
    transformer_config: TransformerConfig
    enable_gradient_checkpointing: bool = False
    input: ImageInputSpec
    n_layer: int = 12
    num_patches: int
    n_channels: int
    feature_extractor_gate_sizes: Optional[Tuple[int, ...]] = None
    feature_extractor_kernel_size: Tuple[int, int] = (4, 4)


class PretrainedViTConfig(EncoderConfig):
    """torchvision ViT-B/16 backbone + heads (reference configs/models.py:91-96): ``models/encoder.py::PretrainedViT`` on
    ``engine_vit.ViTEncoder`` (slot-MLP / PEER / LSH heads; learnable LSH projections are refused)."""
    refine_base_model: bool = True
    n_embd_out_vit: int
    peer_config: Optional[PeerConfig] = None
    lsh_config: Optional[LshConfig] = None
    gate_sizes: Optional[Tuple[int, ...]] = None


class ModelType(Enum):
    GPT2 = 'gpt2'
    GPT2_MEDIUM = 'gpt2-medium'
    GPT2_LARGE = 'gpt2-large'
    GPT2_XL = 'gpt2-xl'


class DecoderConfig(BaseModel):
    lora_spec: Optional[LoraSpec] = None
    enable_gradient_checkpointing: bool = False
    vocab_size: int


class TransformerDecoderConfig(DecoderConfig):
    """nanoGPT-style decoder (reference configs/models.py:112-119, models/decoder.py:161-282)."""
    transformer_config: TransformerConfig
    use_advanced_pos_emb: bool = False
    advanced_pos_emb_gate_sizes: Optional[Tuple[int, ...]] = None
    pretrained_model: Optional[ModelType] = None
    n_layer: int
    skip_alternate_cross_attn: bool = True
    block_size: int


class HuggingfaceDecoderConfig(DecoderConfig):
    """HF causal-LM decoders (reference configs/models.py:122-128): the GPT-2, Llama-2 / Qwen2 and Falcon plugins of ``models/decoder.py``
    run on the hot path (``engine.py``, ``engine_llama.py``); ``load_in_4bit`` (bitsandbytes NF4) is refused unless I2T_4BIT_AS_FP8=1."""
    use_cross_attn: bool
    model_str: str
    extra_tokens: int
    load_in_4bit: bool
    prepare_for_kbit_training: bool
    use_auth_token: bool = False


class VisionEncoderDecoderConfig(BaseModel):
    """reference configs/models.py:131-138"""
    vision_encoder_config: Union[VisionTransformerEncoderConfig, PretrainedViTConfig]
    decoder_config: Union[TransformerDecoderConfig, HuggingfaceDecoderConfig]
    loose_match_decoder_state_dict: bool = False
    chkpt_path: Optional[str] = None
    use_cross_attn: bool = False
    use_soft_prompting: bool = True
    no_repeat_n_grams: Tuple[int, ...] = (2, 3, 4, 5)
