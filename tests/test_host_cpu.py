"""Host-side logic that needs no GPU: config schema parses the reference's YAML presets, state-dict keys, factories
refuse out-of-scope families, the drop-in aliasing, and the product path fails loudly without a GPU."""
import os

import pytest
import torch
import yaml

from image2text_amd.configs.models import PretrainedViTConfig, TransformerDecoderConfig, VisionTransformerEncoderConfig
from image2text_amd.configs.trainer import TrainerWrapperConfig, TrainingConfig
from image2text_amd.lib import I2TError
from image2text_amd.models.decoder import Decoder
from image2text_amd.models.encoder import Encoder
from image2text_amd.models.utils import PatternMatcher, mutate_transformer_config
from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
from image2text_amd.synth import fake_tokenizer, nano224_config, synthetic_batch, tiny_config

NANO_YAML = """
tokenizer_str: 'gpt2'
trainer: {}
optimizers:
  - lr: 6e-4
    betas: [0.9, 0.95]
    target_modules: ['decoder*.transformer.h.*.cross_attn.*', 'decoder*.transformer.h.*.ln_3.*']
batch_size: 8
gradient_accumulation_steps: 4
precision: 'no'
model:
  use_cross_attn: True
  use_soft_prompting: True
  no_repeat_n_grams: [2, 3, 4, 5]
  vision_encoder_config:
    n_embd_out_vit: 768
    n_cls: 8
    refine_base_model: False
    enable_gradient_checkpointing: True
  decoder_config:
    pretrained_model: gpt2
    n_layer: 12
    block_size: 256
    vocab_size: 50257
    transformer_config:
      is_cross_attn: True
      is_causal: True
      attn_config: {attn_dropout: 0.1, bias: True, dropout: 0.1, n_head: 12, n_embd: 768, attn_type: multi_head}
      rotator_config: {ff_mult: 4}
"""


def test_yaml_schema_parses_like_the_reference(monkeypatch):
    monkeypatch.setenv('I2T_VIT_B16_CHECKPOINT', 'random')
    cfg = TrainingConfig.model_validate(yaml.safe_load(NANO_YAML))
    assert isinstance(cfg.model.vision_encoder_config, PretrainedViTConfig)      # unknown key silently ignored
    assert isinstance(cfg.model.decoder_config, TransformerDecoderConfig)
    assert cfg.model.decoder_config.pretrained_model.value == 'gpt2'
    assert cfg.optimizers[0].betas == (0.9, 0.95) and cfg.trainer == TrainerWrapperConfig()
    enc = Encoder.from_config(cfg.model.vision_encoder_config)                    # torchvision ViT-B/16 + 8 slot MLPs (engine_vit)
    assert enc.num_outputs == 8 and enc.output_embed_dim == 768 and not enc.refine
    assert sum(p.numel() for n, p in enc.named_parameters() if n.startswith('model.')) == 85_798_656      # torchvision vit_b_16 without its head
    monkeypatch.delenv('I2T_VIT_B16_CHECKPOINT')
    monkeypatch.setattr(torch.hub, 'get_dir', lambda: '/nonexistent')
    with pytest.raises(FileNotFoundError):       # no checkpoint, no silent random backbone
        Encoder.from_config(cfg.model.vision_encoder_config)
    with pytest.raises(AssertionError):          # pretrained_model: gpt2 with block_size 256 and loose = False (reference decoder.py:61)
        Decoder.from_config(cfg.model.decoder_config)


def test_nano224_state_dict_layout_and_param_groups():
    m = VisionEncoderDecoder(nano224_config())
    sd = m.state_dict()
    assert sum(p.numel() for p in m.parameters()) == 161_759_384
    for k in ('encoder.0.cls_token', 'encoder.0.feature_extractor.model.4.weight', 'encoder.0.ln_input.weight',
              'encoder.0.transformer.h.5.mlp.c_proj.weight', 'encoder.1.weight', 'decoder.transformer.wte.weight',
              'decoder.transformer.h.0.cross_attn.in_proj_weight', 'decoder.transformer.h.10.ln_3.bias', 'decoder.lm_head.weight'):
        assert k in sd, k
    assert 'decoder.transformer.h.1.cross_attn.in_proj_weight' not in sd          # cross-attention on even depths only
    assert 'encoder.0.projector.bias' not in sd                                   # bias: False in the encoder
    assert sd['decoder.lm_head.weight'].data_ptr() == sd['decoder.transformer.wte.weight'].data_ptr()
    matcher = PatternMatcher(['decoder*.transformer.h.*.cross_attn.*', 'decoder*.transformer.h.*.ln_3.*'])
    picked = [n for n, _ in m.named_parameters() if matcher.match(n)]
    assert len(picked) == 6 * 4 + 6 * 2
    assert m.space_for_prompt == 64 and m.decoder.block_size == 256 and m.decoder.n_embd == 768
    assert m.encoder[0].num_outputs == 64 and m.encoder[0].output_embed_dim == 512


def test_misconfiguration_errors_match_the_reference():
    with pytest.raises(ValueError):
        VisionEncoderDecoder(tiny_config(use_cross_attn=False, use_soft_prompting=False))
    cfg = tiny_config()
    tc = cfg.decoder_config.transformer_config
    assert mutate_transformer_config(tc, 1, True).is_cross_attn is False and mutate_transformer_config(tc, 2, True).is_cross_attn is True
    assert tc.is_cross_attn is True                                               # original left untouched (deepcopy)


def test_no_cpu_path_fails_loudly():
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    cfg = tiny_config()
    w = ModelTrainerWrapper(cfg, fake_tokenizer(cfg.decoder_config.vocab_size), TrainerWrapperConfig())
    images, labels = synthetic_batch(2, 32, 16, cfg.decoder_config.vocab_size)
    with pytest.raises(I2TError):
        w.train_step(images, labels)
    with pytest.raises(I2TError):
        w.model(images=images, ids=labels.clamp(min=0))
    with pytest.raises(NotImplementedError):
        w.model.decoder.transformer.h[0](torch.zeros(1, 4, 128))                  # blocks are parameter containers


def test_trainer_options_are_wired_or_refused():
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    cfg = tiny_config()
    wc = ModelTrainerWrapper(cfg, fake_tokenizer(384), TrainerWrapperConfig(add_contrastive_loss=True, training_contrastive_temperature=0.7))
    assert wc.add_contrastive_loss and wc.contrastive_temperature == 0.7          # wrapper.py:35,42
    with pytest.raises(ValueError):                         # MLM corruption needs a mask token (trainer.py:124-125 adds one)
        ModelTrainerWrapper(cfg, fake_tokenizer(384), TrainerWrapperConfig(mask_fraction=0.15))
    # momentum distillation builds the twin and starts it from the model's weights (wrapper.py:30-33,46-50)
    w = ModelTrainerWrapper(cfg, fake_tokenizer(384), TrainerWrapperConfig(moco_momentum=0.995, moco_alpha=0.4))
    assert w.is_momentum and w.model_m is not None
    for (n1, p1), (n2, p2) in zip(w.model.named_parameters(), w.model_m.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    assert sum(1 for n, _ in w.named_parameters() if n.startswith('model_m.')) == sum(1 for n, _ in w.named_parameters() if n.startswith('model.'))


def test_loss_weights_match_the_oracle():
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    from oracle import reference_model as orc
    cfg = tiny_config()
    _, labels = synthetic_batch(5, 32, 16, 384, seed=4)
    for kw in (dict(), dict(weight_fn='inverse_sqrt_position'), dict(eos_token_weight=3.0)):
        w = ModelTrainerWrapper(cfg, fake_tokenizer(384), TrainerWrapperConfig(**kw))
        ref = orc.loss_weights(labels, -100, kw.get('weight_fn', 'constant'), 383, kw.get('eos_token_weight'))
        assert torch.allclose(w.get_weights(labels), ref)


# the first-party imports of the reference's trainer.py (trainer.py:10-15) and of the modules it pulls in -- kept here as DATA
# (module, names) so that the test also runs where /root/reference does not exist; when it does, the import block itself is
# parsed out of trainer.py and executed
TRAINER_IMPORTS = [('models.optimizer', ['SNRAdam']), ('configs.trainer', ['TrainingConfig']),
                   ('configs.models', ['PretrainedViTConfig']), ('training.wrapper', ['ModelTrainerWrapper']),
                   ('training.utils', ['train_loop', 'val_loop', 'WrapperDataLoader']), ('models.utils', ['PatternMatcher']),
                   ('models.vision_encoder_decoder', ['VisionEncoderDecoder']), ('models.generation_utils', ['BeamSearchTokenGenerator']),
                   ('models.encoder', ['Encoder']), ('models.decoder', ['Decoder']), ('models.layers', ['TransformerBlock']),
                   ('models.functions', ['normalize_gradients']), ('object_models', ['VisionEncoderDecoderModelOutput']),
                   ('training.utils', ['normalize_label', 'unpack_batch'])]


def test_dropin_resolves_every_first_party_import_of_trainer_py():
    import ast
    import importlib
    import sys
    import types
    import image2text_amd.dropin as dropin
    names = list(dropin._ALIASES) + ['deeplake', 'torchvision', 'torchvision.models', 'torchvision.transforms']
    saved = {k: sys.modules.get(k) for k in names}
    try:
        dropin.install()
        for mod, attrs in TRAINER_IMPORTS:
            m = importlib.import_module(mod)
            assert m.__name__.startswith('image2text_amd.'), (mod, m.__name__)
            for a in attrs:
                assert hasattr(m, a), f'{mod}.{a}'
        from models import optimizer as o2                                          # noqa: package-attribute form
        from models.vision_encoder_decoder import VisionEncoderDecoder as V2        # noqa
        from configs.trainer import TrainingConfig as T2                            # noqa
        assert V2 is VisionEncoderDecoder and T2 is TrainingConfig and o2.SNRAdam is not None
        ref_trainer = '/root/reference/trainer.py'
        if os.path.exists(ref_trainer):
            # network-only third-party packages trainer.py imports are absent here: name-only stand-ins
            for name, attrs in (('deeplake', ['load', 'Dataset']), ('torchvision', []), ('torchvision.models', ['ViT_B_16_Weights']),
                                ('torchvision.transforms', [])):
                if name not in sys.modules or sys.modules[name] is None:
                    stub = types.ModuleType(name)
                    stub.__spec__ = importlib.machinery.ModuleSpec(name, None)
                    for a in attrs:
                        setattr(stub, a, object)
                    sys.modules[name] = stub
            sys.modules['torchvision'].models = sys.modules['torchvision.models']
            sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
            tree = ast.parse(open(ref_trainer).read())
            block = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
            assert len(block) >= 10
            ns = {}
            exec(compile(ast.Module(body=block, type_ignores=[]), ref_trainer, 'exec'), ns)       # the import block, verbatim
            assert ns['SNRAdam'].__module__.startswith('image2text_amd.')
            assert ns['train_loop'].__module__ == 'image2text_amd.training.utils'
            assert ns['ModelTrainerWrapper'].__module__ == 'image2text_amd.training.wrapper'
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_label_normalisation_and_caption_fanout_loader():
    """normalize_label keeps ids through the first pad position (= the EOS to predict) and ignores the rest; WrapperDataLoader
    yields every (image, caption) pair exactly once per epoch in batch_size pieces (reference training/utils.py:16-60)."""
    from image2text_amd.training.utils import WrapperDataLoader, normalize_label
    ids = torch.tensor([[5, 6, 7, 9, 9, 9], [1, 2, 3, 4, 5, 6], [9, 9, 9, 9, 9, 9]])
    att = torch.tensor([[1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0]])
    lab = normalize_label(ids, att, -100)
    assert lab.tolist() == [[5, 6, 7, 9, -100, -100], [1, 2, 3, 4, 5, 6], [9, -100, -100, -100, -100, -100]]
    torch.manual_seed(0)
    batches = []
    for b in range(2):
        d = {'image': torch.arange(3).float().view(3, 1, 1, 1) + 10 * b}
        for k in range(5):
            d[f'input_ids_{k}'] = torch.full((3, 4), 100 * b + 10 * k) + torch.arange(3).view(3, 1)
            d[f'attn_mask_{k}'] = torch.ones(3, 4, dtype=torch.long)
        batches.append(d)
    dl = WrapperDataLoader(batches, batch_size=4, ignore_idx=-100, epochs=2)
    assert len(dl) == 10
    out = list(dl)
    assert [x[0].shape[0] for x in out] == [4, 4, 4, 3] * 4                 # 15 pairs per loader batch, 2 batches, 2 epochs
    pairs = sorted((float(im.view(-1)[i]), int(lb[i, 0])) for im, lb in out[:8] for i in range(im.shape[0]))
    want = sorted((float(img + 10 * b), 100 * b + 10 * k + img) for b in range(2) for k in range(5) for img in range(3))
    assert pairs == want


def test_dropout_rule_statistics():
    """The counter-based dropout rule (rng.py = csrc/common.h: four 8-bit uniforms per lowbias32 hash): drop rate = thr8 / 256
    within sampling error (and within 1/512 of the requested p), the four decisions drawn from one hash are pairwise
    uncorrelated, so are neighbouring hashes, elements a row apart in the layouts the kernels use, and the masks of different
    sites; the scale makes the mask mean-preserving.  (A one-multiply mixer fails this screen by hundreds of sigma at lags
    4, 8, 16, 64, 260: the two multiplies of lowbias32 are what the decisions need.)"""
    import math
    from image2text_amd import rng
    n = 1 << 20
    for p in (0.1, 0.05, 0.5):
        thr = rng.threshold(p)
        p_eff = thr / 256.0
        assert abs(p_eff - p) <= 1 / 512 + 1e-9
        k0, k1 = rng.site_key(12345, 7), rng.site_key(12345, 8)
        m0, m1 = rng.keep_mask(k0, n, thr).double(), rng.keep_mask(k1, n, thr).double()
        sigma = math.sqrt(p_eff * (1 - p_eff) / n)
        assert abs((1 - m0.mean().item()) - p_eff) < 5 * sigma
        assert abs(m0.mean().item() * rng.scale(thr) - 1.0) < 5 * sigma * rng.scale(thr)
        d0, d1 = 1 - m0, 1 - m1

        def corr(a, b):
            return ((a * b).mean().item() - a.mean().item() * b.mean().item()) / (p_eff * (1 - p_eff))
        lim = 5 / math.sqrt(n / 4)
        for i in range(4):                                   # the four byte fields of one hash, pairwise
            for j in range(i + 1, 4):
                assert abs(corr(d0[i::4], d0[j::4])) < lim, (p, i, j)
        assert abs(corr(d0[3:-1:4], d0[4::4])) < lim        # last decision of one hash / first of the next
        for lag in (4, 8, 16, 64, 260, 768, 4096):           # a row apart in the layouts the kernels index (Tk 64 / 260, d 768 ...)
            assert abs(corr(d0[:-lag], d0[lag:])) < 5 / math.sqrt(n - lag), (p, lag)
        assert abs(corr(d0, d1)) < 5 / math.sqrt(n)         # two sites of the same step
        assert torch.equal(rng.keep_mask(k0, 64, thr, offset=37), rng.keep_mask(k0, 101, thr)[37:])
    assert rng.site_key(1, 2) != rng.site_key(2, 1) and rng.site_key(1 << 40, 3) != rng.site_key(0, 3)


def _tiny_hf_gpt2(monkeypatch):
    """a randomly initialised 2-layer GPT-2 standing in for the checkpoint GPT2LMHeadModel.from_pretrained would fetch"""
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(3)
    hf = GPT2LMHeadModel(GPT2Config(n_layer=2, n_head=2, n_embd=128, n_positions=64, vocab_size=384, resid_pdrop=0.0, embd_pdrop=0.0,
                                    attn_pdrop=0.0, bos_token_id=383, eos_token_id=383)).eval()
    monkeypatch.setattr(GPT2LMHeadModel, 'from_pretrained', classmethod(lambda cls, name, **kw: hf))
    return hf


def _local_hf_gpt2(tmp_path, monkeypatch, name='gpt2-tiny', **kw):
    """a randomly initialised 2-layer GPT-2 saved under ./gpt2-tiny: what HuggingfaceDecoderConfig(model_str='gpt2-tiny') loads
    through transformers' own from_pretrained (a local directory: no network, no monkeypatching of the loader)"""
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(5)
    args = dict(n_layer=2, n_head=2, n_embd=128, n_positions=64, vocab_size=380, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                bos_token_id=379, eos_token_id=379)
    args.update(kw)
    hf = GPT2LMHeadModel(GPT2Config(**args)).eval()
    with torch.no_grad():                                  # biases and LayerNorm parameters away from their 0 / 1 initialisation
        for n_, p_ in hf.named_parameters():
            if n_.endswith('.bias') or '.ln_' in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    hf.save_pretrained(str(tmp_path / name))
    monkeypatch.chdir(tmp_path)
    return hf


def _hf_decoder_config(name='gpt2-tiny', **kw):
    from image2text_amd.configs.models import HuggingfaceDecoderConfig
    args = dict(vocab_size=380, use_cross_attn=False, model_str=name, extra_tokens=4, load_in_4bit=False, prepare_for_kbit_training=False)
    args.update(kw)
    return HuggingfaceDecoderConfig(**args)


def test_gpt2_huggingface_decoder_plugin(tmp_path, monkeypatch):
    """HuggingfaceDecoderConfig(model_str='gpt2*') (reference decoder.py:119-121, 285-382): the checkpoint's weights land in the hot
    path's decoder, embeddings resized by extra_tokens, the state dict speaks Hugging Face's names and Conv1D layout in both
    directions (a transformers GPT2LMHeadModel loads it strictly), cross-attention layers appear on request; everything else of
    the family is refused by name"""
    from image2text_amd.configs.models import LoraSpec
    from image2text_amd.models.decoder import GPT2HuggingfaceDecoder
    from transformers import GPT2Config, GPT2LMHeadModel
    hf = _local_hf_gpt2(tmp_path, monkeypatch)
    d = Decoder.from_config(_hf_decoder_config(), space_for_prompt=3)
    assert isinstance(d, GPT2HuggingfaceDecoder) and d.n_embd == 128 and d.block_size == 64 and not d.use_cross_attn
    sd, sh = d.state_dict(), hf.state_dict()
    assert set(sd) == {'backbone.' + k for k in sh}
    for k, v in sh.items():
        mine = sd['backbone.' + k]
        assert mine.shape[1:] == v.shape[1:] and mine.shape[0] == (384 if k in ('transformer.wte.weight', 'lm_head.weight') else v.shape[0]), k
        assert torch.equal(mine[:v.shape[0]], v), k                       # rows 380..383 are transformers' fresh rows
    assert d.lm_head.weight is d.transformer.wte.weight and d.get_inputs_embeds(torch.tensor([[1, 2]])).shape == (1, 2, 128)
    # internal layout: nn.Linear [out, in] (what the GEMMs read); the Conv1D transposes happen at the state-dict boundary
    assert torch.equal(d.transformer.h[1].mlp.c_fc.weight, sh['transformer.h.1.mlp.c_fc.weight'].t())
    # ... and back: transformers loads our state dict strictly, we load transformers' strictly
    hf2 = GPT2LMHeadModel(GPT2Config(n_layer=2, n_head=2, n_embd=128, n_positions=64, vocab_size=384))
    hf2.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items()}, strict=True)
    d2 = Decoder.from_config(_hf_decoder_config())
    with torch.no_grad():
        for p_ in d2.parameters():
            p_.zero_()
    d2.load_state_dict(sd, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(d.parameters(), d2.parameters()))
    # cross-attention on request: Hugging Face's q_attn / c_attn / c_proj / ln_cross_attn <-> in_proj / out_proj / ln_3
    dc = Decoder.from_config(_hf_decoder_config(use_cross_attn=True))
    sdc = dc.state_dict()
    blk = dc.transformer.h[0]
    assert dc.use_cross_attn and dc.hot_config.transformer_config.is_cross_attn and not dc.hot_config.skip_alternate_cross_attn
    assert torch.equal(sdc['backbone.transformer.h.0.crossattention.q_attn.weight'], blk.cross_attn.in_proj_weight[:128].t())
    assert torch.equal(sdc['backbone.transformer.h.0.crossattention.c_attn.weight'], blk.cross_attn.in_proj_weight[128:].t())
    assert torch.equal(sdc['backbone.transformer.h.0.crossattention.c_attn.bias'], blk.cross_attn.in_proj_bias[128:])
    assert torch.equal(sdc['backbone.transformer.h.1.crossattention.c_proj.weight'], dc.transformer.h[1].cross_attn.out_proj.weight.t())
    assert torch.equal(sdc['backbone.transformer.h.1.ln_cross_attn.weight'], dc.transformer.h[1].ln_3.weight)
    hf3 = GPT2LMHeadModel(GPT2Config(n_layer=2, n_head=2, n_embd=128, n_positions=64, vocab_size=384, add_cross_attention=True))
    hf3.load_state_dict({k[len('backbone.'):]: v for k, v in sdc.items()}, strict=True)
    assert float(blk.cross_attn.in_proj_weight.detach().std()) > 0.01               # initialised by transformers (normal, 0.02), not left at zero
    # inside the full model the keys sit under decoder.backbone.* as in the reference
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from image2text_amd.synth import tiny_config
    cfg = tiny_config(dec_d=128, dec_heads=2, dec_layers=2, block_size=64).model_copy(update=dict(decoder_config=_hf_decoder_config(use_cross_attn=True)))
    m = VisionEncoderDecoder(cfg)
    keys = [k for k in m.state_dict() if k.startswith('decoder.')]
    assert keys and all(k.startswith('decoder.backbone.') for k in keys)
    assert m._engine.dec.V == 384 and m._engine.dec_cross == [True, True] and m._engine.cross_inputs
    m_soft = VisionEncoderDecoder(cfg.model_copy(update=dict(decoder_config=_hf_decoder_config())))
    assert not m_soft._engine.cross_inputs                                  # model asks for cross-attention, the decoder has none: dropped
    # refusals
    with pytest.raises(NotImplementedError, match='4-bit'):
        Decoder.from_config(_hf_decoder_config(load_in_4bit=True))
    with pytest.raises(NotImplementedError, match='free-form'):
        Decoder.from_config(_hf_decoder_config(name='mistralai/Mistral-7B', vocab_size=32000))
    hf_relu = _local_hf_gpt2(tmp_path, monkeypatch, name='gpt2-relu', activation_function='relu')
    with pytest.raises(NotImplementedError, match='activation'):
        Decoder.from_config(_hf_decoder_config(name='gpt2-relu'))


def _lora_spec(**kw):
    from image2text_amd.configs.models import LoraSpec
    args = dict(r=4, lora_alpha=16, lora_dropout=0.0, target_modules=['c_attn', 'mlp.c_fc', 'mlp.c_proj'],
                force_enable_update_modules=['*.wpe.*', '*.wte.*', '*.crossattention.*', '*.ln_cross_attn.*'])      # gpu/gpt2-xl.yaml:55-60
    args.update(kw)
    return LoraSpec(**args)


def test_gpt2_huggingface_decoder_lora(tmp_path, monkeypatch):
    """lora_spec on the GPT-2 plugin (reference decoder.py:133-134 -> models/utils.py:46-65 -> peft LoraModel): adapters on the modules
    peft's suffix rule selects (c_attn = self- AND cross-attention's), lora_B zero, everything else of the decoder frozen, the
    force-enable patterns matched against the LoraModel's parameter names, and a state dict in the LoraModel's keys"""
    from image2text_amd.engine import _arena_order
    _local_hf_gpt2(tmp_path, monkeypatch)
    d = Decoder.from_config(_hf_decoder_config(use_cross_attn=True, lora_spec=_lora_spec()))
    assert d.lora.sites == ('attn_c_attn', 'xattn_c_attn', 'mlp_c_fc', 'mlp_c_proj') and d.lora.scale == 4.0 and d.lora.r == 4
    sd = d.state_dict()
    p_ = 'backbone.model.transformer.h.1.'
    shapes = {p_ + 'attn.c_attn.base_layer.weight': (128, 384), p_ + 'attn.c_attn.base_layer.bias': (384,), p_ + 'attn.c_attn.lora_A.default.weight': (4, 128),
              p_ + 'attn.c_attn.lora_B.default.weight': (384, 4), p_ + 'crossattention.c_attn.base_layer.weight': (128, 256),
              p_ + 'crossattention.c_attn.lora_A.default.weight': (4, 128), p_ + 'crossattention.c_attn.lora_B.default.weight': (256, 4),
              p_ + 'crossattention.q_attn.weight': (128, 128), p_ + 'mlp.c_fc.base_layer.weight': (128, 512), p_ + 'mlp.c_fc.lora_B.default.weight': (512, 4),
              p_ + 'mlp.c_proj.lora_A.default.weight': (4, 512), p_ + 'mlp.c_proj.base_layer.bias': (128,), p_ + 'attn.c_proj.weight': (128, 128),
              p_ + 'ln_cross_attn.weight': (128,), 'backbone.model.transformer.wte.weight': (384, 128), 'backbone.model.lm_head.weight': (384, 128)}
    for k, shp in shapes.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    assert not any(k.startswith('backbone.transformer.') or 'lora_params' in k for k in sd)
    assert float(sd[p_ + 'attn.c_attn.lora_B.default.weight'].abs().max()) == 0.0 and float(sd[p_ + 'attn.c_attn.lora_A.default.weight'].std()) > 0.01
    grads = {n: p.requires_grad for n, p in d.named_parameters()}
    on = ('lora_params.h0_attn_c_attn_A', 'lora_params.h1_mlp_c_proj_B', 'transformer.wte.weight', 'transformer.wpe.weight',
          'transformer.h.0.cross_attn.in_proj_weight', 'transformer.h.0.cross_attn.out_proj.bias', 'transformer.h.1.ln_3.weight')
    off = ('transformer.h.0.attn.c_attn.weight', 'transformer.h.0.attn.c_attn.bias', 'transformer.h.0.attn.c_proj.weight',
           'transformer.h.1.mlp.c_fc.weight', 'transformer.h.1.ln_1.weight', 'transformer.ln_f.bias')
    assert all(grads[n] for n in on) and not any(grads[n] for n in off)
    # the adapters' A matrices are followed by zero pad rows in the arena (rank -> 128: the adapter GEMMs' K / N panel)
    order = _arena_order(list(d.named_parameters()))
    i = [n for n, *_ in order].index('lora_params.h0_attn_c_attn_A')
    assert order[i + 1][0] == 'lora_params.h0_attn_c_attn_A.<pad>' and order[i + 1][1] is None and order[i + 1][2] == 124 * 128
    # round trip in the LoraModel's keys; a checkpoint of the un-adapted model (plain transformers keys) loads too (loose)
    d2 = Decoder.from_config(_hf_decoder_config(use_cross_attn=True, lora_spec=_lora_spec()))
    d2.load_state_dict(sd, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(d.parameters(), d2.parameters()))
    plain = Decoder.from_config(_hf_decoder_config(use_cross_attn=True)).state_dict()
    missing, unexpected = d2.load_state_dict(plain, strict=False)
    assert not unexpected and all('lora_params' in k for k in missing)
    # fnmatch patterns written against the reference's parameter names select the hot path's parameters (optimizer target_modules,
    # checkpoint matchers): the plugin registers the translation, PatternMatcher tries both names
    from image2text_amd.models.utils import PatternMatcher, reference_names
    names = ['model.decoder.' + n for n, _ in d.named_parameters()]
    pick = lambda pats: [n for n in names if PatternMatcher(pats).match(n)]
    assert pick(['*.crossattention.*']) == [n for n in names if '.cross_attn.' in n or 'xattn_c_attn' in n]
    assert pick(['*decoder*lora*']) == [n for n in names if 'lora_params' in n] and len(pick(['*lora_A*'])) == 8
    assert pick(['*.ln_cross_attn.*']) == [n for n in names if '.ln_3.' in n] and pick(['*.attn.c_attn.base_layer.weight']) == [
        'model.decoder.transformer.h.0.attn.c_attn.weight', 'model.decoder.transformer.h.1.attn.c_attn.weight']
    assert {'model.decoder.backbone.model.transformer.h.0.crossattention.q_attn.weight',
            'model.decoder.backbone.model.transformer.h.0.crossattention.c_attn.base_layer.weight'} <= set(
        reference_names('model.decoder.transformer.h.0.cross_attn.in_proj_weight'))
    from image2text_amd.models.utils import state_dict_keys_of_parameters
    holder = torch.nn.Module()
    holder.decoder = d
    keys = state_dict_keys_of_parameters(holder)
    assert sorted(r for refs in keys.values() for r in refs) == sorted(k for k in holder.state_dict() if k != 'decoder.backbone.model.lm_head.weight')
    # partial checkpoint of what the patterns select (training/utils.py::save_checkpoint): stored under the reference's keys, loadable
    from types import SimpleNamespace
    from image2text_amd.training.utils import save_checkpoint
    from image2text_amd.models.utils import update_state_dict_from_partial_checkpoint
    ck = str(tmp_path / 'partial.pt')
    save_checkpoint(holder, ck, SimpleNamespace(save=torch.save), matchers=[PatternMatcher(['*lora_B*', '*.crossattention.q_attn.*'])])
    part = torch.load(ck, weights_only=True)
    assert sorted(part) == sorted(k for k in holder.state_dict() if 'lora_B' in k or '.crossattention.' in k and '.lora_' not in k and '.c_proj.' not in k)
    holder2 = torch.nn.Module()
    holder2.decoder = Decoder.from_config(_hf_decoder_config(use_cross_attn=True, lora_spec=_lora_spec()))
    with torch.no_grad():
        for n_, p_ in d.named_parameters():
            if 'lora_params' in n_ and n_.endswith('_B'):
                p_.fill_(0.5)
        save_checkpoint(holder, ck, SimpleNamespace(save=torch.save), matchers=[PatternMatcher(['*lora_B*'])])
    update_state_dict_from_partial_checkpoint(holder2, ck)
    assert all(float(p_.min()) == 0.5 for n_, p_ in holder2.decoder.named_parameters() if 'lora_params' in n_ and n_.endswith('_B'))
    # no force-enable list: adapters only
    d3 = Decoder.from_config(_hf_decoder_config(use_cross_attn=True, lora_spec=_lora_spec(force_enable_update_modules=None, target_modules=None)))
    assert d3.lora.sites == ('attn_c_attn', 'xattn_c_attn')                  # peft's default target for gpt2: c_attn
    assert [n for n, p in d3.named_parameters() if p.requires_grad] == [n for n, _ in d3.named_parameters() if n.startswith('lora_params.')]
    # prepare_for_kbit_training without 4-bit loading (local/llama2-7b.yaml): peft's call freezes the base model -- and nothing else
    d4 = Decoder.from_config(_hf_decoder_config(prepare_for_kbit_training=True))
    assert not any(p.requires_grad for p in d4.parameters())
    d5 = Decoder.from_config(_hf_decoder_config(use_cross_attn=True, prepare_for_kbit_training=True, lora_spec=_lora_spec()))
    assert {n: p.requires_grad for n, p in d5.named_parameters()} == grads
    with pytest.raises(NotImplementedError, match='target_modules'):
        Decoder.from_config(_hf_decoder_config(lora_spec=_lora_spec(target_modules=['attn.c_proj'])))
    with pytest.raises(NotImplementedError, match='splits'):
        Decoder.from_config(_hf_decoder_config(use_cross_attn=True, lora_spec=_lora_spec(force_enable_update_modules=['*.q_attn.*'])))
    with pytest.raises(NotImplementedError, match='rank'):
        Decoder.from_config(_hf_decoder_config(lora_spec=_lora_spec(r=200)))


def _local_hf_llama(tmp_path, monkeypatch, kind='llama'):
    """a randomly initialised 2-layer Llama-2 / Qwen2 checkpoint in a local directory whose name satisfies the reference's dispatch
    (decoder.py:124-127, 404-440: 'meta-llama/Llama-2*' with vocab >= 32000, '*Qwen*' with vocab >= 151936)"""
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(7)
    if kind == 'llama':     # 4 query heads of 64 on 2 K/V heads, no biases, untied head
        name, vocab = 'meta-llama/Llama-2-tiny', 32000
        hf = LlamaForCausalLM(LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                          num_key_value_heads=2, vocab_size=vocab, max_position_embeddings=128, rms_norm_eps=1e-5))
    else:                   # 2 query heads of 128 on 1 K/V head, q / k / v biases, tied head
        name, vocab = 'Qwen-tiny', 151936
        hf = Qwen2ForCausalLM(Qwen2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                                          num_key_value_heads=1, vocab_size=vocab, max_position_embeddings=128, tie_word_embeddings=True))
    with torch.no_grad():
        for n_, p_ in hf.named_parameters():
            if n_.endswith('.bias') or 'norm' in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    hf.save_pretrained(str(tmp_path / name))
    monkeypatch.chdir(tmp_path)
    return hf.eval(), name, vocab


def test_llama_qwen2_huggingface_decoder_plugins(tmp_path, monkeypatch):
    """Llama2HuggingfaceDecoder / Qwen2HuggingfaceDecoder (reference decoder.py:124-127, 404-440): the transformers module is the
    parameter container (state-dict keys = the reference's), the arena lays q | k | v (and their biases) and gate | up next to each
    other so that the fused projections are single views, the rotary table holds transformers' own cos / sin values"""
    from image2text_amd.engine import _arena_order
    from image2text_amd.models.decoder import Llama2HuggingfaceDecoder, Qwen2HuggingfaceDecoder
    for kind, cls, extra in (('llama', Llama2HuggingfaceDecoder, 4), ('qwen', Qwen2HuggingfaceDecoder, 0)):
        hf, name, vocab = _local_hf_llama(tmp_path, monkeypatch, kind)
        d = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra))
        assert isinstance(d, cls) and d.n_embd == 256 and d.block_size == (4096 if kind == 'llama' else 128)
        ls = d.llama_spec
        assert (ls.H, ls.Hkv, ls.hd, ls.L, ls.ff, ls.V) == ((4, 2, 64) if kind == 'llama' else (2, 1, 128)) + (2, 512, vocab + extra)
        assert ls.qkv_bias == (kind == 'qwen') and ls.tied == (kind == 'qwen')
        sd, sh = d.state_dict(), hf.state_dict()
        assert set(sd) == {'backbone.' + k for k in sh}
        assert all(torch.equal(sd['backbone.' + k][:v.shape[0]], v) for k, v in sh.items())
        assert d.get_inputs_embeds(torch.tensor([[1, 2]])).shape == (1, 2, 256)
        order = [n for n, *_ in _arena_order(list(d.named_parameters()))]
        i = order.index('backbone.model.layers.1.self_attn.q_proj.weight')
        want = [f'backbone.model.layers.1.self_attn.{x}_proj.weight' for x in 'qkv']
        if kind == 'qwen':
            want += [f'backbone.model.layers.1.self_attn.{x}_proj.bias' for x in 'qkv']
        assert order[i:i + len(want)] == want
        j = order.index('backbone.model.layers.0.mlp.gate_proj.weight')
        assert order[j + 1] == 'backbone.model.layers.0.mlp.up_proj.weight' and sorted(order) == sorted(n for n, _ in d.named_parameters())
        # rotary table against transformers' module: [cos | sin] halves
        tab = d.rope_table(16)
        cos, sin = hf.model.rotary_emb(torch.zeros(1, 1), torch.arange(16)[None])
        assert tab.shape == (16, ls.hd) and torch.equal(tab[:, :ls.hd // 2], cos[0, :, :ls.hd // 2]) and torch.equal(tab[:, ls.hd // 2:], sin[0, :, ls.hd // 2:])
        frozen = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra, prepare_for_kbit_training=True))
        assert not any(p.requires_grad for p in frozen.parameters())
        with pytest.raises(ValueError, match='cross attention'):
            Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra, use_cross_attn=True))
    with pytest.raises(AssertionError):
        Decoder.from_config(_hf_decoder_config(name='meta-llama/Llama-2-tiny', vocab_size=1000))          # 'vocab should not shrink' (decoder.py:407)


def _local_hf_falcon(tmp_path, monkeypatch, **cfg_kw):
    """a randomly initialised 2-layer checkpoint of the falcon-7b architecture (parallel attention + MLP, multi-query, rotary, no
    biases) in a local directory whose name satisfies the reference's dispatch (decoder.py:122-123, 385-386: 'tiiuae/falcon*', vocab >= 65024)"""
    from transformers import FalconConfig, FalconForCausalLM
    torch.manual_seed(11)
    name, vocab = 'tiiuae/falcon-tiny', 65024
    args = dict(hidden_size=256, num_attention_heads=4, num_hidden_layers=2, vocab_size=vocab, multi_query=True, parallel_attn=True,
                new_decoder_architecture=False, bias=False, alibi=False, max_position_embeddings=128)
    args.update(cfg_kw)
    hf = FalconForCausalLM(FalconConfig(**args))
    with torch.no_grad():
        for n_, p_ in hf.named_parameters():
            if 'layernorm' in n_ or 'ln_f' in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    hf.save_pretrained(str(tmp_path / name))
    monkeypatch.chdir(tmp_path)
    return hf.eval(), name, vocab


def test_falcon_huggingface_decoder_plugin(tmp_path, monkeypatch):
    """FalconHuggingfaceDecoder (reference decoder.py:122-123, 383-400): the transformers module is the parameter container, the spec
    the engine reads describes the falcon-7b block, LoRA on the targets of gpu/falcon-7b.yaml:55-60 with its force-enable patterns, the
    LoraModel-keyed state dict; architectures of the family the hot path does not run are refused by name."""
    from image2text_amd.configs.models import LoraSpec
    from image2text_amd.models.decoder import FalconHuggingfaceDecoder
    hf, name, vocab = _local_hf_falcon(tmp_path, monkeypatch)
    d = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=1))
    assert isinstance(d, FalconHuggingfaceDecoder) and d.n_embd == 256 and d.block_size == 2048
    ls = d.llama_spec
    assert (ls.arch, ls.H, ls.Hkv, ls.hd, ls.L, ls.ff, ls.V, ls.tied) == ('falcon', 4, 1, 64, 2, 1024, vocab + 1, True)
    sd, sh = d.state_dict(), hf.state_dict()
    assert set(sd) == {'backbone.' + k for k in sh} and all(torch.equal(sd['backbone.' + k][:v.shape[0]], v) for k, v in sh.items())
    assert d.get_inputs_embeds(torch.tensor([[1, 2]])).shape == (1, 2, 256)           # (the reference raises AttributeError here)
    tab = d.rope_table(16)
    cos, sin = hf.transformer.rotary_emb(torch.zeros(1, 1), torch.arange(16)[None])
    assert tab.shape == (16, 64) and torch.equal(tab[:, :32], cos[0, :, :32]) and torch.equal(tab[:, 32:], sin[0, :, 32:])
    spec = LoraSpec(r=4, lora_alpha=16, lora_dropout=0.1, target_modules=['query_key_value', 'dense', 'dense_h_to_4h', 'dense_4h_to_h'],
                    force_enable_update_modules=['*.word_embeddings.*', '*.lm_head.*'])
    dl = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=1, lora_spec=spec))
    assert dl.lora.sites == ('qkv', 'o', 'gu', 'dn')
    sdl = dl.state_dict()
    p_ = 'backbone.model.transformer.h.1.'
    for k, shp in {p_ + 'self_attention.query_key_value.base_layer.weight': (384, 256), p_ + 'self_attention.query_key_value.lora_A.default.weight': (4, 256),
                   p_ + 'self_attention.query_key_value.lora_B.default.weight': (384, 4), p_ + 'self_attention.dense.lora_B.default.weight': (256, 4),
                   p_ + 'mlp.dense_h_to_4h.lora_B.default.weight': (1024, 4), p_ + 'mlp.dense_4h_to_h.lora_A.default.weight': (4, 1024),
                   p_ + 'input_layernorm.bias': (256,), 'backbone.model.transformer.word_embeddings.weight': (vocab + 1, 256),
                   'backbone.model.lm_head.weight': (vocab + 1, 256)}.items():
        assert k in sdl and tuple(sdl[k].shape) == shp, k
    on = {n for n, p in dl.named_parameters() if p.requires_grad}
    assert on == {n for n, _ in dl.named_parameters() if n.startswith('lora_params.')} | {'backbone.transformer.word_embeddings.weight'}
    d2 = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=1, lora_spec=spec))
    d2.load_state_dict(sdl, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(dl.parameters(), d2.parameters()))
    missing, unexpected = d2.load_state_dict(sd, strict=False)
    assert not unexpected and missing and all('lora_' in k for k in missing)
    with pytest.raises(ValueError, match='cross attention'):
        Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, use_cross_attn=True))
    with pytest.raises(NotImplementedError, match='4-bit'):              # gpu/falcon-7b.yaml loads in 4 bits (bitsandbytes: not in this image)
        Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, load_in_4bit=True))
    # ... unless the caller opts into the fp8 stand-in: the decoder_config of gpu/falcon-7b.yaml:46-60 as shipped (4-bit, kbit preparation,
    # LoRA on the four linears, embeddings / head left trainable)
    monkeypatch.setenv('I2T_4BIT_AS_FP8', '1')
    d4 = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=1, load_in_4bit=True, prepare_for_kbit_training=True,
                                                 enable_gradient_checkpointing=True, lora_spec=spec))
    assert d4.fp8_request and d4.lora is not None
    on4 = {n for n, p in d4.named_parameters() if p.requires_grad}
    assert on4 == {n for n, _ in d4.named_parameters() if n.startswith('lora_params.')} | {'backbone.transformer.word_embeddings.weight'}
    monkeypatch.delenv('I2T_4BIT_AS_FP8')
    _local_hf_falcon(tmp_path, monkeypatch, alibi=True)
    with pytest.raises(NotImplementedError, match='alibi'):
        Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab))


def test_llama_qwen2_huggingface_decoder_lora(tmp_path, monkeypatch):
    """lora_spec on the Llama-2 / Qwen2 plugins (reference gpu/llama2-13b.yaml:35-39; decoder.py:133-134 -> models/utils.py:46-65): adapters
    on the block linears peft's suffix rule selects, the adapters of a fused projection stacked in one lora_A parameter, everything else
    frozen, state dict in the LoraModel's per-module keys (backbone.model.model.layers...base_layer / lora_A.default / lora_B.default)."""
    from image2text_amd.configs.models import LoraSpec
    from image2text_amd.engine import _arena_order
    targets = ['q_proj', 'k_proj', 'v_proj', 'o_proj', 'up_proj', 'down_proj']
    for kind, extra in (('llama', 4), ('qwen', 0)):
        hf, name, vocab = _local_hf_llama(tmp_path, monkeypatch, kind)
        mk = lambda **kw: Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra, lora_spec=LoraSpec(
            **{**dict(r=4, lora_alpha=16, lora_dropout=0.1, target_modules=targets), **kw})))
        d = mk()
        ls = d.llama_spec
        assert d.lora.sites == ('qkv', 'o', 'gu', 'dn') and d.lora.members['gu'] == ['mlp.up_proj'] and d.lora.scale == 4.0
        sd = d.state_dict()
        p_ = 'backbone.model.model.layers.1.'
        kv = ls.Hkv * ls.hd
        shapes = {p_ + 'self_attn.q_proj.base_layer.weight': (256, 256), p_ + 'self_attn.q_proj.lora_A.default.weight': (4, 256),
                  p_ + 'self_attn.k_proj.lora_A.default.weight': (4, 256), p_ + 'self_attn.k_proj.lora_B.default.weight': (kv, 4),
                  p_ + 'self_attn.v_proj.lora_B.default.weight': (kv, 4), p_ + 'self_attn.o_proj.lora_A.default.weight': (4, 256),
                  p_ + 'mlp.up_proj.lora_B.default.weight': (512, 4), p_ + 'mlp.gate_proj.weight': (512, 256),
                  p_ + 'mlp.down_proj.lora_A.default.weight': (4, 512), p_ + 'mlp.down_proj.base_layer.weight': (256, 512),
                  p_ + 'input_layernorm.weight': (256,), 'backbone.model.model.embed_tokens.weight': (vocab + extra, 256),
                  'backbone.model.lm_head.weight': (vocab + extra, 256)}
        if kind == 'qwen':
            shapes[p_ + 'self_attn.q_proj.base_layer.bias'] = (256,)
        for k, shp in shapes.items():
            assert k in sd and tuple(sd[k].shape) == shp, k
        assert not any('lora_params' in k or k.startswith('backbone.model.layers.') for k in sd)
        assert set(k for k in sd if '.lora_' not in k) == {'backbone.model.' + k.replace('_proj.', '_proj.base_layer.') if 'gate_proj' not in k
                                                           else 'backbone.model.' + k for k in hf.state_dict()}
        # the stacked lora_A: q rows, then k rows, then v rows
        A = d.lora_params['h1_qkv_A']
        assert A.shape == (12, 256) and torch.equal(sd[p_ + 'self_attn.k_proj.lora_A.default.weight'], A[4:8])
        assert float(d.lora_params['h0_q_B'].abs().max()) == 0.0 and float(A.std()) > 0.01
        grads = {n: p.requires_grad for n, p in d.named_parameters()}
        assert all(v == n.startswith('lora_params.') for n, v in grads.items())
        order = _arena_order(list(d.named_parameters()))
        i = [n for n, *_ in order].index('lora_params.h0_qkv_A')
        assert order[i + 1][0] == 'lora_params.h0_qkv_A.<pad>' and order[i + 1][2] == (128 - 12) * 256
        # round trips: the LoraModel's keys (strict), and an un-adapted checkpoint in transformers' keys
        d2 = mk()
        d2.load_state_dict(sd, strict=True)
        assert all(torch.equal(a, b) for a, b in zip(d.parameters(), d2.parameters()))
        plain = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra)).state_dict()
        missing, unexpected = d2.load_state_dict(plain, strict=False)
        assert not unexpected and missing and all('lora_' in k for k in missing)
        # force_enable_update_modules is matched against the LoraModel's names; the defaults are peft's (q_proj, v_proj)
        d3 = mk(force_enable_update_modules=['*.embed_tokens.*', '*.q_proj.base_layer.*', '*norm*'])
        on = {n for n, p in d3.named_parameters() if p.requires_grad and not n.startswith('lora_params.')}
        assert 'backbone.model.embed_tokens.weight' in on and 'backbone.model.layers.0.self_attn.q_proj.weight' in on
        assert 'backbone.model.norm.weight' in on and 'backbone.model.layers.0.self_attn.k_proj.weight' not in on
        d4 = mk(target_modules=None)
        assert d4.lora.sites == ('qkv',) and d4.lora.members['qkv'] == ['self_attn.q_proj', 'self_attn.v_proj']
        assert d4.lora_params['h0_qkv_A'].shape == (8, 256) and 'h0_k_B' not in d4.lora_params
        from image2text_amd.models.utils import PatternMatcher
        names = ['model.decoder.' + n for n, _ in d.named_parameters()]
        pick = lambda pats: [n for n in names if PatternMatcher(pats).match(n)]
        assert pick(['*.k_proj.lora_A.*']) == ['model.decoder.lora_params.h0_qkv_A', 'model.decoder.lora_params.h1_qkv_A']
        assert pick(['*.o_proj.base_layer.*']) == [f'model.decoder.backbone.model.layers.{l}.self_attn.o_proj.weight' for l in (0, 1)]
        with pytest.raises(NotImplementedError, match='target_modules'):
            mk(target_modules=['q_proj', 'lm_head'])
        with pytest.raises(NotImplementedError, match='rank'):
            mk(r=64)


def _gpt2_decoder_config(**kw):
    from image2text_amd.configs.models import MLPConfig, ModelType, SelfAttentionConfig, SelfAttentionType, TransformerConfig
    tc = TransformerConfig(rotator_config=MLPConfig(ff_mult=4), is_causal=True, is_cross_attn=True,
                           attn_config=SelfAttentionConfig(attn_dropout=0.0, bias=True, dropout=0.0, n_head=2, n_embd=128,
                                                           attn_type=SelfAttentionType.MULTI_HEAD))
    args = dict(transformer_config=tc, n_layer=2, block_size=64, vocab_size=384, pretrained_model=ModelType.GPT2)
    args.update(kw)
    return TransformerDecoderConfig(**args)


def test_gpt2_weight_import(monkeypatch):
    """pretrained_model: gpt2 (reference decoder.py:45-117): Conv1D weights transposed on the way in, embeddings tied, keys the
    checkpoint lacks (cross-attention, ln_3) left at their initialisation; strict mode insists on GPT-2's own shapes"""
    hf = _tiny_hf_gpt2(monkeypatch)
    d = Decoder.from_config(_gpt2_decoder_config(), loose=True)
    sd, sh = d.state_dict(), hf.state_dict()
    for k in ('attn.c_attn.weight', 'attn.c_proj.weight', 'mlp.c_fc.weight', 'mlp.c_proj.weight'):
        assert torch.equal(sd[f'transformer.h.1.{k}'], sh[f'transformer.h.1.{k}'].t())
    for k in ('transformer.wpe.weight', 'transformer.h.0.ln_1.weight', 'transformer.h.0.attn.c_attn.bias', 'transformer.ln_f.bias'):
        assert torch.equal(sd[k], sh[k])
    assert torch.equal(sd['lm_head.weight'], sh['transformer.wte.weight']) and d.lm_head.weight is d.transformer.wte.weight
    assert 'transformer.h.0.cross_attn.in_proj_weight' in sd and 'transformer.h.0.cross_attn.in_proj_weight' not in sh
    with pytest.raises(AssertionError):
        Decoder.from_config(_gpt2_decoder_config(), loose=False)              # 2 x 128 is not GPT-2's 12 x 768


# the vision_encoder_config blocks of the reference's seven PretrainedViT presets (training_configs/*/*.yaml), as data
SHIPPED_VIT_PRESETS = {
    'local/nano.yaml': dict(n_embd_out_vit=768, n_cls=8, gate_sizes=[1024], refine_base_model=False,
                            lsh_config=dict(num_bins=[4, 8, 20], num_proj=32, learnable=False)),
    'gpu/nano.yaml': dict(n_embd_out_vit=1600, n_cls=8, refine_base_model=False,
                          peer_config=dict(num_units_sqrt=256, topk=8, nhead=4, query_dim=128)),
    'local/nano-mini.yaml': dict(n_embd_out_vit=768, n_cls=16, gate_sizes=[1024], refine_base_model=False),
    'local/gpt2.yaml': dict(n_embd_out_vit=768, n_cls=16, gate_sizes=[1024], refine_base_model=True),
    'local/llama2-7b.yaml': dict(n_embd_out_vit=4096, n_cls=16, gate_sizes=[2048], refine_base_model=False),
    'local/qwen-1.5b-deepseek-distill.yaml': dict(n_embd_out_vit=4096, n_cls=16, gate_sizes=[2048], refine_base_model=False),
    'gpu/llama2-13b.yaml': dict(n_embd_out_vit=5120, n_cls=16, gate_sizes=[2560], refine_base_model=False),
}


@pytest.mark.parametrize('preset', list(SHIPPED_VIT_PRESETS))
def test_every_shipped_pretrained_vit_preset_builds(preset, monkeypatch):
    """SURVEY 8 f3 / VERDICT r2 item 1: the encoder of 7 of the reference's 11 yamls.  Each builds (random backbone: the SWAG checkpoint
    is not in the image), exposes the reference's state-dict keys, and -- when the reference's files are present -- the yaml itself
    parses into the same encoder config."""
    monkeypatch.setenv('I2T_VIT_B16_CHECKPOINT', 'random')
    kw = SHIPPED_VIT_PRESETS[preset]
    cfg = PretrainedViTConfig.model_validate(kw)
    path = os.path.join('/root/reference/training_configs', preset)
    if os.path.exists(path):
        with open(path) as fh:
            shipped = TrainingConfig.model_validate(yaml.safe_load(fh)).model.vision_encoder_config
        assert isinstance(shipped, PretrainedViTConfig) and shipped == cfg
    enc = Encoder.from_config(cfg)
    keys = set(enc.state_dict())
    assert {'model.conv_proj.weight', 'model.class_token', 'model.encoder.pos_embedding', 'model.encoder.ln.bias', 'peer_proj_wt',
            'model.encoder.layers.encoder_layer_11.self_attention.in_proj_weight', 'model.encoder.layers.encoder_layer_0.mlp.3.bias'} <= keys
    assert enc.num_outputs == kw['n_cls'] and enc.output_embed_dim == kw['n_embd_out_vit']
    if 'peer_config' in kw:
        assert enc.state_dict()['peer.emb_out.weight'].shape == (256 * 256, 1600) and enc.state_dict()['peer_proj_wt'].shape == (768, 768, 8)
    elif 'lsh_config' in kw:
        assert enc.state_dict()['lsh_emb.7.emb.2.emb.weight'].shape == (21 * 32, 768) and not enc.refine
        assert enc.state_dict()['lsh_emb.0.emb.0.projection_mat'].shape == (768, 32)
        matcher = PatternMatcher(['encoder*.lsh_emb.*'])                               # the yaml's own optimizer pattern (local/nano.yaml:9)
        assert sum(matcher.match('encoder.' + n) for n, _ in enc.named_parameters()) == 8 * 3
    else:
        last = kw['n_cls'] - 1
        assert enc.state_dict()[f'proj.models.{last}.model.2.weight'].shape == (kw['n_embd_out_vit'], kw['gate_sizes'][0])
        assert (f'proj.models.{last}.residual_connector.weight' in keys) == (kw['n_embd_out_vit'] != 768)
