#!/usr/bin/env python3
"""One-GPU rehearsal of the data-parallel exchange over a REAL 1-rank RCCL group (run under torch.distributed.run).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/dp_selfcheck.py

Checks, on the engine-hooked path (overlap=True) against the un-hooked one (overlap=False):
  * the gradients after all_reduce_mean agree (a 1-rank mean is the identity; the dW GEMMs reduce their K slices with fp32
    atomics, so two backward passes of the same step differ in the last bits whatever the exchange does: the bound is the
    one two identical un-hooked steps meet, not bit-equality);
  * the CU reservation is in force while the decoder slice is in flight and restored afterwards;
  * a priming backward that nobody finishes and a no_sync() accumulation window leave the exchange state clean.
Prints DP_SELFCHECK_OK on success; any failure raises.  Started as a fresh child process by tests/test_dp_gpu.py (a process
that has initialised the GPU must never re-exec).
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# Two backward passes of the SAME step are not bit-identical: the gradient normaliser's sum and the dW slices are reduced with fp32
# atomics (order-dependent in the last bit), and every bf16 re-quantisation of the gradient stream turns a relative perturbation eps
# into rounding noise of about sqrt(eps x 2^-8) -- 1e-7 -> 2e-5 -> 3e-4 -> 1e-3 over three stages, saturating at bf16 precision
# (tools/diag_accum.py shows the growth entry by entry on the tiny model).  Two identical steps often DO agree to 1e-7 (same launch
# order), so a measured floor can be far below what a third pass shows; the bound below is therefore bf16-level.  The defects this
# check exists for are O(1): a slice reduced twice or not at all, a mean applied twice, a micro-batch lost.
TOL = 5e-3
# I2T_DETERMINISTIC=1 (fixed-order reductions, csrc/common.h): the same comparisons must then hold EXACTLY -- the bound above is for
# the default mode's atomics only (tests/test_round3_gpu.py::test_deterministic_mode_makes_two_backward_passes_bit_equal).
if os.environ.get('I2T_DETERMINISTIC', '0') not in ('', '0'):
    TOL = 0.0


def main():
    from image2text_amd.training.dp import RCCL_CUS, DataParallelGrads, configure_rccl_env
    configure_rccl_env()
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    # I2T_DP_SELFCHECK_PG=gloo: torch.distributed as the control plane only (what bench.py and train_loop do) -- the package's own RCCL
    # communicator is then the only one the process creates; I2T_DP_COMM=torch on top of it: the exchange falls back to a torch NCCL group
    if os.environ.get('I2T_DP_SELFCHECK_PG', 'nccl') == 'gloo':
        dist.init_process_group('gloo')
    else:
        dist.init_process_group('nccl', device_id=dev)
    from image2text_amd import ops
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch
    from image2text_amd.training.wrapper import ModelTrainerWrapper

    if os.environ.get('I2T_DP_SELFCHECK_MODEL') == 'gpt2_lora':
        # the Hugging Face GPT-2 plugin with LoRA adapters (frozen base weights inside the decoder's slice of the arena, the prefixed
        # decoder sequence): a randomly initialised checkpoint in a scratch directory, loaded through from_pretrained
        import tempfile
        from transformers import GPT2Config, GPT2LMHeadModel
        from image2text_amd.configs.models import HuggingfaceDecoderConfig, LoraSpec
        from image2text_amd.synth import tiny_config
        os.chdir(tempfile.mkdtemp(prefix='i2t_dp_'))
        torch.manual_seed(0)
        GPT2LMHeadModel(GPT2Config(n_layer=2, n_head=4, n_embd=256, n_positions=128, vocab_size=1000, resid_pdrop=0.0, embd_pdrop=0.0,
                                   attn_pdrop=0.0)).save_pretrained('gpt2-dp')
        lora = LoraSpec(r=8, lora_alpha=16, lora_dropout=0.0, target_modules=['c_attn', 'mlp.c_fc', 'mlp.c_proj'],
                        force_enable_update_modules=['*.wte.*', '*.crossattention.*'])
        dcfg = HuggingfaceDecoderConfig(vocab_size=1000, use_cross_attn=True, model_str='gpt2-dp', extra_tokens=0, load_in_4bit=False,
                                        prepare_for_kbit_training=False, lora_spec=lora)
        cfg = tiny_config(dec_d=256, dec_heads=4).model_copy(update=dict(decoder_config=dcfg, use_cross_attn=True, use_soft_prompting=True))
        V, img, cap = 1000, 32, 24
        wrapper = ModelTrainerWrapper(cfg, fake_tokenizer(V), TrainerWrapperConfig(), ignore_index=-100)
        with torch.no_grad():
            for n, p in wrapper.model.decoder.lora_params.items():
                if n.endswith('_B'):
                    p.normal_(0.0, 0.05)
    else:
        cfg = nano224_config(dropout=0.0)
        V, img, cap = cfg.decoder_config.vocab_size, 224, 64
        wrapper = ModelTrainerWrapper(cfg, fake_tokenizer(V), TrainerWrapperConfig(), ignore_index=-100)
        det_init_(wrapper.model, seed=3, style='reference')
    wrapper = wrapper.to(dev).train()
    images, labels = synthetic_batch(16, img, cap, V, seed=5)
    images, labels = images.to(dev), labels.to(dev)
    model = wrapper.model
    eng = model._engine

    def zero():
        for p in model.parameters():
            p.grad = None

    def grads():
        return eng.arena.g32.clone()

    # un-hooked reference: two identical steps give the atomic-order noise floor
    dp0 = DataParallelGrads(model, overlap=False)
    with dp0.no_sync():
        wrapper.train_step(images[:1], labels[:1])[0].backward()          # priming: builds the arena
    zero()
    dp0.broadcast_parameters()
    ref = []
    for _ in range(2):
        wrapper.train_step(images, labels)[0].backward()
        dp0.all_reduce_mean()
        ref.append(grads())
        zero()
    scale = float(ref[0].abs().max())
    floor = float((ref[0] - ref[1]).abs().max()) / scale

    # hooked path: the decoder slice is launched from inside backward
    dp1 = DataParallelGrads(model, overlap=True)
    seen = []
    eng.grad_ready_hooks.append(lambda which: seen.append((which, ops.gemm_reserved_cus(), dp1._reduced_upto)))
    wrapper.train_step(images[:1], labels[:1])[0].backward()              # priming backward, never finished ...
    zero()
    dp1.broadcast_parameters()                                            # ... must be drained here
    assert ops.gemm_reserved_cus() == 0 and dp1._reduced_upto is None and not dp1._pending
    seen.clear()
    wrapper.train_step(images, labels)[0].backward()
    at_encoder = [s for s in seen if s[0] == 'encoder']
    assert at_encoder and at_encoder[0][1] == RCCL_CUS and at_encoder[0][2] is not None, seen
    dp1.all_reduce_mean()
    assert ops.gemm_reserved_cus() == 0, 'CU reservation not restored'
    got = grads()
    zero()
    err = float((got - ref[0]).abs().max()) / scale
    assert err <= max(4 * floor, TOL), f'hooked vs un-hooked gradients: {err:.3e} (noise floor {floor:.3e})'

    # accumulation window: micro-batch 1 under no_sync, micro-batch 2 exchanges; equals the un-hooked accumulation
    with dp1.no_sync():
        wrapper.train_step(images[:8], labels[:8])[0].backward()
    assert not dp1._pending and ops.gemm_reserved_cus() == 0
    wrapper.train_step(images[8:], labels[8:])[0].backward()
    dp1.all_reduce_mean()
    acc1 = grads()
    zero()
    eng.grad_ready_hooks.clear()
    wrapper.train_step(images[:8], labels[:8])[0].backward()
    wrapper.train_step(images[8:], labels[8:])[0].backward()
    acc0 = grads()
    err2 = float((acc1 - acc0).abs().max()) / float(acc0.abs().max())
    if err2 > max(4 * floor, TOL):            # name the arena entries that differ (diagnostic for the assertion below)
        worst = sorted(((float((acc1[o:o + n] - acc0[o:o + n]).abs().max()), name) for name, (o, n, _) in eng.arena.entries.items()), reverse=True)[:6]
        print('DP_SELFCHECK accumulation mismatch, worst entries:', worst, 'max |g| =', float(acc0.abs().max()), flush=True)
    assert err2 <= max(4 * floor, TOL), f'accumulation window: {err2:.3e} (noise floor {floor:.3e})'
    # transport: the package's own RCCL communicator behind the C ABI (i2t_comm_*), fp32 on the wire; its bf16 wire form rounds once
    if os.environ.get('I2T_DP_COMM', 'rccl') == 'torch':      # the fallback transport was asked for: a torch NCCL group even under a gloo control plane
        assert dp1.comm is None
        fb = dp1._cuda_group()
        print(f'DP_SELFCHECK transport=torch-{dist.get_backend(fb)}', flush=True)
        assert dist.get_backend(fb) == 'nccl'
        print(f'DP_SELFCHECK deterministic={int(TOL == 0.0)} tol={TOL}', flush=True)
        print(f'DP_SELFCHECK_OK floor={floor:.2e} hooked={err:.2e} accumulate={err2:.2e} nchannels={os.environ.get("NCCL_MAX_NCHANNELS")}', flush=True)
        dp1.close()
        dist.destroy_process_group()
        return
    assert dp1.comm is not None, 'the C-ABI RCCL communicator was not created (torch.distributed fallback in use)'
    t = torch.randn(1 << 16, device=dev)
    want32, want16 = t.clone(), t.bfloat16().float()
    dp1.comm.all_reduce_mean_async(t).wait()
    torch.cuda.synchronize()
    assert torch.equal(t, want32), 'fp32 wire: a 1-rank mean must be the identity'
    dp1.comm.wire_bf16 = True
    dp1.comm.all_reduce_mean_async(t).wait()
    torch.cuda.synchronize()
    assert torch.equal(t, want16), 'bf16 wire: a 1-rank mean must be the bf16 rounding'
    dp1.comm.wire_bf16 = False
    spans = dp1._spans(eng.arena, 0, eng.arena.total)
    frozen = sum(n for name, (o, n, _) in eng.arena.entries.items() if name in eng.arena.params and not eng.arena.trainable(name))
    sent = sum(b - a for a, b in spans)
    assert sent <= eng.arena.total and (frozen == 0 or sent < eng.arena.total), (sent, frozen, eng.arena.total)
    torch.cuda.synchronize()
    print(f'DP_SELFCHECK transport=rccl-abi spans={len(spans)} floats_sent={sent} of {eng.arena.total} (frozen {frozen})', flush=True)
    print(f'DP_SELFCHECK deterministic={int(TOL == 0.0)} tol={TOL}', flush=True)
    print(f'DP_SELFCHECK_OK floor={floor:.2e} hooked={err:.2e} accumulate={err2:.2e} nchannels={os.environ.get("NCCL_MAX_NCHANNELS")}', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
