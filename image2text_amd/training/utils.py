"""Training / validation loops and the caption-fan-out loader behind ``trainer.py`` (reference training/utils.py).

Same names, arguments and return values as the reference (`normalize_label` :16-20, `unpack_batch` :23-37,
`WrapperDataLoader` :40-60, `train_loop` :63-123, `val_loop` :126-164), so ``trainer.py`` calls them unchanged.  The
``accelerator`` argument is used through the same attributes the reference touches (``device``, ``autocast``,
``accumulate``, ``backward``, ``sync_gradients``, ``gather``, ``save``, ``unwrap_model``, ``wait_for_everyone``,
``is_local_main_process``), so a real ``accelerate.Accelerator`` or any object with those members works.

What differs, on purpose (MI355X-first):
  * **gradient exchange**: the reference relies on DDP, whose reducer never fires with its call pattern (SURVEY.md
    section 5).  Here, when ``torch.distributed`` is initialised with more than one rank, the loop drives
    ``training.dp.DataParallelGrads`` itself: micro-batches of an accumulation window run under ``no_sync()``, the last
    one starts the decoder slice's RCCL all-reduce inside its backward and ``all_reduce_mean()`` finishes the exchange
    before ``optimizer.step()``.  A DDP wrapper around the model is unwrapped (its hooks never see the HIP path's
    gradients, which are written straight into the flat arena).
  * **metrics without a per-step host sync**: the reference calls ``.cpu().item()`` on every metric after every step
    (:99-102), which drains the HIP stream once per step.  Here each step's metrics are copied to pinned host memory
    asynchronously behind an event and *reported one step late* (progress bar, ``logging_callback`` -- the callback still
    receives the right ``batch`` index); only the last step of the loop waits.
  * ``disable_flash`` is accepted and ignored: there is no SDPA kernel selection, attention is always the hand-written
    HIP kernel.
"""
from contextlib import nullcontext
from typing import Iterator, List, Optional, Tuple, Union

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.utils.data

from ..models.utils import PatternMatcher
from .wrapper import ModelTrainerWrapper

try:                                          # the reference writes checkpoints through smart_open (local paths and URLs);
    from smart_open import open as _open      # absent from the image -> plain files
except ImportError:                           # pragma: no cover - depends on the environment
    _open = open

try:
    from tqdm.auto import tqdm
except ImportError:                           # pragma: no cover
    tqdm = None


def normalize_label(input_ids, attn_mask, ignore_index):
    """Labels of one caption: the ids up to AND INCLUDING the first pad position (the tokenizer pads with EOS, so that
    position is the EOS the model must learn), ``ignore_index`` after it (reference :16-20)."""
    width = attn_mask.size(-1)
    last = attn_mask.sum(dim=-1).clamp(0, width - 1).unsqueeze(-1)            # index of the first pad = number of real tokens
    keep = torch.arange(width, device=last.device).unsqueeze(0) <= last
    return torch.where(keep, input_ids, torch.full_like(input_ids, ignore_index))


def unpack_batch(batch, ignore_index: int = -100):
    """A Flickr30K loader batch (one image, five tokenised captions) -> (images, labels_0, ..., labels_4) (reference :23-37)."""
    labels = tuple(normalize_label(batch[f'input_ids_{k}'], batch[f'attn_mask_{k}'], ignore_index) for k in range(5))
    return (batch['image'],) + labels


class WrapperDataLoader:
    """Fans every image out to its five captions, shuffles the 5x batch and yields it in ``batch_size`` pieces, for
    ``epochs`` passes over the underlying loader (reference :40-60)."""

    def __init__(self, dataloader: torch.utils.data.DataLoader, batch_size: int, ignore_idx: int, epochs: int):
        self.dataloader = dataloader
        self.batch_size = batch_size
        self.ignore_idx = ignore_idx
        self.epochs = epochs

    def __len__(self):
        return 5 * len(self.dataloader)

    def __iter__(self):
        for _ in range(self.epochs):
            for batch in self.dataloader:
                images, *caps = unpack_batch(batch, ignore_index=self.ignore_idx)
                images = torch.cat([images] * len(caps), dim=0)
                labels = torch.cat(caps, dim=0)
                order = torch.randperm(images.size(0))
                images, labels = images[order], labels[order]
                yield from zip(torch.split(images, self.batch_size, dim=0), torch.split(labels, self.batch_size, dim=0))


# ------------------------------------------------------------------------------------------------------------ helpers
def _unwrap(model_wrapper) -> ModelTrainerWrapper:
    return model_wrapper.module if isinstance(model_wrapper, nn.parallel.DistributedDataParallel) else model_wrapper


def _exchange_for(wrapper: ModelTrainerWrapper):
    """The data-parallel gradient exchange of this wrapper's model (created on first use: parameters are broadcast from rank 0
    then, as DDP does at wrap time); None in a single-process run."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return None
    dp = getattr(wrapper, '_i2t_dp', None)
    if dp is None:
        from .dp import DataParallelGrads
        wrapper.model._engine.prepare(False)                  # the flat arenas must exist before they can be broadcast
        dp = DataParallelGrads(wrapper.model)
        dp.broadcast_parameters(0)
        object.__setattr__(wrapper, '_i2t_dp', dp)
    return dp


class _LateMetrics:
    """Per-step metrics copied to pinned host memory behind an event; ``pop`` returns the PREVIOUS step's values as floats."""

    def __init__(self):
        self._slot = None          # (step, {name: pinned tensor}, event)

    def push(self, step: int, metrics):
        prev = self.pop()
        host, ev = {}, None
        for k, v in metrics.items():
            v = v.detach()
            if v.is_cuda:
                buf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                buf.copy_(v, non_blocking=True)
                host[k] = buf
            else:
                host[k] = v
        if any(v.is_cuda for v in metrics.values()):
            ev = torch.cuda.Event()
            ev.record()
        self._slot = (step, host, ev)
        return prev

    def pop(self):
        if self._slot is None:
            return None
        step, host, ev = self._slot
        self._slot = None
        if ev is not None:
            ev.synchronize()       # recorded a whole step ago (or the loop is ending): normally already complete
        return step, {k: float(v) for k, v in host.items()}


def _bar(n, accelerator):
    if tqdm is None:
        return nullcontext(range(n))
    return tqdm(range(n), unit='batch', disable=not getattr(accelerator, 'is_local_main_process', True))


def save_checkpoint(model: nn.Module, chckpt_fname: str, accelerator, matchers: List[PatternMatcher] = ()):
    """Full state dict, or -- when optimizer target patterns exist -- only the parameters some pattern selects (fine-tuning a
    large model stores just what was trained; reference :111-123).  Loadable by ``update_state_dict_from_partial_checkpoint``."""
    sd = model.state_dict()
    if len(matchers) > 0:
        # (a parameter's state-dict key(s) are its reference name(s): the same string except under a GPT2HuggingfaceDecoder)
        from ..models.utils import state_dict_keys_of_parameters
        keys = state_dict_keys_of_parameters(model)
        sd = {r: sd[r] for k, _ in model.named_parameters() if any(m.match(k) for m in matchers) for r in keys[k]}
    with _open(chckpt_fname, mode='wb') as fh:
        accelerator.save(sd, fh)


# -------------------------------------------------------------------------------------------------------------- loops
def train_loop(model_wrapper: Union[nn.parallel.DistributedDataParallel, ModelTrainerWrapper],
               optimizer: torch.optim.Optimizer,
               train_iter: Iterator[Tuple[torch.Tensor, torch.Tensor]],
               epoch: int,
               num_steps: Optional[int],
               accelerator,
               disable_flash: bool = False,
               reset_moco_after_k_epochs: Optional[List[int]] = None,
               logging_callback=None,
               chckpt_fname=None,
               matchers: List[PatternMatcher] = []):
    """One epoch of at most ``num_steps`` (default 100) steps; returns True when the iterator ran dry (reference :63-123)."""
    model_wrapper.train()
    wrapper = _unwrap(model_wrapper)
    dp = _exchange_for(wrapper)
    device = accelerator.device
    main = getattr(accelerator, 'is_local_main_process', True)
    stop = False
    late = _LateMetrics()

    def report(item, bar):
        if item is None:
            return
        step, vals = item
        if hasattr(bar, 'set_postfix'):
            bar.set_postfix(**vals)
        if main and logging_callback is not None:
            logging_callback(vals, batch=step, epoch=epoch)

    with _bar(100 if num_steps is None else num_steps, accelerator) as bar:
        for step in bar:
            if hasattr(bar, 'set_description'):
                bar.set_description(f'Epoch: {epoch}')
            try:
                images, labels = next(train_iter)
            except StopIteration:
                stop = True
                break
            images, labels = images.to(device, non_blocking=True), labels.to(device, non_blocking=True)
            with accelerator.autocast():
                with accelerator.accumulate(model_wrapper):
                    # accumulate() decided on entry whether this micro-batch closes its window
                    sync = bool(getattr(accelerator, 'sync_gradients', True))
                    loss, metrics = wrapper.train_step(images, labels)
                    with (dp.no_sync() if (dp is not None and not sync) else nullcontext()):
                        accelerator.backward(loss)
                    if dp is not None and sync:
                        dp.all_reduce_mean()
                    optimizer.step()           # accelerate's optimizer wrapper skips these two on non-sync micro-batches
                    optimizer.zero_grad()
            report(late.push(step, metrics), bar)
        report(late.pop(), bar)

    if reset_moco_after_k_epochs is not None and (epoch + 1) in reset_moco_after_k_epochs:
        wrapper.copy_momentum_params()

    if chckpt_fname is not None:
        accelerator.wait_for_everyone()
        save_checkpoint(accelerator.unwrap_model(model_wrapper).model, chckpt_fname, accelerator, matchers)
    return stop


def val_loop(model_wrapper: Union[nn.parallel.DistributedDataParallel, ModelTrainerWrapper],
             val_iter: Iterator[Tuple[torch.Tensor, torch.Tensor]],
             epoch: int,
             num_val_steps: Optional[int],
             accelerator,
             disable_flash: bool = False):
    """``num_val_steps`` (default 100) validation steps; returns (mean loss over steps and ranks, {metric: mean})
    (reference :126-164).  Losses stay on the device until the loop ends: one host sync per call, not per step."""
    model_wrapper.eval()
    wrapper = _unwrap(model_wrapper)
    device = accelerator.device
    n = 100 if num_val_steps is None else num_val_steps
    losses, sums = [], {}
    with _bar(n, accelerator) as bar:
        for _ in bar:
            if hasattr(bar, 'set_description'):
                bar.set_description(f'Epoch: {epoch}')
            images, labels = next(val_iter)
            images, labels = images.to(device, non_blocking=True), labels.to(device, non_blocking=True)
            with torch.no_grad(), accelerator.autocast():
                loss, metrics = wrapper.val_step(images, labels)
                losses.append(accelerator.gather(loss))
                for k, v in accelerator.gather(metrics).items():
                    m = v.float().mean()
                    sums[k] = m if k not in sums else sums[k] + m
    loss = torch.stack([l.float().reshape(-1).mean() for l in losses]).mean().item() if losses else float('nan')
    return loss, {k: float(v) / n for k, v in sums.items()}
