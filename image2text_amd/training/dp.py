"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce (mean) of the flat gradient arena.

The reference wraps the trainer in accelerate/DDP (trainer.py:108-114,173-174) but its call pattern bypasses
``DDP.forward`` so no gradient is ever reduced; this implements the intended semantics:
every rank runs forward/backward on its own shard (per-replica ``normalize_gradients`` norms, loss divided by the
local batch), then parameters see the MEAN over ranks of the per-shard gradients.

MI355X mapping: gradients live in ONE contiguous fp32 arena, so the exchange is two large collectives instead of
DDP's 25 MB buckets: the decoder slice (80 % of the bytes) is reduced on RCCL's stream as soon as the decoder backward
has finished, overlapped with the whole encoder backward; the encoder slice follows.  xGMI is point-to-point, large
messages keep every link busy; RCCL picks the rings/trees.  Transport: on the GPU the package's own RCCL communicator behind
the C ABI (``i2t_comm_*``, csrc/comm.cpp; its unique id travels over the existing ``torch.distributed`` group, the collectives run
on a side HIP stream; ``I2T_DP_COMM=torch`` keeps ``torch.distributed.all_reduce`` instead, ``I2T_DP_WIRE=bf16`` sends bf16); on CPU
the same code runs over gloo (tests).  Frozen parameters (LoRA's base weights, a ``prepare_for_kbit_training`` decoder, a backbone
run under no_grad) are not exchanged: the reduce ranges are the coalesced spans of trainable arena entries.

Protocol (what a training loop does; ``training/utils.train_loop`` and ``bench.py`` follow it):

    with dp.no_sync():                 # micro-batches 1 .. k-1 of an accumulation window: nothing is exchanged
        loss.backward()
    loss.backward()                    # last micro-batch: the decoder slice's all-reduce starts inside this backward
    dp.all_reduce_mean()               # finishes the exchange (encoder slice + wait); then optimizer.step()

A backward that is neither inside ``no_sync`` nor followed by ``all_reduce_mean`` (e.g. a priming backward that only
builds the arena) is tolerated: the next backward's start (engine hook 'begin'), ``broadcast_parameters`` and
``all_reduce_mean`` all drain whatever is still in flight and reset the window, so an early reduce is never skipped
because of stale state and never races the next backward's accumulation into the arena.
"""
import contextlib
import os
from typing import Optional

import torch
import torch.distributed as dist

# CUs left free for RCCL while a collective overlaps the backward pass.  The persistent GEMM kernels occupy whole CUs with a
# static share of the tiles each; a collective's workgroups (one per channel) need whole CUs too and keep them for its whole
# duration, so without this every GEMM launched during the window would wait for a second round of workgroups (2x its
# time).  NCCL_MAX_NCHANNELS caps RCCL at the same number (only set when the user has not chosen a value).
RCCL_CUS = int(os.environ.get('I2T_RCCL_CUS', '16'))


def configure_rccl_env():
    """Call BEFORE ``dist.init_process_group``: RCCL reads NCCL_MAX_NCHANNELS when a communicator is created, and
    ``init_process_group(device_id=...)`` creates it eagerly.  A value the user exported wins."""
    if RCCL_CUS > 0:
        os.environ.setdefault('NCCL_MAX_NCHANNELS', str(RCCL_CUS))


GAP_FLOATS = 1 << 18        # frozen gaps up to 1 MB between two trainable spans travel with them (cheaper than one more collective)


class _Done:
    """What ``dist.all_reduce(async_op=True)`` returns, for a collective issued on the communicator's side stream."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def _exchange_unique_id(group, uid: Optional[bytes]) -> bytes:
    """Rank 0 of ``group`` hands RCCL's 128-byte unique id to the others WITHOUT a device collective: through the process group's
    store (``TCPStore.set`` / ``get``; the get blocks until the key exists), else as a pickled host object.  A device
    ``dist.broadcast`` would make torch create an RCCL communicator of its own just for these 128 bytes -- with it every process
    held two communicators (two sets of channels and proxy threads on the same xGMI links)."""
    rank = dist.get_rank(group)
    RcclComm._created += 1
    key = f'i2t_comm_uid/{dist.get_world_size(group)}/{RcclComm._created}'      # every rank creates its communicators in the same order
    try:
        store = dist.distributed_c10d._get_default_store()
        if group is not None and group is not dist.group.WORLD:
            key += '/' + '-'.join(str(r) for r in dist.get_process_group_ranks(group))
        if rank == 0:
            store.set(key, uid)
            return uid
        return bytes(store.get(key))
    except Exception:                       # no reachable store (a custom rendezvous): one host-side object broadcast
        box = [uid]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return bytes(box[0])


def _all_agree(group, ok: bool, device) -> bool:
    """True iff ``ok`` holds on EVERY rank (a MIN all-reduce over the existing group: host tensor under gloo)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend(group) == 'nccl' else 'cpu')
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


class RcclComm:
    """The C-ABI communicator (include/i2t.h ``i2t_comm_*``): created collectively by all ranks of ``group``.  ``group`` is only
    the control plane (unique id, agreement on success): it may be a gloo group -- then this is the process's ONLY RCCL communicator
    and NCCL_MAX_NCHANNELS caps the one that matters (bench.py runs that way)."""
    _created = 0

    def __init__(self, group, device, max_floats: int = 0):
        import ctypes as C
        from .. import lib as _l
        self._l, self._lib = _l, _l.load()
        self.handle = None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        err = None
        uid = None
        if not self._lib.i2t_comm_available():
            err = 'RCCL is not available in this process'
        elif rank == 0:
            buf = C.create_string_buffer(128)
            if self._lib.i2t_comm_unique_id(buf, 128) != 0:
                err = 'i2t_comm_unique_id failed'
            uid = bytes(buf.raw)
        # every rank must take the same branch from here on: ncclCommInitRank is collective, a rank that skipped it would leave
        # the others waiting in it forever
        if not _all_agree(group, err is None, device):
            raise RuntimeError(err or 'RCCL is unavailable on another rank')
        uid = _exchange_unique_id(group, uid)
        handle = C.c_void_p()
        with torch.cuda.device(device):
            rc = self._lib.i2t_comm_init(uid, world, rank, C.byref(handle))
        if not _all_agree(group, rc == 0, device):
            if rc == 0:
                self._lib.i2t_comm_destroy(handle)
            raise RuntimeError('i2t_comm_init failed' + ('' if rc == 0 else f' on this rank: {_l.last_error()}') + ' (no rank keeps a communicator)')
        self.handle, self.world, self.device = handle, world, device
        self.stream = torch.cuda.Stream(device=device)
        self.wire_bf16 = os.environ.get('I2T_DP_WIRE', 'f32') == 'bf16'
        # bf16 wire image: ONE buffer for the communicator's lifetime, sized for the largest span it will ever carry (the arena).
        # (Re-allocating it when a later span is larger would hand the old block back to the caching allocator while a collective
        # on the side stream may still be reading it.)
        self._staging = None
        self._max_floats = int(max_floats)

    def _staging_for(self, n: int, device):
        if self._staging is None or self._staging.numel() < n:
            if self._staging is not None:
                self._staging.record_stream(self.stream)           # the allocator may reuse it only after the side stream is done with it
            self._staging = torch.empty(max(n, self._max_floats), dtype=torch.bfloat16, device=device)
        return self._staging

    def all_reduce_async(self, t: torch.Tensor, mean: bool = True):
        """Mean (or sum) over ranks of the fp32 tensor ``t`` (contiguous view of the arena), in place, on the side stream."""
        assert t.dtype == torch.float32 and t.is_contiguous() and self.handle is not None
        n = t.numel()
        staging = self._staging_for(n, t.device) if (self.wire_bf16 and mean and n % 4 == 0) else None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)                      # the gradients being sent are final on the compute stream
        self._l.check(self._lib.i2t_comm_allreduce(self.handle, self.stream.cuda_stream, t.data_ptr(), n, 1 if mean else 0,
                                                   staging.data_ptr() if staging is not None else None), 'i2t_comm_allreduce')
        done = torch.cuda.Event()
        done.record(self.stream)
        return _Done(done)

    def all_reduce_mean_async(self, t: torch.Tensor):
        return self.all_reduce_async(t, mean=True)

    def close(self):
        if self.handle is not None:
            self.stream.synchronize()
            self._lib.i2t_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DataParallelGrads:
    def __init__(self, model, group: Optional[dist.ProcessGroup] = None, overlap: bool = True):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.overlap = overlap
        self.comm = None                # RcclComm, created with the first exchange on a GPU arena (see _transport)
        self._comm_tried = False
        self._fb_group = None
        self._pending = []              # (work, tensor view, finishing step or None)
        self._reduced_upto = None       # arena offset from which the current window's gradients are already in flight/reduced
        self._reserved = False
        self._sync = True               # False inside no_sync(): accumulate locally, exchange nothing
        if dist.get_backend(group) == 'nccl':
            configure_rccl_env()        # too late for an eagerly created communicator (see configure_rccl_env), harmless otherwise
        eng = getattr(model, '_engine', None)
        if eng is not None and overlap:
            eng.grad_ready_hooks.append(self._on_grads_ready)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside this context only accumulate into the local arena (DDP.no_sync semantics)."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def _arena(self):
        arena = self.model._engine.arena if hasattr(self.model, '_engine') else self.model.arena
        if arena is None:
            raise RuntimeError('no parameter arena yet: run one forward/backward first')
        return arena

    def _split(self, arena):
        """[lo, hi): the arena range that holds the decoder's parameters (named_parameters lays the towers out one after the other;
        which one comes first depends on the module's registration order, so both ends are looked up)."""
        offs = [(off, off + ((n + 7) // 8) * 8) for name, (off, n, _) in arena.entries.items() if name.startswith('decoder.')]
        if not offs:
            return arena.total, arena.total
        lo, hi = min(o for o, _ in offs), max(e for _, e in offs)
        # (the arena pads the start of every matrix to a line: gaps inside the range are fine, another tower's entry is not)
        if any(lo <= off < hi for name, (off, n, _) in arena.entries.items() if not name.startswith('decoder.')):
            raise RuntimeError('decoder parameters are not contiguous in the arena: the overlapped gradient exchange needs them to be')
        return lo, min(hi, arena.total)

    def _transport(self, t: torch.Tensor):
        """The package's RCCL communicator for GPU arenas (created once, collectively: every rank reaches its first exchange
        together; the process group -- nccl OR gloo -- is only its control plane); None = torch.distributed (a CPU arena over gloo,
        I2T_DP_COMM=torch, or RCCL not bindable on some rank: the ranks agree on the outcome, so all of them take the same transport)."""
        if not self._comm_tried:
            self._comm_tried = True
            if t.is_cuda and os.environ.get('I2T_DP_COMM', 'rccl') != 'torch':
                try:
                    self.comm = RcclComm(self.group, t.device, max_floats=self._arena().total)
                except Exception as e:      # (raised on EVERY rank or on none: RcclComm agrees on each step before the collective init)
                    import warnings
                    warnings.warn(f'image2text_amd: C-ABI RCCL communicator unavailable ({e}); using torch.distributed.all_reduce')
        return self.comm

    def _spans(self, arena, lo: int, hi: int):
        """Coalesced [a, b) ranges of TRAINABLE arena entries inside [lo, hi): frozen parameters have no gradient to exchange."""
        spans = []
        for name, (off, n, _) in sorted(arena.entries.items(), key=lambda kv: kv[1][0]):
            a, b = max(off, lo), min(off + ((n + 7) // 8) * 8, hi, arena.total)
            if b <= a or name not in arena.params or not arena.trainable(name):
                continue
            if spans and a - spans[-1][1] <= GAP_FLOATS:
                spans[-1][1] = b
            else:
                spans.append([a, b])
        return [(a, b) for a, b in spans]

    def _reduce(self, t: torch.Tensor, async_op: bool):
        """Start the mean over ranks of ``t`` in place.  Returns (work, post): ``post`` (or None) finishes the mean after ``work.wait()``."""
        comm = self._transport(t)
        if comm is not None:
            return comm.all_reduce_mean_async(t), None
        group = self._cuda_group() if t.is_cuda else self.group
        backend = dist.get_backend(group)
        if os.environ.get('I2T_DP_WIRE', 'f32') == 'bf16':
            # the bf16 wire form on torch.distributed (what csrc/comm.cpp does around ncclAllReduce: round, sum in bf16, widen x 1/world)
            wire = t.to(torch.bfloat16)
            work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            return work, (lambda: t.copy_(wire.to(torch.float32) / self.world))
        if backend == 'nccl':
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=async_op), None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op), (lambda: t.div_(self.world))

    def _cuda_group(self):
        """The group a GPU arena is reduced over when the package's own communicator is not in use: the control group if it is an NCCL
        (= RCCL) one, else an NCCL group over the same ranks created here, once and collectively (every rank reaches its first fallback
        exchange together, having agreed that the communicator is unavailable) -- a gloo all-reduce of a GPU arena would stage 647 MB
        through the host on every step.  If that group cannot be made either, the control group (slow, correct)."""
        if self._fb_group is None:
            self._fb_group = self.group
            if dist.get_backend(self.group) != 'nccl':
                try:
                    ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
                    self._fb_group = dist.new_group(ranks=ranks, backend='nccl')
                except Exception as e:
                    import warnings
                    warnings.warn(f'image2text_amd: no NCCL group for the fallback exchange ({e}); reducing GPU gradients over {dist.get_backend(self.group)}')
        return self._fb_group

    def _drain(self):
        """Wait for every collective in flight, finish its mean, release the CU reservation, close the window."""
        for work, _, post in self._pending:
            if work is not None:
                work.wait()
            if post is not None:
                post()
        self._pending.clear()
        self._reduced_upto = None
        if self._reserved:
            from .. import ops
            ops.gemm_reserve_cus(0)
            self._reserved = False

    def _on_grads_ready(self, which: str):
        """Engine callback.  'begin': a backward is about to write the gradient arena.  'decoder': the decoder's gradients
        are final for this backward (fires before the encoder backward starts)."""
        if which == 'begin':
            self._drain()               # an exchange nobody finished must not race this backward's accumulation
            return
        if which != 'decoder' or not self._sync:
            return
        arena = self._arena()
        lo, hi = self._split(arena)
        if hi <= lo:
            return
        if arena.g32.is_cuda and RCCL_CUS > 0 and (self._transport(arena.g32) is not None or dist.get_backend(self._cuda_group()) == 'nccl'):
            from .. import ops
            ops.gemm_reserve_cus(RCCL_CUS)          # the encoder backward's GEMMs leave room for the collective
            self._reserved = True
        for a, b in self._spans(arena, lo, hi):
            work, post = self._reduce(arena.g32[a:b], async_op=True)
            self._pending.append((work, arena.g32[a:b], post))
        self._reduced_upto = (lo, hi)

    def all_reduce_mean(self):
        """Finish the exchange: after this every rank holds mean-over-ranks gradients in its arena / p.grad."""
        arena = self._arena()
        lo, hi = self._reduced_upto if self._reduced_upto is not None else (0, 0)
        for a0, b0 in ((0, lo), (hi, arena.total)):      # everything the 'decoder' hook has not already put on the wire
            for a, b in self._spans(arena, a0, b0):
                work, post = self._reduce(arena.g32[a:b], async_op=True)
                self._pending.append((work, arena.g32[a:b], post))
        self._drain()

    def broadcast_parameters(self, src: int = 0):
        """Initial parameter sync (what DDP does at wrap time).  Also closes any exchange window left open by a priming
        backward (its gradients are the caller's to discard: zero_grad)."""
        self._drain()
        arena = self._arena()
        comm = self._transport(arena.p32)
        if comm is not None:
            # on the package's own communicator: every rank but ``src`` contributes zeros to a SUM (exact: x + 0 + ... + 0)
            if dist.get_rank(self.group) != src:
                arena.p32.zero_()
            comm.all_reduce_async(arena.p32, mean=False).wait()
        else:
            dist.broadcast(arena.p32, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
        arena._versions = None          # force a bf16 shadow refresh on the next forward
        arena.generation += 1           # parameter VALUES changed under every cache keyed on them (fp8 weight images, merged LoRA weights)
        eng = getattr(self.model, '_engine', None)
        if eng is not None and hasattr(eng, '_sub_cache'):
            for k in [k for k in eng._sub_cache if isinstance(k, tuple) and k and k[0] == 'fp8w']:
                del eng._sub_cache[k]

    def close(self):
        """Finish whatever is in flight and destroy the communicator (collective teardown: call on every rank, before
        ``dist.destroy_process_group``)."""
        self._drain()
        eng = getattr(self.model, '_engine', None)
        if eng is not None and self._on_grads_ready in getattr(eng, 'grad_ready_hooks', []):
            eng.grad_ready_hooks.remove(self._on_grads_ready)
        if self.comm is not None:
            self.comm.close()
            self.comm = None
