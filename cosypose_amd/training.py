"""The loop around the training step: learning-rate schedule, checkpoints, epochs.

Contract (cosypose/training/train_pose.py):
  * schedule (:282-299): Adam at `lr`; during the first `n_epochs_warmup` epochs the rate ramps linearly PER BATCH,
    lr * (b + 1) / n_batches_warmup after the b-th optimizer step (a LambdaLR stepped once per batch); from then on it is
    divided by 10 every `lr_epoch_decay` epochs (a StepLR stepped at the end of every epoch >= n_epochs_warmup).  Both torch
    schedulers act on the same optimizer in "chained" form; `LRSchedule` is the same state machine -- including the state a
    resumed run starts from (:286, :297-298: warm-up counter = start_epoch * batches_per_epoch, StepLR stepped once).
  * checkpoint (:54-61, :260-275): `checkpoint.pth.tar` = {'state_dict': module.state_dict(), 'epoch': epoch}, written after
    every epoch; resuming loads the state_dict and continues at epoch + 1.  Optimizer moments are NOT part of the format
    (a resumed run restarts Adam's statistics, as in the reference).
  * step (:317-331): zero_grad -> h_pose -> backward -> clip_grad_norm_(clip_grad_norm) -> Adam.step, then the warm-up
    scheduler; with `FlatAdam` clip + Adam are two fused launches on the flat buffers and the gradient all-reduce of a
    multi-rank run is one RCCL call (or pass a DistributedDataParallel model and let DDP do it).
  * upload (:242 pin_memory=True + pose_forward_loss.py:24-27 .cuda() inside the step): the reference uploads a batch at the
    top of the step that consumes it -- 59 MB of uint8 frames for 64 x 480x640, ~1 ms of PCIe in front of the first kernel.
    `DevicePrefetcher` uploads batch i+1 on a copy stream (SDMA, no compute units) WHILE step i computes; h_pose's .cuda() is
    then the identity.  Same batches, same order, same values.
Datasets, samplers and evaluation are out of scope: `train_loop` takes any iterable of batches.
"""
import copy
import pathlib
from collections import defaultdict

import torch

from . import train_engine
from .pose_forward_loss import h_pose


class LRSchedule:
    """The learning rate of every optimizer step, as a small state machine that reproduces what the reference's two chained
    torch schedulers do to the optimizer (train_pose.py:284-299, :331-334), INCLUDING what happens on resume:

        fresh run   : ramp lr*(s+1)/n_warm_batches over the warm-up steps s, ending one notch ABOVE lr (the LambdaLR is
                      stepped once more after the last warm-up batch: lr*(n+1)/n), then x0.1 every `lr_epoch_decay` epochs
                      counted from the END of the warm-up;
        resumed run : the LambdaLR's constructor leaves lr at lr/n_warm_batches and nothing resets it once the warm-up is
                      over, and the StepLR restarts its counter at start_epoch (one decay at most is applied at start).

    `faithful=True` (default) is that behaviour, bit for bit (tests/test_host_logic.py drives torch's schedulers in the
    reference's order beside it).  `faithful=False` is the evident intent: the fresh-run schedule as a pure function of the
    global step, so that a resumed run continues where the interrupted one left off."""

    def __init__(self, lr, n_epochs_warmup, batches_per_epoch, lr_epoch_decay, start_epoch=0, gamma=0.1, faithful=True):
        self.base, self.n_warm, self.bpe, self.decay, self.gamma, self.faithful = float(lr), int(n_epochs_warmup), int(batches_per_epoch), int(lr_epoch_decay), gamma, faithful
        self.nbw = self.n_warm * self.bpe
        self.start_epoch = int(start_epoch)
        # state after the reference's construction sequence
        self.warm_counter = self.start_epoch * self.bpe                 # LambdaLR.last_epoch (overwritten after construction)
        self.lr = self.base * self._lam(0)                              # LambdaLR's constructor applied lambda(0)
        self.step_counter = self.start_epoch                            # StepLR.last_epoch after `last_epoch = start - 1; step()`
        if self.step_counter > 0 and self.step_counter % self.decay == 0:
            self.lr *= self.gamma

    def _lam(self, counter):
        return 1.0 if self.n_warm == 0 else (counter + 1) / self.nbw

    def intended(self, epoch, batch):
        """the fresh-run schedule as a function of the position in training"""
        if self.n_warm and epoch < self.n_warm:
            return self.base * (epoch * self.bpe + batch + 1) / self.nbw
        top = self.base * ((self.nbw + 1) / self.nbw if self.n_warm else 1.0)
        return top * self.gamma ** ((epoch - self.n_warm) // self.decay)

    def current(self, epoch, batch):
        return self.lr if self.faithful else self.intended(epoch, batch)

    def after_batch(self, epoch):
        if epoch < self.n_warm:                                         # lr_scheduler_warmup.step()
            self.warm_counter += 1
            self.lr = self.base * self._lam(self.warm_counter)

    def after_epoch(self, epoch):
        if epoch >= self.n_warm:                                        # lr_scheduler.step(), chainable form
            self.step_counter += 1
            if self.step_counter % self.decay == 0:
                self.lr *= self.gamma

    def apply(self, optimizer, epoch, batch):
        v = self.current(epoch, batch)
        if isinstance(optimizer, train_engine.FlatAdam):
            optimizer.lr = v
        else:
            for g in optimizer.param_groups:
                g['lr'] = v
        return v


class DevicePrefetcher:
    """Iterates `batches` and hands every batch out with its tensor fields already in HBM.  One batch ahead: when batch j is handed
    out, the upload of batch j+1 has just been enqueued on a dedicated copy stream, so it runs beside step j's kernels; the
    consumer's stream waits for the upload's event (the host does not).  The device tensors live in TWO persistent slots (no
    allocator traffic, no frees that wait for another stream): batch j+1 goes into the slot batch j-1 used, after an event recorded
    on the consumer's stream when batch j is requested -- by then all of step j-1 has been enqueued.  So a batch's device tensors
    are valid until the NEXT batch is requested (requesting batch j enqueues the upload of batch j+1 into the slot batch j-1 used, gated
    only on the work enqueued so far); a consumer that keeps them longer -- logging the previous batch during the next step -- must clone them.  Batches are shallow
    copies: the caller's objects keep their host tensors.  Host tensors should be page-locked (DataLoader(pin_memory=True)) --
    pageable ones are uploaded synchronously by the runtime and only the ordering benefit remains."""
    FIELDS = ('images', 'K', 'TCO', 'bboxes')

    def __init__(self, batches, device=None, fields=FIELDS):
        if not torch.cuda.is_available():
            raise RuntimeError('DevicePrefetcher uploads to an MI355X: no GPU visible')
        self.batches, self.fields = batches, tuple(fields)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots, self._free = [{}, {}], [None, None]

    def __len__(self):
        return len(self.batches)

    def _upload(self, batch, slot):
        held, moved = self._slots[slot], {}
        consumer = torch.cuda.current_stream(self.device)        # captured BEFORE the copy-stream context: inside it current_stream is the copy stream
        with torch.cuda.stream(self.stream):
            if self._free[slot] is not None:
                self.stream.wait_event(self._free[slot])            # the step that read this slot's previous batch is through
            for name in self.fields:
                t = getattr(batch, name, None)
                if not torch.is_tensor(t) or t.is_cuda:
                    continue
                buf = held.get(name)
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    if buf is not None:
                        buf.record_stream(consumer)      # a replaced buffer may still be read by the consumer's queued kernels
                    buf = held[name] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                buf.copy_(t, non_blocking=True)
                moved[name] = buf
            ready = torch.cuda.Event()
            ready.record(self.stream)
        if hasattr(batch, '_replace'):                 # a namedtuple batch
            return batch._replace(**moved), ready
        out = copy.copy(batch)
        for name, t in moved.items():
            setattr(out, name, t)
        return out, ready

    def __iter__(self):
        source = iter(self.batches)
        # a re-iteration (train_loop keeps ONE prefetcher across epochs) must not inherit the previous pass's slot events: the last step of that pass may
        # still be reading either slot, so both are gated on everything the consumer has enqueued up to now
        start = torch.cuda.Event()
        start.record(torch.cuda.current_stream(self.device))
        self._free = [start, start]
        try:
            ahead = self._upload(next(source), 0)
        except StopIteration:
            return
        j = 0
        while ahead is not None:
            (batch, ready), ahead = ahead, None
            consumer = torch.cuda.current_stream(self.device)
            done = torch.cuda.Event()
            done.record(consumer)                      # everything enqueued so far = every step up to j-1
            self._free[(j + 1) % 2] = done
            try:
                ahead = self._upload(next(source), (j + 1) % 2)     # enqueued BEFORE the consumer enqueues step j: runs beside it
            except StopIteration:
                pass
            consumer.wait_event(ready)
            yield batch
            j += 1


class LazyMeters(defaultdict):
    """Meters (name -> object with .add(float)) that take values living on the device WITHOUT stopping the host: `defer` starts a
    non-blocking copy into page-locked memory and the numbers are added, in order, once their copy has finished (checked at every
    later `defer`, forced by `flush`).  h_pose uses `defer` when its `meters` has one: the reference's .item() calls in the middle of
    the step (pose_forward_loss.py:74-83) make the host wait for the forward pass and the GPU then idle while the backward's first
    kernels are being enqueued (~1 ms of a 33 ms step)."""

    def __init__(self, factory=None, max_pending=64):
        super().__init__(factory if factory is not None else _Mean)
        self._pending, self.max_pending = [], max_pending

    def defer(self, names, values):
        """values: 1-D device tensor; names[i]: the meter names that receive values[i]"""
        if not values.is_cuda:                       # nothing to wait for
            self._drain(force=len(self._pending))
            for keys, v in zip(names, values.detach().tolist()):
                for k in keys:
                    self[k].add(v)
            return
        host = torch.empty(values.shape, dtype=values.dtype, pin_memory=True)
        host.copy_(values.detach(), non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(values.device))
        self._pending.append((names, host, done))
        self._drain(force=len(self._pending) - self.max_pending)

    def _drain(self, force=0):
        while self._pending and (force > 0 or self._pending[0][2].query()):
            names, host, done = self._pending.pop(0)
            done.synchronize()
            for keys, v in zip(names, host.tolist()):
                for k in keys:
                    self[k].add(v)
            force -= 1

    def flush(self):
        self._drain(force=len(self._pending))
        return self


def checkpoint_path(save_dir):
    return pathlib.Path(save_dir) / 'checkpoint.pth.tar'


def save_checkpoint(model, epoch, save_dir):
    """{'state_dict', 'epoch'} exactly as the reference writes it (train_pose.py:54-61); DDP wrappers are unwrapped."""
    save_dir = pathlib.Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    module = model.module if hasattr(model, 'module') else model
    path = checkpoint_path(save_dir)
    torch.save({'state_dict': {k: v.detach().cpu() for k, v in module.state_dict().items()}, 'epoch': int(epoch)}, path)
    return path


def load_checkpoint(path_or_dir, model, strict=True):
    """Loads a reference-format checkpoint into `model`; returns the epoch to continue with (saved epoch + 1)."""
    p = pathlib.Path(path_or_dir)
    if p.is_dir():
        p = checkpoint_path(p)
    save = torch.load(p, map_location='cpu')
    module = model.module if hasattr(model, 'module') else model
    module.load_state_dict(save['state_dict'], strict=strict)
    return int(save['epoch']) + 1


def train_loop(model, mesh_db, cfg, batches, n_epochs, save_dir=None, start_epoch=0, optimizer=None, on_epoch_end=None,
               input_generator='fixed', faithful_schedule=True, prefetch=True, lazy_meters=True):
    """Runs epochs [start_epoch, n_epochs) of the reference's training loop on `batches` (a callable epoch -> iterable of
    batch objects with images / K / TCO / objects / bboxes, or a re-iterable).  cfg: lr, weight_decay, n_epochs_warmup,
    lr_epoch_decay, clip_grad_norm, n_iterations (+ what h_pose needs).  prefetch: upload batch i+1 while step i runs
    (DevicePrefetcher); lazy_meters: read the loss / gradient-norm values back without stopping the host (LazyMeters).  Returns {epoch: mean loss}."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    is_ddp = hasattr(model, 'module')
    if optimizer is None:
        # direct_grads (the backward writes straight into FlatAdam's flat buffer and returns None per parameter) must be OFF under a
        # DistributedDataParallel wrapper: DDP averages through per-parameter autograd hooks, which would never fire
        optimizer = train_engine.FlatAdam(model.module if is_ddp else model, lr=cfg.lr, weight_decay=getattr(cfg, 'weight_decay', 0.0),
                                          clip_grad_norm=cfg.clip_grad_norm, direct_grads=not is_ddp,
                                          overlap_allreduce=world > 1 and not is_ddp)      # the data-parallel step: buckets launched from inside the backward
    flat = isinstance(optimizer, train_engine.FlatAdam)
    if flat and is_ddp and optimizer.direct_grads:
        raise ValueError('train_loop: a FlatAdam with direct_grads=True under DistributedDataParallel would skip DDP\'s gradient hooks (the ranks '
                         'would apply un-averaged gradients): build it with direct_grads=False, or pass the bare module and let train_loop '
                         'average with train_engine.allreduce_gradients')
    if world > 1 and not is_ddp and not flat:
        raise ValueError('train_loop: %d ranks, but the model is not DistributedDataParallel and the optimizer is not FlatAdam: '
                         'the ranks would train independently (wrap the model in DDP or use train_engine.FlatAdam)' % world)
    if start_epoch > 0 and faithful_schedule:
        import warnings
        warnings.warn('train_loop: resuming with faithful_schedule=True reproduces the reference\'s resume behaviour '
                      '(train_pose.py:283-298: the warm-up LambdaLR is rebuilt on resume): the learning rate restarts at '
                      'lr / n_warm_batches and never recovers; pass faithful_schedule=False for the intended schedule')
    bpe = getattr(cfg, 'batches_per_epoch', None)      # the reference's warm-up length: epoch_size // batch_size (train_pose.py:296)
    if bpe is None and getattr(cfg, 'epoch_size', None) and getattr(cfg, 'batch_size', None):
        bpe = cfg.epoch_size // cfg.batch_size
    history = {}
    schedule = None
    prefetcher = None
    for epoch in range(start_epoch, n_epochs):
        it = batches(epoch) if callable(batches) else batches
        items = list(it) if schedule is None and not hasattr(it, '__len__') else it
        if schedule is None:
            schedule = LRSchedule(cfg.lr, cfg.n_epochs_warmup, bpe or len(items), cfg.lr_epoch_decay, start_epoch=start_epoch, faithful=faithful_schedule)
        model.train()
        meters = LazyMeters(_Mean) if lazy_meters and torch.cuda.is_available() else defaultdict(_Mean)
        if prefetch and torch.cuda.is_available():       # ONE prefetcher (copy stream + two persistent slots) for all epochs
            if prefetcher is None:
                prefetcher = DevicePrefetcher(items)
            else:
                prefetcher.batches = items
            feed = prefetcher
        else:
            feed = items
        for b, sample in enumerate(feed):
            schedule.apply(optimizer, epoch, b)
            optimizer.zero_grad()
            loss = h_pose(model=model, mesh_db=mesh_db, data=sample, meters=meters, cfg=cfg, n_iterations=getattr(cfg, 'n_iterations', 1),
                          input_generator=input_generator)
            loss.backward()
            if flat:
                if world > 1 and not is_ddp:
                    train_engine.allreduce_gradients(optimizer)      # collects detached .grad tensors first
                norm = optimizer.step()
                if hasattr(meters, 'defer') and torch.is_tensor(norm) and norm.is_cuda:
                    meters.defer([('grad_norm',)], norm.reshape(1))         # read back when it is there: the host goes on to the next step
                else:
                    meters['grad_norm'].add(float(norm))
            else:
                params = (model.module if is_ddp else model).parameters()
                meters['grad_norm'].add(float(torch.nn.utils.clip_grad_norm_(params, max_norm=cfg.clip_grad_norm, norm_type=2)))
                optimizer.step()
            schedule.after_batch(epoch)
        schedule.after_epoch(epoch)
        if hasattr(meters, 'flush'):
            meters.flush()
        history[epoch] = meters['loss_total'].mean
        if save_dir is not None and (not dist.is_initialized() or dist.get_rank() == 0):
            save_checkpoint(model, epoch, save_dir)
        if on_epoch_end is not None:
            on_epoch_end(epoch, {k: m.mean for k, m in meters.items()})
    return history


class _Mean:
    def __init__(self):
        self.total, self.n = 0.0, 0

    def add(self, v):
        self.total += float(v); self.n += 1

    @property
    def mean(self):
        return self.total / max(self.n, 1)
