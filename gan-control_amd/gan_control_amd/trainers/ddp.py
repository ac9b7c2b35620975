"""One-process-per-GPU data parallelism: bucketed gradient all-reduce overlapped with backward.

Replaces the reference's single-process nn.DataParallel (generator_trainer.py:195-199), whose
implicit collectives are scatter / replicate / gather / reduce-add per forward+backward
(SURVEY.md 2.4).  Here every rank holds full replicas; the only data-path collective is the
mean all-reduce of the gradients, launched per bucket from ``post_accumulate_grad`` hooks while
the rest of backward is still running.  Backend ``nccl`` is RCCL over xGMI on ROCm; ``gloo``
runs the same code on CPUs (tests).

Bucket size: on the fully connected 8-GPU xGMI mesh (7 links x ~153 GB/s per GPU) an
all-reduce moves 2*(N-1)/N of the payload per GPU; 32 MiB buckets are large enough to be
bandwidth- rather than latency-bound yet leave 4 buckets per network in flight to hide behind
the remaining backward.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    """True when gradients must be exchanged.  GANCONTROL_FORCE_DDP=1 also turns the collective path on for a
    single-rank group, so the RCCL code path can be exercised on a one-GPU machine (tests)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('GANCONTROL_FORCE_DDP') == '1'


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def broadcast_module(module, src=0):
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def all_reduce_mean_(t):
    """In-place mean over ranks of a small tensor (path-length running mean, ADA statistics)."""
    if is_dist():
        dist.all_reduce(t)
        t.div_(dist.get_world_size())
    return t


class _Bucket:
    __slots__ = ('params', 'pending', 'work', 'flat', 'ready')

    def __init__(self, params):
        self.params, self.pending, self.work, self.flat, self.ready = params, 0, None, None, []


class GradientReducer:
    """Mean-reduces ``.grad`` of ``module``'s parameters across ranks, bucket by bucket.

    Usage per optimiser step:  ``reducer.begin()`` -> backward(s) -> ``reducer.finish()`` ->
    optimiser.step().  With gradient accumulation call ``begin(sync=False)`` for all but the
    last micro-batch.  Parameters whose grad stays None (identical on all ranks by construction)
    are skipped.
    """

    def __init__(self, module, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.enabled = False
        params = [p for p in module.parameters()]
        # reverse registration order approximates the order in which backward produces gradients
        self.buckets, cur, size = [], [], 0
        for p in reversed(params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(_Bucket(cur)); cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        self._bucket_of = {}
        self._handles = []
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def begin(self, sync=True):
        self.enabled = sync and is_dist()
        for b in self.buckets:
            b.pending = sum(1 for p in b.params if p.requires_grad)
            b.work, b.flat, b.ready = None, None, []

    def _on_grad(self, p):
        if not self.enabled:
            return
        b = self._bucket_of[p]
        b.ready.append(p)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        seen = {id(p) for p in b.ready}
        ready = [p for p in b.params if p.grad is not None and id(p) in seen]
        b.ready = ready
        if not ready:
            return
        b.flat = torch.cat([p.grad.reshape(-1) for p in ready])
        b.flat.div_(dist.get_world_size(self.group))
        b.work = dist.all_reduce(b.flat, group=self.group, async_op=True)

    def finish(self):
        """Launch what backward did not complete (parameters without gradients), wait, scatter back."""
        if not self.enabled:
            return
        for b in self.buckets:
            if b.work is None and b.flat is None:
                b.ready = [p for p in b.params if p.grad is not None]
                self._launch(b)
        for b in self.buckets:
            if b.work is None:
                continue
            b.work.wait()
            off = 0
            for p in b.ready:
                n = p.numel()
                if p.grad is not None:          # set_grad_none may have dropped it after the launch
                    p.grad.copy_(b.flat[off:off + n].view_as(p.grad))
                off += n
            b.work, b.flat = None, None
        self.enabled = False
