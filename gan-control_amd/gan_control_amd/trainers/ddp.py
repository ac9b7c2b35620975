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

Storage: every bucket owns ONE persistent flat buffer.  When the last expected gradient of a
bucket arrives, the gradients are gathered into it by a single multi-tensor copy, the buffer is
all-reduced in place (RCCL averages in the collective itself), and ``.grad`` of each parameter
is re-pointed at its slice of the buffer: no ``cat``, no copy back.

Bucket order: buckets are first laid out in reverse registration order (an approximation of the order in which backward
produces gradients) and RE-LAID, once, in the order the gradients were actually observed to arrive during the first reduced pass
(G's backward interleaves to_rgbs / convs / modulation layers, which registration order does not follow): a bucket then holds
gradients that arrive back to back and launches as soon as its own share of backward is done instead of waiting for a straggler
registered next to its other members.  Rank 0's order is broadcast so that every rank builds the same buckets.

Which gradients to wait for: the four backward passes of an iteration (D step, R1, G step,
path-length) each touch a different subset of the parameters (R1 and path-length run under
``activation_grads_only`` for their first-order pass and leave the additive biases without a
gradient).  The subset of a pass is a property of the graph, so the reducer learns it per
``phase`` the first time that phase runs (that first run reduces from ``finish()``); from then
on a bucket launches from the hook of the last gradient *that phase* produces.
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
    from ..models.op import weight_cache
    weight_cache.invalidate(module.parameters())     # written through .data: drop whatever was derived from the old values


def all_reduce_mean_(t):
    """In-place mean over ranks of a small tensor (path-length running mean, ADA statistics)."""
    if is_dist():
        dist.all_reduce(t)
        t.div_(dist.get_world_size())
    return t


class _Bucket:
    __slots__ = ('params', 'offsets', 'numel', 'flat', 'pending', 'work', 'ready', 'late', 'from_hook', 'launched_at')

    def __init__(self, params):
        self.params = params
        self.offsets, off = {}, 0
        for p in params:
            self.offsets[p] = off
            off += p.numel()
        self.numel = off
        self.flat = None          # allocated on first use (the parameters may still move to a device after construction)
        self.pending, self.work, self.ready, self.late, self.from_hook, self.launched_at = 0, None, [], [], False, None

    def view(self, p):
        off = self.offsets[p]
        return self.flat[off:off + p.numel()].view_as(p)


class GradientReducer:
    """Mean-reduces ``.grad`` of ``module``'s parameters across ranks, bucket by bucket.

    Usage per optimiser step:  ``reducer.begin(phase=...)`` -> backward(s) -> ``reducer.finish()`` ->
    optimiser.step().  With gradient accumulation call ``begin(sync=False)`` ... ``finish()`` for all but the
    last micro-batch: the record of which parameters hold a gradient survives those non-synchronising
    ``finish()`` calls and is cleared by the synchronising one.  Parameters whose grad stays None (identical on all ranks by construction)
    are skipped.  After ``finish()`` the ``.grad`` tensors are views into the bucket buffers.
    """

    def __init__(self, module, bucket_bytes=None, group=None, order_by_arrival=True, last_bucket_bytes=None):
        """bucket_bytes: default 32 MiB (GANCONTROL_BUCKET_MB).  last_bucket_bytes (GANCONTROL_LAST_BUCKET_MB, default = bucket_bytes): size of the
        bucket holding the gradients that arrive LAST -- the one collective backward cannot hide; a smaller one shortens the exposed tail at the
        price of one more launch.  Both are the first knobs of the 8-GPU session (bench.py --bucket-mb / --last-bucket-mb)."""
        self.group = group
        self.enabled = self._armed = False
        env_mb = lambda name: (float(os.environ[name]) if os.environ.get(name) else None)
        if bucket_bytes is None:
            bucket_bytes = int((env_mb('GANCONTROL_BUCKET_MB') or 32) * (1 << 20))
        if last_bucket_bytes is None:
            mb = env_mb('GANCONTROL_LAST_BUCKET_MB')
            last_bucket_bytes = int(mb * (1 << 20)) if mb else bucket_bytes
        self.bucket_bytes = max(1, bucket_bytes)
        self.last_bucket_bytes = max(1, min(last_bucket_bytes, self.bucket_bytes))
        self._params = [p for p in module.parameters()]
        # reverse registration order approximates the order in which backward produces gradients; replaced by the OBSERVED order of
        # the first reduced pass (_reorder)
        self._layout(list(reversed(self._params)))
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self._params]
        self._order_pending = bool(order_by_arrival)
        self._arrival = []        # parameters in the order their gradients arrived in the current pass
        self._expected = {}       # phase -> set of parameters that received a gradient the last time the phase ran
        self._phase = None
        self._fired = set()
        self.report = {}          # phase -> {'hook': buckets launched from a hook, 'finish': buckets launched in finish()}
        self.max_gap = (64 << 10) // 4   # elements: gaps up to 64 KiB ride along in one collective
        self.measure = False      # bench.py: tally payload bytes and the exposed (not overlapped) wait of every finish()
        self.bytes_reduced, self._stalls = 0, []

    def _layout(self, ordered, live=None):
        """Buckets over ``ordered``; the first ``live`` parameters (default: all) are the ones that produce gradients, and the last of THOSE form
        the small tail bucket when last_bucket_bytes < bucket_bytes."""
        live = len(ordered) if live is None else live
        nbytes = lambda p: p.numel() * p.element_size()
        tail_at = live
        if self.last_bucket_bytes < self.bucket_bytes and live > 1:
            size = 0
            while tail_at > 1 and size + nbytes(ordered[tail_at - 1]) <= self.last_bucket_bytes:
                tail_at -= 1
                size += nbytes(ordered[tail_at])
            tail_at = min(tail_at, live - 1)             # at least one parameter
        self.buckets = []
        for part in (ordered[:tail_at], ordered[tail_at:live], ordered[live:]):
            cur, size = [], 0
            for p in part:
                cur.append(p)
                size += nbytes(p)
                if size >= self.bucket_bytes:
                    self.buckets.append(_Bucket(cur)); cur, size = [], 0
            if cur:
                self.buckets.append(_Bucket(cur))
        self._bucket_of = {p: b for b in self.buckets for p in b.params}

    def _reorder(self):
        """Re-lay the buckets in the arrival order of the pass that has just run (once; before any bucket buffer is in use by a collective).
        Parameters that gave no gradient in that pass keep their reverse-registration position at the tail.  Rank 0 decides for everybody."""
        self._order_pending = False
        index = {p: i for i, p in enumerate(self._params)}
        seen, order = set(), []
        for p in self._arrival:
            if p not in seen:
                seen.add(p); order.append(index[p])
        live = len(order)
        order += [index[p] for p in reversed(self._params) if p not in seen]
        if dist.get_world_size(self.group) > 1:
            dev = self._params[0].device if dist.get_backend(self.group) == 'nccl' else 'cpu'
            t = torch.tensor(order + [live], dtype=torch.int64, device=dev)
            dist.broadcast(t, dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            *order, live = t.tolist()
        # gradients an accumulation pass has already moved into an old bucket buffer keep their values: .grad owns its storage view
        self._layout([self._params[i] for i in order], live)
        self.arrival_order = order

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def begin(self, sync=True, phase=None):
        """Arm the reducer for one backward pass.  ``phase`` names the pass (any hashable); passes with the same name must
        produce gradients for the same parameters (a pass that does not is still reduced correctly, just later)."""
        self.enabled = sync and is_dist()
        self._armed = is_dist()
        self._phase = phase
        expected = self._expected.get(phase)
        self._cur_expected = expected if expected is not None else ()
        self._arrival = []
        for b in self.buckets:
            if expected is None:
                b.pending = -1                       # unknown pass: reduce everything from finish()
            else:
                b.pending = sum(1 for p in b.params if p in expected)
            b.work, b.ready, b.late, b.from_hook, b.launched_at = None, [], [], False, None

    def _on_grad(self, p):
        if not self._armed:
            return
        self._fired.add(p)           # also during accumulation passes (sync=False): the gradient exists from then on
        self._arrival.append(p)
        if not self.enabled:
            return
        b = self._bucket_of[p]
        if b.work is not None:                       # not in the learned set of this phase: reduced on its own in finish()
            b.late.append(p)
            return
        b.ready.append(p)
        if b.pending > 0 and p in self._cur_expected:
            b.pending -= 1
            if b.pending == 0:
                self._launch(b, b.ready)
                b.from_hook = True
                b.launched_at = len(self._arrival)       # gradients seen so far in this pass (tests: how early the collectives start)

    def _reduce(self, t):
        world = dist.get_world_size(self.group)
        if dist.get_backend(self.group) == 'nccl':
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        t.div_(world)                                # gloo has no averaging reduction
        return dist.all_reduce(t, group=self.group, async_op=True)

    def _launch(self, b, ready):
        ready = [p for p in ready if p.grad is not None]
        b.ready = ready
        if not ready:
            return
        g0 = ready[0].grad
        if b.flat is None or b.flat.device != g0.device or b.flat.dtype != g0.dtype:
            # zero-filled ONCE: a gap between two reduced gradients (a parameter this pass gives no gradient) rides along in the
            # collective and must never hold uninitialised memory (a NaN there would trip any NaN-checking debug mode)
            b.flat = torch.zeros(b.numel, dtype=g0.dtype, device=g0.device)
        views = [b.view(p) for p in ready]
        src = [p.grad for p in ready]
        moved = [(v, s) for v, s in zip(views, src) if v.data_ptr() != s.data_ptr()]     # accumulation passes already live in the buffer
        if moved:
            torch._foreach_copy_([v for v, _ in moved], [s for _, s in moved])
        for p, v in zip(ready, views):
            p.grad = v
        # Reduce the spans that hold gradients.  Small gaps (the additive biases R1 / path-length leave without a gradient: a few KiB)
        # ride along -- one collective is cheaper than two -- larger ones (a whole layer without a gradient) split the span.
        spans = []
        for lo, hi in sorted((b.offsets[p], b.offsets[p] + p.numel()) for p in ready):
            if spans and lo - spans[-1][1] <= self.max_gap:
                spans[-1][1] = max(spans[-1][1], hi)
            else:
                spans.append([lo, hi])
        b.work = [self._reduce(b.flat[lo:hi]) for lo, hi in spans]
        self.bytes_reduced += sum(hi - lo for lo, hi in spans) * b.flat.element_size()

    def finish(self):
        """Launch what the hooks did not (unknown pass, stragglers), wait for every bucket."""
        if not self.enabled:
            # An accumulation pass (begin(sync=False)) of a distributed run keeps its record: the gradients it produced exist from now on and
            # must be part of the reduction the final, synchronising pass launches, even if that pass does not touch them again.
            if not is_dist():
                self._fired = set()
            self._armed = False
            return
        if self._order_pending and all(b.work is None for b in self.buckets):
            self._reorder()
        hook = fin = 0
        for b in self.buckets:
            if b.work is None:
                self._launch(b, [p for p in b.params if p.grad is not None and p in self._fired])
                fin += b.work is not None
            else:
                hook += 1
        ev = None
        if self.measure and torch.cuda.is_available():
            # how long the compute stream stalls for the collectives that backward did not hide: events either side of the waits
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        late = []
        for b in self.buckets:
            if b.work is not None:
                for w in b.work:
                    w.wait()
                b.work = None
            for p in b.late:
                if p.grad is not None:
                    late.append(self._reduce(p.grad))
                    self.bytes_reduced += p.grad.numel() * p.grad.element_size()
        for w in late:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._stalls.append(ev)
        if self._phase is not None:
            self._expected[self._phase] = set(self._fired)
        self.report[self._phase] = {'hook': hook, 'finish': fin, 'late': len(late), 'gradients': len(self._arrival),
                                    'launched_at': [b.launched_at for b in self.buckets]}
        self._fired = set()
        self.enabled = self._armed = False        # gradient hooks of a backward outside begin() .. finish() are not this reducer's business

    def comm_summary(self, reset=True):
        """{'bytes': payload handed to the collectives, 'exposed_ms': compute-stream stall in finish()} since the last reset
        (``measure`` must be on; call after a device synchronise)."""
        out = {'bytes': self.bytes_reduced, 'exposed_ms': sum(a.elapsed_time(b) for a, b in self._stalls)}
        if reset:
            self.bytes_reduced, self._stalls = 0, []
        return out
