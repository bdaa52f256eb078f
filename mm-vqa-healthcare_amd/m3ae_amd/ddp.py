"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI; gloo on CPU for tests).

The reference's only parallelism is PyTorch-Lightning DDP (main.py:59-63): replicate weights, shard the global
batch, all-reduce gradients.  Here the gradients already live in ONE flat fp32 buffer (ParamStore.grad), so a
bucket is a contiguous byte range of it -- no gather/scatter copies -- and buckets are cut by BYTES, sized for
xGMI: RCCL rings are per-link bound (7 links x ~153 GB/s per GPU), so buckets are large (default 64 MiB) to
amortise launch + ring latency.  The wgrad kernels accumulate straight into the bucket memory and report each
finished parameter through `ops.grad_ready_hook`; when the last parameter of a bucket reports, its all-reduce is
launched asynchronously (torch.distributed orders it after the producing kernels and runs it on RCCL's own stream),
overlapping the rest of backward.  `finish()` launches whatever is left (parameters whose gradient flows through
autograd glue never report) and waits.  A parameter can receive SEVERAL in-place contributions per step (the three
`infer` passes of a pre-training step, an in-projection used for queries and for keys/values): the reducer LEARNS, on its
first step, how many reports each parameter makes (no early launches on that step) and afterwards launches a bucket only
when every parameter in it has reported that many times; a report that arrives after its bucket was launched is an
error (`relearn()`), never a silently half-reduced gradient.  Averaging (1 / world_size) is folded into the fused AdamW kernel's
`grad_scale`, not applied to the buffer.  The 6 tensors that never receive a gradient are not in the
buffer at all -- no `find_unused_parameters` pass.

Bucket order (round 3).  The flat buffer is laid out optimizer group by optimizer group, and inside a group in module
(= forward execution) order, so backward completes a group's parameters from its END towards its start.  Buckets are
therefore cut walking every group BACKWARDS from its last parameter, a full bucket every `bucket_bytes`, except that the
group's first ~`tail_bytes` -- the parameters whose gradients arrive last (patch embedding, first encoder blocks) -- are a
bucket of their own: the all-reduce that starts after the last backward kernel, the only one nothing overlaps, moves
~`tail_bytes` instead of up to `bucket_bytes`.  `finish_order` records the order in which buckets completed on the
last step (tests; DESIGN.md 7 states the exposed-communication prediction this is meant to falsify).

`grad_dtype="bf16"`: a bucket is rounded to bf16 into a staging buffer, all-reduced at half the bytes (0.64 GB instead of
1.29 GB per step at M3AE-base) and written back as fp32 before the optimizer step.  Every rank's addend is rounded once
(relative 2^-9) and the sum is accumulated in bf16 by the collective: the reduced gradient differs from the fp32 all-reduce
by <= ~1e-2 in relative L2 norm (stated tolerance, held by the two-virtual-rank GPU test; single elements whose gradient is
rounding noise can differ by their whole value, which AdamW's m / sqrt(v) turns into a full +-lr step for those elements).
"""
import os
import torch
import torch.distributed as dist

from . import ops



def rccl_group_options():
    """Options for dist.init_process_group("nccl", pg_options=...): RCCL's streams at HIGH priority.  The step's launch stream is
    high priority (ops.launch_stream): at normal priority the collectives' few workgroups would be dispatched only in the compute
    kernels' tails and could surface at the end of backward.  None when this torch build has no such option."""
    try:
        from torch.distributed import ProcessGroupNCCL
        opts = ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:  # noqa: BLE001
        return None

class FlatGradReducer:
    def __init__(self, store, bucket_bytes=64 << 20, group=None, overlap=True, collective=None, world=None,
                 tail_bytes=8 << 20, grad_dtype="fp32", update_in_backward=False):
        """`collective(tensor) -> handle with .wait()` replaces `dist.all_reduce(SUM, async)` (tests: a summing stand-in
        that plays the other ranks); `world` then names the emulated world size.
        `update_in_backward`: the optimizer runs bucket by bucket while backward still runs (see `arm_update`); also without
        any data parallelism (world == 1: the buckets are then only the optimizer's work units)."""
        self.store, self.group, self.overlap = store, group, overlap
        self.update_in_backward = update_in_backward
        self._hyper = None          # this step's hyper-parameter record (arm_update); None: the caller steps the optimizer
        self._opt_stream = None
        self.collective = collective
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        assert grad_dtype in ("fp32", "bf16")
        self.grad_dtype = grad_dtype
        self._stage = None     # bf16 staging buffer (grad_dtype == "bf16"), allocated on first use
        n = store.trainable_end
        per, tail = max(1, bucket_bytes // 4), max(1, min(tail_bytes, bucket_bytes) // 4)
        params = []
        for gi in range(6):
            for name, p in store.groups[gi]:
                if store.offset[id(p)] < n:
                    params.append((store.offset[id(p)], gi, p))
        params.sort(key=lambda t: t[0])
        cuts = {0, n}
        i = 0
        while i < len(params):
            j = i
            while j < len(params) and params[j][1] == params[i][1]:
                j += 1
            offs = [t[0] for t in params[i:j]]                   # one optimizer group: [g0, g1) of the flat buffer
            g0, g1 = offs[0], (params[j][0] if j < len(params) else n)
            cuts.update((g0, g1))
            # the group's first parameters (>= `tail` elements of them) are the gradients that arrive last: their own bucket
            t_cut = next((off for off in offs if off - g0 >= tail), g1)
            cuts.add(t_cut)
            hi = g1                                              # the rest: walk the group backwards (= backward execution
            for off in reversed(offs):                           # order), a full bucket every `per` elements
                if off <= t_cut:
                    break
                if hi - off >= per:
                    cuts.add(off)
                    hi = off
            i = j
        params.sort(key=lambda t: t[0])
        bounds = sorted(cuts)
        self.bounds = bounds
        self.nb = len(bounds) - 1
        self.bucket_of = {}
        self.bucket_group = [0] * self.nb     # buckets never span optimizer groups
        bi = 0
        for off, gi, p in params:
            while off >= bounds[bi + 1]:
                bi += 1
            self.bucket_of[id(p)] = bi
            self.bucket_group[bi] = gi
        params = [(off, p) for off, _, p in params]
        self.expected = None   # reports per parameter per step, learned on the first step
        self.exposed_ms = None  # a list when the caller wants finish()'s exposed-communication times (bench.py)
        self.late = None
        self.glue = set()      # parameters that (also) receive a gradient through autograd's AccumulateGrad
        self._params = [p for _, p in params]
        self._hooks = []
        self.reset()

    def relearn(self):
        """Forget the learned report counts (the set of objectives / the graph changed)."""
        self.expected = None
        self._bucket_streams = [set() for _ in range(self.nb)]
        self.reset()

    def reset(self):
        self.count = {}
        if self.expected is None:
            self.pending = [-1] * self.nb          # learning step: nothing launches before finish()
        else:
            self.pending = [0] * self.nb
            for pid, bi in self.bucket_of.items():
                n = 0 if pid in self.glue else self.expected.get(pid, 0)
                # a parameter that never reports gets its gradient through autograd glue at an unknown time: its
                # bucket is left to finish()
                self.pending[bi] = -1 if (n == 0 or self.pending[bi] < 0) else self.pending[bi] + n
        self.launched = [False] * self.nb
        self.handles = []
        self._order = []
        if not hasattr(self, "_bucket_streams"):
            self._bucket_streams = [set() for _ in range(self.nb)]
        self._bucket_events = [dict() for _ in range(self.nb)]
        self._multi_stream = bool(getattr(self.store, "streams", None))
        self._complete = []        # (bucket, handle) in completion order, for the in-backward optimizer
        self._updated = [False] * self.nb

    def attach(self):
        ops.grad_ready_hook = self.on_grad_ready if (self.world > 1 or self.update_in_backward) else None
        if self.world > 1 and os.environ.get("M3AE_DDP_KEEP_PERSISTENT") != "1":   # (the env: A/B runs of the rehearsal only)
            # RCCL's kernels run next to backward and hold some CUs: the persistent NT kernel (static tile lists, one
            # workgroup per CU) would wait for them with a whole tile list in hand; every GEMM descriptor issued while the
            # reducer is attached asks for the one-tile-per-workgroup launch (restored by detach())
            self._saved_no_persist = ops.NT_NO_PERSISTENT
            ops.NT_NO_PERSISTENT = True
        if (self.world > 1 or self.update_in_backward) and not self._hooks:
            # a gradient that autograd itself accumulates (glue ops around the kernels, e.g. `x + positional_embedding`
            # in the masked-image pass) arrives at a time the kernels' reports say nothing about: such parameters keep
            # their bucket for finish(), even when they ALSO report in-place contributions
            # (a tensor hook fires only when autograd really delivers a gradient tensor for the leaf; the kernels'
            # Functions return None for their parameters)
            for p in self._params:
                if p.requires_grad:
                    self._hooks.append(p.register_hook(lambda g, p=p: self._on_autograd_grad(p, g)))
        return self

    def _on_autograd_grad(self, p, g):
        pid = id(p)
        if g is not None and pid in self.bucket_of:
            if self.expected is not None and pid not in self.glue and self.launched[self.bucket_of[pid]]:
                self.late = self.store.names.get(pid, "?") if hasattr(self.store, "names") else "?"
            self.glue.add(pid)
        return None

    def detach(self):
        ops.grad_ready_hook = None
        if getattr(self, "_saved_no_persist", None) is not None:
            ops.NT_NO_PERSISTENT, self._saved_no_persist = self._saved_no_persist, None
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _launch(self, bi):
        if self.launched[bi]:
            return
        self.launched[bi] = True
        self._order.append(bi)
        a, b = self.bounds[bi], self.bounds[bi + 1]
        if b > a and self.world <= 1:
            self._complete.append((bi, None))       # nothing to exchange: the bucket is only the optimizer's work unit
        elif b > a:
            # the bucket's gradients may have been written from more than one stream (the module's text half runs on a side
            # stream): the launching stream -- which the collective's own stream waits for -- first waits for the last report
            # of this bucket on every OTHER stream (an event recorded right behind that wgrad launch: nothing later on that
            # stream is waited for, so the two halves keep running beside each other)
            evs = self._bucket_events[bi]
            if evs:
                cur = torch.cuda.current_stream()
                here = cur.cuda_stream or 0
                for sid, (st, ev) in evs.items():
                    if sid != here:
                        if ev is not None:
                            cur.wait_event(ev)
                        else:
                            cur.wait_stream(st)
            buf = self.store.grad[a:b]
            if self.grad_dtype == "bf16":
                if self._stage is None:
                    self._stage = torch.empty(self.store.trainable_end, dtype=torch.bfloat16, device=self.store.grad.device)
                buf = self._stage[a:b]
                buf.copy_(self.store.grad[a:b])      # one rounding per addend; stream-ordered before the collective
            if self.collective is not None:
                h = self.collective(buf)
            else:
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append(h)
            self._complete.append((bi, h))

    # ---- the optimizer inside backward -----------------------------------------------------------------------------------
    def arm_update(self, **kw):
        """Call before backward (every step): this step's optimizer hyper-parameters (arguments of ParamStore.begin_update).
        From the second step on, a bucket's AdamW update is then issued on a separate stream as soon as the NEXT bucket has
        completed -- at that point every backward kernel that reads the bucket's weights has been enqueued, and the update
        waits for all of them (events on the compute streams) and for the bucket's all-reduce -- so the 9.9 GB of optimizer
        traffic (HBM-bound, ~30 registers per lane: its waves fit beside the GEMMs' on a CU) runs under the compute-bound
        rest of backward instead of after it.  `finish()` updates what is left, refreshes the transposed weight copies and
        joins; the caller must NOT call the optimizer's step afterwards."""
        assert self.update_in_backward
        # the early update of a bucket assumes every later reader of its weights is ordered behind events already recorded: true when
        # the GEMMs read the bf16 shadows / transposed copies the update rewrites LAST (finish()), not the fp32 masters.  A "late
        # contribution" error raised by this mode leaves the parameters partially stepped (the buckets updated so far).
        assert self.store.compute_dtype == torch.bfloat16, "update_in_backward needs the bf16 shadow / transposed-copy layout"
        self._hyper = self.store.begin_update(**kw)

    def _update_complete(self, keep_last):
        todo = [(bi, h) for bi, h in self._complete if not self._updated[bi]]
        if keep_last:
            todo = todo[:-keep_last]
        if not todo:
            return
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream()
        os_ = self._opt_stream
        cur = torch.cuda.current_stream()
        for st in {cur, *getattr(self.store, "streams", ())}:
            os_.wait_event(st.record_event())       # everything enqueued so far on the compute streams
        with torch.cuda.stream(os_):
            for bi, h in todo:
                a, b = self.bounds[bi], self.bounds[bi + 1]
                if h is not None:
                    h.wait()                        # stream-ordered behind the bucket's all-reduce
                if self.grad_dtype == "bf16" and self._stage is not None and self.world > 1:
                    self.store.grad[a:b].copy_(self._stage[a:b])
                self.store.adamw_range(a, b, self.bucket_group[bi], self._hyper)
                self._updated[bi] = True

    def _note_stream(self, bi):
        """A report for bucket `bi` on the current stream (called right behind the wgrad launch)."""
        sid = ops._stream().value or 0           # raw handle: no Stream object on the common path (~700 reports per step)
        known = self._bucket_streams[bi]
        if len(known) > 1:                       # a bucket both halves write into: remember where its last write on this stream is
            cur = torch.cuda.current_stream()
            self._bucket_events[bi][sid] = (cur, cur.record_event())
        elif sid not in known:                   # the schedule changed after the learning step: fall back to whole-stream waits
            known.add(sid)
            for s2 in getattr(self.store, "streams", ()):
                self._bucket_events[bi].setdefault(s2.cuda_stream or 0, (s2, None))

    def on_grad_ready(self, p):
        pid = id(p)
        if pid not in self.bucket_of:
            return
        c = self.count.get(pid, 0) + 1
        self.count[pid] = c
        if self.expected is None:
            if getattr(self.store, "streams", None):    # learning step: which streams write into which bucket
                self._bucket_streams[self.bucket_of[pid]].add(ops._stream().value or 0)
            return
        bi = self.bucket_of[pid]
        if self._multi_stream:
            self._note_stream(bi)
        if c > self.expected.get(pid, 0):
            if self.launched[bi]:   # a contribution landed after the bucket's all-reduce was issued
                self.late = self.store.names.get(pid, "?") if hasattr(self.store, "names") else "?"
            return
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and self.overlap:
            self._launch(bi)
            if self._hyper is not None:
                self._update_complete(keep_last=1)

    def finish(self):
        """After backward: reduce every bucket not yet launched, wait for all, re-arm for the next step."""
        late_any = self.late is not None
        if self.world > 1:
            for bi in range(self.nb):
                self._launch(bi)
            if self.collective is None:
                # the error below must be raised on EVERY rank or on none: a rank that raises alone leaves the others
                # waiting in their next collective.  One more (4-byte) all-reduce in the same queue as the buckets.
                flag = torch.tensor([1.0 if late_any else 0.0], device=self.store.grad.device)
                self.handles.append(dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
            ev = None
            if self.exposed_ms is not None:             # bench.py: how long the caller's stream waits for collectives backward did not cover
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for h in self.handles:
                if h is not None:
                    h.wait()
            if ev is not None:
                ev[1].record()
            if self.grad_dtype == "bf16" and self._stage is not None:
                for bi in self._order:                # reduced bf16 -> the fp32 buffer the optimizer reads
                    a, b = self.bounds[bi], self.bounds[bi + 1]
                    if b > a and not self._updated[bi]:
                        self.store.grad[a:b].copy_(self._stage[a:b])
            if self.collective is None:
                late_any = flag.item() > 0
            if ev is not None:
                ev[1].synchronize()
                self.exposed_ms.append(ev[0].elapsed_time(ev[1]))
        if self._hyper is not None:
            if self.world <= 1:
                for bi in range(self.nb):
                    self._launch(bi)
            done_early = sum(self._updated)
            cur = torch.cuda.current_stream()
            if self._opt_stream is not None:
                cur.wait_stream(self._opt_stream)       # updates issued during backward
            for bi, _ in self._complete:                # the rest, here: backward is over, the collectives were waited for
                if not self._updated[bi]:
                    self.store.adamw_range(self.bounds[bi], self.bounds[bi + 1], self.bucket_group[bi], self._hyper)
                    self._updated[bi] = True
            assert all(self._updated[bi] or self.bounds[bi + 1] == self.bounds[bi] for bi in range(self.nb))
            self.store.sync_shadows(cast=False)
            self.updated_in_backward = done_early       # (tests / bench: how many buckets ran under backward)
            self._hyper = None
        if late_any:
            name, self.late = self.late or "<on another rank>", None
            self.reset()
            raise RuntimeError(f"gradient of {name!r} was accumulated after its bucket had been all-reduced (more "
                               "contributions than on the reducer's first step): call reducer.relearn() when the set of "
                               "objectives changes")
        if self.expected is None:
            self.expected = dict(self.count)
        self.finish_order = list(self._order)      # buckets in the order their all-reduce was issued on this step
        self.reset()

    def bucket_bytes_list(self):
        return [(self.bounds[i + 1] - self.bounds[i]) * 4 for i in range(self.nb)]

    @property
    def grad_scale(self):
        return 1.0 / self.world
