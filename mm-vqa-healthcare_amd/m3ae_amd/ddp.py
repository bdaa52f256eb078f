"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI; gloo on CPU for tests).

The reference's only parallelism is PyTorch-Lightning DDP (main.py:59-63): replicate weights, shard the global
batch, all-reduce gradients.  Here the gradients already live in ONE flat fp32 buffer (ParamStore.grad), so a
bucket is a contiguous byte range of it -- no gather/scatter copies -- and buckets are cut by BYTES, sized for
xGMI: RCCL rings are per-link bound (7 links x ~153 GB/s per GPU), so buckets are large (default 64 MiB) to
amortise launch + ring latency.  The wgrad kernels accumulate straight into the bucket memory and report each
finished parameter through `ops.grad_ready_hook`; when the last parameter of a bucket reports, its all-reduce is
launched asynchronously (torch.distributed orders it after the producing kernels and runs it on RCCL's own stream),
overlapping the rest of backward.  `finish()` launches whatever is left (parameters whose gradient flows through
autograd glue never report) and waits.  Averaging (1 / world_size) is folded into the fused AdamW kernel's
`grad_scale`, not applied to the buffer.  The 6 tensors that never receive a gradient are not in the
buffer at all -- no `find_unused_parameters` pass.
"""
import torch
import torch.distributed as dist

from . import ops


class FlatGradReducer:
    def __init__(self, store, bucket_bytes=64 << 20, group=None, overlap=True):
        self.store, self.group, self.overlap = store, group, overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = store.trainable_end
        per = max(1, bucket_bytes // 4)
        # bucket boundaries on parameter boundaries
        bounds, cur = [0], 0
        params = []
        for gi in range(6):
            for name, p in store.groups[gi]:
                params.append((store.offset[id(p)], p))
        params.sort(key=lambda t: t[0])
        for off, p in params:
            if off - bounds[-1] >= per:
                bounds.append(off)
        bounds.append(n)
        self.bounds = bounds
        self.nb = len(bounds) - 1
        self.bucket_of, self.pending0 = {}, [0] * self.nb
        bi = 0
        for off, p in params:
            while off >= bounds[bi + 1]:
                bi += 1
            self.bucket_of[id(p)] = bi
            self.pending0[bi] += 1
        self.reset()

    def reset(self):
        self.pending = list(self.pending0)
        self.seen = set()
        self.launched = [False] * self.nb
        self.handles = []

    def attach(self):
        ops.grad_ready_hook = self.on_grad_ready if self.world > 1 else None
        return self

    def detach(self):
        ops.grad_ready_hook = None

    def _launch(self, bi):
        if self.launched[bi]:
            return
        self.launched[bi] = True
        a, b = self.bounds[bi], self.bounds[bi + 1]
        if b > a:
            h = dist.all_reduce(self.store.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append(h)

    def on_grad_ready(self, p):
        pid = id(p)
        if pid in self.seen or pid not in self.bucket_of:
            return
        self.seen.add(pid)
        bi = self.bucket_of[pid]
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and self.overlap:
            self._launch(bi)

    def finish(self):
        """After backward: reduce every bucket not yet launched, wait for all, re-arm for the next step."""
        if self.world > 1:
            for bi in range(self.nb):
                self._launch(bi)
            for h in self.handles:
                h.wait()
        self.reset()

    @property
    def grad_scale(self):
        return 1.0 / self.world
