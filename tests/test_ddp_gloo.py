"""CPU, 2 processes over gloo: the data-parallel gradient exchange (m3ae_amd/ddp.py) -- bucket cutting on the flat
gradient buffer, launch-on-last-gradient overlap, finish() draining, SUM semantics + 1/world in the optimizer scale.
(The same reducer runs over RCCL on the GPUs; only the process group differs.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from m3ae_amd.ddp import FlatGradReducer
from m3ae_amd.param_store import ALIGN, param_group_of


class FakeStore:
    """Host-side stand-in for ParamStore's layout (same grouping / offset rules, CPU buffers)."""

    def __init__(self, named):
        self.groups = [[] for _ in range(7)]
        for n, p in named:
            self.groups[param_group_of(n)].append((n, p))
        self.offset, off = {}, 0
        for g in self.groups[:6]:
            for n, p in g:
                self.offset[id(p)] = off
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.trainable_end = off
        self.grad = torch.zeros(off)
        for g in self.groups[:6]:
            for n, p in g:
                o = self.offset[id(p)]
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


def make_named():
    torch.manual_seed(0)
    shapes = {
        "vision_encoder.visual.transformer.resblocks.0.mlp.c_fc.weight": (96, 32),
        "vision_encoder.visual.transformer.resblocks.0.mlp.c_fc.bias": (96,),
        "language_encoder.encoder.layer.0.output.dense.weight": (32, 96),
        "language_encoder.encoder.layer.0.output.LayerNorm.weight": (32,),
        "multi_modal_vision_layers.0.attention.self.query.weight": (32, 32),
        "multi_modal_vision_layers.0.attention.self.key.weight": (32, 32),
        "multi_modal_vision_layers.0.attention.self.query.bias": (32,),
        "vqa_head.0.weight": (64, 64),
        "vqa_head.0.bias": (64,),
        "vqa_head.3.weight": (50, 64),
    }
    return [(n, torch.nn.Parameter(torch.zeros(s))) for n, s in shapes.items()]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, out, grad_dtype="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        named = make_named()
        store = FakeStore(named)
        red = FlatGradReducer(store, bucket_bytes=bucket_bytes, grad_dtype=grad_dtype).attach()
        assert red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
        for step in range(3):
            store.grad.zero_()
            # "backward": gradients appear in reverse parameter order; one parameter receives TWO in-place
            # contributions per step (a weight used twice) and reports after each
            launched_before_finish = 0
            for i, (n, p) in enumerate(reversed(named)):
                if n == "vqa_head.0.bias":
                    p.grad.fill_(float(rank + 1) * (i + 1) + step)
                    continue  # a gradient that flows through autograd glue never reports; finish() must cover it
                if n == "language_encoder.encoder.layer.0.output.dense.weight":
                    p.grad.fill_(1000.0)                       # first contribution
                    red.on_grad_ready(p)
                    bi = red.bucket_of[id(p)]
                    assert not red.launched[bi]                # ... must not release the bucket
                p.grad.fill_(float(rank + 1) * (i + 1) + step)
                red.on_grad_ready(p)
            launched_before_finish = sum(red.launched)
            if step == 0:
                assert launched_before_finish == 0             # learning step: everything waits for finish()
            glue = dict(named)["vqa_head.0.bias"]              # never reports: its bucket must wait for finish()
            assert not red.launched[red.bucket_of[id(glue)]]
            red.finish()
            for i, (n, p) in enumerate(reversed(named)):
                expect = sum(float(r + 1) * (i + 1) + step for r in range(world))
                assert torch.all(p.grad == expect), (n, p.grad.flatten()[0].item(), expect)
            # alignment padding between parameters stays zero
            used = torch.zeros_like(store.grad, dtype=torch.bool)
            for n, p in named:
                o = store.offset[id(p)]
                used[o:o + p.numel()] = True
            assert torch.all(store.grad[~used] == 0)
            assert sum(red.launched) == 0  # re-armed
        out.put((rank, red.nb, launched_before_finish))
    finally:
        red.detach()
        dist.destroy_process_group()


@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("bucket_bytes", [1 << 30, 4096])
def test_flat_grad_reducer_two_ranks_gloo(bucket_bytes, grad_dtype):
    """grad_dtype = "bf16": the buckets travel as bf16 (half the bytes); the test's gradients are small integers, exact in
    bf16, so the reduced values are still exact."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out, grad_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(out.get(timeout=10) for _ in range(2))
    nb = res[0][1]
    ngroups = len({param_group_of(n) for n, _ in make_named()})
    if bucket_bytes == 4096:
        assert nb > ngroups                # several buckets per group, on parameter boundaries
        assert res[0][2] >= nb - 2         # most buckets were launched during "backward" (overlap), not in finish()
    else:
        assert nb == ngroups               # buckets never span optimizer groups (each group is walked from its end)
        assert res[0][2] == nb - 1         # ... and only the bucket of the unreported gradient waits for finish()


def test_buckets_are_cut_from_the_end_of_each_group_and_the_last_to_finish_is_small():
    """Backward completes a group's parameters from its end towards its start: full buckets are cut walking the group
    backwards, what is left at its start (the gradients that arrive last) in pieces of at most tail_bytes -- the all-reduce
    that nothing overlaps is the small one (ddp.py, "Bucket order")."""
    torch.manual_seed(0)
    named = [(f"vision_encoder.visual.transformer.resblocks.{i}.mlp.c_fc.weight", torch.nn.Parameter(torch.zeros(64, 40)))
             for i in range(12)]
    named += [(f"vision_encoder.visual.transformer.resblocks.{i}.mlp.c_fc.bias", torch.nn.Parameter(torch.zeros(64)))
              for i in range(12)]
    store = FakeStore(named)
    per_param = 64 * 40 * 4
    red = FlatGradReducer(store, bucket_bytes=4 * per_param, tail_bytes=per_param, world=2,
                          collective=lambda t: None)
    sizes = red.bucket_bytes_list()
    g_w = param_group_of(named[0][0])
    w_offs = sorted(store.offset[id(p)] for n, p in named if param_group_of(n) == g_w)
    first_bucket = red.bucket_of[id(named[0][1])]            # block 0's weight: the group's start, finishes last
    assert sizes[first_bucket] <= per_param + ALIGN * 4
    assert max(sizes) <= 5 * per_param                       # a full bucket + at most one parameter
    assert red.bounds[0] == 0 and red.bounds[-1] == store.trainable_end
    red.attach()
    try:
        for step in range(2):
            for n, p in reversed(named):                     # "backward": reverse module order
                red.on_grad_ready(p)
            red.finish()
        # on the second step buckets were released as they completed: the weight group's buckets in descending offset order,
        # its small first bucket last
        order_w = [bi for bi in red.finish_order if red.bounds[bi] in w_offs or red.bounds[bi] == w_offs[0]]
        assert order_w[-1] == first_bucket
        assert order_w == sorted(order_w, reverse=True)
    finally:
        red.detach()


def test_bucket_boundaries_are_parameter_aligned_and_cover_buffer():
    named = make_named()
    store = FakeStore(named)
    red = FlatGradReducer(store, bucket_bytes=2048)
    assert red.bounds[0] == 0 and red.bounds[-1] == store.trainable_end
    offs = {store.offset[id(p)] for _, p in named}
    assert all(b in offs or b == store.trainable_end for b in red.bounds)
    assert len(red.bucket_of) == len(named) and all(0 <= b < red.nb for b in red.bucket_of.values())
