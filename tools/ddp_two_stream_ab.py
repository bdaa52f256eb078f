"""A/B of the data-parallel reducer beside the two-stream schedule (one GPU, rehearsal mode of bench.py): which part of the
reducer costs step time when the text half runs on its own stream.   python tools/ddp_two_stream_ab.py <variant>
variants: base | no_overlap (all buckets at finish()) | no_wait (no cross-stream waits: UNSAFE, timing only) |
          no_collective (buckets counted, nothing launched) | no_hooks (reducer world = 1)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
from m3ae_amd import ddp  # noqa: E402

R = ddp.FlatGradReducer
if variant == "no_overlap":
    orig = R.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.overlap = False
    R.__init__ = init
elif variant == "no_wait":
    R._note_stream = lambda self, bi: None
elif variant == "no_collective":
    class H:
        def wait(self):
            pass
    orig = R.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.collective = lambda buf: H()
    R.__init__ = init
elif variant == "pg_only":        # RCCL initialised, no reducer hooks
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29545")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
elif variant == "hooks_only":     # reducer hooks and bucket bookkeeping, no process group, nothing launched
    class H2:
        def wait(self):
            pass
    orig = R.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.world = 2
        self.collective = lambda buf: H2()
    R.__init__ = init
import bench  # noqa: E402

sys.argv = ["bench.py", "--no-secondary", "--no-cpu-baseline", "--no-roofline"] + ([] if variant in ("no_hooks", "pg_only", "hooks_only") else ["--rehearse-ddp"])
bench.main()
