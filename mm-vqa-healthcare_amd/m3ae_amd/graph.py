"""One training step (zero_grad -> training_step -> backward -> fused AdamW) as ONE hipGraph, replayed per step.

Why: at the per-GPU batch the reference's run scripts name (32: run_scripts/finetune_m3ae_decoder.sh:1-2, pretrain_m3ae.sh:1-2) a
step is ~1000 launches of 10-80 us each and the host needs longer to enqueue them (35 ms of Python + ctypes) than the device
needs to run them: the step is host-bound (profiles/r03_host_bound_B32.log).  A replayed graph costs the host one launch.

What has to live in device memory for a replay to be a NEW step (kernel arguments are frozen at capture):
  * dropout: every site's seed is frozen; the kernels fold `ops.DROPOUT_SALT` (a device uint32 this class bumps per replay) into
    the mask key (include/m3ae_hip.h, ABI 3), so every replay draws fresh masks at every site;
  * AdamW: learning rate and bias-corrected step size per optimizer group come from `ParamStore.hyper_dev` (uploaded per replay:
    the host still evaluates the reference's schedule, m3ae_utils.py:212-240, exactly as the eager step does);
  * the batch: the caller's batch tensors are the graph's static inputs (copy new data INTO them between replays); everything
    the step reads must already be a device tensor -- e.g. `batch["vqa_targets"]` (objectives.build_vqa_targets): a host-built
    tensor uploaded inside the captured region would be read from a stale host address at every replay.

The model's second HIP stream (modules/m3ae_module.py::_fusion_two_streams) forks from and joins the capturing stream through
events, so the graph keeps the text half and the image half as parallel branches.  Data-parallel runs (bucketed all-reduce from
backward hooks) stay eager: this class is for N = 1 or for the steps between gradient exchanges.
"""
import torch

from . import ops


class GraphedStep:
    def __init__(self, model, batch, max_steps, grad_scale=1.0):
        self.model, self.store, self.batch = model, model.store, batch
        self.max_steps, self.grad_scale = max_steps, grad_scale
        dev = self.store.flat.device
        self.salt = torch.zeros(1, dtype=torch.int32, device=dev)
        self.hyper_dev = torch.zeros(6, 2, dtype=torch.float32, device=dev)
        self._salt_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._hyper_host = torch.zeros(6, 2, dtype=torch.float32).pin_memory()
        self.graph, self.loss = None, None
        self.replays = 0

    # the captured body: exactly the eager step of bench.py / trainer.py
    def _body(self, hyper):
        self.store.zero_grad()
        out = self.model.training_step(self.batch)
        loss = out["loss"] if isinstance(out, dict) else out
        loss.backward()
        self.store.adamw_apply(hyper)
        return loss.detach()

    def _upload(self, hyper):
        self._hyper_host.copy_(torch.tensor(self.store.hyper_values(hyper), dtype=torch.float32))
        self._salt_host[0] = hyper["step"] & 0x7FFFFFFF
        self.hyper_dev.copy_(self._hyper_host, non_blocking=True)
        self.salt.copy_(self._salt_host, non_blocking=True)

    def capture(self):
        """Capture the step.  Call after at least one eager step (lazy buffers, kernel attributes and the optimizer state exist)."""
        st = self.store
        if "vqa_labels" in self.batch and "vqa_targets" not in self.batch:
            raise ValueError('a graphed step needs batch["vqa_targets"] on the device (objectives.build_vqa_targets)')
        if st.exp_avg is None:
            raise RuntimeError("run one eager step before capturing (the optimizer state is allocated lazily)")
        hyper = st.begin_update(self.max_steps, self.grad_scale)   # this first graphed step is a real step
        self._upload(hyper)
        torch.cuda.synchronize()
        old_salt, old_hd = ops.DROPOUT_SALT, st.hyper_dev
        ops.DROPOUT_SALT, st.hyper_dev = self.salt, self.hyper_dev
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self.loss = self._body(hyper)
        except Exception:
            ops.DROPOUT_SALT, st.hyper_dev = old_salt, old_hd
            st.step_count -= 1
            raise
        ops.DROPOUT_SALT, st.hyper_dev = old_salt, old_hd
        self.graph = g
        g.replay()          # capture only records: run the step it stands for
        self.replays = 1
        return self.loss

    def step(self):
        if self.graph is None:
            return self.capture()
        self._upload(self.store.begin_update(self.max_steps, self.grad_scale))
        self.graph.replay()
        self.replays += 1
        return self.loss
