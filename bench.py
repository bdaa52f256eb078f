"""bench.py -- image-question pairs/s of the M3AE Med-VQA fine-tuning step on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic batch resident in HBM: forward (ViT-B/16 @384 -> RoBERTa-base
-> 6 co-attention layers -> poolers -> vqa_head -> BCE), backward, bucketed gradient all-reduce (N > 1) and the
fused AdamW update, bf16 storage / fp32 accumulation (configs[1] of BASELINE.json).  Per-GPU batch is fixed as N
grows (weak scaling, as DDP shards a global batch).  Rank 0 prints ONE JSON line.

Extra legs after the timed region (rank 0):
  roofline     -- every GEMM / attention launch of two further steps is bracketed by HIP events on its launch
                  stream; `roofline` reports the dominant kernel (the bf16 MFMA "NT" GEMM): algorithmic FLOPs of
                  its launches / their summed durations vs the 2.5 PFLOP/s dense bf16 peak, plus the fused
                  cross-attention forward (north_star's kernel) at the batch it ran.
  cpu_baseline -- the oracle (CPU restatement of the reference path, fp32) timed on this node's host cores on a
                  bounded sample (B = 2, fwd + bwd), N = 1 only.  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "mm-vqa-healthcare_amd"), ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP initialises: see m3ae_amd/__init__.py (second HIP stream beside RCCL's)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DOM_KINDS = ("gemm:mfma_nt_pp2", "gemm:mfma_nt_pp")   # ops.PROFILE record kinds of the ping-pong NT GEMM kernels
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
FWD_GFLOP_PER_SAMPLE = 183.8
STEP_GFLOP_PER_SAMPLE = 551.3  # fwd + bwd
XATTN_FWD_GFLOP_PER_SAMPLE = 17.922  # 6 layers x 2 directions, projections + SDPA


_T0 = time.time()


def log(msg):
    """Progress to stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Threads for the CPU legs: every core this process may actually use -- the affinity mask, capped by the cgroup CPU
    quota (a GPU box hands one GPU's share of the host, e.g. 16 of its cores; 200 threads on a 16-core quota thrash)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    # without a readable quota: one GPU's share of a GPU box is 16 host cores (more threads than that thrash: a 7-minute
    # stall was measured with the whole affinity mask); M3AE_CPU_THREADS overrides
    n = min(n, int(os.environ.get("M3AE_CPU_THREADS", 16)))
    return max(1, n)


def profile_commit(path):
    """Short hash of the commit that last touched `path` (HEAD when path is None); "unknown" outside a git checkout."""
    import subprocess
    try:
        cmd = ["git", "-C", ROOT, "log", "-1", "--format=%h"] + (["--", path] if path else [])
        return subprocess.run(cmd, capture_output=True, text=True, timeout=10).stdout.strip() or "unknown"
    except Exception:  # noqa: BLE001
        return "unknown"


def to_dev(batch, dev):
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev)
        elif isinstance(v, list) and v and isinstance(v[0], torch.Tensor):
            out[k] = [t.to(dev) for t in v]
        else:
            out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("M3AE_BENCH_BATCH", 256)), help="per-GPU batch")
    ap.add_argument("--head", choices=["cls", "t5", "pretrain"], default="cls",
                    help="cls: configs[1] full fine-tune with the classification head (default, the timed metric); "
                         "t5: configs[2] frozen M3AE + T5 generative head (main_t5_m3ae.py recipe)")
    ap.add_argument("--t5", default="t5-small", help="t5-small (reference-faithful) | t5-base (BASELINE configs[2])")
    ap.add_argument("--arch", choices=["base", "large"], default="base",
                    help="large: configs[4] towers (ViT-L/16 + RoBERTa-large, 512x512; fusion stays 768 / 6 layers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary line (configs[2]: frozen M3AE + t5-base head at per-GPU batch 64, run as a child "
                         "process after the main measurement; N = 1 default run only)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size bf16-vs-reference-fixture error report")
    ap.add_argument("--no-dropout", action="store_true", help="eval-mode semantics (A/B of the dropout cost)")
    ap.add_argument("--ddp-grad-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="gradient bucket dtype of the data-parallel all-reduce (bf16: half the bytes over xGMI, ddp.py)")
    ap.add_argument("--optimizer-in-backward", action="store_true",
                    help="AdamW bucket by bucket on its own stream under backward (ddp.FlatGradReducer.arm_update; measured "
                         "+0.7 %% at per-GPU batch 256, -1.8 %% at 32: not the default)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as ONE hipGraph (m3ae_amd/graph.py: dropout salt and AdamW hyper-parameters in device memory). "
                         "auto = off: on ROCm 7.2 a replay of this ~1000-node graph costs the host MORE than the eager launches and "
                         "serialises the two HIP streams (profiles/r04_hipgraph_replay_B32_measured.log)")
    ap.add_argument("--rehearse-ddp", action="store_true",
                    help="single GPU only: run the N > 1 code path (one-rank RCCL group, bucketed async all-reduces from the "
                         "backward hooks, one-tile-per-workgroup NT launches) -- a rehearsal of the scaling run, not a metric")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: everything else this process (or a library in it: RCCL prints a version
    # banner to stdout when its first communicator is created) writes to fd 1 goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from m3ae_amd.ddp import rccl_group_options   # (imported before anything initialises HIP: m3ae_amd sets GPU_MAX_HW_QUEUES)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=rccl_group_options())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.rehearse_ddp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=rccl_group_options())

    from m3ae_amd import ops, synth
    from m3ae_amd.config import finetune_vqa_rad_config
    from m3ae_amd.ddp import FlatGradReducer
    from m3ae_amd.modules import M3AETransformerSS
    from m3ae_amd.modules.objectives import build_vqa_targets

    torch.set_num_threads(host_cores())
    ops.use_launch_stream()   # the whole run is issued on the library's high-priority stream (M3AE_LAUNCH_PRIORITY=normal: the default stream)
    log(f"rank {rank}/{world} on {dev}: building model (launch stream priority {torch.cuda.current_stream().priority})")
    cfg = finetune_vqa_rad_config(compute_dtype="bf16", t5_model_name=args.t5)
    if args.head == "pretrain":  # configs[3]: MLM + MIM + ITM @384, 64 text tokens (task_pretrain_m3ae, roberta vocabulary)
        from m3ae_amd.config import compose
        cfg = compose("task_pretrain_m3ae", "clip16", "text_roberta", image_size=384, max_text_len=64, compute_dtype="bf16")
        args.no_cpu_baseline = args.no_roofline = True
    if args.arch == "large":  # configs[4]: no named config changes the fusion stack (config.py:45-51)
        from m3ae_amd.config import compose
        cfg = compose("task_finetune_vqa_vqa_rad", "clip16", "text_roberta", vit="ViT-L/16", tokenizer="roberta-large",
                      input_image_embed_size=1024, input_text_embed_size=1024, image_size=512, compute_dtype="bf16",
                      t5_model_name=args.t5)
        args.no_cpu_baseline = args.no_roofline = True
    if args.head == "t5":
        from m3ae_amd.modules import T5VQA_MMEncoderInput
        model = T5VQA_MMEncoderInput(cfg)
        model.unfreeze_top_layers(4, 4)  # run_scripts/finetune_m3ae.sh: unfreeze_num_{encoder,decoder}_layers=4
        args.no_cpu_baseline = True      # the cpu_baseline leg times the configs[1] oracle
    else:
        model = M3AETransformerSS(cfg)
    synth.fill_deterministic(model)  # random-init weights of the named architecture (no checkpoints offline)
    model.finalize(dev, torch.bfloat16)
    # the reference trains in train() mode with drop_rate = 0.1 (m3ae/config.py:74; T5Config.dropout_rate 0.1): dropout
    # is part of the step in every mode
    model.train(not args.no_dropout)
    store = model.store
    reducer = FlatGradReducer(store, grad_dtype=args.ddp_grad_dtype, update_in_backward=args.optimizer_in_backward)
    if args.rehearse_ddp and world == 1:
        reducer.world = 2          # take the hook / bucket path; the sum over one rank is the identity
    reducer.attach()
    ddp_scale = 1.0 if (args.rehearse_ddp and world == 1) else None

    log("model resident; generating synthetic batch")
    B = args.batch
    batch = to_dev(synth.synthetic_batch(B, text_len=cfg["max_text_len"], image_size=cfg["image_size"], rank=rank,
                                         pretrain=args.head == "pretrain"), dev)
    batch["vqa_targets"] = build_vqa_targets(batch, cfg["vqa_label_size"], dev)
    if args.head == "t5":
        lab = synth.det_randint("t5_labels", 2, 32128, (B, 6), salt=31 + rank)
        lab[:, -1] = 1  # eos
        batch["t5_labels"] = lab.to(dev)
    max_steps = args.steps + args.warmup + 16

    def train_loss():
        out = model.training_step(batch)
        return out["loss"] if isinstance(out, dict) else out

    def step():
        store.zero_grad()
        gs = ddp_scale if ddp_scale is not None else reducer.grad_scale
        if reducer.update_in_backward:
            # AdamW runs bucket by bucket on its own stream while backward still runs (ddp.FlatGradReducer.arm_update):
            # every update of the step is issued and joined inside finish(), i.e. inside the timed region
            reducer.arm_update(max_steps=max_steps, grad_scale=gs)
        loss = train_loss()
        loss.backward()
        reducer.finish()
        if not reducer.update_in_backward:
            store.adamw_step(max_steps=max_steps, grad_scale=gs)
        return loss

    graphed, graph_note = None, "eager launches"
    want_graph = args.graph == "on"   # "auto" = off: measured on ROCm 7.2 a replay of the ~1000-node two-branch graph costs the
    # host 35 ms and serialises the two streams (44 ms/step against 34.6 eager at per-GPU batch 32, profiles/r04_hipgraph_replay_B32_measured.log)
    if want_graph and (world > 1 or args.rehearse_ddp or args.optimizer_in_backward):
        want_graph, graph_note = False, "eager launches (graph replay is single-GPU only: the data-parallel step exchanges buckets from backward hooks)"
    for i in range(args.warmup):
        loss = step()
        torch.cuda.synchronize()
        log(f"warm-up step {i} done, loss {loss.item():.4f}")
    if want_graph:
        from m3ae_amd.graph import GraphedStep
        reducer.detach()
        try:
            graphed = GraphedStep(model, batch, max_steps=max_steps)
            loss = graphed.step()          # capture + first replay (a real step)
            torch.cuda.synchronize()
            log(f"step captured as one hipGraph, loss {loss.item():.4f}")
            eager_step = step
            step = graphed.step
            loss = step()
            torch.cuda.synchronize()
            graph_note = "ONE hipGraph replay per step (m3ae_amd/graph.py; per-replay dropout salt and AdamW hyper-parameters in device memory)"
        except Exception as e:  # noqa: BLE001 -- keep measuring: the eager step is the fallback, and the line says so
            log(f"graph capture failed ({e!r}): eager step")
            graphed, graph_note = None, f"eager launches (graph capture failed: {e!r})"
            reducer.attach()

    # data-parallel runs: what RCCL sees and what the links deliver, recorded BEFORE the timed region so that the scaling curve can
    # be read against it (DESIGN.md 7's prediction): a 20-iteration all-reduce of 64 MiB fp32 (one gradient bucket)
    dist_info = None
    if world > 1 or args.rehearse_ddp:
        probe = torch.ones(16 * 2 ** 20, dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_reduce(probe)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(probe)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 20
        wsz = dist.get_world_size()
        dist_info = {"backend": dist.get_backend(), "world_size_seen_by_rccl": wsz,
                     "allreduce_probe": {"bytes": probe.numel() * 4, "iters": 20, "ms": round(ar_ms, 4),
                                         "algbw_GBps": round(probe.numel() * 4 / (ar_ms * 1e-3) / 1e9, 1),
                                         "busbw_GBps": round(probe.numel() * 4 / (ar_ms * 1e-3) / 1e9 * 2 * (wsz - 1) / max(wsz, 1), 1)},
                     "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
        del probe
        reducer.exposed_ms = []       # finish() appends the time its caller's stream waited for the collectives of a step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step device times (current stream)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i].record()
        loss = step()
    ev[args.steps].record()
    fence()
    dt = time.perf_counter() - t0
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(round(q * (len(step_ms) - 1))))]
    step_stats = {"median": round(pct(0.5), 3), "p10": round(pct(0.1), 3), "p90": round(pct(0.9), 3),
                  "min": round(step_ms[0], 3), "max": round(step_ms[-1], 3),
                  "how": "HIP events on the launch stream around every timed step of this rank"}
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    if dist_info is not None and getattr(reducer, "exposed_ms", None):
        ex = sorted(reducer.exposed_ms[-args.steps:])
        dist_info["exposed_comm_ms_per_step"] = {"median": round(ex[len(ex) // 2], 3), "max": round(ex[-1], 3),
                                                 "how": "HIP events on the caller's stream around FlatGradReducer.finish()'s waits "
                                                        "for the step's bucket all-reduces (what backward did not cover), this rank"}
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    final_loss = loss.item()
    log(f"timed region: {ms_per_step:.2f} ms/step, {value:.1f} pairs/s")

    roofline, xattn, kern_table = None, None, None
    if dist_info is None:
        dist_info = {"world_size_seen_by_rccl": 1, "note": "single process, no process group"}
    if rank == 0 and not args.no_roofline:
        log("roofline leg")
        reducer.detach()
        m3r = model.m3ae if args.head == "t5" else model

        def profiled_steps():
            ops.PROFILE = []
            for _ in range(2):
                store.zero_grad()
                train_loss().backward()
            torch.cuda.synchronize()
            r, ops.PROFILE = ops.PROFILE, None
            return r

        recs = profiled_steps()          # as the timed region ran: the text half on its own HIP stream beside the image half
        alone = None
        if getattr(m3r, "two_streams", False):
            m3r.two_streams = False      # the same launches with nothing beside them: the kernel's own rate
            ra = [(k, d, a_, b_) for k, d, a_, b_ in profiled_steps() if k in DOM_KINDS]
            m3r.two_streams = True
            ms_a = sum(a_.elapsed_time(b_) for _, _, a_, b_ in ra)
            fl_a = sum(2.0 * d[0] * d[1] * d[2] * d[3] for _, d, _, _ in ra)
            if ms_a > 0:
                alone = {"achieved": round(fl_a / (ms_a * 1e-3) / 1e12, 1), "frac": round(fl_a / (ms_a * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                         "avg_launch_ms": round(ms_a / len(ra), 4),
                         "how": "same launches, single-stream schedule (M3AE_TWO_STREAMS=0): no text-side kernel shares the chip"}
        agg = {}
        for kind, dims, e0, e1 in recs:
            ms = e0.elapsed_time(e1)
            if kind.startswith("gemm"):
                M, N, K, nb = dims
                fl = 2.0 * M * N * K * nb
            elif kind.startswith("xattn"):   # the reference formulation's FLOPs of one sub-block, x2 for the backward
                fl = XATTN_FWD_GFLOP_PER_SAMPLE / 12 * 1e9 * dims[0] * (1.0 if kind == "xattn_fwd" else 2.0)
            else:
                Bq, H, Lq, Lk, Dh = dims
                fl = 4.0 * Bq * H * Lq * Lk * Dh * (1.0 if kind == "attn_fwd" else 2.5)
            a = agg.setdefault(kind, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += fl
        shp = {}
        for kind, dims, e0, e1 in recs:
            if kind.startswith("gemm:mfma"):
                a = shp.setdefault((kind, dims[:3]), [0, 0.0])
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
        for (kind, (M_, N_, K_)), (n_, ms_) in sorted(shp.items(), key=lambda kv: -kv[1][1])[:16]:
            log(f"  {kind:12s} M={M_:6d} N={N_:5d} K={K_:6d}: {n_:4d} launches, avg {ms_ / n_ * 1e3:8.1f} us, "
                f"{2.0 * M_ * N_ * K_ * n_ / (ms_ * 1e-3) / 1e12:7.1f} TF/s, total {ms_ / 2:7.2f} ms/step")
        kern_table = {k: {"launches": v[0], "avg_ms": v[1] / v[0], "tflops": v[2] / (v[1] * 1e-3) / 1e12}
                      for k, v in agg.items() if v[1] > 0}
        # the dominant kernel of the step: gemm_nt_pp2_kernel (round 4; gemm_nt_pp(_persistent)_kernel before), all epilogue classes
        dom = max(DOM_KINDS, key=lambda k_: agg.get(k_, [0, 0.0, 0.0])[1])
        if dom in agg:
            n, ms, fl = agg[dom]
            ach = fl / (ms * 1e-3) / 1e12
            # data-parallel runs use the one-tile-per-workgroup launch of the same kernel body (ddp.FlatGradReducer.attach)
            roofline = {"bound": "mfma", "kernel": "gemm_nt_pp2_kernel" if dom.endswith("pp2") else
                        ("gemm_nt_pp_persistent_kernel" if world == 1 else "gemm_nt_pp_kernel"),
                        "achieved": round(ach, 1),
                        "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                        "traffic": None, "launches": n // 2, "avg_launch_ms": round(ms / n, 4),
                        "flops_per_launch": fl / n}
            if alone is not None:
                roofline["how"] = ("HIP events on the launching stream around every launch of the kernel, in the schedule of the timed "
                                   "region (the text half of the model runs on a second HIP stream, so some launches share the chip "
                                   "with text-side kernels)")
                roofline["single_stream"] = alone
            # HBM-side bytes per launch from the committed PMC passes of this same command (tools/pmc_traffic.py;
            # FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as read) -- only when the workload matches
            import glob
            tps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            tp = tps[-1] if tps else os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(tp):
                with open(tp) as f:
                    tj = json.load(f)
                if tj.get("per_gpu_batch") == B and tj.get("head") == args.head:
                    roofline["traffic"] = tj["gemm_nt_pp_kernel"]["bytes_per_launch"]
                    # the snapshot on a GPU box has no .git: the profile carries the commit of the code it measured
                    roofline["traffic_source"] = ("committed profile " + tp[len(ROOT) + 1:] + ", taken at commit " +
                                                  str(tj.get("commit", profile_commit(tp))) +
                                                  " (NOT measured in this run; HEAD is " + profile_commit(None) + ")")
                    roofline["traffic_unit"] = "bytes/launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE; " + tp[len(ROOT) + 1:] + ")"
                    # MFMA-pipe busy fraction and effective clock of the same kernel from the committed counter pass
                    # (tools/pmc_mfma.py: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))
                    mp = tp.replace("_pmc_traffic.json", "_pmc_mfma.json")
                    if os.path.exists(mp):
                        with open(mp) as f:
                            mj = json.load(f)
                        mk = [v for k, v in mj["kernels"].items() if k.startswith("gemm_nt_pp")]
                        ms_ = sum(v["total_ms"] for v in mk)
                        if ms_ > 0:
                            roofline["pmc_source"] = ("committed profile " + mp[len(ROOT) + 1:] + ", taken at commit " +
                                                      str(mj.get("commit", profile_commit(mp))) +
                                                      " (NOT measured in this run)")
                            roofline["mfma_busy_pmc"] = round(sum(v["mfma_util"] * v["total_ms"] for v in mk) / ms_, 3)
                            roofline["clock_mhz_pmc"] = round(sum(v["clock_mhz"] * v["total_ms"] for v in mk) / ms_)
        # fused cross-attention forward (all 6 layers, both directions), HIP events around the sub-blocks
        m3 = model.m3ae if args.head == "t5" else model
        with torch.no_grad():
            dt_ = torch.bfloat16
            x = torch.randn(B, 32, 768, device=dev).to(dt_)
            y = torch.randn(B, 577, 768, device=dev).to(dt_)
            mt = m3.language_encoder.get_extended_attention_mask(batch["text_masks"]).contiguous()

            def xattn_all():
                for tl, il in zip(m3.multi_modal_language_layers, m3.multi_modal_vision_layers):
                    tl.crossattention(x, None, y, None)
                    il.crossattention(y, None, x, mt)
            def timed(fn):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 5

            ms = timed(xattn_all)
            ms_two = None
            if getattr(m3, "two_streams", False):
                # the same 12 sub-blocks the way M3AETransformerSS.infer issues them: text queries on the side stream, image
                # queries on the caller's stream, a layer's two directions beside each other, layers in order on each stream
                side = m3._side()

                def xattn_two_streams():
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    for tl, il in zip(m3.multi_modal_language_layers, m3.multi_modal_vision_layers):
                        with torch.cuda.stream(side):
                            tl.crossattention(x, None, y, None)
                        il.crossattention(y, None, x, mt)
                    main.wait_stream(side)
                ms_two = timed(xattn_two_streams)
            tf = XATTN_FWD_GFLOP_PER_SAMPLE * B / 1e3
            xattn = {"batch": B, "ms": round(ms, 3), "tflop": round(tf, 3), "achieved": round(tf / (ms * 1e-3), 1),
                     "unit": "TFLOP/s", "frac": round(tf / (ms * 1e-3) / PEAK_BF16_TFLOPS, 4),
                     "path": "m3ae_xattn_fwd (long-side projection absorbed into the 32-token side; image queries: ONE launch, scores "
                             "and probabilities on chip, csrc/xflash.hip; text queries: csrc/xattn.hip)"
                             if ops.XATTN != "off" else "composition (GEMM + flash attention + GEMM + LayerNorm)",
                     "note": "6 layers x 2 directions, eval-mode forward, includes the output dense + residual + LayerNorm; "
                             "tflop = the reference formulation's 17.922 GFLOP / sample (the fused path executes ~9.9); ms / frac: "
                             "the 12 sub-blocks one after the other on ONE stream (the kernels' own rate)"}
            if ms_two is not None:
                xattn["two_streams"] = {"ms": round(ms_two, 3), "achieved": round(tf / (ms_two * 1e-3), 1),
                                        "frac": round(tf / (ms_two * 1e-3) / PEAK_BF16_TFLOPS, 4),
                                        "how": "the same 12 sub-blocks issued as M3AETransformerSS.infer issues them: text queries on "
                                               "the side stream beside the image queries of the same layer"}
        reducer.attach()

    parity = None
    if rank == 0 and world == 1 and args.head == "cls" and args.arch == "base" and not args.no_parity:
        # the kernels just timed (bf16 storage, MFMA, fused cross-attention) against the reference fixture at FULL size: configs[1]
        # dimensions, B = 2, eval mode; tests/golden/full_vqa.npz holds the unmodified reference's logits / loss / gradient norms
        gpath = os.path.join(ROOT, "tests", "golden", "full_vqa.npz")
        if os.path.exists(gpath):
            import numpy as np
            from m3ae_amd.parity import parity_report
            log("parity leg (full size, B = 2, bf16 MFMA path vs the reference fixture)")
            pm = M3AETransformerSS(finetune_vqa_rad_config(compute_dtype="bf16"))
            synth.fill_deterministic(pm)
            pm.finalize(dev, torch.bfloat16)
            pm.eval()
            pb = to_dev(synth.synthetic_batch(2, text_len=32, image_size=384, vocab_size=50265, rank=0), dev)
            old_rule = ops.XATTN_TRAIN_MIN_BATCH
            parity = {"what": "bf16 MFMA path vs tests/golden/full_vqa.npz (reference-generated: configs[1] dimensions, B = 2, eval "
                              "mode, one forward + backward); tolerance north_star states for fp32 parity mode: logits rtol 1e-3 -- "
                              "the bf16 path is reported as observed, not held to it (DESIGN.md 4)"}
            for label, rule in (("fused_cross_attention", 0), ("batch_rule_default", old_rule)):
                ops.XATTN_TRAIN_MIN_BATCH = rule
                r = parity_report(pm, np.load(gpath, allow_pickle=False), pb)
                parity[label] = {k: (round(v, 8) if isinstance(v, float) else v) for k, v in r.items()
                                 if k not in ("p99_rel_err_param_grad_norms",)}
            ops.XATTN_TRAIN_MIN_BATCH = old_rule
            del pm, pb
            torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import m3ae_oracle as O
        from oracle_util import make_sd, oracle_cfg
        log("cpu_baseline leg (oracle on host cores)")
        torch.set_num_threads(host_cores())
        sd = make_sd(cfg, requires_grad=True)
        oc = oracle_cfg(cfg)
        samples = {}
        for cbatch, iters in ((2, 5), (8, 3)):   # B = 2 and B = 8, all host cores; ~20 s of CPU work in all
            cb = synth.synthetic_batch(cbatch, text_len=32, image_size=384, rank=0)
            times = []
            for i in range(iters + 1):
                for p in sd.values():
                    p.grad = None
                tt = time.perf_counter()
                l, _, _ = O.training_loss(sd, oc, cb)
                l.backward()
                times.append(time.perf_counter() - tt)
                log(f"cpu_baseline B={cbatch} iter {i}: {times[-1]:.1f}s")
            samples[f"B={cbatch}"] = round(cbatch / min(times[1:]), 4)
        cpu = {"value": max(samples.values()), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
               "pairs_per_s": samples,
               "sample": "fwd+bwd fp32 (no optimizer) of configs[1] at B=2 (1 warm-up + 5 timed) and B=8 (1 + 3), best "
                         "iteration each, value = the better batch; oracle/m3ae_oracle.py (CPU restatement of the reference "
                         "path) on PyTorch-CPU; threads = this process's CPU share (cgroup quota / 16-core GPU-box share, bench.host_cores)"}

    secondary = None
    if rank == 0 and world == 1 and args.head == "cls" and args.arch == "base" and not args.no_secondary \
            and not args.rehearse_ddp:
        # the "+T5" half of BASELINE.json's metric, timed in the same driver run (configs[2]'s recipe on one GPU), and the
        # small-per-GPU-batch regime the reference's run scripts name (per_gpu_batchsize 32: finetune_m3ae_decoder.sh:1-2,
        # pretrain_m3ae.sh:1-2) for configs[1] and configs[2]: child processes, a few steps each
        import subprocess
        del loss
        torch.cuda.empty_cache()
        secondary = []
        for label, extra in (("configs[2] recipe, t5-base head, per-GPU batch 64", ["--head", "t5", "--t5", "t5-base", "--batch", "64"]),
                             ("configs[1], per-GPU batch 32", ["--batch", "32"]),
                             ("configs[2] recipe, t5-base head, per-GPU batch 32", ["--head", "t5", "--t5", "t5-base", "--batch", "32"])):
            log("secondary line: " + label + " (child process)")
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", *extra,
                   "--steps", str(min(args.steps, 8)), "--warmup", str(min(args.warmup, 2)), "--no-roofline", "--no-cpu-baseline",
                   "--no-secondary", "--no-parity"] + (["--no-dropout"] if args.no_dropout else [])
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
                rows = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                if pr.returncode == 0 and rows:
                    sl = json.loads(rows[-1])
                    secondary.append({"what": label, **{k: sl[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                                                                          "ms_per_step", "step_ms", "dtype", "data", "config",
                                                                          "final_loss")}})
                else:
                    secondary.append({"what": label, "error": f"child rc {pr.returncode}", "stderr_tail": pr.stderr[-400:]})
            except Exception as e:  # noqa: BLE001
                secondary.append({"what": label, "error": repr(e)})

    if rank == 0:
        line = {
            "metric": "image-question pairs/sec, M3AE-base fine-tune step (fwd+bwd+AdamW) @384px" if args.head == "cls"
            else ("image-text pairs/sec, M3AE-base pre-training step (MLM + MIM + ITM, fwd+bwd+AdamW) @384px"
                  if args.head == "pretrain" else
                  f"image-question pairs/sec, M3AE-base(frozen)+{args.t5} generative-head fine-tune step @384px"),
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]: M3AE-base (ViT-B/16 + RoBERTa-base + 6 co-attention layers) VQA-RAD "
                                    "classification fine-tune, 384x384, 32 text tokens, 498 answers") if args.head == "cls"
                       else ("configs[3]: M3AE-base pre-training, three infer passes per step (MLM, MIM at 75 % masking, ITM), "
                             "384x384, 64 text tokens, RoBERTa vocabulary 50265") if args.head == "pretrain"
                       else (f"configs[2] recipe: frozen M3AE-base forward + {args.t5} encoder (512 padded tokens) / "
                             "teacher-forced decoder / tied LM head, top-4 encoder + top-4 decoder attention blocks "
                             "trainable (main_t5_m3ae.py); the reference's extra beam-4 generate() inside every training "
                             "step (string metrics, m3ae_t5_mm_encoder_input.py:252-261) is EXCLUDED"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "hip_streams": 2 if getattr(model.m3ae if args.head == "t5" else model, "two_streams", False) else 1,
                       "launch_stream_priority": torch.cuda.current_stream().priority,
                       "launch": graph_note,
                       "optimizer": ("AdamW bucket by bucket on its own stream under backward (%d of %d buckets issued before "
                                     "backward ended)" % (getattr(reducer, "updated_in_backward", 0), reducer.nb))
                       if reducer.update_in_backward else "one AdamW pass after backward",
                       "dropout": ("p=0.1 at every dropout site of the model (train mode, as the reference)" if model.training
                                   else "off (eval-mode semantics)"), "weights": "random-init (synthetic, deterministic)",
                       **({"rehearsal": "N > 1 code path on one GPU (one-rank RCCL group); not the metric's configuration"}
                          if args.rehearse_ddp else {}),
                       **({"ddp_buckets": {"count": reducer.nb, "dtype": args.ddp_grad_dtype,
                                           "largest_MiB": round(max(reducer.bucket_bytes_list()) / 2 ** 20, 1)}}
                          if (world > 1 or args.rehearse_ddp) else {})},
            "step_tflops_per_gpu": round(STEP_GFLOP_PER_SAMPLE * B / 1e3 / (ms_per_step * 1e-3), 1) if args.head == "cls" else None,
            "mfma_frac_whole_step": round(STEP_GFLOP_PER_SAMPLE * B / 1e3 / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS, 4)
            if args.head == "cls" else None,
            "final_loss": round(final_loss, 4), "step_ms": step_stats,
            "roofline": roofline, "cross_attention_fwd": xattn, "kernels": kern_table, "cpu_baseline": cpu,
            "parity": parity, "distributed": dist_info,
            "secondary": secondary,
            "notes": {"optimizer": "AdamW update rule restated from transformers==4.6.0 (third party, not installable here): "
                                   "group membership / lr / weight decay / schedule are pinned by reference fixtures, the "
                                   "update arithmetic itself is not (DESIGN.md 3)"},
        }
        if args.arch == "large":   # configs[4] towers: relabel (the numbers are not the headline metric)
            line["metric"] = line["metric"].replace("M3AE-base", "M3AE (ViT-L/16 + RoBERTa-large towers)").replace("@384px", "@512px")
            line["config"]["workload"] = ("configs[4] towers: ViT-L/16 (23 blocks, width 1024) + RoBERTa-large + 6 co-attention "
                                          "layers (768), 512x512 (1025 image tokens), head: " + args.head +
                                          (" " + args.t5 if args.head == "t5" else ""))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()  # rank 0 runs the roofline / cpu_baseline legs alone; the others wait here, not in teardown
        dist.destroy_process_group()
    elif args.rehearse_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
