#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of TRAINING Restormer base at 256x256 (BASELINE.json `metric`).

One step = one pass of the hot path over one synthetic batch: forward, L1 loss, backward, gradient
all-reduce (N>1) and the AdamW update, inputs already resident in HBM.  BASELINE.json quotes the metric at
bs 32 per GPU ("Restormer 256^2 bs=32 at 1/2/4/8 GPU"), which fits one MI355X (139 GiB of 288), so every N runs
32 images per GPU (weak scaling; N>1 = configs[2] with the RCCL gradient all-reduce).  `--batch 8` gives
configs[1] (bs 8 on one GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel (largest share of device time in a profiled step of the same workload),
                 its ALGORITHMIC bytes (or flops) per launch / its average launch duration, measured with HIP
                 events recorded on the kernel's own stream by the library's profiler (mi_prof_*), against the
                 MI355X peak (HBM 8 TB/s; bf16 MFMA 2.5 PFLOP/s dense; fp32 MFMA 157.3 TFLOP/s);
  cpu_baseline : the CPU oracle (oracle/restormer_ref.py, "port") timed on this box's host cores on a bounded
                 sample of the same workload (one 1x3x256x256 training step), rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
HBM_KERNELS_BOUND = "hbm"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32; 8 = BASELINE configs[1])")
    ap.add_argument("--patch", type=int, default=256)
    ap.add_argument("--graph", type=int, default=0,
                    help="1: replay the step as one HIP graph (N=1 only).  Off by default: with the packed-weight cache "
                         "and the fused small kernels the eager step is no longer launch-bound (61.3 vs 61.5 ms at "
                         "bs 8, 206.7 vs 205.7 ms at bs 32), and the graph's private pool doubles peak memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-json", default="", help="also write the per-kernel table of the profiled step here")
    return ap.parse_args()


def make_step(model, trainer, noisy, clean, use_dev_scalars):
    from image_restoration_amd import ops

    loss_buf = {}

    def step():
        trainer.zero_grad()
        out = model(noisy)
        loss, dout = ops.l1_loss(out, clean, want_grad=True)
        out.backward(dout)
        trainer.reduce_gradients()
        trainer.optimizer_step(use_dev_scalars=use_dev_scalars)
        loss_buf["loss"] = loss
        return loss
    return step, loss_buf


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE in separate runs of this same command; FETCH_SIZE doubled per the gfx950 calibration in
    profiles/r01_q_pmc_calibration_*.txt).  Counters cannot be collected from inside the timed process, so this is the
    last profiled value, or None when no profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    import re
    pat = re.compile(rf"{re.escape(kernel)}(_res|_dma|_stream|_wave|_wave_xres|_wave_stream)?_kernel")   # every variant booked under this profiler key
    tot = n = 0.0
    for name, row in table.items():
        if pat.search(name):
            tot += row["hbm_bytes_per_launch"] * row["launches"]
            n += row["launches"]
    return round(tot / n) if n else None


def cpu_baseline(cfg, patch: int):
    """One oracle training step (fwd + L1 + bwd) on a single patch, fp32, all host threads."""
    from oracle import restormer_ref as R
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    sd = {k: v.requires_grad_(True) for k, v in R.make_restormer_state(cfg, seed=0).items()}
    g = torch.Generator().manual_seed(1234)
    clean = torch.rand((1, 3, patch, patch), generator=g)
    noisy = R.degrade_sigma(clean, 25.0, seed=4321)
    t0 = time.perf_counter()
    out = R.restormer_forward(noisy, sd, cfg)
    loss = (out - clean).abs().mean()
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": round(patch * patch / dt / 1e6, 6), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": f"oracle.restormer_forward + L1 + backward, 1x3x{patch}x{patch} fp32, one step, {dt:.1f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_ONE_DEVICE=1: rehearsal of the N>1 code path on a single-GPU box - every rank uses cuda:0 and the collectives
    # go through gloo (RCCL refuses two ranks on one device).  Not a measurement mode.
    one_dev = os.environ.get("BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import __graft_entry__ as entry
    if world > 1:
        # one rank runs make (the others would race it on the same object files), everybody loads the result
        if rank == 0:
            entry.build()
        dist.barrier()
    entry.build()
    import image_restoration_amd as m
    from image_restoration_amd import ops
    from image_restoration_amd.trainer import FlatTrainer
    from oracle import restormer_ref as R  # only for the cpu_baseline leg and the synthetic degradation recipe

    cfg = R.RESTORMER_BASE
    batch = args.batch or 32
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = m.Restormer(**cfg).to(dev)
    trainer = FlatTrainer(model, lr=2e-4)

    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    clean = torch.rand((batch, 3, args.patch, args.patch), generator=gen)
    noise = torch.randn(clean.shape, generator=gen)
    noisy = torch.clamp(torch.round(clean * 255.0) + 25.0 * noise, 0, 255) / 255.0   # sigma=25 recipe, BASELINE.md s.3
    clean = clean.to(dev).to(act)
    noisy = noisy.to(dev).to(act)

    use_graph = args.graph == 1 and world == 1
    step, loss_buf = make_step(model, trainer, noisy, clean, use_dev_scalars=use_graph)

    def run_eager(n):
        for _ in range(n):
            if use_graph:
                trainer.set_step_scalars(trainer.step_count + 1)
            step()

    graph = None
    if use_graph:
        # a few eager steps first (allocator warm-up), then capture ONE full step and replay it
        run_eager(2)
        torch.cuda.synchronize()
        try:
            graph = torch.cuda.CUDAGraph()
            trainer.set_step_scalars(trainer.step_count + 1)
            with torch.cuda.graph(graph):
                step()
        except Exception as e:  # capture is an optimisation, never a requirement
            if rank == 0:
                print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            use_graph = False
            torch.cuda.synchronize()
            step, loss_buf = make_step(model, trainer, noisy, clean, use_dev_scalars=False)
    else:
        run_eager(2)  # allocator settle: the first steps hipMalloc ~4 GB of activations per image (never timed)

    def one():
        if graph is not None:
            trainer.set_step_scalars(trainer.step_count + 1)
            trainer.step_count += 1
            graph.replay()
        else:
            step()

    for _ in range(args.warmup):
        one()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss_buf["loss"].float().item()) if "loss" in loss_buf else float("nan")

    pixels = world * batch * args.patch * args.patch * args.steps
    value = pixels / elapsed / 1e6

    roofline = None
    nprof = 2
    if not args.no_roofline:
        # profiled pass: same workload, eager, every kernel bracketed by HIP events on its own stream.  EVERY rank runs
        # these steps (they contain the gradient all-reduce); only rank 0 records and reports.
        step_e, _ = make_step(model, trainer, noisy, clean, use_dev_scalars=False)
        step_e()
        torch.cuda.synchronize()
        if rank == 0:
            ops.prof_enable(True)
        for _ in range(nprof):
            step_e()
        torch.cuda.synchronize()
    if not args.no_roofline and rank == 0:
        table = ops.prof_collect()
        ops.prof_enable(False)
        tot_ms = sum(v["ms"] for v in table.values())
        name, dom = max(table.items(), key=lambda kv: kv[1]["ms"])
        gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        tfs = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        mfma_kernel = name in ("pw_gemm", "gram")
        mfma_peak = MFMA_PEAK_TFLOPS[args.dtype]
        # the bound is whichever roof the kernel's algorithmic intensity puts nearer
        use_mfma = mfma_kernel and (tfs / mfma_peak) > (gbs / HBM_PEAK_GBS)
        roofline = {
            "kernel": name, "bound": "mfma" if use_mfma else "hbm",
            "achieved": round(tfs if use_mfma else gbs, 2), "peak": mfma_peak if use_mfma else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if use_mfma else "GB/s",
            "frac": round((tfs / mfma_peak) if use_mfma else (gbs / HBM_PEAK_GBS), 4),
            "traffic": pmc_traffic(name), "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc passes)",
            "launches_per_step": dom["launches"] // nprof,
            "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
            "alg_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
            "alg_flops_per_launch": round(dom["flops"] / dom["launches"]),
            "share_of_kernel_time": round(dom["ms"] / tot_ms, 4),
            "kernel_ms_per_step": round(tot_ms / nprof, 3),
        }
        if args.profile_json:
            rows = {k: {**v, "ms_per_step": v["ms"] / nprof, "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                        "TFLOPps": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in table.items()}
            os.makedirs(os.path.dirname(os.path.abspath(args.profile_json)), exist_ok=True)
            with open(args.profile_json, "w") as f:
                json.dump({"steps_profiled": nprof, "kernels": rows}, f, indent=1)

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, args.patch)

    if rank == 0:
        line = {
            "metric": "Mpixels/sec train (Restormer base 256x256, bs 32/GPU)" if batch == 32 else
                      f"Mpixels/sec train (Restormer base 256x256, bs {batch}/GPU)", "value": round(value, 4), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Restormer base (dim 48, blocks [4,6,6,8], 26.13M params) train step: fwd + L1 + bwd + "
                                   f"AdamW, {args.patch}x{args.patch} patches, bs {batch}/GPU, {args.dtype} activations, "
                                   f"fp32 params/grads/optimizer",
                       "per_gpu_batch": batch, "global_batch": batch * world, "patch": args.patch,
                       "parallelism": f"dp{world}", "hip_graph": bool(graph is not None), "final_loss": final_loss,
                       "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
