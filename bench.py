#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of TRAINING Restormer base at 256x256 (BASELINE.json `metric`).

One step = one pass of the hot path over one synthetic batch: forward, L1 loss, backward, gradient
all-reduce (N>1) and the AdamW update, inputs already resident in HBM.  BASELINE.json quotes the metric at
bs 32 per GPU ("Restormer 256^2 bs=32 at 1/2/4/8 GPU"), which fits one MI355X, so every N runs 32 images per GPU
(weak scaling; N>1 = configs[2] with the RCCL gradient all-reduce).  `--batch 8` gives configs[1] (bs 8 on one GPU);
`--model moce` trains MoCE-IR (configs[3]: noisy top-1 router, experts, balance loss) instead.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline         : the dominant kernel (largest share of device time in a profiled pass of the same workload): its
                     ALGORITHMIC bytes (or flops) per launch / its average launch duration, measured with HIP events
                     recorded on the kernel's own stream by the library's profiler (mi_prof_*), against the MI355X peak
                     (HBM 8 TB/s; bf16 MFMA 2.5 PFLOP/s dense; fp32 MFMA 157.3 TFLOP/s); `traffic` = PMC-measured HBM bytes
                     per launch from the committed rocprofv3 --pmc passes of this command (`traffic_source` names them);
  step_roofline    : the WHOLE step against both roofs with SURVEY 8(d)'s per-pixel figures (14.4 MFLOP and, with perfect
                     fusion, 3 x 4 C N s bytes per block): this is the number fusion moves, the per-kernel one is not;
  mdta_contraction : the MDTA contraction (q k^T and attn v of every block, forward) by itself: flops, ms, MFMA and HBM
                     fractions - the north_star's 40 % MFMA target is quoted against this;
  inference_forward: the no_grad forward of the same model and batch, with the fused MDTA pass A kernel's own MFMA / HBM fractions
                     (q, k and qkv0 never reach HBM on that path);
  fp32_line        : a short run of the same step with fp32 activations (the exact-MFMA parity path, the reference's
                     mainline precision), reported beside the bf16 headline, never instead of it;
  cpu_baseline     : the CPU oracle (oracle/restormer_ref.py, "port") timed on this box's host cores: warm-up + median of
                     3 of forward + L1 + backward on one 1x3x256x256 patch at the thread count a 128x128 sweep picks;
                     the whole leg stays within ~30 s (rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
TRAFFIC_FILE = os.path.join("profiles", "r04_pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", choices=["restormer", "moce", "adair"], default="restormer")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 Restormer / 8 MoCE-IR; 8 = BASELINE configs[1])")
    ap.add_argument("--patch", type=int, default=0, help="patch size (default 256 Restormer / 128 MoCE-IR)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: replay the step as one HIP graph (N=1; Restormer, or MoCE-IR whose small-batch dispatch keeps the "
                         "segment sizes on the device).  Off by default for Restormer: the eager step is not launch-bound, and "
                         "the graph's private pool doubles peak memory")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="reduce-scatter + sharded AdamW + all-gather instead of the overlapped bucketed all-reduce (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-line", action="store_true")
    ap.add_argument("--profile-json", default="", help="also write the per-kernel table of the profiled step here")
    return ap.parse_args()


def make_step(model, trainer, noisy, clean, use_dev_scalars, moce):
    from image_restoration_amd import ops

    loss_buf = {}

    def step():
        trainer.zero_grad()
        out = model(noisy)
        loss, dout = ops.l1_loss(out, clean, want_grad=True)
        if moce:          # L1 + 0.01 * balance loss (MoCE-IR-main/src/train.py:62-71)
            aux = model.total_loss
            torch.autograd.backward([out, aux], [dout, torch.full_like(aux, 0.01)])
        else:
            out.backward(dout)
        trainer.reduce_gradients()
        trainer.optimizer_step(use_dev_scalars=use_dev_scalars)
        loss_buf["loss"] = loss
        return loss
    return step, loss_buf


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE in separate runs of this same command; FETCH_SIZE doubled per the gfx950 calibration in
    profiles/r01_q_pmc_calibration_*.txt).  Counters cannot be collected from inside the timed process, so this is the value
    of the last profiled build (the file records the commit it was taken at), or None when no profile is committed."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        table = json.load(f)
    meta = table.pop("_meta", {})
    import re
    pat = re.compile(rf"{re.escape(kernel)}(_res|_dma|_stream|_wave|_wave_xres|_wave_stream)?_kernel")   # every variant booked under this profiler key
    tot = n = 0.0
    for name, row in table.items():
        if pat.search(name):
            tot += row["hbm_bytes_per_launch"] * row["launches"]
            n += row["launches"]
    return (round(tot / n) if n else None), meta.get("commit")


def cpu_baseline():
    """The CPU oracle's training step, fp32, on this box's host cores (SURVEY 8(d) / BASELINE.md s.3: 1 x 3 x 256^2).
    A thread sweep on a 128^2 patch (16 / 8 threads: oneDNN's small convolutions do not scale to a whole socket - round 2
    measured all 64 threads 5-10x slower) picks the thread count; the quoted value is forward + L1 + backward on ONE
    256 x 256 patch at that setting, warm-up + median of 2, with the forward-only rate beside it.  ~30 s in all."""
    from image_restoration_amd.configs import RESTORMER_BASE as cfg
    from oracle import restormer_ref as R
    sd = {k: v.requires_grad_(True) for k, v in R.make_restormer_state(cfg, seed=0).items()}

    def data(patch):
        g = torch.Generator().manual_seed(1234)
        clean = torch.rand((1, 3, patch, patch), generator=g)
        return clean, R.degrade_sigma(clean, 25.0, seed=4321)

    def fwd(clean, noisy):
        with torch.no_grad():
            return R.restormer_forward(noisy, sd, cfg)

    def train(clean, noisy):
        for v in sd.values():
            v.grad = None
        (R.restormer_forward(noisy, sd, cfg) - clean).abs().mean().backward()

    def med(fn, args, n=3):
        fn(*args)                                       # warm-up
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn(*args)
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)
    t_start = time.perf_counter()
    small = data(128)
    sweep = {}
    for threads in (16, 8):                             # (all 64 host threads is 5-10x slower than 16 on these small convolutions)
        torch.set_num_threads(threads)
        sweep[threads] = {"fwd_s": med(fwd, small), "train_s": med(train, small)}
    best = min((t for t in sweep if "train_s" in sweep[t]), key=lambda t: sweep[t]["train_s"])
    torch.set_num_threads(best)
    big = data(256)
    fwd256, train256 = med(fwd, big), med(train, big, n=2)   # (6 s per training step on this host: warm-up + 2)
    px128, px256 = 128 * 128 / 1e6, 256 * 256 / 1e6
    return {"value": round(px256 / train256, 6), "unit": "Mpixels/s", "cores": best, "kind": "port",
            "sample": f"oracle.restormer_forward + L1 + backward (Restormer base, fp32) on one 1x3x256x256 patch, {best} threads, "
                      f"warm-up + median of 2 (forward: of 3); thread count picked by a 128x128 sweep; whole leg {time.perf_counter() - t_start:.1f} s",
            "fwd_mpix_s_256": round(px256 / fwd256, 6),
            "sweep_128": {"train_mpix_s": {str(t): round(px128 / r["train_s"], 6) for t, r in sweep.items() if "train_s" in r},
                          "fwd_mpix_s": {str(t): round(px128 / r["fwd_s"], 6) for t, r in sweep.items()}}}


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
    # MI_FORCE_COMM=1 under a one-process torchrun: the RCCL process group, the trainer's bucketed all-reduces and the barriers
    # run in a one-rank group (rehearsal of the N > 1 code path on a one-GPU box; not a measurement mode)
    dist_on = world > 1 or (os.environ.get("MI_FORCE_COMM") == "1" and "MASTER_ADDR" in os.environ)
    # The library's freshness is decided in Python and any build runs HERE, before this process touches the GPU (and never
    # under rocprofv3: _build refuses a stale library there) - no child process is started once the GPU is initialised.
    from image_restoration_amd import _build
    if rank == 0:
        _build.build()
    else:
        _build.wait_fresh()
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import image_restoration_amd as m
    from image_restoration_amd import configs, ops
    from image_restoration_amd.trainer import FlatTrainer

    moce = args.model == "moce"
    batch = args.batch or (8 if moce else 32)
    patch = args.patch or (128 if moce else 256)
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    def build_model():
        torch.manual_seed(0)
        if moce:
            from image_restoration_amd.moce_ir import MoCEIR
            return MoCEIR(**configs.MOCEIR_BASE).to(dev).train()
        if args.model == "adair":
            from image_restoration_amd.adair import AdaIR
            return AdaIR(**configs.ADAIR_BASE).to(dev).train()
        return m.Restormer(**configs.RESTORMER_BASE).to(dev)

    if args.graph == 1:
        os.environ.setdefault("MI_MOCE_DISPATCH", "capacity")   # MoCE-IR: segment sizes stay on the device (no .tolist())
    model = build_model()
    n_params = sum(p.numel() for p in model.parameters())
    trainer = FlatTrainer(model, lr=2e-4, shard_optimizer=args.shard_optimizer)

    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    clean32 = torch.rand((batch, 3, patch, patch), generator=gen)
    noise = torch.randn(clean32.shape, generator=gen)
    noisy32 = torch.clamp(torch.round(clean32 * 255.0) + 25.0 * noise, 0, 255) / 255.0   # sigma=25 recipe, BASELINE.md s.3
    clean = clean32.to(dev).to(act)
    noisy = noisy32.to(dev).to(act)

    use_graph = args.graph == 1 and world == 1 and args.model in ("restormer", "moce")
    step, loss_buf = make_step(model, trainer, noisy, clean, use_dev_scalars=use_graph, moce=moce)

    def run_eager(n):
        for _ in range(n):
            if use_graph:
                trainer.set_step_scalars(trainer.step_count + 1)
            step()

    graph = None
    if use_graph:
        # FlatTrainer.capture_step: warm-up steps on a side stream (allocator warm-up; AccumulateGrad nodes must not bind to the
        # legacy default stream), then ONE full step captured and replayed
        try:
            graph = trainer.capture_step(step, warmup=2)
        except Exception as e:  # capture is an optimisation, never a requirement
            if rank == 0:
                print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph = None
            use_graph = False
            torch.cuda.synchronize()
            step, loss_buf = make_step(model, trainer, noisy, clean, use_dev_scalars=False, moce=moce)
    else:
        run_eager(2)  # allocator settle: the first steps hipMalloc the activations (never timed)

    def one():
        if graph is not None:
            trainer.replay_step(graph)
        else:
            step()

    for _ in range(args.warmup):
        one()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss_buf["loss"].float().item()) if "loss" in loss_buf else float("nan")
    peak_gib = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)

    pixels_per_step = world * batch * patch * patch
    ms_per_step = elapsed / args.steps * 1e3
    value = pixels_per_step * args.steps / elapsed / 1e6

    roofline = step_roofline = contraction = None
    nprof = 2
    if not args.no_roofline:
        # profiled pass: same workload, eager, every kernel bracketed by HIP events on its own stream.  EVERY rank runs
        # these steps (they contain the gradient all-reduce); only rank 0 records and reports.
        step_e, _ = make_step(model, trainer, noisy, clean, use_dev_scalars=False, moce=moce)
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
        mfma_peak = MFMA_PEAK_TFLOPS[args.dtype]
        # the bound is whichever roof the kernel's algorithmic intensity puts nearer
        use_mfma = name in ("pw_gemm", "gram", "mdta_qk", "mdta_av") and (tfs / mfma_peak) > (gbs / HBM_PEAK_GBS)
        # PMC traffic exists for the headline workload only (Restormer base, bs 32, 256^2, bf16: tools/profile_bench.sh);
        # any other workload reports null rather than another workload's bytes
        headline = args.model == "restormer" and batch == 32 and patch == 256 and args.dtype == "bf16"
        traffic, traffic_commit = pmc_traffic(name) if headline else (None, None)
        roofline = {
            "kernel": name, "bound": "mfma" if use_mfma else "hbm",
            "achieved": round(tfs if use_mfma else gbs, 2), "peak": mfma_peak if use_mfma else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if use_mfma else "GB/s",
            "frac": round((tfs / mfma_peak) if use_mfma else (gbs / HBM_PEAK_GBS), 4),
            "traffic": traffic,
            "traffic_source": (f"{TRAFFIC_FILE} (rocprofv3 --pmc passes of this command at commit {traffic_commit})" if traffic
                               else "none: no PMC pass committed for this workload"),
            "launches_per_step": dom["launches"] // nprof,
            "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
            "alg_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
            "alg_flops_per_launch": round(dom["flops"] / dom["launches"]),
            "share_of_kernel_time": round(dom["ms"] / tot_ms, 4),
            "kernel_ms_per_step": round(tot_ms / nprof, 3),
        }
        if args.model == "restormer":
            es = 2 if args.dtype == "bf16" else 4
            flops = 3.0 * configs.RESTORMER_BASE_FWD_FLOP_PER_PIXEL * (pixels_per_step / world)
            fused_bytes = 3.0 * configs.RESTORMER_BASE_FWD_FUSED_BYTES_PER_PIXEL_BF16 * (es / 2) * (pixels_per_step / world)
            sec = ms_per_step * 1e-3
            step_roofline = {
                "flops_per_step": flops, "tflops": round(flops / sec / 1e12, 2), "mfma_frac": round(flops / sec / 1e12 / mfma_peak, 4),
                "perfect_fusion_bytes_per_step": fused_bytes, "gbs_vs_perfect_fusion": round(fused_bytes / sec / 1e9, 1),
                "hbm_frac_vs_perfect_fusion": round(fused_bytes / sec / 1e9 / HBM_PEAK_GBS, 4),
                "note": "14.4 MFLOP / pixel and 3 x 4 C N s bytes per block (SURVEY 8(d)); per GPU, on the timed ms_per_step"}
            qk, av = table.get("mdta_qk"), table.get("mdta_av")
            if qk and av:
                c_ms = (qk["ms"] + av["ms"]) / nprof
                c_fl = (qk["flops"] + av["flops"]) / nprof
                c_by = (qk["bytes"] + av["bytes"]) / nprof
                contraction = {
                    "what": "q k^T (+ row norms) and (W_o attn) v of all 44 blocks, forward; q, k, v read from HBM",
                    "flops": c_fl, "ms": round(c_ms, 3), "tflops": round(c_fl / c_ms / 1e9, 1),
                    "mfma_frac": round(c_fl / c_ms / 1e9 / mfma_peak, 4), "hbm_gbs": round(c_by / c_ms / 1e6, 1),
                    "hbm_frac": round(c_by / c_ms / 1e6 / HBM_PEAK_GBS, 4), "target_mfma_frac": 0.40,
                    "note": "c/2 flop per byte (24 at c=48, 48 at c=96): HBM-bound while q, k, v come from HBM; see DESIGN.md"}
        if args.profile_json:
            rows = {k: {**v, "ms_per_step": v["ms"] / nprof, "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                        "TFLOPps": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in table.items()}
            os.makedirs(os.path.dirname(os.path.abspath(args.profile_json)), exist_ok=True)
            with open(args.profile_json, "w") as f:
                json.dump({"steps_profiled": nprof, "kernels": rows}, f, indent=1)

    infer = None
    if world == 1 and rank == 0 and args.model == "restormer" and args.dtype == "bf16" and not args.no_roofline:
        # the no_grad forward of the same model and batch: the path whose MDTA contraction is fused (pass A: LN -> qkv -> dw3x3 ->
        # q k^T partials + v in one launch, csrc/fused_mdta.hip) - q, k and qkv0 never reach HBM there
        model.eval()
        with torch.no_grad():
            model(noisy)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                model(noisy)
            torch.cuda.synchronize()
            dt_inf = (time.perf_counter() - t1) / 3
            ops.prof_enable(True)
            model(noisy)
            torch.cuda.synchronize()
            tab = ops.prof_collect()
            ops.prof_enable(False)
        model.train()
        fa, av = tab.get("mdta_fused_a"), tab.get("mdta_av") or tab.get("pw_gemm")
        gf = tab.get("gdfn_fused_fwd")
        infer = {"what": "no_grad forward, same model and batch (one-launch LN+GDFN half-blocks; MDTA pass A fused where C is 48 / 96)",
                 "ms": round(dt_inf * 1e3, 3), "mpix_s": round(batch * patch * patch / dt_inf / 1e6, 3)}
        if gf:
            infer["gdfn_fused"] = {"launches": gf["launches"], "ms": round(gf["ms"], 3), "tflops": round(gf["flops"] / gf["ms"] / 1e9, 1),
                                   "mfma_frac": round(gf["flops"] / gf["ms"] / 1e9 / MFMA_PEAK_TFLOPS["bf16"], 4),
                                   "hbm_gbs": round(gf["bytes"] / gf["ms"] / 1e6, 1)}
        if fa:
            infer["mdta_pass_a_fused"] = {
                "what": "LN -> qkv 1x1 -> dw3x3 -> q k^T partials + row norms + v, one launch per block (24 of 44 blocks); HBM: x in, v out",
                "launches": fa["launches"], "ms": round(fa["ms"], 3), "tflops": round(fa["flops"] / fa["ms"] / 1e9, 1),
                "mfma_frac": round(fa["flops"] / fa["ms"] / 1e9 / MFMA_PEAK_TFLOPS["bf16"], 4),
                "hbm_gbs": round(fa["bytes"] / fa["ms"] / 1e6, 1), "hbm_frac": round(fa["bytes"] / fa["ms"] / 1e6 / HBM_PEAK_GBS, 4),
                "target_mfma_frac": 0.40,
                "note": "C = 48: fourth form (depthwise conv on the matrix cores, wave-local Gram), bound by its vector-instruction count "
                        "and one GEMM1 -> conv barrier per tile; C = 96: the round-3 form; neither roof - DESIGN.md section 7d"}

    fp32_line = None
    if world == 1 and rank == 0 and args.dtype == "bf16" and not args.no_fp32_line and not moce:
        # the parity path (exact fp32 MFMA, the reference's mainline precision) timed beside the headline: bs 8, 5 steps
        trainer.close()                     # the packed-weight cache is process-global: hand it to the fp32 trainer (idempotent)
        b32 = min(batch, 8)
        model32 = build_model()
        tr32 = FlatTrainer(model32, lr=2e-4)
        st32, _ = make_step(model32, tr32, noisy32[:b32].to(dev), clean32[:b32].to(dev), use_dev_scalars=False, moce=False)
        for _ in range(3):
            st32()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            st32()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 5
        fp32_line = {"value": round(b32 * patch * patch / dt / 1e6, 4), "unit": "Mpixels/s", "ms_per_step": round(dt * 1e3, 3),
                     "per_gpu_batch": b32, "steps": 5, "dtype": "fp32"}
        tr32.close()

    # BASELINE configs[1] (Restormer base, bs 8 on one GPU) and configs[3] (MoCE-IR base, the reference README's per-GPU batch and
    # patch) as short legs of the default line, so that the driver times them too: value, ms per step, library launches per step
    extra = {}
    if world == 1 and rank == 0 and args.dtype == "bf16" and args.model == "restormer" and not args.no_fp32_line:
        trainer.close()

        def short_leg(kind, b_, p_, steps=10, warm=3):
            torch.cuda.reset_peak_memory_stats()
            torch.manual_seed(0)
            if kind == "moce":
                from image_restoration_amd.moce_ir import MoCEIR
                net = MoCEIR(**configs.MOCEIR_BASE).to(dev).train()
            else:
                net = m.Restormer(**configs.RESTORMER_BASE).to(dev)
            tr_ = FlatTrainer(net, lr=2e-4)
            g_ = torch.Generator(device="cpu").manual_seed(4321)
            cl = torch.rand((b_, 3, p_, p_), generator=g_)
            nz = torch.clamp(torch.round(cl * 255.0) + 25.0 * torch.randn(cl.shape, generator=g_), 0, 255) / 255.0
            st_, _ = make_step(net, tr_, nz.to(dev).to(act), cl.to(dev).to(act), use_dev_scalars=False, moce=(kind == "moce"))
            for _ in range(warm):
                st_()
            torch.cuda.synchronize()
            t1_ = time.perf_counter()
            for _ in range(steps):
                st_()
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - t1_) / steps
            ops.prof_enable(True)
            st_()
            torch.cuda.synchronize()
            tab_ = ops.prof_collect()
            ops.prof_enable(False)
            tr_.close()
            return {"value": round(b_ * p_ * p_ / dt_ / 1e6, 4), "unit": "Mpixels/s", "ms_per_step": round(dt_ * 1e3, 3),
                    "per_gpu_batch": b_, "patch": p_, "steps": steps, "dtype": args.dtype,
                    "library_launches_per_step": int(sum(v["launches"] for v in tab_.values())),
                    "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
        extra["c2_bs8"] = dict(short_leg("restormer", 8, 256), what="BASELINE configs[1]: Restormer base 256x256, bs 8, one GPU, train step")
        extra["c4_moce"] = dict(short_leg("moce", 8, 128), what="BASELINE configs[3] per-GPU workload: MoCE-IR base 128x128, bs 8, train step "
                                                               "(L1 + 0.01 balance loss)")

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        if moce:
            metric = f"Mpixels/sec train (MoCE-IR base {patch}x{patch}, bs {batch}/GPU)"
            workload = (f"MoCE-IR (dim 48, enc [4,6,6,8], dec [2,4,4], 4 experts top-1, {n_params / 1e6:.2f}M params) train step: "
                        f"fwd + L1 + 0.01 balance loss + bwd + AdamW, {patch}x{patch} patches, bs {batch}/GPU, {args.dtype} "
                        f"activations, fp32 params/grads/optimizer")
        elif args.model == "adair":
            metric = f"Mpixels/sec train (AdaIR base {patch}x{patch}, bs {batch}/GPU)"
            workload = (f"AdaIR (Restormer base U-Net + 3 FreModules, {n_params / 1e6:.2f}M params) train step: fwd + L1 + bwd + "
                        f"AdamW, {patch}x{patch} patches, bs {batch}/GPU, {args.dtype} activations, fp32 params/grads/optimizer")
        else:
            metric = ("Mpixels/sec train (Restormer base 256x256, bs 32/GPU)" if (batch == 32 and patch == 256) else
                      f"Mpixels/sec train (Restormer base {patch}x{patch}, bs {batch}/GPU)")
            workload = (f"Restormer base (dim 48, blocks [4,6,6,8], {n_params / 1e6:.2f}M params) train step: fwd + L1 + bwd + "
                        f"AdamW, {patch}x{patch} patches, bs {batch}/GPU, {args.dtype} activations, fp32 params/grads/optimizer")
        line = {
            "metric": metric, "value": round(value, 4), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "per_gpu_batch": batch, "global_batch": batch * world, "patch": patch,
                       "parallelism": f"dp{world}", "hip_graph": bool(graph is not None), "final_loss": final_loss,
                       "peak_hbm_gib": peak_gib},
            "roofline": roofline, "step_roofline": step_roofline, "mdta_contraction": contraction, "inference_forward": infer,
            "fp32_line": fp32_line, **extra,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
