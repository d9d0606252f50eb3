#!/usr/bin/env python
"""bench.py — volumetric patches/sec (fwd+bwd) of the MI355X-native 3D U-Net hot path.

Workload (BASELINE.json configs[1]/[2]): UNet3D in=1,out=1,f_maps=32,'gcr',num_groups=8, per-GPU batch
2x1x64x128x128 fp32, BCEDiceLoss on the logits, synthetic N(0,1) patches / Bernoulli(0.5) targets, random-init
weights.  A "step" = forward + loss + backward (+ gradient all-reduce when N>1) + Adam update, inputs resident in
HBM.  One process per GPU (torch.distributed over RCCL); weak scaling: the per-GPU batch is fixed.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3      # launches its own 8 ranks through torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus 8 --steps 20 --warmup 3

Rank 0 prints ONE JSON line (contract in the task description) extended with
  "roofline":     fp32-MFMA roofline of the dominant kernel family, timed with HIP events on the launching stream
  "cpu_baseline": the CPU oracle (port of the reference path on ATen CPU operators) timed on this box's host cores
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "pytorch-3dunet_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 256 FLOP/clk x 2.4 GHz
PATCH = (64, 128, 128)
PER_GPU_BATCH = 2
MODEL_CFG = dict(in_channels=1, out_channels=1, f_maps=32, layer_order="gcr", num_groups=8, final_sigmoid=True)


def conv_flops_per_patch(model):
    """2*Cin*Cout*27*D*H*W per 3^3 conv at its resolution (473.822 GFLOP fwd per patch)."""
    total = 0.0
    d, h, w = PATCH
    dims = []
    for i, enc in enumerate(model.encoders):
        if i > 0:
            d, h, w = d // 2, h // 2, w // 2
        dims.append((d, h, w))
        for sc in (enc.basic_module.SingleConv1, enc.basic_module.SingleConv2):
            total += 2.0 * sc.conv.in_channels * sc.conv.out_channels * 27 * d * h * w
    for j, dec in enumerate(model.decoders):
        d, h, w = dims[len(dims) - 2 - j]
        for sc in (dec.basic_module.SingleConv1, dec.basic_module.SingleConv2):
            total += 2.0 * sc.conv.in_channels * sc.conv.out_channels * 27 * d * h * w
    return total


def cpu_baseline(sample_iters=3):
    """The CPU oracle (oracle/unet3d_oracle.py = the reference's module graph on ATen CPU operators) on a bounded
    sample of the same workload: batch 1 of the same patch, 1 warm-up + `sample_iters` timed fwd+bwd."""
    import unet3d_oracle as orc
    from pytorch3dunet_amd.unet3d.model import UNet3D

    # thread count: tools/cpu_thread_scan.py on the MI355X box (256 logical cores) gives 8:0.352 s, 16:0.266 s,
    # 32:0.356 s, 64:0.620 s, 128:1.014 s, 256:34.7 s per iteration on a 1x1x32x64x64 patch -> 16 threads is the
    # fastest ATen/oneDNN configuration; the default (all 256) would understate the CPU path 100x.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = UNet3D(**MODEL_CFG)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(1, 1, *PATCH)
    target = (torch.rand(1, 1, *PATCH) > 0.5).float()
    orc.forward_backward(sd, x, target, MODEL_CFG["num_groups"])  # warm-up
    times = []
    for _ in range(sample_iters):
        t0 = time.perf_counter()
        orc.forward_backward(sd, x, target, MODEL_CFG["num_groups"])
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    cpu_model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name"))
    except Exception:
        pass
    # kind "port": /root/reference does not exist on the GPU box, so what is timed is oracle/unet3d_oracle.py — the reference's
    # module graph restated on the same ATen CPU operators.  tests/test_oracle.py::test_port_and_live_reference_same_speed times
    # both side by side in the build container (same numerics to 1e-6, same throughput within noise).
    return {"value": round(1.0 / med, 4), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"PORT of the reference path (oracle/unet3d_oracle.py, not the reference module tree itself): batch "
                      f"1x1x{PATCH[0]}x{PATCH[1]}x{PATCH[2]} fwd+bwd, 1 warm-up + {sample_iters} timed iterations (median {med:.3f} s), "
                      f"torch {torch.__version__} CPU operators, {torch.get_num_threads()} of {os.cpu_count()} logical cores of {cpu_model}"}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the newest committed PMC summary (profiles/*_pmc_traffic.json,
    written by tools/prof_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this
    very command; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  PMC counters cannot be read from inside the timed
    process, hence the file; None if absent."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            fam = json.load(fh)["families"][family]
        return round(fam["bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher: re-exec this command line under torch.distributed.run with N
    local ranks (the reference itself is single-process `nn.DataParallel`, unet3d/trainer.py:202-205; here it is one process per
    GPU).  Returns the launcher's exit code; the ranks inherit stdout, so rank 0's JSON line is the only line printed."""
    import socket
    import subprocess

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    have = torch.cuda.device_count()
    assert have >= n, f"bench.py --gpus {n}: only {have} GPU(s) visible"
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.pop("U3D_BENCH_SELF_LAUNCH", None)  # (test hook: forces this path at --gpus 1 on a one-GPU box)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")             # (the launcher would set 1 and print a warning)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the HIP-event per-kernel timing")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch (default = BASELINE config 2)")
    ap.add_argument("--compute-dtype", default="fp32", choices=["fp32", "fp32_split"],
                    help="arithmetic of the TIMED model (default fp32 = fp32 MFMA; fp32_split is reported as an extra leg anyway)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs outside the contract (reference step order, fp32_split, partial bf16)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("U3D_BENCH_SELF_LAUNCH") == "1"):
        # plain `python bench.py --gpus N`: launch our own ranks (one process per GPU, the same command line the driver's
        # torch.distributed.run form uses) and let rank 0's single JSON line through
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from pytorch3dunet_amd import _native as nat
    from pytorch3dunet_amd import parallel
    from pytorch3dunet_amd.unet3d.model import UNet3D

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    nat.call("u3d_check_device", local_rank)
    for kv in os.environ.get("U3D_TUNE", "").split(","):  # e.g. U3D_TUNE=0:0 switches the start-phase stagger off
        if ":" in kv:
            nat.call("u3d_set_tuning", int(kv.split(":")[0]), int(kv.split(":")[1]))
    # launched through torch.distributed.run (RANK set) the RCCL path runs even with ONE rank: same hooks, same
    # collectives, so the N>1 code path is exercised on a single-GPU box (`--nproc-per-node 1`)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    torch.manual_seed(0)  # identical initial weights on every rank (also broadcast below)
    model = UNet3D(compute_dtype=args.compute_dtype, **MODEL_CFG).to(dev).train()
    sync = parallel.attach(model, force_single=True) if use_dist else None
    # Adam with the hyper-parameters of 3DUnet_confocal_boundary/train_config.yml (the reference's create_optimizer, utils.py:246-316,
    # builds torch.optim.Adam): the same update as ONE launch over all 44 parameters (pytorch3dunet_amd.optim.FusedAdam; torch's
    # multi-tensor form is 8 launches / 0.17 ms of the 17 ms step; U3D_BENCH_TORCH_ADAM=1 times that one instead)
    from pytorch3dunet_amd.optim import FusedAdam

    Adam = torch.optim.Adam if os.environ.get("U3D_BENCH_TORCH_ADAM") == "1" else FusedAdam
    opt = Adam(model.parameters(), lr=2e-4, weight_decay=1e-5)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)  # per-rank synthetic shard
    B = args.batch
    x = torch.randn((B, 1, *PATCH), device=dev, generator=g)
    target = (torch.rand((B, 1, *PATCH), device=dev, generator=g) > 0.5).float()

    from pytorch3dunet_amd.unet3d.losses import BCEDiceLoss

    criterion = BCEDiceLoss()  # fused HIP kernels on the logits (u3d_bce_dice_fwd/_bwd)

    # (experiment hooks, off by default: U3D_AB_SLEEP = idle cycles / U3D_AB_COPY = MiB of device-to-device copy inserted per step —
    # how the step time responds says whether the chip is time- or power-bound)
    ab_sleep = int(os.environ.get("U3D_AB_SLEEP", "0"))
    ab_copy = int(os.environ.get("U3D_AB_COPY", "0"))
    ab_buf = torch.empty((2, ab_copy << 18), device=dev) if ab_copy else None

    def step():
        if ab_sleep:
            torch.cuda._sleep(ab_sleep)
        if ab_buf is not None:
            ab_buf[0].copy_(ab_buf[1])
        probs, logits = model(x, return_logits=True)
        loss = criterion(logits, target)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    # u3d_conv3d_ex is u3d_conv3d with a scratch buffer, u3d_conv3d_ex_reps that with replica rows of the statistics table (same kernels):
    # one family; likewise the weight gradient's strided / job-carrying forms
    FAMILY = {"u3d_conv3d_ex": "u3d_conv3d", "u3d_conv3d_ex_reps": "u3d_conv3d", "u3d_conv3d_wgrad_job": "u3d_conv3d_wgrad",
              "u3d_conv3d_wgrad_strided": "u3d_conv3d_wgrad"}
    dominant = {"u3d_conv3d", "u3d_conv3d_ex", "u3d_conv3d_ex_reps"}
    calls_per_step, fam_calls = 32, {"u3d_conv3d": 128}
    prof = None

    def host_prep():
        """Everything host-side that has to happen once before the clock starts: the timed region's HIP events and the collector pass.
        It takes 0.1-0.3 s during which the GPU sits idle and drops to its idle clocks — so it runs after the FIRST warm-up step, and the
        remaining warm-up steps bring the chip back to its working state before the barrier (with it between the last warm-up step and
        the barrier, as until round 6, the first timed steps ran on ramping clocks: 20 timed steps read 0.07 ms per step more than 100,
        same box, whatever --warmup was)."""
        nonlocal prof
        if not args.no_roofline and rank == 0:
            # inside the timed region only the DOMINANT MFMA family is bracketed by HIP events (the `roofline` object describes that
            # family); every other entry point is timed in a few extra steps after it
            # (its events are created HERE, before the clock starts: first-time event creation is 0.1-0.2 ms of host time each)
            bracket_all = os.environ.get("U3D_BENCH_BRACKET_ALL") == "1"
            prof = nat.EventProfiler(flops_only=True, only=None if bracket_all else dominant,
                                     prealloc=2 * args.steps * ((sum(fam_calls.values()) if bracket_all else calls_per_step) + 4))
        # Python's cyclic collector: a full (generation 2) pass over this process's objects takes ~75 ms of HOST time, and whether one falls
        # into the timed region depended on --steps / --warmup through the number of event objects allocated above (measured: 17.4 -> 24-28 ms
        # per step at --steps 10 --warmup 1..3, the whole difference in the first timed step's enqueue).  Collect now and move everything
        # alive into the permanent generation; the collector itself stays ON during the timed steps (their garbage is young and cheap).
        gc.collect()
        gc.freeze()

    prep_early = os.environ.get("U3D_BENCH_PREP_LAST") != "1"  # (A/B: 1 = the preparation between the last warm-up step and the barrier)
    for w in range(args.warmup):
        if w == (0 if prep_early else args.warmup - 1) and not args.no_roofline and rank == 0:
            # the first warm-up step finds the dominant MFMA family (HIP events around every call that declares FLOPs), so that the
            # TIMED region only brackets that family: an event pair costs ~2 us of device time, ~110 MFMA calls per step
            # would cost ~0.4 ms (2.5 %) of every timed step
            scout = nat.EventProfiler(flops_only=True)
            nat.profiler = scout
            step()
            torch.cuda.synchronize()
            nat.profiler = None
            fam_ms, fam_calls = {}, {}
            for k, v in scout.summary().items():
                fam_ms[FAMILY.get(k, k)] = fam_ms.get(FAMILY.get(k, k), 0.0) + v["ms"]
                fam_calls[FAMILY.get(k, k)] = fam_calls.get(FAMILY.get(k, k), 0) + v["calls"]
            if fam_ms:
                top = max(fam_ms, key=fam_ms.get)
                dominant = {top} | {k for k, f in FAMILY.items() if f == top}
                calls_per_step = fam_calls[top]
        else:
            step()
        if w == 0 and prep_early:
            host_prep()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if not prep_early or args.warmup == 0:
        host_prep()
    barrier()
    t0 = time.perf_counter()
    host_marks = []
    # HIP events around the dominant family's launches cost ~7 us of stream time per bracketed launch (measured on one box, round 6: 16.71 ms
    # per step with every launch of every timed step bracketed, 16.51 without any) — that is inside `value`.  The roofline figure needs
    # an average launch duration over the timed region, not every launch of it: every `bracket_every`-th timed step is bracketed
    # (steps 0, 10, 20, ..), the others run as a user's step does.  U3D_BENCH_BRACKET_EVERY=1: every step (the round-5 behaviour).
    bracket_every = max(1, int(os.environ.get("U3D_BENCH_BRACKET_EVERY", "10")))
    # (the bracketed steps are every/2, every/2 + every, ..: NOT the first timed step — right after the barrier the host has no lead over the
    # GPU, the launches of that step arrive spaced out and its kernels run on a chip that has just idled: the family read 0.82 of peak there
    # against 0.77-0.78 in steady-state steps and rocprofv3's 0.78-0.79; U3D_BENCH_BRACKET_PHASE)
    bracket_phase = int(os.environ.get("U3D_BENCH_BRACKET_PHASE", str(bracket_every // 2))) % bracket_every
    if args.steps <= bracket_phase:
        bracket_phase = 0
    n_bracketed = len([i for i in range(args.steps) if i % bracket_every == bracket_phase])
    for i_step in range(args.steps):
        if prof is not None:
            nat.profiler = prof if i_step % bracket_every == bracket_phase else None
        loss = step()
        if os.environ.get("U3D_BENCH_HOST_TRACE") == "1":
            host_marks.append(time.perf_counter() - t0)  # (debugging aid: when each step's launches were all enqueued)
    barrier()
    elapsed = time.perf_counter() - t0
    if host_marks and rank == 0:
        print("host enqueue marks (ms):", [round(1000 * v, 1) for v in host_marks], "elapsed", round(1000 * elapsed, 1), file=sys.stderr)
    nat.profiler = None
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    final_loss = loss.item()

    if rank == 0:
        patches = world * B * args.steps
        value = patches / elapsed
        f_fwd = conv_flops_per_patch(model)
        out = {
            "metric": "volumetric patches/sec (fwd+bwd), UNet3D f_maps=32 patch 64x128x128",
            "value": round(value, 3),
            "unit": "patches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.compute_dtype == "fp32" else "f32 (3xbf16 operand split, 6 partial products, fp32 accumulate)",
            "data": "synthetic",
            # what torch.distributed actually saw: ranks, backend, gradient collectives issued per step by the engine hooks
            "ranks_seen": ({"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                            "allreduce_per_step": sync.launched // (args.steps + args.warmup)} if use_dist
                           else {"world_size": 1, "backend": None, "allreduce_per_step": 0}),
            "config": {"workload": f"UNet3D in=1 out=1 f_maps=32 gcr num_groups=8, per-GPU batch {B}x1x64x128x128 fp32, "
                                   "BCEDiceLoss, fwd+loss+bwd+Adam step (" + ("torch.optim.Adam" if Adam is torch.optim.Adam else "one-launch FusedAdam") + "), random-init weights",
                       "global_batch": world * B, "parallelism": f"dp{world}",
                       "conv_gflop_per_patch_fwd": round(f_fwd / 1e9, 3),
                       # 3x the forward convolution FLOPs of the REFERENCE formulation per patch over the
                       # step time; the sub-pixel decoder kernels execute fewer (roofline.executed_tflops_whole_step)
                       "achieved_conv_tflops_whole_step": round(value / world * 3 * f_fwd / 1e12, 2),
                       "final_loss": round(final_loss, 5)},
        }
        if prof is not None:
            summ = prof.summary()
            measured = {k: dict(v) for k, v in summ.items()}  # what the events actually covered: n_bracketed of the timed steps
            for v in summ.values():  # per-step figures below divide by args.steps: scale the sampled totals to the whole region
                v["calls"] = v["calls"] * args.steps // n_bracketed
                v["ms"] = v["ms"] * args.steps / n_bracketed
                v["flops"] = v["flops"] * args.steps / n_bracketed
            if not use_dist:  # (a distributed step contains a collective: rank 0 must not run extra ones alone)
                full = nat.EventProfiler()  # untimed: the complete per-entry-point table
                nat.profiler = full
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                nat.profiler = None
                for k, v in full.summary().items():
                    if k not in summ:
                        summ[k] = {"calls": v["calls"] * args.steps // 3, "ms": v["ms"] * args.steps / 3.0,
                                   "flops": v["flops"] * args.steps / 3.0}
            for k, f in FAMILY.items():  # one family under the name the PMC summaries use
                if k in summ:
                    ex = summ.pop(k)
                    base = summ.setdefault(f, {"calls": 0, "ms": 0.0, "flops": 0.0})
                    for kk in ("calls", "ms", "flops"):
                        base[kk] += ex[kk]
            dom = FAMILY.get(sorted(dominant)[0], sorted(dominant)[0])  # the family bracketed INSIDE the timed region
            d = summ[dom]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(dom)
            executed = sum(v["flops"] for v in summ.values())
            # the matrix pipe's own ceiling on this box, measured now: a launch of nothing but fp32 MFMAs on register operands, ~8 ms each
            # (the sustained rate depends on the operand data; random operands are what activations look like)
            import ctypes

            def mfma_rate(mode):
                sink = torch.zeros(4, device=dev)
                flop = ctypes.c_double(0.0)
                st_ = torch.cuda.current_stream(dev).cuda_stream
                for _ in range(3):  # ~25 ms of the same load first: the rate depends on the chip's recent power history
                    nat.call("u3d_debug_mfma_f32_rate", local_rank, st_, sink.data_ptr(), 4000, mode, ctypes.byref(flop))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                nat.call("u3d_debug_mfma_f32_rate", local_rank, st_, sink.data_ptr(), 4000, mode, ctypes.byref(flop))
                e1.record()
                torch.cuda.synchronize()
                return flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12

            ceil_rand, ceil_const = mfma_rate(2), mfma_rate(1)
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                "frac_source": f"HIP events on the launching stream around every launch of the family in {n_bracketed} of the {args.steps} timed "
                               f"steps of this run (every {bracket_every}th, from timed step {bracket_phase} on; an event pair costs ~7 us of stream time, which is inside `value`)",
                # (the same family by rocprofv3 kernel durations is in profiles/*_tables.md, generated from a profile of this command; it is
                # not repeated here: a number read from a committed file would sit beside the live one as if it described HEAD — ADVICE r05)
                # NOT the contract's peak: what a bare fp32-MFMA stream sustains on this box right now, by operand data
                "mfma_ceiling_random_operands": round(ceil_rand, 1), "mfma_ceiling_constant_operands": round(ceil_const, 1),
                "frac_of_random_operand_ceiling": round(achieved / ceil_rand, 4),
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
                "launches": sum(v["calls"] for k, v in measured.items() if FAMILY.get(k, k) == dom),
                "avg_launch_ms": round(d["ms"] / d["calls"], 4),
                "gflop_per_launch": round(d["flops"] / d["calls"] / 1e9, 3),
                "executed_tflops_whole_step": round(executed / elapsed / 1e12, 2),
                "families": {k: {"calls": v["calls"], "ms_per_step": round(v["ms"] / args.steps, 3),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None}
                             for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])},
            }
        if world == 1 and not use_dist and not args.no_extras:
            # EXTRA leg, not the contract's `value`: the SAME model, optimizer state and batch driven in the reference trainer's step
            # order (unet3d/trainer.py:231-246): forward -> loss -> `loss.item()` (a host synchronisation EVERY iteration, :241) ->
            # zero_grad -> backward -> optimizer step.  The headline loop above never synchronises inside a step; this leg says what
            # the drop-in delivers through the unmodified loop ("What's missing" 3).
            def step_ref_order():
                probs, logits = model(x, return_logits=True)   # trainer.py:362
                loss = criterion(logits, target)               # :365
                val = loss.item()                              # :241 train_losses.update(loss.item(), batch size)
                opt.zero_grad(set_to_none=True)                # :244
                loss.backward()                                # :245
                opt.step()                                     # :246
                return val

            gc.collect()  # (before the warm-up steps, not between them and the clock: idle GPU time right before a timed region is paid
            for _ in range(3):  # back as two steps on clocks ramping up from idle, profiles/r06f_step_family_times.txt)
                step_ref_order()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                last = step_ref_order()
            torch.cuda.synchronize()
            el_ref = time.perf_counter() - t1
            out["extra_reference_step_order"] = {
                "value": round(B * args.steps / el_ref, 3), "unit": "patches/s", "ms_per_step": round(1000.0 * el_ref / args.steps, 3),
                "vs_value": round((B * args.steps / el_ref) / value, 4), "final_loss": round(last, 5),
                "loop": "forward, loss, loss.item() host sync, zero_grad, backward, Adam step (unet3d/trainer.py:231-246), eager launches",
            }
        if world == 1 and not use_dist and not args.no_extras and args.compute_dtype == "fp32":
            # EXTRA legs, not the contract's `value`: the same step (same weights, same batch, same loss + Adam step) in the two opt-in
            # arithmetics.  fp32_split: fp32 operands split exactly into three bf16 values, six partial products accumulated in fp32 on
            # the bf16 MFMA pipe (forward / data gradient of the 3x3x3 convolutions; weight gradients and the sub-pixel decoder kernels
            # stay on the fp32 MFMA).  bf16: REDUCED precision — bf16 MFMA operands wherever the bf16 kernels cover a layer of THIS
            # model: the decoders' first convolutions run on a materialised concat (no bf16 sub-pixel kernels exist), the first two
            # layers stay on the fp32 MFMA.
            EXTRAS = (
                ("fp32_split", "model key compute_dtype: fp32_split (or U3D_F32_SPLIT=1)",
                 "fwd/dgrad 3x3x3 convs: 3xbf16 exact operand split, 6 bf16 MFMAs per fp32 multiply-add, fp32 accumulation; everything "
                 "else as the default path"),
                ("bf16", "model key compute_dtype: bf16 (or U3D_BF16=1)",
                 "REDUCED precision, PARTIAL coverage on this model: bf16 MFMA operands (fp32 accumulation, fp32 tensors in HBM) for every "
                 "3x3x3 convolution with Cin % 32 == 0 and Cout % 32 == 0 in forward / data gradient (the decoders' first convolutions on a "
                 "materialised torch.cat((skip, upsampled)) instead of the fp32 path's virtual concat + sub-pixel kernels) and in the weight "
                 "gradient; the first two layers (1 -> 16 -> 32) run on the fp32 MFMA"),
            )
            for mode, opt_in, arithmetic in EXTRAS:
                torch.manual_seed(0)
                model2 = UNet3D(compute_dtype=mode, **MODEL_CFG).to(dev).train()
                opt2 = Adam(model2.parameters(), lr=2e-4, weight_decay=1e-5)
                # same weights, same batch, before any update: how far the two arithmetics are apart on the logits
                torch.manual_seed(0)
                model1 = UNet3D(**MODEL_CFG).to(dev).train()
                with torch.no_grad():
                    l1 = model1(x, return_logits=True)[1]
                    l2 = model2(x, return_logits=True)[1]
                    logits_rel = float((l1 - l2).norm() / l1.norm())
                del model1, l1, l2

                def step2():
                    probs2, logits2 = model2(x, return_logits=True)
                    loss2 = criterion(logits2, target)
                    opt2.zero_grad(set_to_none=True)
                    loss2.backward()
                    opt2.step()
                    return loss2

                gc.collect()
                for _ in range(max(args.warmup, 3)):
                    step2()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    loss2 = step2()
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t1
                out["extra_" + mode] = {
                    "value": round(B * args.steps / el2, 3), "unit": "patches/s", "ms_per_step": round(1000.0 * el2 / args.steps, 3),
                    "opt_in": opt_in, "final_loss": round(loss2.item(), 5),
                    "logits_rel_l2_vs_fp32_mfma_same_weights": float(f"{logits_rel:.3e}"), "arithmetic": arithmetic,
                }
                del model2, opt2
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        # a rank that dies says WHICH rank it was before the launcher tears the others down (its own summary names the first failure only by
        # local rank and exit code); the non-zero exit code travels through torch.distributed.run / self_launch to the caller
        import traceback

        traceback.print_exc()
        print(f"[bench rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')}, local rank {os.environ.get('LOCAL_RANK', '0')}] "
              f"failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.exit(1)
