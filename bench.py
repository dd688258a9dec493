"""ALI training throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full ALI iteration (E+G update, two D updates, diagnostics: everything in the reference's
image_scms/mnist.py:204-248) on one synthetic MorphoMNIST-shaped batch of 512 images per GPU, inputs already
resident in HBM.  Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, timed live with HIP events on
the launch stream in an extra instrumented iteration) and `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "imagecfgen-pytorch_amd"))

BS_PER_GPU = 512
# SURVEY.md 8(d) / BASELINE.md 3: algorithmic work per image of the minimal schedule at the config's per-GPU batch (fp32
# storage): (bytes / image, FLOP / image)
STEP_WORK = {"mnist": (8108239.0, 1.0026e9), "audio": (102121035.0, 3.0046e10), "whale": (335299959.0, 9.8432e10),
             "esrf": (1946750058.0, 6.0378e11)}
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (v_mfma_f32_32x32x16_f16)
PEAK_HBM_GBS = 8000.0
# committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of `python bench.py --workload W --precision P` (scratch/collect_profiles.sh)
TRAFFIC_CSV = "profiles/r03_pmc_hbm_traffic_{workload}_{precision}.csv"


def synth_batch(bs, device, seed):
    """MorphoMNIST-shaped synthetic batch: images in [-1,1], one-hot digit, 3 scaled attributes."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (bs, 28, 28), generator=g).float() * (torch.rand(bs, 28, 28, generator=g) > 0.8)
    images = (2 * x / 255 - 1).reshape(bs, 1, 28, 28)
    c = {"digit": torch.nn.functional.one_hot(torch.randint(0, 10, (bs,), generator=g), 10).float(),
         "thickness": torch.rand(bs, 1, generator=g) * 2 - 1, "intensity": torch.rand(bs, 1, generator=g) * 2 - 1,
         "slant": torch.rand(bs, 1, generator=g) * 2 - 1}
    z = torch.randn(bs, 512, 1, 1, generator=g)
    return images.to(device), {k: v.to(device) for k, v in c.items()}, z.to(device)


SPECT = {   # image side, categorical attrs, continuous attr, module  (reference audio_mnist.py:22-30, whalecalls.py:14-19,
    # esrf_acoustic.py:17-20); default per-GPU batch of BASELINE.json configs 3-5
    "audio": (128, {"country_of_origin": 13, "native_speaker": 2, "accent": 15, "digit": 10, "age": 5, "gender": 2}, None,
              "image_scms.audio_mnist", 256),
    "whale": (256, {"call_type": 3}, None, "image_scms.whalecalls", 128),
    "esrf": (512, {"has_boat": 2}, "closest_boat", "image_scms.esrf_acoustic", 64),
}


def synth_spect_batch(workload, bs, device, seed):
    """images ~ clip(N(0,1),-3,3)/3 (what spect_to_img yields, audio_mnist.py:361-363), uniform one-hot attributes."""
    import torch
    side, cats, cont, _, _ = SPECT[workload]
    g = torch.Generator().manual_seed(seed)
    images = torch.clip(torch.randn(bs, 1, side, side, generator=g), -3, 3) / 3
    c = {k: torch.nn.functional.one_hot(torch.randint(0, v, (bs,), generator=g), v).float() for k, v in cats.items()}
    if cont:
        c[cont] = torch.rand(bs, 1, generator=g) * 2 - 1
    z = torch.randn(bs, 512, 1, 1, generator=g)
    return images.to(device), {k: v.to(device) for k, v in c.items()}, z.to(device)


def cpu_baseline(bs, budget_s=25.0):
    """The CPU oracle (torch-CPU restatement of the reference iteration, bit-identical to the reference in the
    build container) timed on this host: 1 warm-up + as many iterations as fit in the budget (>= 2)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ali_oracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # the GPU box exposes 256 logical CPUs but a cgroup quota of 16: use what we may actually run on
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    E, G, D = orc.build_models("mnist")
    oe, od = orc.build_optimizers(E, G, D)
    for m in (E, G, D):
        m.train()
    images, c, z = synth_batch(bs, "cpu", 1)
    orc.ali_step(E, G, D, oe, od, images, c, z)
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 8):
        orc.ali_step(E, G, D, oe, od, images, c, z)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(bs * n / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} ALI iterations at bs={bs} after 1 warm-up, torch {torch.__version__} CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the BASELINE.json config's)")
    ap.add_argument("--workload", default="mnist", choices=["mnist", "audio", "whale", "esrf"],
                    help="mnist = BASELINE.json configs[1] (the headline metric); the others are configs[2..4]")
    ap.add_argument("--mode", default="stepper", choices=["stepper", "autograd"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"],
                    help="f16 = fp16-MFMA forward / data-gradient GEMMs with fp32 accumulation (BASELINE config 5: esrf)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pipeline-reduce", action="store_true",
                    help="N > 1: overlap the last all-reduce of an iteration with the next iteration's E(x) / G(z) forward "
                         "(AliStepper(pipeline_reduce=True); same arithmetic, tests/test_gpu_dp.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one rank per GPU over RCCL.  Rehearsal on a box with fewer GPUs than ranks (ALI_DIST_BACKEND=gloo): the ranks
    # share the devices round-robin and exchange over gloo -- same schedule, same collective calls, not a measurement.
    backend = os.environ.get("ALI_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    import ali_hip
    import image_scms.mnist as pm
    from ali_hip import ops
    from ali_hip.step import AliStepper
    from image_scms.training_utils import ali_step
    ali_hip.load()

    torch.manual_seed(1)                       # identical replicas; per-rank data / z / dropout streams
    if args.workload != "mnist":
        import importlib
        pm = importlib.import_module(SPECT[args.workload][3])
    E, G, D = pm.Encoder(), pm.Generator(), pm.Discriminator()
    for m in (E, G, D):
        m.apply(pm.init_weights)
        m.to(dev).train()
    ali_hip.manual_seed(1234 + rank)
    bs = args.batch or (BS_PER_GPU if args.workload == "mnist" else SPECT[args.workload][4])
    if args.workload == "mnist":
        batches = [synth_batch(bs, dev, 100 + rank * 17 + i) for i in range(4)]
    else:
        ops.set_workspace_bytes(2 << 30)
        batches = [synth_spect_batch(args.workload, bs, dev, 100 + rank * 17 + i) for i in range(2)]
    betas = (0.5, 0.999) if args.workload == "mnist" else (0.5, 0.9)

    if args.mode == "stepper":
        stepper = AliStepper(E, G, D, betas=betas, process_group=pg, capture=not args.no_graph, precision=args.precision,
                             pipeline_reduce=args.pipeline_reduce and world > 1)

        def one(i):
            images, c, z = batches[i % len(batches)]
            return stepper.step(images, c, z, ahead=batches[(i + 1) % len(batches)] if stepper.pipeline_reduce else None)
    else:
        assert world == 1, "autograd mode is single-GPU (reference schedule through torch.autograd)"
        oe = torch.optim.Adam(list(E.parameters()) + list(G.parameters()), lr=1e-4, betas=betas)
        od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=betas)

        def one(i):
            images, c, z = batches[i % len(batches)]
            return ali_step(E, G, D, oe, od, images, c, z)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        r = one(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = one(args.warmup + i)
    fence()
    dt = dt_local = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    losses = {k: float(v) for k, v in r.items()}

    # ---- roofline leg: ONE instrumented eager iteration on every rank (collectives stay matched); on rank 0 every GEMM
    # launch is timed once, in sequence, with HIP events on the launch stream (ops.KernelProfile).  The stream is kept
    # busy with replays of the timed region's own iteration while the host enqueues the instrumented one, so its kernels
    # run back to back and at steady-state clocks; the event / dispatch overhead of an interval is calibrated on spin
    # kernels of known length, behind the same kind of busy stream, and subtracted.
    prof = ops.KernelProfile()
    step_ms = dt_local / args.steps * 1e3
    graph_mode = args.mode == "stepper" and not args.no_graph

    def eager(i):
        if args.mode == "stepper":
            keep, stepper.capture = stepper.capture, False
        try:
            return one(i)
        finally:
            if args.mode == "stepper":
                stepper.capture = keep

    def busy(ms):
        """enqueue about ``ms`` milliseconds of the timed region's own work (same on every rank)"""
        if graph_mode:
            for i in range(int(ms / step_ms) + 1):
                one(i)
        elif rank == 0:
            prof.hold(ms)

    eager(0)                    # un-instrumented eager pass first (code objects / allocator warm for this mode) ...
    fence()
    th = time.perf_counter()
    eager(1)                    # ... and one to learn how long the host needs to enqueue an iteration
    host_ms = (time.perf_counter() - th) * 1e3
    fence()
    cover = min(2.5 * host_ms + 20.0, 1500.0)
    if rank == 0:
        prof.spin_rate()
    if rank == 0 or graph_mode:
        busy(30.0)
    if rank == 0:
        prof.calibrate(busy=lambda: None)          # (the stream is busy already)
    busy(cover)
    if rank == 0:
        ops.set_profile(prof)
    eager(2)
    ops.set_profile(None)
    fence()
    rank_ms = [dt_local / args.steps * 1e3]
    if world > 1:
        t = torch.tensor(rank_ms, device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        rank_ms = [float(v.item()) for v in allr]

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = bs * world * args.steps / dt
        per_gpu = value / world
        fam = prof.summary() if prof.records else {}
        for src, dst in (("gconv_t", "gconv"), ("wgrad_multi", "wgrad")):
            # conv-forward and transposed / data-gradient launches are one kernel; so are the weight-gradient GEMMs
            # launched alone and the jobs of a combined launch (wgrad_fast_kernel / wgrad_fast_multi_kernel)
            if src in fam:
                g0 = fam.setdefault(dst, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
                for k in g0:
                    g0[k] += fam[src][k]
                del fam[src]
        traffic_csv = TRAFFIC_CSV.format(workload=args.workload, precision=args.precision)

        def pmc_traffic(kernel_substr):
            """HBM bytes per launch of a kernel family from the COMMITTED rocprofv3 PMC passes of this command
            (FETCH_SIZE / WRITE_SIZE in KB per launch, separate --pmc passes; on gfx950 FETCH_SIZE reports half of wide
            coalesced reads, MI355X_MICROARCH.md).  Counters cannot be read from inside the timed process: the figure
            is a property of the kernels at the benched shapes, not of this run -- ``traffic_source`` says so.  None if
            no committed pass covers this workload / precision."""
            path = os.path.join(ROOT, traffic_csv)
            if world != 1 or args.mode != "stepper" or not os.path.exists(path):
                return None
            n = tot = 0.0
            for line in open(path):
                if line.startswith("#") or not any(k in line for k in kernel_substr):
                    continue
                name, launches, fetch_kb, write_kb = line.rsplit(",", 3)
                n += float(launches)
                tot += float(launches) * (2.0 * float(fetch_kb) + float(write_kb)) * 1024.0
            return round(tot / n) if n else None

        roof = None
        f16 = args.precision == "f16"
        if fam:
            gemm = [k for k in fam if k in ("gconv", "wgrad")] or list(fam)
            name = max(gemm, key=lambda k: fam[k]["ms"])
            f = fam[name]
            ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
            peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
            ksub = {"gconv": ("gconv_kernel", "gconv_multi_kernel", "conv_first_kernel"), "wgrad": ("wgrad_fast", "conv_first_wgrad")}

            def pair(k):      # algorithmic bytes next to the counters' traffic, per launch, for one kernel family
                v = fam[k]
                return {"alg_bytes_per_launch": round(v["bytes"] / v["launches"]), "traffic": pmc_traffic(ksub[k])}

            traffic = pmc_traffic(ksub.get(name, (name,)))
            roof = {"bound": "mfma", "kernel": {"gconv": "ali::gconv_kernel", "wgrad": "ali::wgrad_fast_kernel"}.get(name, name),
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic,
                    "traffic_source": (traffic_csv + " (committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command;"
                                       " not measured in this run)") if traffic is not None else None,
                    "alg_bytes_per_launch": round(f["bytes"] / f["launches"]),
                    "alg_flop_per_launch": round(f["flops"] / f["launches"]),
                    "launches_per_step": f["launches"], "avg_launch_us": round(f["ms"] * 1e3 / f["launches"], 2),
                    "method": "every launch of one eager iteration timed once, in sequence, [e0] k [e1] with HIP events on "
                              "the launch stream, which is kept busy with replays of the timed iteration while the host "
                              "enqueues (back-to-back execution at steady-state clocks); minus the interval's event/dispatch "
                              "overhead calibrated on spin kernels; algorithmic FLOP on live channels",
                    "event_overhead_us": round(prof.overhead_ms * 1e3, 2),
                    "families": {k: dict({"launches": v["launches"], "ms": round(v["ms"], 3),
                                          "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)},
                                         **(pair(k) if k in ksub else {})) for k, v in fam.items()}}
        names = {"mnist": "image_scms/mnist.py ALI iteration (EG step + 2 D steps + diagnostics), MorphoMNIST 28x28x1",
                 "audio": "image_scms/audio_mnist.py ALI iteration, AudioMNIST log-spectrogram 128x128x1",
                 "whale": "image_scms/whalecalls.py BiGAN iteration, whale-call spectrogram 256x256x1",
                 "esrf": "image_scms/esrf_acoustic.py ALI iteration, ESRF spectrogram 512x512x1"}
        arith = "fp16 MFMA (fp32 accumulate, fp32 master weights)" if f16 else "fp32 MFMA"
        alg_bytes, alg_flop = STEP_WORK[args.workload]
        step_roof = {"hbm_frac": round(per_gpu * alg_bytes / (PEAK_HBM_GBS * 1e9), 4),
                     ("mfma_f16_frac" if f16 else "mfma_f32_frac"):
                         round(per_gpu * alg_flop / ((PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS) * 1e12), 4),
                     "alg_bytes_per_img": alg_bytes, "alg_flop_per_img": alg_flop}
        if bs != (BS_PER_GPU if args.workload == "mnist" else SPECT[args.workload][4]):
            step_roof = None      # the per-image figures are those of the config's batch (weights / Adam amortised over it)
        out = {
            "metric": "ALI training images/sec (E+G+D step), MorphoMNIST bs=512/GPU" if args.workload == "mnist" else
                      f"ALI training images/sec (E+G+D step), {args.workload} bs={bs}/GPU",
            "value": round(value, 1),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{names[args.workload]} synthetic, bs={bs}/GPU, {arith}, mode={args.mode}"
                                   f"{'' if args.no_graph or args.mode != 'stepper' else ('+hipgraph' if world == 1 else '+hipgraph-segments')}",
                       "global_batch": bs * world,
                       "parallelism": f"dp{world}" + ("+pipelined-allreduce" if args.pipeline_reduce and world > 1 else "")},
            "roofline": roof,
            "step_roofline": step_roof,
            "ms_per_step_ranks": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
            "losses_last_step": losses,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "mnist":
            out["cpu_baseline"] = cpu_baseline(bs)
            if bs != 64:      # BASELINE.json configs[0]: the reference's own CPU-runnable case
                out["cpu_baseline"]["config0_bs64"] = cpu_baseline(64, budget_s=8.0)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
