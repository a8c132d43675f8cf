#!/usr/bin/env python3
"""Benchmark of the sketch-guided sampler hot path (BASELINE.json metric, config[1]).

One "step" = one complete pass of the hot path over one batch: 8 independent sketch-guided samples per
GPU, SD1.5 architecture (synthetic seeded weights), 512x512 (64x64 latents), 50 DDIM steps, CFG 7.5, LGP
guidance on steps 0..25.  Inputs (weights, text embeddings, sketch targets, initial latents) are resident
in HBM before the timed region.  N > 1: one process per GPU (torch.distributed / RCCL), rank 0's weights
are broadcast once, every rank samples its own 8 images (weak scaling, no per-step collective) and the
final latents are gathered on rank 0 inside the timed region.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py --gpus 8

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event
timed in an extra instrumented pass) and `cpu_baseline` (the CPU oracle timed on this box's host cores on a
bounded sample of the same workload; baseline only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F_IMG_TFLOP = 107.65          # algorithmic TFLOP per image at 50 steps / 26 guided, SURVEY.md section 8(d)


def f_img_tflop(T):
    """SURVEY 8(d) per-image work for T DDIM steps: UNet fwd 2 rows/step, + per guided step (i <= 0.5*T) the
    cond-row UNet backward and the LGP forward (2 rows) + backward (1 row).  T = 50 -> 107.65."""
    guided = sum(1 for i in range(T) if not (i > 0.5 * T))
    return (T * 2 * 803.27 + guided * (929.33 + 3 * 40.50)) / 1e3
PEAK_FP16_TFLOPS = 2500.0     # dense fp16 MFMA peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples-per-gpu", type=int, default=8)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed, separately reported) VAE decode")
    ap.add_argument("--shape-report", default=None, help="write a per-shape table of the GEMM / conv launches of one batch")
    return ap.parse_args()


class LaunchTimer:
    """Times every skg_gemm_f16 / skg_conv3x3_f16 launch with HIP events on the launch stream."""

    def __init__(self, ops):
        self.ops, self.rec = ops, []
        self._gemm, self._conv = ops.gemm, ops.conv3x3

    def __enter__(self):
        from sketch2img_amd._lib import lib
        ops = self.ops

        MODES = {"DIRECT": 0, "S1": 1, "S2": 2, "UP2": 3, "S2T": 4}

        def kname(variant, mode):
            """The name rocprofv3 prints for the instantiation that ran (template arguments spelled out)."""
            bn = variant % 1000
            if variant >= 2000:
                stages = 3 if variant >= 10000 else 2
                return f"gemm2_kernel<{256 if bn == 320 else 128}, {bn}, 2, {4 if bn == 320 else 2}, {MODES[mode]}, {stages}>"
            return f"gemm_kernel<{bn}, {MODES[mode]}>"

        def gemm(A, B, *a, **k):
            M, K = A.shape
            N = B.shape[0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._gemm(A, B, *a, **k)
            e1.record()
            nbytes = 2.0 * (M * K + N * K + M * N * (2 if k.get("out_f32") else 1) + (M * N if k.get("residual") is not None else 0))
            tag = "+res" * (k.get("residual") is not None) + "+geglu" * bool(k.get("geglu")) + "+f32" * bool(k.get("out_f32"))
            self.rec.append((kname(lib.skg_gemm_variant(M, N, K, 0, 0), "DIRECT"), 2.0 * M * N * K, e0, e1, nbytes,
                             f"gemm M{M} N{N} K{K}{tag}"))
            return out

        def conv(X, Wp, rows, IH, IW, mode=0, *a, **k):
            Cin, Cout = X.shape[1], Wp.shape[0]
            OH = IH if mode == 0 else (IH // 2 if mode == 1 else IH * 2)
            M = rows * OH * OH
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._conv(X, Wp, rows, IH, IW, mode, *a, **k)
            e1.record()
            name = ("S1", "S2", "UP2", "S2T")[mode]
            nbytes = 2.0 * (X.shape[0] * Cin + Cout * 9 * Cin + M * Cout + (M * Cout if k.get("residual") is not None else 0))
            self.rec.append((kname(lib.skg_gemm_variant(M, Cout, 9 * Cin, Cin, 1 + mode), name), 2.0 * M * Cout * 9 * Cin, e0, e1, nbytes,
                             f"conv {name} M{M} Cin{Cin} Cout{Cout}" + "+res" * (k.get("residual") is not None)))
            return out

        ops.gemm, ops.conv3x3 = gemm, conv
        return self

    def __exit__(self, *exc):
        self.ops.gemm, self.ops.conv3x3 = self._gemm, self._conv

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, fl, e0, e1, nb, _ in self.rec:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += nb
        return agg

    def shape_report(self, path):
        """Per-shape table (launches, total ms, avg us, TFLOP/s, algorithmic TB/s), sorted by total time."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, e0, e1, nb, shape in self.rec:
            a = agg.setdefault((shape, name), [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += nb
        rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
        tot = sum(v[2] for _, v in rows)
        with open(path, "w") as f:
            f.write(f"GEMM / conv launches of one batch, HIP events around each launch: {tot * 1e3:.1f} ms in {len(self.rec)} launches\n")
            f.write(f"{'shape':48s} {'kernel':36s} {'n':>6s} {'ms':>8s} {'avg us':>8s} {'TF/s':>7s} {'TB/s':>6s} {'%':>5s}\n")
            for (shape, name), (n, fl, sec, nb) in rows:
                f.write(f"{shape:48s} {name:36s} {n:6d} {sec * 1e3:8.2f} {sec / n * 1e6:8.1f} {fl / sec / 1e12:7.0f} "
                        f"{nb / sec / 1e12:6.2f} {100 * sec / tot:5.1f}\n")


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc passes (profiles/
    r01_hbm_counters.json: FETCH_SIZE and WRITE_SIZE collected in separate passes over the same workload).
    Units are KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md
    section HBM; re-checked here on GEGLU / GroupNorm-apply whose read:write byte ratio is known) -> doubled."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_counters.json")
    if not os.path.exists(path):
        return None
    data = json.load(open(path))
    for k, v in data.items():
        if kernel_name in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
            return (2.0 * f["sum"] / f["launches"] + w["sum"] / w["launches"]) * 1024.0
    return None


def cpu_baseline(sd_unet, sd_lgp, ehs2, latent0, target0):
    """The CPU oracle (a port of the reference's formulation: eager fp32, autograd through BOTH CFG rows,
    materialised 9320-channel tensor) on a bounded sample of the same workload: ONE 512x512 sample, one
    guided and one unguided DDIM step, extrapolated to 26 guided + 24 unguided steps."""
    from oracle import ddim as oddim, guidance as og, unet as ounet
    # 32 threads is the fastest setting on the GPU box's 256-thread host for this eager fp32 graph (measured
    # with tools/cpu_sweep.py: 5.3 s / UNet eval at 32 threads, 7.2 s at 64, 11.1 s at 128)
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = ounet.SD15
    tab = oddim.make_tables(50)
    t = int(tab.timesteps[0])
    x = latent0.clone()
    times = {}
    for guided in (True, False):
        t0 = time.time()
        x_in = torch.cat([x] * 2).requires_grad_(guided)
        with torch.enable_grad() if guided else torch.no_grad():
            eps, taps = ounet.unet_forward(cfg, sd_unet, x_in, t, ehs2)
        eu, ec = eps.detach().chunk(2)
        nxt = oddim.ddim_step(tab, eu + 7.5 * (ec - eu), t, x)
        if guided:
            nxt = og.apply_anti_gradient(taps, sd_lgp, tab.alphas_cumprod, x_in, nxt, latent0, t, target0, 1.6)
        times[guided] = time.time() - t0
    per_image = 26 * times[True] + 24 * times[False]
    return dict(value=1.0 / per_image, unit="images/s", cores=cores, kind="port",
                sample=f"1 sample 512x512, SD1.5 fp32 eager PyTorch on {cores} host threads: 1 guided step "
                       f"({times[True]:.1f} s) + 1 unguided step ({times[False]:.1f} s) measured, extrapolated to "
                       f"26 guided + 24 unguided = {per_image:.0f} s/image")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    # test-only switches: run the N > 1 code path with several ranks on ONE GPU (gloo carries CUDA tensors)
    backend = os.environ.get("SKG_BENCH_BACKEND", "nccl")
    if "SKG_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SKG_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)    # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from sketch2img_amd import ops, synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.dist import broadcast_state_dict, gather_latents
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet

    S, h, T = args.samples_per_gpu, 64, args.ddim_steps
    t_setup = time.time()
    sd_unet = synthetic.unet_state_dict(SD15) if rank == 0 else None
    sd_lgp = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15)) if rank == 0 else None
    if world > 1:
        sd_unet = broadcast_state_dict(sd_unet, synthetic.unet_param_shapes(SD15), dev, src=0)
        sd_lgp = broadcast_state_dict(sd_lgp, None, dev, src=0)
    net = HipUNet(SD15, sd_unet, dev)
    lgp = HipLGP(sd_lgp, tap_channels(SD15), dev)
    ehs = synthetic.text_embeddings(S)
    net.prepare_context(ehs)
    tab = DDIMTables.make(T)
    net.prepare_timesteps(tab.timesteps.tolist())
    lat0 = synthetic.initial_latents(rank * S, S, h).to(dev)
    target = synthetic.sketch_targets(rank * S, S, h).to(dev)
    sampler = HipSampler(net, lgp)
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    def one_batch():
        x = sampler.sample(lat0, target, T, tables=tab)
        if world > 1:
            gather_latents(x, world, dst=0)
        return x

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        out = one_batch()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_batch()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    finite = bool(torch.isfinite(out).all())
    value = world * S * args.steps / dt

    roof, cpu = None, None
    if rank == 0 and not args.no_roofline:
        with LaunchTimer(ops) as lt:
            sampler.sample(lat0, target, T, tables=tab)
        agg = lt.summary()
        if args.shape_report:
            lt.shape_report(args.shape_report)
        name, (n, fl, sec, nb) = max(agg.items(), key=lambda kv: kv[1][2])
        tot_sec = sum(v[2] for v in agg.values())
        roof = dict(bound="mfma", kernel=name, achieved=fl / sec / 1e12, peak=PEAK_FP16_TFLOPS, unit="TFLOP/s",
                    frac=fl / sec / 1e12 / PEAK_FP16_TFLOPS, traffic=pmc_traffic(name), algorithmic_bytes=nb / n,
                    traffic_source="profiles/r01_hbm_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 10-step run; "
                                   "bytes per launch = (2*FETCH + WRITE)*1024)", launches=n,
                    avg_launch_us=sec / n * 1e6, avg_launch_gflop=fl / n / 1e9,
                    all_gemm_conv_tflops=sum(v[1] for v in agg.values()) / tot_sec / 1e12,
                    gemm_conv_share_of_step=tot_sec / (dt / args.steps),
                    per_kernel={k: dict(launches=v[0], tflops=v[1] / v[2] / 1e12, seconds=v[2]) for k, v in
                                sorted(agg.items(), key=lambda kv: -kv[1][2])})
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd_unet, sd_lgp, ehs[[0, S]], lat0[:1].cpu(), target[:1].cpu())

    vae_info = None
    if rank == 0 and not args.no_vae:
        # latents -> pixels (modules/pipeline.py:118) is outside the per-step path and outside `value` (SURVEY 8d:
        # F_img excludes the 2.51 TFLOP VAE decode); timed separately on this rank's S final latents
        from sketch2img_amd.config import SD_VAE
        from sketch2img_amd.vae import AutoencoderKL
        vae = AutoencoderKL(SD_VAE).to(dev)
        img = vae.decode_latents(out)
        torch.cuda.synchronize()
        tv = time.perf_counter()
        img = vae.decode_latents(out)
        torch.cuda.synchronize()
        tv = time.perf_counter() - tv
        vae_info = {"ms_per_image": tv / S * 1e3, "tflops": 2.5145 * S / tv, "images": S, "out_shape": list(img.shape),
                    "finite": bool(torch.isfinite(img).all()), "note": "SD VAE decoder on the HIP kernels, synthetic "
                    "seeded weights; not included in value / ms_per_step"}

    if rank == 0:
        res = {
            "metric": "sketch-guided images/sec whole-node, SD1.5 512px 50-step DDIM",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: SD1.5 architecture (synthetic seeded weights), "
                                   f"{S} independent samples per GPU, 512x512 (64x64 latents), {T} DDIM steps, "
                                   f"CFG 7.5, LGP sketch guidance on steps 0..{int(0.5 * T)} (beta 1.6)",
                       "samples_per_gpu": S, "global_batch": world * S, "ddim_steps": T,
                       "parallelism": f"replicas x{world} (samples sharded, weights broadcast, latents gathered)"},
            "achieved_tflops_per_gpu": value / world * f_img_tflop(T),
            "outputs_finite": finite, "setup_s": t_setup,
            "roofline": roof, "cpu_baseline": cpu, "vae_decode": vae_info,
        }
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
