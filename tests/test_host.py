"""CPU tests (-m "not gpu") of the host side: the C-ABI library loads and exports every symbol include/skg.h
declares with the argument list the ctypes binding uses, scheduler / schedule host logic is bit-exact against
the oracle, synthetic weights equal the oracle's, and the N > 1 plumbing works under gloo with world_size 2.
No kernel is launched here."""
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_header():
    """name -> (return_letter, arg_letters) from include/skg.h using the binding's letter code."""
    src = open(os.path.join(ROOT, "include", "skg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(skg_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        letters = ""
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    letters += "p"
                elif a.startswith("float"):
                    letters += "f"
                elif a.startswith("unsigned"):
                    letters += "u"
                elif a.startswith("size_t"):
                    letters += "z"
                elif a.startswith("int"):
                    letters += "i"
                else:
                    raise AssertionError(f"unparsed argument {a!r} of {name}")
        out[name] = ({"int": "i", "size_t": "z", "const char*": "s"}[ret], letters)
    return out


def test_library_exports_every_declared_symbol_with_matching_signature():
    from sketch2img_amd import _lib
    decl = parse_header()
    assert len(decl) >= 35
    assert set(decl) == set(_lib.SIGNATURES), (set(decl) ^ set(_lib.SIGNATURES))
    for name, sig in decl.items():
        assert _lib.SIGNATURES[name] == sig, (name, _lib.SIGNATURES[name], sig)
        assert hasattr(_lib.lib, name)
    assert _lib.lib.skg_abi_version() == _lib.ABI_VERSION == 5
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (skg_\w+)", nm))
    assert set(decl) <= exported


def test_host_side_argument_checks_do_not_need_a_gpu():
    """Precondition failures return SKG_E_BADARG before any HIP call."""
    from sketch2img_amd._lib import lib
    assert lib.skg_gemm_f16(None, 0, None, 0, None, 0, 1, 8, 32, None, None, 0, 1.0, 0, None) == -1
    assert lib.skg_gemm_f16(16, 24, 16, 24, 16, 8, 4, 8, 24, None, None, 0, 1.0, 0, None) == -1      # K % 32
    assert lib.skg_attn_fwd(16, 8, 16, 8, 16, 8, 16, 8, None, 1, 1, 8, 8, 8, 24, 1.0, None) == -2      # dh = 24
    assert lib.skg_conv3x3_f16(16, 32, 16, 16, 8, 1, 4, 4, 32, 8, 9, None, None, 0, 1.0, 0, None) == -2  # mode 9
    assert lib.skg_ff_block_f16(16, 320, 16, 320, 128, 640, 1280, 16, 16, 1e-5, 16, 16, 16, None, None) == -1   # C = 320 only
    assert lib.skg_ff_block_f16(16, 320, 16, 320, 128, 320, 1296, 16, 16, 1e-5, 16, 16, 16, None, None) == -1   # F % 32
    assert lib.skg_xattn_block_f16(16, 320, 16, 320, 4096, 4096, 320, 8, 81, 16, 16, 1e-5, 16, 16, 16, 0.158, None) == -1   # Nkv <= 80
    assert lib.skg_xattn_block_f16(16, 320, 16, 320, 4096, 1000, 320, 8, 77, 16, 16, 1e-5, 16, 16, 16, 0.158, None) == -1  # HW % 128
    # conv2 + folded shortcut: an X2 operand of 2 GiB or more is DECLINED (-2: the caller runs the two launches), not a bad argument
    sc_args = lambda rows, k2: (16, 320, 16, k2, k2, 16, 16, None, 320, rows, 64, 64, 320, 320, None, 0, None, 0, None)
    assert lib.skg_conv3x3_sc_f16(*sc_args(140, 1920)) == -2 and lib.skg_conv3x3_sc_f16(*sc_args(16, 1920 + 8)) == -1      # K2 % 64
    assert lib.skg_gemm_variant(65536, 320, 2880, 320, 2) == 2160
    assert lib.skg_gemm_variant(4096, 64, 96, 32, 2) == 1064
    # GroupNorm statistics in the producer's epilogue: which launches fuse them (the rest run the stand-alone pass)
    assert lib.skg_gemm_gn_fused(65536, 320, 2880, 320, 1, 4096, 32) == 1      # 64 x 64 conv: the 256 x 320 ping-pong tile
    assert lib.skg_gemm_gn_fused(65536, 320, 320, 0, 0, 4096, 32) == 1         # proj_out GEMM: 128 x 160 tile
    assert lib.skg_gemm_gn_fused(16384, 640, 5760, 640, 1, 1024, 32) == 1      # 32 x 32 conv
    assert lib.skg_gemm_gn_fused(65536, 320, 288, 32, 1, 4096, 32) == 0         # conv_in: Cin = 32 -> generic kernel
    assert lib.skg_gemm_gn_fused(65536, 960, 2880, 320, 1, 4096, 32) == 0       # 30-wide groups straddle the 160-column tile
    assert lib.skg_gemm_gn_fused(65536, 320, 2880, 320, 1, 4000, 32) == 0       # chunks must be whole 128-row blocks
    assert lib.skg_gemm_f16_gn(16, 32, 16, 32, 16, 320, 256, 320, 32, None, None, 0, 1.0, 0, None, 128, 32, None) == -1  # no buffer
    from sketch2img_amd import ops
    assert ops.gn_fusable(65536, 320, 4096, 32) and not ops.gn_fusable(65536, 320, 4000, 32)
    assert ops.gn_concat_ok(320, 320, 32, 32, 32) and ops.gn_concat_ok(640, 640, 32, 32, 32)
    assert not ops.gn_concat_ok(640, 320, 32, 32, 32) and not ops.gn_concat_ok(1280, 640, 32, 32, 32)


def test_ddim_tables_bit_exact_vs_oracle():
    from oracle import ddim as oddim
    from sketch2img_amd.sampler import DDIMTables, guided_step
    for T in (4, 10, 50):
        a, b = DDIMTables.make(T), oddim.make_tables(T)
        assert a.timesteps.dtype == np.int64 and np.array_equal(a.timesteps, b.timesteps)
        assert torch.equal(a.alphas_cumprod, b.alphas_cumprod) and a.final_alpha_cumprod == b.final_alpha_cumprod
        for t in a.timesteps.tolist():
            assert a.coeffs(t) == oddim.step_coeffs(b, t)
    assert DDIMTables.make(50).timesteps.tolist() == list(range(981, 0, -20))
    assert [i for i in range(50) if guided_step(i, 50)] == list(range(26))
    assert [i for i in range(10) if guided_step(i, 10)] == list(range(6))


def test_synthetic_weights_equal_oracle_init_and_configs_agree():
    from oracle import lgp as olgp, unet as ounet
    from sketch2img_amd import config, synthetic
    for a, b in ((config.SD15, ounet.SD15), (config.SD21, ounet.SD21), (config.TINY, ounet.TINY)):
        assert vars(a) == vars(b)
        assert list(synthetic.unet_param_shapes(a).items()) == list(ounet.param_shapes(b).items())
    assert sum(int(np.prod(s)) for s in synthetic.unet_param_shapes(config.SD15).values()) == 859_520_964
    A, B = synthetic.unet_state_dict(config.TINY), ounet.init_weights(ounet.TINY)
    assert list(A) == list(B) and all(torch.equal(A[k], B[k]) for k in A)
    n = synthetic.lgp_input_dim(config.SD15)
    assert n == 9320
    la, lb = synthetic.lgp_state_dict(488), olgp.init_state_dict(488)
    assert list(la) == list(lb) and all(torch.equal(la[k], lb[k]) for k in la)
    assert config.tap_channels(config.SD15) == ounet.tap_channels(ounet.SD15)
    assert config.up_block_plan(config.SD15) == ounet.up_block_plan(ounet.SD15)
    # per-sample inputs depend only on the GLOBAL sample index (placement independent sharding)
    assert torch.equal(synthetic.initial_latents(3, 2, 8)[1], synthetic.initial_latents(4, 1, 8)[0])
    assert torch.equal(synthetic.sketch_targets(3, 2, 8)[1], synthetic.sketch_targets(4, 1, 8)[0])


def test_weight_packs_match_conv_definitions():
    """The implicit-GEMM weight packs (forward and dgrad) against F.conv2d / its autograd, on the CPU."""
    import torch.nn.functional as F
    from sketch2img_amd.unet import pack_conv, pack_conv_dgrad
    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 5, 3, 3, generator=g)
    x = torch.randn(1, 5, 7, 7, generator=g)
    wp = pack_conv(w, "cpu").float().reshape(6, 3, 3, 5)
    xp = F.pad(x, (1, 1, 1, 1))
    y = torch.zeros(1, 6, 7, 7)
    for ky in range(3):
        for kx in range(3):
            y += torch.einsum("oc,bchw->bohw", wp[:, ky, kx], xp[:, :, ky:ky + 7, kx:kx + 7])
    assert torch.allclose(y, F.conv2d(x.half().float(), w.half().float(), padding=1), atol=2e-2)
    gy = torch.randn(1, 6, 7, 7, generator=g).half().float()
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w.half().float(), padding=1).backward(gy)
    wd = pack_conv_dgrad(w, "cpu").float().reshape(5, 3, 3, 6)
    gp = F.pad(gy, (1, 1, 1, 1))
    gx = torch.zeros(1, 5, 7, 7)
    for ky in range(3):
        for kx in range(3):
            gx += torch.einsum("co,bohw->bchw", wd[:, ky, kx], gp[:, :, ky:ky + 7, kx:kx + 7])
    assert torch.allclose(gx, xr.grad, atol=2e-2)


def test_injection_routing_and_names_match_oracle():
    from oracle import attn_inject
    from sketch2img_amd import inject
    from sketch2img_amd.config import SD15
    from oracle import unet as ounet
    assert inject.transformer_block_paths(SD15) == ounet.transformer_block_paths(ounet.SD15)
    assert inject.block_dims(SD15) == attn_inject.block_dims(ounet.SD15)
    rs = [tuple(torch.full((1,), 10 * i + j) for j in range(3 if i < 3 else 2)) for i in range(4)]
    assert [int(t) for t in inject.route_res_samples(rs)] == [int(t) for t in attn_inject.route_res_samples(rs)]
    assert inject.module_name("mid_block.attentions.0.transformer_blocks.0") == \
        "sketch_attn_mid_block_attentions_0_transformer_blocks_0"


def test_module_mirror_state_dict_layouts():
    """The drop-in nn.Modules expose the reference's checkpoint key layouts (LGP: golden manifest)."""
    import json
    from modules.latent_predictor import LatentEdgePredictor
    from tests.util import GOLDEN
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    m = LatentEdgePredictor(9320, 4, 9)
    sd = m.state_dict()
    assert sorted(sd) == sorted(meta["manifest"]) and m.training
    for k, v in sd.items():
        assert list(v.shape) == meta["manifest"][k][0] and str(v.dtype) == meta["manifest"][k][1]
    assert sum(p.numel() for p in m.parameters()) == meta["n_params"]
    assert torch.count_nonzero(m.layers[0].bias) == 0                      # zeros_ init (:35)


WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import TINY
    from sketch2img_amd.dist import broadcast_state_dict, gather_latents, shard_range
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    ref = synthetic.unet_state_dict(TINY)
    sd = broadcast_state_dict(ref if rank == 0 else None, synthetic.unet_param_shapes(TINY), "cpu", src=0)
    assert list(sd) == list(ref) and all(torch.equal(sd[k].float(), ref[k]) for k in ref), "unet broadcast"
    lref = synthetic.lgp_state_dict(488)
    lsd = broadcast_state_dict(lref if rank == 0 else None, None, "cpu", src=0)
    assert list(lsd) == list(lref)
    assert all(torch.equal(lsd[k].float(), lref[k].float()) for k in lref), "lgp broadcast"
    assert lsd["layers.2.num_batches_tracked"].dtype == torch.int64
    total = 5
    first, count = shard_range(total, rank, world)
    x = synthetic.initial_latents(first, count, 8) + 1.0
    if count < 3:
        x = torch.cat([x, torch.zeros(3 - count, 4, 8, 8)])            # equal-size gather
    got = gather_latents(x, world, dst=0)
    if rank == 0:
        allx = torch.cat([g[:shard_range(total, r, world)[1]] for r, g in enumerate(got)])
        assert torch.equal(allx, synthetic.initial_latents(0, total, 8) + 1.0), "gather"
    else:
        assert got is None
    # the final gather of decoded images (uint8, the payload north_star names)
    from sketch2img_amd.dist import gather_images
    img = torch.full((2, 8, 8, 3), 10 * rank + 1, dtype=torch.uint8)
    gi = gather_images(img, world, dst=0)
    if rank == 0:
        assert len(gi) == 2 and all(g.dtype == torch.uint8 for g in gi)
        assert int(gi[0][0, 0, 0, 0]) == 1 and int(gi[1][0, 0, 0, 0]) == 11
    else:
        assert gi is None
    # the LGP-training collective: bucketed mean all-reduce of a flat gradient vector
    from sketch2img_amd.dist import allreduce_mean_
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_mean_(g, bucket_bytes=1024)                              # 4 buckets of 256 floats
    assert torch.equal(g, torch.arange(1000, dtype=torch.float32) * 1.5), "allreduce mean"
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_two_process_gloo_broadcast_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o[-2000:]


def test_bench_gpus_n_launches_itself():
    """`python bench.py --gpus 2` outside a launcher re-execs under torch.distributed.run (one rank per GPU),
    relays rank 0's single JSON line on stdout and the exit code.  --plumbing-check keeps the hot path out (no GPU here): rendezvous
    on 127.0.0.1, barrier, max over ranks.  A mismatching external launcher still trips the assert."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--plumbing-check", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                  # ONE line on stdout: the contract line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["self_launched"] and d["ranks"] == [[0, 0], [1, 1]] and d["steps"] == 3 and d["max_over_ranks"] >= 1.0
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--plumbing-check"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr


def test_bench_workload_strings_survive_a_120_character_cut():
    """The driver's parser keeps scalars and strings of at most 120 characters inside `config`: the workload name must fit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for C in (2, 4, 5):
        s = b.WORKLOADS[C].format(S=8, T=50, G=25, sched="DPM-Solver++ 2M")
        assert len(s) <= 120 and f"configs[{C - 1}]" in s
    assert abs(b.f_img_tflop(2, 50) - 107.65) < 0.01 and b.EPS_BOUND == 1e-3


def test_shard_range_covers_everything():
    from sketch2img_amd.dist import shard_range
    for total in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            nxt = 0
            for f, c in spans:
                assert f == nxt
                nxt += c


def test_dpm_tables_bit_exact_vs_oracle_and_scheduler_selection():
    from types import SimpleNamespace
    from oracle import dpmsolver as odpm
    from sketch2img_amd.sampler import DDIMTables, DPMTables
    from sketch2img_amd.schedulers import DDIMScheduler, DPMSolverMultistepScheduler
    from sketch2img_amd.modules.pipeline import AntiGradientPipeline
    for N in (10, 25, 50):
        a, b = DPMTables.make(N), odpm.make_tables(N)
        assert np.array_equal(a.timesteps, b.timesteps) and a.timesteps.dtype == np.int64
        assert torch.equal(a.lambda_t, b.lambda_t)
        seen = 0
        for i in range(N):
            od = a.order(i, seen)
            assert od == odpm.step_order(b, i, seen)
            assert a.coeffs(i, od) == odpm.step_coeffs(b, i, od)
            seen = min(seen + 1, 2)
    # the scheduler object decides which tables the pipeline builds (constructor as app.py:13-25)
    dpm = DPMSolverMultistepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                      num_train_timesteps=1000, trained_betas=None, predict_epsilon=True,
                                      thresholding=False, algorithm_type="dpmsolver++", solver_type="midpoint",
                                      lower_order_final=True)
    p = AntiGradientPipeline.__new__(AntiGradientPipeline)
    p.scheduler = dpm
    assert isinstance(p._tables(25), DPMTables)
    p.scheduler = DDIMScheduler()
    assert isinstance(p._tables(50), DDIMTables) and p._tables(50).timesteps[0] == 981
    p.scheduler = None
    assert isinstance(p._tables(50), DDIMTables)
    p.scheduler = DPMSolverMultistepScheduler(solver_type="heun")
    with pytest.raises(NotImplementedError):
        p._tables(25)
    p.scheduler = type("EulerDiscreteScheduler", (), {"config": SimpleNamespace()})()
    with pytest.raises(NotImplementedError):
        p._tables(25)


# ---------------------------------------------------------------------------------------- loading: nothing silent
def test_from_pretrained_never_falls_back_to_random_weights(tmp_path):
    """A path that does not resolve to weights raises (a hub id, as app.py:32 passes, is not a local folder); seeded
    synthetic weights / pseudo text embeddings need the explicit opt-in (None or synthetic=True)."""
    import json
    from sketch2img_amd.config import TINY
    from sketch2img_amd.modules.pipeline import AntiGradientPipeline, _config_from_folder
    from sketch2img_amd.vae import AutoencoderKL
    with pytest.raises(FileNotFoundError):
        AntiGradientPipeline.from_pretrained("runwayml/stable-diffusion-v1-5")
    with pytest.raises(FileNotFoundError):
        AutoencoderKL.from_pretrained("runwayml/stable-diffusion-v1-5", subfolder="vae")
    (tmp_path / "unet").mkdir()
    with pytest.raises(FileNotFoundError):                       # folder exists, weights do not
        AntiGradientPipeline.from_pretrained(str(tmp_path))
    p = AntiGradientPipeline.from_pretrained(None, unet_config=TINY)            # explicit opt-in
    assert p.allow_pseudo_text and p._encode_prompt("a", "cpu", 1, True, None).shape == (2, 77, TINY.cross_attention_dim)
    p.allow_pseudo_text = False
    with pytest.raises(RuntimeError):
        p._encode_prompt("a", "cpu", 1, True, None)
    # a real (tiny) checkpoint folder loads, its config.json decides the architecture, a wrong one is rejected
    from safetensors.torch import save_file
    from sketch2img_amd import synthetic
    sd = synthetic.unet_state_dict(TINY)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    cj = dict(block_out_channels=list(TINY.block_out_channels), attention_head_dim=list(TINY.num_heads), layers_per_block=2,
              cross_attention_dim=TINY.cross_attention_dim, norm_num_groups=TINY.norm_groups, sample_size=32)
    json.dump(cj, open(tmp_path / "unet" / "config.json", "w"))
    assert _config_from_folder(str(tmp_path)) == TINY
    q = AntiGradientPipeline.from_pretrained(str(tmp_path), text_encoder=lambda prompts: torch.zeros(len(prompts), 77, 64))
    assert not q.allow_pseudo_text and q.unet.cfg == TINY and set(q.unet.state_dict()) == set(sd)
    json.dump(dict(cj, block_out_channels=[32, 64, 128, 256]), open(tmp_path / "unet" / "config.json", "w"))
    with pytest.raises(ValueError):                              # config and weights disagree
        AntiGradientPipeline.from_pretrained(str(tmp_path))
    json.dump(dict(cj, class_embed_type="timestep"), open(tmp_path / "unet" / "config.json", "w"))
    with pytest.raises(NotImplementedError):
        AntiGradientPipeline.from_pretrained(str(tmp_path))


def test_scheduler_options_that_change_the_maths_are_rejected():
    from types import SimpleNamespace
    from sketch2img_amd.modules.pipeline import AntiGradientPipeline
    p = AntiGradientPipeline.__new__(AntiGradientPipeline)
    DDIM = type("DDIMScheduler", (), {})
    for bad in (dict(prediction_type="sample"), dict(predict_epsilon=False), dict(clip_sample=True), dict(beta_schedule="linear"),
                dict(trained_betas=[0.1, 0.2])):
        s = DDIM()
        s.config = SimpleNamespace(**bad)
        p.scheduler = s
        with pytest.raises(NotImplementedError):
            p._tables(50)
    s = DDIM()
    s.config = SimpleNamespace(prediction_type="epsilon", clip_sample=False, steps_offset=1)
    p.scheduler = s
    assert p._tables(50).timesteps[0] == 981 and not p._tables(50).v_prediction
    # the public SD2.1-768 scheduler_config.json: v-prediction is carried into the tables (the step kernels take it as a flag)
    s.config = SimpleNamespace(prediction_type="v_prediction", clip_sample=False, steps_offset=1)
    assert p._tables(50).v_prediction and p._tables(50).timesteps[0] == 981
    DPM = type("DPMSolverMultistepScheduler", (), {})
    s = DPM()
    s.config = SimpleNamespace(prediction_type="v_prediction", algorithm_type="dpmsolver++", solver_type="midpoint")
    p.scheduler = s
    assert p._tables(25).v_prediction and len(p._tables(25).timesteps) == 25


def test_vae_attention_keys_both_diffusers_spellings():
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import TINY_VAE
    from sketch2img_amd.vae import _HipVAEBlocks
    sd = dict(synthetic.vae_decoder_state_dict(TINY_VAE))
    legacy = [k for k in sd if ".attentions." in k]
    assert any(".query." in k for k in legacy) and any(".proj_attn." in k for k in legacy)
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    new = {}
    for k, v in sd.items():
        for old, nw in ren.items():
            if f".{old}." in k and ".attentions." in k:
                k = k.replace(f".{old}.", f".{nw}.")
                if k.endswith(".weight"):
                    v = v[:, :, None, None]                     # some re-saves keep the 1x1-conv shape
        new[k] = v
    back = _HipVAEBlocks.normalise_attention_keys(new)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    assert _HipVAEBlocks.normalise_attention_keys(sd).keys() == sd.keys()


def test_binding_loads_torch_before_libskg():
    """libskg.so must bind to the HIP runtime torch ships (one runtime per process): importing the binding alone has to
    import torch first.  (build() followed by smoke() in one process used to launch through a second runtime.)"""
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); import sketch2img_amd._lib as L; "
                        "ks = list(sys.modules); assert ks.index('torch') < ks.index('sketch2img_amd._lib'); print('order ok')" % ROOT],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "order ok" in r.stdout, r.stderr[-1500:]


def test_polyphase_packs_reproduce_upsample_conv_and_its_gradient():
    """unet.pack_conv_up2 / pack_conv_up2_dgrad (what skg_conv3x3_up2_f16 / skg_conv4x4s2_f16 consume), emulated with torch on the
    CPU: four 4-tap convolutions over the low-res map with pre-summed weights == F.interpolate(nearest, 2x) + 3x3 conv, and one
    4 x 4 stride-2 convolution over dY with the transposed pre-summed weights == its autograd gradient.  Weights on a 1/8
    grid, so the fp16 rounding of the sums is exact and the identity can be checked to fp32 accuracy."""
    import torch.nn.functional as F
    from sketch2img_amd.unet import pack_conv_up2, pack_conv_up2_dgrad
    g = torch.Generator().manual_seed(3)
    co, ci, H, W = 6, 5, 4, 7
    w = torch.randint(-8, 9, (co, ci, 3, 3), generator=g).float() / 8
    x = torch.randn(2, ci, H, W, generator=g, requires_grad=True)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    Wpp = pack_conv_up2(w, "cpu").float()                      # [4 phases 2a+b][co][4 taps * ci]
    assert Wpp.shape == (4, co, 4 * ci)
    xp = F.pad(x.detach(), (1, 1, 1, 1))
    y = torch.zeros_like(ref)
    for a in (0, 1):
        for b in (0, 1):
            acc = 0
            for ty in (0, 1):
                for tx in (0, 1):
                    wt = Wpp[2 * a + b][:, (2 * ty + tx) * ci:(2 * ty + tx + 1) * ci]
                    acc = acc + torch.einsum("oc,bchw->bohw", wt, xp[:, :, a + ty:a + ty + H, b + tx:b + tx + W])
            y[:, :, a::2, b::2] = acc
    assert torch.allclose(y, ref.detach(), atol=1e-5)
    dy = torch.randn(ref.shape, generator=g)
    gref, = torch.autograd.grad(ref, x, dy)
    W16 = pack_conv_up2_dgrad(w, "cpu").float()                # [ci][16 taps ky*4+kx][co]
    assert W16.shape == (ci, 16 * co)
    dyp = F.pad(dy, (1, 1, 1, 1))
    dx = 0
    for ky in range(4):
        for kx in range(4):
            wt = W16[:, (4 * ky + kx) * co:(4 * ky + kx + 1) * co]
            dx = dx + torch.einsum("co,bohw->bchw", wt, dyp[:, :, ky:ky + 2 * H:2, kx:kx + 2 * W:2])
    assert torch.allclose(dx, gref, atol=1e-5)



def test_winograd_pack_reproduces_conv_and_its_data_gradient():
    """unet.pack_conv_wino (U = G g G^T, component c = 4 i + j at columns [c Cin, (c + 1) Cin)) with the transforms csrc/wino.hip applies -
    V = B^T d B per 4 x 4 tile (stride 2, zero halo), M_c = V_c U_c^T, Y = A^T M A - reproduces F.conv2d(padding=1) and, with dgrad=True,
    its data gradient (autograd), to the fp16 rounding of U."""
    import torch.nn.functional as F
    from sketch2img_amd.unet import pack_conv_wino
    g = torch.Generator().manual_seed(0)
    rows, H, Cin, Cout = 2, 8, 64, 128
    x = torch.randn(rows, Cin, H, H, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).half().float()
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

    def wino(inp, U):                                   # inp [rows, C, H, H], U [O, 16 * C]
        C, O = inp.shape[1], U.shape[0]
        d = F.pad(inp, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                    # [rows, C, H/2, H/2, 4, 4]
        V = torch.einsum("ij,bcthjk,lk->bthilc", BT, d, BT).reshape(-1, 16, C)        # [Mt, 16, C]: the kernel's K order
        M = torch.einsum("mkc,okc->kmo", V, U.float().reshape(O, 16, C)).reshape(4, 4, rows, H // 2, H // 2, O)
        return torch.einsum("ai,ijrtso,bj->rtasbo", AT, M, AT).reshape(rows, H, H, O).permute(0, 3, 1, 2)

    ref = F.conv2d(x, w, padding=1)
    assert float((wino(x, pack_conv_wino(w, "cpu")) - ref).norm() / ref.norm()) < 6e-4
    dy = torch.randn(rows, Cout, H, H, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, padding=1).backward(dy)
    Ud = pack_conv_wino(w, "cpu", dgrad=True)
    assert Ud.shape == (Cin, 16 * Cout)
    assert float((wino(dy, Ud) - xr.grad).norm() / xr.grad.norm()) < 6e-4


def test_ff_block_pack_reproduces_the_geglu_feed_forward():
    """unet.pack_ff_block (what skg_ff_block_f16 consumes), emulated on the CPU with the kernel's own index arithmetic
    (csrc/ffblock.hip): a 1 KB piece is an MFMA A operand, lane 16 g + l holds A[l][8 g + i]; the accumulator lane (l, g) holds
    rows 4 g + r; the gated accumulators of two hidden tiles are the B operand of the second product with k-slot 8 g + i <->
    hidden unit 16 (i >> 2) + 4 g + (i & 3).  Against diffusers' FeedForward (GEGLU: value, gate = proj(x).chunk(2)) in fp32;
    weights and inputs on coarse grids so that fp16 storage is exact."""
    import torch.nn.functional as F
    from sketch2img_amd.unet import pack_ff_block
    g = torch.Generator().manual_seed(5)
    C, Fh, M = 320, 96, 16
    w1 = torch.randint(-8, 9, (2 * Fh, C), generator=g).float() / 64
    b1 = torch.randint(-8, 9, (2 * Fh,), generator=g).float() / 8
    w2 = torch.randint(-8, 9, (C, Fh), generator=g).float() / 64
    a = torch.randint(-16, 17, (M, C), generator=g).float() / 16
    pack, bias1 = pack_ff_block(w1, b1, w2, "cpu")
    nch, KS, NU = Fh // 32, C // 32, C // 16
    assert pack.shape == (nch, 60, 512) and pack.dtype == torch.float16 and bias1.shape == (nch, 4, 16)

    def a_operand(piece):                       # [512] lane-major -> A [16 rows l][32 k-slots 8 g + i]
        return piece.float().reshape(4, 16, 8).permute(1, 0, 2).reshape(16, 32)

    y = torch.zeros(C, M)
    for c in range(nch):
        h = []
        for t in range(4):
            acc = bias1[c, t][:, None].expand(16, M).clone()
            for ks in range(KS):
                acc = acc + a_operand(pack[c, t * KS + ks]) @ a[:, 32 * ks:32 * ks + 32].t()
            h.append(acc)                       # [16 hidden units of tile t][M rows]
        gated = [h[t] * F.gelu(h[2 + t]) for t in range(2)]
        # B operand of the second product: k-slot 8 g + i <-> (tile i >> 2, unit 4 g + (i & 3))
        bop = torch.zeros(32, M)
        for gg in range(4):
            for i in range(8):
                bop[8 * gg + i] = gated[i >> 2][4 * gg + (i & 3)]
        for u in range(NU):
            y[16 * u:16 * u + 16] += a_operand(pack[c, 4 * KS + u]) @ bop
    hidden = a @ w1.t() + b1
    ref = (hidden[:, :Fh] * F.gelu(hidden[:, Fh:])) @ w2.t()
    assert torch.allclose(y.t(), ref, atol=2e-4, rtol=1e-5), float((y.t() - ref).abs().max())


def test_ff_pack_proj_chunks_reproduce_proj_out():
    """unet.pack_ff_block(..., w_proj=) (what skg_ff_block_proj_f16 streams behind the feed-forward chunks), emulated on the CPU with
    the kernel's own index arithmetic (csrc/ffblock.hip, PROJ): the block output sits in accumulator tiles [16 channels, rows] with
    lane (l, g) holding rows 4 g + r; tiles 2 ks, 2 ks + 1 become the B operand of k-step ks with k-slot 8 g + i <-> (tile i >> 2,
    row 4 g + (i & 3)); chunk j's piece (t, ks) is the A operand [16 x 32] of output tile 4 j + t.  Against W_proj @ p3 exactly
    (inputs on coarse grids); the feed-forward chunks in front are untouched and the 20 W2 slots of the proj chunks are zero."""
    from sketch2img_amd.unet import pack_ff_block
    g = torch.Generator().manual_seed(10)
    C, Fh, M = 320, 1280, 16
    w1 = torch.randint(-8, 9, (2 * Fh, C), generator=g).float() / 64
    b1 = torch.randint(-8, 9, (2 * Fh,), generator=g).float() / 64
    w2 = torch.randint(-8, 9, (C, Fh), generator=g).float() / 64
    wp = torch.randint(-8, 9, (C, C), generator=g).float() / 64
    p3 = torch.randint(-16, 17, (M, C), generator=g).float() / 16
    pack0, bias0 = pack_ff_block(w1, b1, w2, "cpu")
    pack, bias = pack_ff_block(w1, b1, w2, "cpu", w_proj=wp)
    nch = Fh // 32
    assert pack.shape == (nch + 5, 60, 512) and torch.equal(pack[:nch], pack0) and torch.equal(bias, bias0)
    assert float(pack[nch:, 40:].abs().max()) == 0

    def a32(piece):      # lane 16 g + l holds A[l][8 g + i]
        return piece.float().reshape(4, 16, 8).permute(1, 0, 2).reshape(16, 32)

    tiles = [p3[:, 16 * u:16 * u + 16].t() for u in range(C // 16)]      # accumulator tiles [16 channels, M rows]
    out = torch.zeros(C, M)
    for j in range(5):
        for t in range(4):
            for ks in range(10):
                b = torch.zeros(32, M)
                for gg in range(4):
                    for i in range(8):
                        b[8 * gg + i] = tiles[2 * ks + (i >> 2)][4 * gg + (i & 3)]
                out[16 * (4 * j + t):16 * (4 * j + t) + 16] += a32(pack[nch + j, t * 10 + ks]) @ b
    assert torch.equal(out.t(), p3 @ wp.t())


def test_xattn_packs_reproduce_cross_attention():
    """unet.pack_xattn_weights / pack_xattn_kv (what skg_xattn_block_f16 consumes), emulated on the CPU with the kernel's own
    index arithmetic (csrc/xattn.hip): K = 32 pieces are A operands [16 x 32] with lane 16 g + l holding A[l][8 g + i], K = 16
    pieces [16 x 16] with A[l][4 g + i]; an accumulator lane (l, g) holds rows 4 g + r, and accumulator tiles feed the next
    product as B operands with k-slot 8 g + i <-> (tile i >> 2, row 4 g + (i & 3)) resp. 4 g + i <-> (last tile, row 4 g + i).
    Against to_out(softmax(q k^T / sqrt(d)) v) in fp32 with 77 valid keys; inputs on coarse grids (fp16 storage exact)."""
    from sketch2img_amd.unet import pack_xattn_kv, pack_xattn_weights
    g = torch.Generator().manual_seed(9)
    C, heads, dh, M, Lp, L, rows = 320, 8, 40, 16, 80, 77, 2
    wq = torch.randint(-8, 9, (C, C), generator=g).float() / 64
    wo = torch.randint(-8, 9, (C, C), generator=g).float() / 64
    a = torch.randint(-16, 17, (M, C), generator=g).float() / 16
    K = torch.randint(-16, 17, (rows * Lp, C), generator=g).float() / 16
    V = torch.randint(-16, 17, (rows * Lp, C), generator=g).float() / 16
    wp = pack_xattn_weights(wq, wo, heads, "cpu")
    kv = pack_xattn_kv(K.half(), V.half(), rows, Lp, L, heads)
    assert wp.shape == (heads, 60, 512) and kv.shape == (rows, heads, 16, 512)
    img = 1
    scale = dh ** -0.5

    def a32(piece):
        return piece.float().reshape(4, 16, 8).permute(1, 0, 2).reshape(16, 32)

    def a16(piece):
        return piece.float().reshape(4, 16, 4).permute(1, 0, 2).reshape(16, 16)

    def b32(t0, t1):                            # two accumulator tiles [16, M] -> B operand [32 k-slots, M]
        out = torch.zeros(32, M)
        for gg in range(4):
            for i in range(8):
                out[8 * gg + i] = (t0, t1)[i >> 2][4 * gg + (i & 3)]
        return out

    y = torch.zeros(C, M)
    for h in range(heads):
        W = wp[h].reshape(-1)
        q = [sum(a32(W[(t * 10 + ks) * 512:(t * 10 + ks + 1) * 512]) @ a[:, 32 * ks:32 * ks + 32].t() for ks in range(10)) for t in range(3)]
        q = [t * scale for t in q]
        kimg, vimg = kv[img, h].reshape(-1)[:4096], kv[img, h].reshape(-1)[4096:]
        s = []
        for kt in range(5):
            acc = a32(kimg[kt * 512:(kt + 1) * 512]) @ b32(q[0], q[1]) + a16(kimg[5 * 512 + kt * 256:5 * 512 + (kt + 1) * 256]) @ q[2]
            s.append(acc)
        S = torch.cat(s)                                        # [80 keys, M]
        S[L:] = float("-inf")
        P = torch.softmax(S, 0)
        pt = [P[16 * kt:16 * kt + 16] for kt in range(5)]
        o = []
        for dt in range(3):
            acc = a32(vimg[(dt * 2) * 512:(dt * 2 + 1) * 512]) @ b32(pt[0], pt[1]) + a32(vimg[(dt * 2 + 1) * 512:(dt * 2 + 2) * 512]) @ b32(pt[2], pt[3]) \
                + a16(vimg[6 * 512 + dt * 256:6 * 512 + (dt + 1) * 256]) @ pt[4]
            o.append(acc)
        wo_img = W[30 * 512:]
        for u in range(20):
            y[16 * u:16 * u + 16] += a32(wo_img[u * 512:(u + 1) * 512]) @ b32(o[0], o[1]) + a16(wo_img[20 * 512 + u * 256:20 * 512 + (u + 1) * 256]) @ o[2]
    # reference
    qr = (a @ wq.t()).reshape(M, heads, dh).permute(1, 0, 2)
    kr = K[img * Lp:img * Lp + L].reshape(L, heads, dh).permute(1, 0, 2)
    vr = V[img * Lp:img * Lp + L].reshape(L, heads, dh).permute(1, 0, 2)
    att = torch.softmax(qr @ kr.transpose(1, 2) * scale, -1) @ vr
    ref = att.permute(1, 0, 2).reshape(M, C) @ wo.t()
    assert torch.allclose(y.t(), ref, atol=2e-4, rtol=1e-4), float((y.t() - ref).abs().max())


def test_unet_pack_stage_runs_on_cpu_and_builds_the_fused_block_packs():
    """HipUNet's weight-pack stage is host code: run it on the CPU for the full SD1.5 layout (no kernel is launched) - the five
    C = 320 transformer blocks get the fragment-major packs of the fused feed-forward and cross-attention launches."""
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15
    from sketch2img_amd.unet import HipUNet
    net = HipUNet(SD15, synthetic.unet_state_dict(SD15), "cpu", need_backward=False)
    xp = sorted(k for k in net.W if k.endswith(".attn2.xpack"))
    fp = sorted(k for k in net.W if k.endswith(".ff.pack"))
    assert len(xp) == 5 and len(fp) == 5
    assert all(net.W[k].shape == (8, 60, 512) and net.W[k].dtype == torch.float16 for k in xp)
    assert all(net.W[k].shape == (40, 60, 512) for k in fp)
    assert all(net.W[k[:-len("pack")] + "bias1"].shape == (40, 4, 16) for k in fp)
