"""Drop-in for the reference's modules/pipeline.py: AntiGradientPipeline.

Same public surface as the reference class: ``from_pretrained(path, vae=, torch_dtype=,
scheduler=)``, ``.to(device)``, ``.unet`` / ``.vae`` / ``.scheduler`` / ``.vae_scale_factor``, ``.setup_lgp(lgp)``
and ``__call__`` with the reference's keyword list (modules/pipeline.py:20-37) and return convention (:127-130:
the bare list of PIL images when ``return_dict`` is true, ``(images, None)`` otherwise).  The sampling loop, the
UNet, the LGP and the guidance gradient run in libskg.so; there is no diffusers dependency.

Around the hot path:
  * ``vae``          ``sketch2img_amd.vae.AutoencoderKL`` decodes / encodes on the HIP kernels; any other object with
                     ``decode(latents) -> tensor | .sample`` (and ``encode`` for app.py:109) is used as a PyTorch module;
  * ``text_encoder`` a callable ``(list[str]) -> (B, 77, D) tensor`` - ``sketch2img_amd.clip_text.PromptEncoder``
                     (CLIP tokenizer + the text transformer on the HIP kernels) is picked up automatically when the
                     checkpoint folder has ``tokenizer/`` and ``text_encoder/``.  Seeded pseudo-embeddings
                     (deterministic in the prompt text) and seeded synthetic UNet weights exist for boxes without
                     checkpoints, but only behind the explicit opt-in ``from_pretrained(None)`` / ``synthetic=True``:
                     a path that does not resolve to weights raises.
Schedulers: DDIM (the BASELINE metric) and DPM-Solver++ 2M (what app.py configures); see `sketch2img_amd.schedulers`.
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
from types import SimpleNamespace
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from ..config import SD15, SD21, UNetConfig, tap_channels
from ..sampler import DDIMTables, DPMTables, HipSampler
from .latent_predictor import LatentEdgePredictor, hook_unet

logger = logging.getLogger(__name__)      # (the reference: diffusers.utils.logging.get_logger(__name__), modules/pipeline.py:11)


class UNetFacade:
    """Stands where ``pipe.unet`` is: the attributes app.py / the reference pipeline touch, plus the engine."""

    def __init__(self, cfg: UNetConfig, state_dict, device, residual_fp32: bool = False):
        self.cfg = cfg
        self.residual_fp32 = bool(residual_fp32)      # HipUNet's accuracy mode (see AntiGradientPipeline.from_pretrained)
        self.config = SimpleNamespace(sample_size=cfg.sample_size, in_channels=cfg.in_channels,
                                      cross_attention_dim=cfg.cross_attention_dim)
        self.in_channels = cfg.in_channels
        self.dtype = torch.float16
        self._state_dict = state_dict
        self._device = torch.device(device)
        self._hip = None
        self._feature_taps = None

    @property
    def device(self):
        return self._device

    def to(self, device=None, dtype=None):
        if device is not None and torch.device(device) != self._device:
            self._device, self._hip = torch.device(device), None
        return self

    def enable_xformers_memory_efficient_attention(self):       # app.py:43 - fused attention is always on
        return None

    @property
    def hip(self):
        if self._hip is None:
            if self._device.type != "cuda":
                raise RuntimeError("sketch2img_amd computes on the GPU only: call pipe.to('cuda') first")
            from ..unet import HipUNet
            self._hip = HipUNet(self.cfg, self._state_dict, self._device, residual_fp32=self.residual_fp32)
        return self._hip

    def state_dict(self):
        return self._state_dict

    def __call__(self, sample, timestep, encoder_hidden_states, **kwargs):
        """UNet2DConditionModel-style call (evaluation.py:94): sample (rows,4,h,w) in the row order
        [uncond rows; cond rows], returns an object with ``.sample`` (fp32 NCHW epsilon).  Fills the hooked
        feature taps (hook_unet) like the reference's forward hooks do."""
        from .. import ops
        from ..unet import CIN_PAD
        rows, _, h, w = sample.shape
        assert h == w
        net = self.hip
        if net.ctx is None or net.ctx.get("src") is not encoder_hidden_states:
            net.prepare_context(encoder_hidden_states)
            net.ctx["src"] = encoder_hidden_states
        x32 = ops.nchw_to_nhwc(sample.to(self._device, torch.float32).contiguous(), CIN_PAD)
        eps, taps = net.forward(x32, int(timestep), rows, h)
        if self._feature_taps is not None:
            for tp, (t, s) in zip(self._feature_taps, taps):
                tp._nhwc = (t, rows, s)
        return SimpleNamespace(sample=ops.nhwc_to_nchw(eps, rows, self.cfg.out_channels, h, w))


WEIGHT_FILES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                "diffusion_pytorch_model.bin", "diffusion_pytorch_model.fp16.bin")


def load_diffusers_weights(folder: str, what: str):
    """state_dict of a diffusers-layout sub-folder (``<root>/unet``, ``<root>/vae``): single-file or sharded
    safetensors / .bin, plain or ``.fp16`` variant.  Raises FileNotFoundError when nothing loadable is there - a
    path that does not resolve to weights must never fall through to random weights (a hub id such as
    "runwayml/stable-diffusion-v1-5", which app.py:32 passes, is not a local folder: there is no network here)."""
    if not os.path.isdir(folder):
        raise FileNotFoundError(
            f"{what}: {folder!r} is not a local diffusers-layout folder (hub ids cannot be resolved: no network). "
            f"Pass a local checkpoint folder, or None / synthetic=True for seeded synthetic weights.")
    for name in WEIGHT_FILES:
        f = os.path.join(folder, name)
        if os.path.exists(f):
            if name.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(f)
            return torch.load(f, map_location="cpu")
    for idx in ("diffusion_pytorch_model.safetensors.index.json", "diffusion_pytorch_model.bin.index.json"):
        f = os.path.join(folder, idx)
        if os.path.exists(f):
            sd = {}
            for shard in sorted(set(json.load(open(f))["weight_map"].values())):
                sp = os.path.join(folder, shard)
                if shard.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd.update(load_file(sp))
                else:
                    sd.update(torch.load(sp, map_location="cpu"))
            return sd
    raise FileNotFoundError(f"{what}: no diffusion_pytorch_model.{{safetensors,bin}} (plain, .fp16 or sharded) in {folder!r}")


def _load_unet_weights(path: Optional[str], cfg: UNetConfig, synthetic_ok: bool):
    """``path`` None (or synthetic=True): seeded synthetic weights, the explicit opt-in used on boxes without
    checkpoints.  Otherwise the folder MUST hold the weights, and their shapes must match ``cfg``."""
    from .. import synthetic
    if path is None or synthetic_ok and not os.path.isdir(os.path.join(str(path), "unet")):
        return synthetic.unet_state_dict(cfg)
    sd = load_diffusers_weights(os.path.join(path, "unet"), "AntiGradientPipeline.from_pretrained")
    want = synthetic.unet_param_shapes(cfg)
    missing = [k for k in want if k not in sd]
    wrong = [k for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])
             and tuple(sd[k].shape) != tuple(want[k]) + (1, 1) and tuple(sd[k].shape) + (1, 1) != tuple(want[k])]
    if missing or wrong:
        raise ValueError(f"unet checkpoint does not match the architecture {cfg}: {len(missing)} missing keys "
                         f"(e.g. {missing[:3]}), {len(wrong)} shape mismatches (e.g. {wrong[:3]})")
    return sd


def _config_from_folder(path: Optional[str]) -> UNetConfig:
    """UNetConfig from ``<path>/unet/config.json``; every field that changes the graph is read and anything this
    implementation does not cover is rejected instead of ignored."""
    if not path:
        return SD15
    cj = os.path.join(path, "unet", "config.json")
    if not os.path.exists(cj):
        return SD15
    c = json.load(open(cj))
    boc = tuple(c.get("block_out_channels", SD15.block_out_channels))
    ahd = c.get("attention_head_dim", 8)            # diffusers' "attention_head_dim" is a head COUNT for these models
    heads = tuple(ahd) if isinstance(ahd, (list, tuple)) else (int(ahd),) * len(boc)
    down = c.get("down_block_types", ["CrossAttnDownBlock2D"] * (len(boc) - 1) + ["DownBlock2D"])
    up = c.get("up_block_types", ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * (len(boc) - 1))
    unsupported = []
    if list(down) != ["CrossAttnDownBlock2D"] * (len(boc) - 1) + ["DownBlock2D"] or \
            list(up) != ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * (len(boc) - 1):
        unsupported.append(f"block types {down} / {up}")
    for k, ok in (("act_fn", "silu"), ("center_input_sample", False), ("flip_sin_to_cos", True), ("freq_shift", 0),
                  ("downsample_padding", 1), ("mid_block_scale_factor", 1), ("dual_cross_attention", False),
                  ("only_cross_attention", False), ("class_embed_type", None), ("num_class_embeds", None),
                  ("upcast_attention", None), ("resnet_time_scale_shift", "default")):
        v = c.get(k, ok)
        if k == "upcast_attention":
            continue                                  # fp32 softmax statistics are what the flash kernel always does
        if v != ok:
            unsupported.append(f"{k}={v!r}")
    if len(heads) != len(boc) or len(boc) != 4:
        unsupported.append(f"block_out_channels {boc} / attention_head_dim {ahd}")
    if unsupported:
        raise NotImplementedError("unet/config.json asks for features outside the SD1.x / SD2.x UNet this package "
                                  "implements: " + "; ".join(unsupported))
    return UNetConfig(in_channels=c.get("in_channels", 4), out_channels=c.get("out_channels", 4), block_out_channels=boc,
                      layers_per_block=c.get("layers_per_block", 2), cross_attention_dim=c.get("cross_attention_dim", 768),
                      num_heads=heads, use_linear_projection=bool(c.get("use_linear_projection", False)),
                      norm_groups=c.get("norm_num_groups", 32), sample_size=c.get("sample_size", 64))


class AntiGradientPipeline:
    def __init__(self, unet: UNetFacade, vae=None, scheduler=None, text_encoder: Optional[Callable] = None,
                 allow_pseudo_text: bool = False):
        self.unet, self.vae, self.scheduler, self.text_encoder = unet, vae, scheduler, text_encoder
        self.allow_pseudo_text = allow_pseudo_text      # seeded pseudo text embeddings: synthetic pipelines only
        self.vae_scale_factor = 8
        self.lgp_model: Optional[LatentEdgePredictor] = None
        self.feature_blocks = None
        self.safety_checker = None

    # ------------------------------------------------------------------ construction (app.py:32-46,67-70)
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, vae=None, torch_dtype=None, scheduler=None,
                        text_encoder=None, unet_config: Optional[UNetConfig] = None, synthetic: bool = False,
                        residual_fp32: bool = False, **kwargs):
        """``pretrained_model_name_or_path``: a local diffusers-layout folder (``unet/``, optionally ``tokenizer/`` +
        ``text_encoder/``).  ``None`` or ``synthetic=True`` is the explicit opt-in to seeded synthetic UNet weights and
        seeded pseudo text embeddings (boxes without checkpoints, tests, bench.py); any other path that does not
        resolve to weights raises FileNotFoundError instead of sampling noise from random weights.

        ``residual_fp32=True`` (also reachable as ``torch_dtype=torch.float32``, what a caller who wants the fp32 reference's
        numbers passes to the reference, app.py:34 being ``torch.float16``): the UNet keeps its residual stream and the
        convolution outputs that feed a norm as (hi, lo) fp16 pairs - max eps deviation from the fp32 reference <= 1e-3
        (north_star's bound; the all-fp16 default, like the reference's own fp16 GPU path, is ~1.7e-3 away).  Works with
        ``setup_lgp`` guidance and with both SatMixin injections."""
        if torch_dtype == torch.float32 and not residual_fp32:
            # NOT fp32 compute: the reference's torch_dtype=float32 means an fp32 model; here it selects the (hi, lo) fp16-pair mode
            # (~12 % slower than fp16, eps within 1e-3 of the fp32 reference) - say so once instead of mapping silently
            logger.warning("AntiGradientPipeline.from_pretrained(torch_dtype=torch.float32): mapped to residual_fp32=True (the accuracy "
                           "mode: fp16 MFMA compute with a (hi, lo) fp16-pair residual stream, eps within 1e-3 of the fp32 reference); "
                           "there is no fp32 compute path")
        residual_fp32 = bool(residual_fp32) or torch_dtype == torch.float32
        root = pretrained_model_name_or_path
        synthetic = synthetic or root is None
        cfg = unet_config or _config_from_folder(root)
        sd = _load_unet_weights(root, cfg, synthetic)
        if text_encoder is None and root and os.path.isdir(os.path.join(root, "tokenizer")) \
                and os.path.isdir(os.path.join(root, "text_encoder")):
            from ..clip_text import PromptEncoder
            text_encoder = PromptEncoder.from_pretrained(root)
        return cls(UNetFacade(cfg, sd, "cpu", residual_fp32=residual_fp32), vae=vae, scheduler=scheduler,
                   text_encoder=text_encoder, allow_pseudo_text=synthetic)

    def to(self, device):
        self.unet.to(device)
        if self.vae is not None and hasattr(self.vae, "to"):
            self.vae.to(device)
        if self.text_encoder is not None and hasattr(self.text_encoder, "to"):
            self.text_encoder.to(device)
        return self

    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    def setup_lgp(self, lgp):                                  # modules/pipeline.py:15-17
        self.lgp_model = lgp
        self.feature_blocks = hook_unet(self.unet)

    # ------------------------------------------------------------------ helpers of the diffusers base class
    def check_inputs(self, prompt, height, width, callback_steps):
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}")

    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt):
        """[uncond rows; cond rows] like the diffusers helper (modules/pipeline.py:55-57)."""
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        if negative_prompt is None:
            negs = [""] * len(prompts)
        elif isinstance(negative_prompt, str):
            negs = [negative_prompt] * len(prompts)
        else:
            negs = list(negative_prompt)
        if len(negs) != len(prompts):
            raise ValueError("`negative_prompt` and `prompt` must have the same batch size")
        if self.text_encoder is None and not self.allow_pseudo_text:
            raise RuntimeError("AntiGradientPipeline: no text encoder - the checkpoint folder has no tokenizer/ + "
                               "text_encoder/; pass text_encoder=<callable (list[str]) -> (B, 77, D)>, or build the "
                               "pipeline with synthetic=True to get seeded pseudo embeddings")
        enc = self.text_encoder or self._pseudo_text_encoder
        cond = enc(prompts).repeat_interleave(num_images_per_prompt, 0)
        if not do_classifier_free_guidance:
            return cond
        unc = enc(negs).repeat_interleave(num_images_per_prompt, 0)
        return torch.cat([unc, cond])

    def _pseudo_text_encoder(self, prompts: List[str]) -> torch.Tensor:
        out = []
        for p in prompts:
            seed = int.from_bytes(hashlib.sha256(p.encode()).digest()[:4], "little")
            g = torch.Generator().manual_seed(seed)
            out.append(torch.randn(77, self.unet.cfg.cross_attention_dim, generator=g))
        return torch.stack(out)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if isinstance(generator, torch.Generator) else "cpu"
            latents = torch.randn(shape, generator=generator if isinstance(generator, torch.Generator) else None,
                                  device=gdev, dtype=torch.float32)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
        return latents.to(device=device, dtype=torch.float32)       # init_noise_sigma = 1 for DDIM

    def decode_latents(self, latents):
        """diffusers decode_latents: /0.18215, VAE decode, /2 + 0.5, clamp, NHWC fp32 numpy."""
        if self.vae is None:
            # latent preview when the pipeline was built without a VAE (pass vae=sketch2img_amd.vae.AutoencoderKL(...))
            m = torch.tensor([[0.298, 0.207, 0.208], [0.187, 0.286, 0.173], [-0.158, 0.189, 0.264],
                              [-0.184, -0.271, -0.473]], device=latents.device)
            img = torch.einsum("bchw,cr->brhw", latents.float(), m)
            img = torch.nn.functional.interpolate(img, scale_factor=8.0, mode="nearest")
        elif hasattr(self.vae, "decode_latents"):
            # sketch2img_amd.vae.AutoencoderKL: the whole of decode_latents on libskg.so kernels, NHWC fp32 out
            return self.vae.decode_latents(latents).cpu().numpy()
        else:
            img = self.vae.decode((latents / 0.18215).to(getattr(self.vae, "dtype", latents.dtype)))
            img = getattr(img, "sample", img)
        img = (img.float() / 2 + 0.5).clamp(0, 1)
        return img.detach().cpu().permute(0, 2, 3, 1).numpy()

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        return [Image.fromarray(im) for im in images]

    def run_safety_checker(self, image, device, dtype):
        return image, None

    # ------------------------------------------------------------------ modules/pipeline.py:19-130
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None,
                 callback_steps: Optional[int] = 1, sketch_image=None):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        if height != width:
            raise RuntimeError("sketch guidance resizes with size=latents.shape[2] only: square images (SURVEY Q8)")
        if eta != 0.0:
            raise NotImplementedError("only eta = 0 (deterministic sampling) is implemented")
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        device = self._execution_device
        if guidance_scale <= 1.0:
            raise NotImplementedError("classifier-free guidance is always on in the hot path (guidance_scale > 1)")
        ehs = self._encode_prompt(prompt, device, num_images_per_prompt, True, negative_prompt)
        S = batch_size * num_images_per_prompt
        tab = self._tables(num_inference_steps)
        lat = self.prepare_latents(S, self.unet.in_channels, height, width, torch.float32, device, generator, latents)

        net = self.unet.hip
        net.prepare_context(ehs)
        lgp = None
        target = None
        if sketch_image is not None:
            if self.lgp_model is None:
                raise AttributeError("'AntiGradientPipeline' object has no attribute 'lgp_model' (call setup_lgp)")
            lgp = self.lgp_model._engine(tap_channels(self.unet.cfg), device)
            target = torch.as_tensor(sketch_image).to(device, torch.float32)
            if target.shape[0] not in (1, S):
                raise RuntimeError(f"The size of tensor a ({target.shape[0]}) must match the size of tensor b ({S})")
        sampler = HipSampler(net, lgp)
        cb = None
        if callback is not None:
            cb = lambda i, t, x: callback(i, t, x) if i % callback_steps == 0 else None
        out = sampler.sample(lat, target, num_inference_steps, guidance_scale, 1.6, callback=cb, tables=tab)
        self.last_aux = sampler.last_aux
        if lgp is not None:
            self.lgp_model._sync_running_stats()
        if self.feature_blocks is not None:
            for tp in self.feature_blocks:          # the reference deletes block.output after use (:149)
                tp._nhwc = None
        if output_type == "latent":
            image = out
        else:
            image = self.decode_latents(out)
            image, _ = self.run_safety_checker(image, device, torch.float16)
            if output_type == "pil":
                image = self.numpy_to_pil(image)
        if not return_dict:
            return (image, None)
        return image                                # sic: the reference returns the bare list (:130, Q10)

    def _tables(self, num_inference_steps: int):
        sch = self.scheduler
        if sch is None:
            return DDIMTables.make(num_inference_steps)
        cfg = getattr(sch, "config", sch)
        get = lambda k, d: (cfg.get(k, d) if isinstance(cfg, dict) else getattr(cfg, k, d))
        name = type(sch).__name__
        # options that change the arithmetic are rejected, never ignored (a diffusers scheduler object carries them)
        # prediction_type "v_prediction" (the public SD2.1-768 checkpoint's scheduler_config.json): the UNet output is v; the
        # latent update kernels derive eps / x0 from it (skg_cfg_ddim_step / skg_cfg_dpmpp2m_step, vpred = 1).  The reference
        # itself always builds its schedulers for epsilon prediction (modules/clip_guided_inf.py:14-26)
        ptype = get("prediction_type", None)
        if ptype is None:
            ptype = "epsilon" if get("predict_epsilon", True) else "sample"
        if ptype not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type={ptype!r}: 'epsilon' and 'v_prediction' are implemented")
        vpred = ptype == "v_prediction"
        if get("beta_schedule", "scaled_linear") != "scaled_linear" or get("trained_betas", None) is not None:
            raise NotImplementedError("scaled_linear betas (Stable Diffusion) only")
        if "DDIM" in name and get("clip_sample", False):
            raise NotImplementedError("DDIM clip_sample=True is not implemented (Stable Diffusion uses clip_sample=False; "
                                      "note that diffusers' DDIMScheduler() default is True)")
        if "DPMSolverMultistep" in name or get("algorithm_type", None) is not None:
            if get("algorithm_type", "dpmsolver++") != "dpmsolver++" or get("solver_type", "midpoint") != "midpoint":
                raise NotImplementedError("DPM-Solver: only algorithm_type='dpmsolver++', solver_type='midpoint'")
            if get("thresholding", False):
                raise NotImplementedError("DPM-Solver++ without thresholding only")
            return DPMTables.make(num_inference_steps, get("num_train_timesteps", 1000), get("beta_start", 0.00085),
                                  get("beta_end", 0.012), get("lower_order_final", True), get("solver_order", 2), vpred)
        if "DDIM" not in name and not isinstance(sch, (dict, SimpleNamespace)):
            raise NotImplementedError(f"{name}: DDIMScheduler and DPMSolverMultistepScheduler are implemented")
        return DDIMTables.make(num_inference_steps, get("num_train_timesteps", 1000), get("beta_start", 0.00085),
                               get("beta_end", 0.012), get("steps_offset", 1), get("set_alpha_to_one", False), vpred)

    # ------------------------------------------------------------------ modules/pipeline.py:132-161
    def get_noise_level(self, noise, timesteps):
        tab = self._tables(50)
        s = (1 - tab.alphas_cumprod[int(timesteps)]) ** 0.5
        return s.reshape(1, 1, 1, 1).to(noise.device) * noise
