"""One LGP training step on the libskg.so kernels (reference trainer.py:208-252).

    taps of a frozen UNet forward on noised latents  ->  LatentEdgePredictor (train-mode BatchNorm over the whole
    batch)  ->  mse_loss(result, sketch latents)  ->  gradients of every LGP parameter  ->  optimizer step,
    with the gradients all-reduced across ranks first (the reference wraps the model in DDP, bucket_cap_mb = 15).

What differs from the reference, on purpose: accelerate's fp16 autocast + GradScaler becomes fp16 compute with a
static power-of-two loss scale and fp32 master weights; bitsandbytes' AdamW8bit becomes plain fp32 AdamW (same
hyper-parameters: train.yaml lr 2e-4, weight_decay 1e-2, eps 1e-8, "constant_with_warmup" over 150 steps).  The
reference's loop also contains a NameError (trainer.py:230 vs :236) and cannot run as published; the
intended computation is restated in tests against PyTorch autograd of the oracle LGP.

Layer 0 keeps the re-association of sketch2img_amd/lgp.py: forward = per-tap GEMM at native resolution + bilinear
gather; its weight gradient is the adjoint: scatter d(pre-activation) back to each tap's resolution, then
dW0[:, tap] = dP_tap^T . F_tap (K = the tap's pixel count), and dW0[:, extras] = dZ0^T . E for the 40 noise-level /
sinusoid channels.  All weight gradients are plain GEMMs on transposed operands with fp32 output.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .lgp import BNS, LIN, LOSS_SCALE, OUT_PAD, SEED_LD

TRAINABLE = [f"layers.{i}.{n}" for i in (0, 2, 3, 5, 6, 8, 9, 11, 12) for n in ("weight", "bias")]


class HipLGPTrainer:
    def __init__(self, state_dict: Dict[str, torch.Tensor], tap_channels: Sequence[int], device="cuda",
                 lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 warmup_steps: int = 150):
        dev = self.dev = torch.device(device)
        self.tap_channels = list(tap_channels)
        self.E = sum(tap_channels)
        self.lr, self.betas, self.eps, self.wd, self.warmup = lr, betas, eps, weight_decay, warmup_steps
        self.step_count = 0
        # flat fp32 master vector in the reference's state_dict order (weights, biases of the 5 Linear and 4 BN)
        self.layout: Dict[str, Tuple[int, torch.Size]] = {}
        off = 0
        for k in TRAINABLE:
            t = state_dict[k]
            self.layout[k] = (off, t.shape)
            off += (t.numel() + 7) // 8 * 8                  # keep every tensor 16-byte aligned in the fp16 copy
        self.n = off
        self.p = torch.zeros(off, device=dev, dtype=torch.float32)
        for k, (o, shp) in self.layout.items():
            self.p[o:o + shp.numel()] = state_dict[k].detach().to(dev, torch.float32).reshape(-1)
        self.p16 = self.p.to(torch.float16)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.running_mean = [state_dict[f"layers.{i}.running_mean"].detach().to(dev, torch.float32).clone() for i in BNS]
        self.running_var = [state_dict[f"layers.{i}.running_var"].detach().to(dev, torch.float32).clone() for i in BNS]
        self.num_batches_tracked = [int(state_dict[f"layers.{i}.num_batches_tracked"]) for i in BNS]
        self.H0 = state_dict["layers.0.weight"].shape[0]
        self.out_dim = state_dict["layers.12.weight"].shape[0]
        assert state_dict["layers.0.weight"].shape[1] == self.E + 40 and self.out_dim <= OUT_PAD

    # ------------------------------------------------------------------------------------------ views
    def w16(self, key: str) -> torch.Tensor:
        o, shp = self.layout[key]
        return self.p16[o:o + shp.numel()].view(shp)

    def grad_view(self, g: torch.Tensor, key: str) -> torch.Tensor:
        o, shp = self.layout[key]
        return g[o:o + shp.numel()].view(shp)

    # ------------------------------------------------------------------------------------------ fwd + bwd
    @torch.no_grad()
    def loss_and_grads(self, taps: Sequence[Tuple[torch.Tensor, int]], noise_level: torch.Tensor,
                       target: torch.Tensor):
        """taps: 9 x (fp16 [B*s*s, C_i], s) of the frozen UNet; noise_level fp32 [B,4,h,h] = sqrt(1-abar_t) * noise
        per sample (trainer.py:196-204); target fp32 [B,4,h,h] (the sketch latents).
        Returns (loss scalar tensor, flat fp32 gradient vector multiplied by LOSS_SCALE)."""
        B, _, h, _ = target.shape
        hw, M = h * h, B * h * h
        dev = self.dev
        W0, b0 = self.w16("layers.0.weight"), self.w16("layers.0.bias")
        noise_level = noise_level.to(dev, torch.float32).contiguous()
        target = target.to(dev, torch.float32).contiguous()
        # ---- forward (rows = B samples, ONE BatchNorm batch of B*hw rows)
        P, sizes, off = [], [], 0
        for (F, s), C in zip(taps, self.tap_channels):
            assert F.shape == (B * s * s, C)
            P.append(ops.gemm(F, W0[:, off:off + C], out_f32=True))
            sizes.append(s)
            off += C
        Ex = ops.lgp_extra_features(noise_level, 1.0, B, B, h, 64)               # [M, 64], 40 valid columns
        if h % 8 == 0 and self.H0 % 128 == 0:
            w0x = torch.nn.functional.pad(W0[:, self.E:], (0, 64 - 40)).contiguous()
            Z = ops.lgp_layer0_gather(P + [ops.gemm(Ex, w0x, out_f32=True)], sizes + [h], None, b0, noise_level, 1.0, B,
                                      h, self.H0, rows=B)
        else:
            Z = ops.lgp_layer0_gather(P, sizes, W0[:, self.E:], b0, noise_level, 1.0, B, h, self.H0, rows=B)
        zs, As, stats = [], [], []
        for l in range(4):
            st = ops.bn_stats(Z, 1, B, hw, 1e-5, self.running_mean[l], self.running_var[l])
            self.num_batches_tracked[l] += 1
            A = ops.bn_apply(Z, 1, B, hw, st, self.w16(f"layers.{BNS[l]}.weight"), self.w16(f"layers.{BNS[l]}.bias"))
            zs.append(Z); As.append(A); stats.append(st)
            w = self.w16(f"layers.{LIN[l + 1]}.weight")
            bias = self.w16(f"layers.{LIN[l + 1]}.bias")
            if l == 3:                                      # 64 -> 4: pad the 4 output channels to 8
                w = torch.nn.functional.pad(w, (0, 0, 0, OUT_PAD - self.out_dim))
                bias = torch.nn.functional.pad(bias, (0, OUT_PAD - self.out_dim))
            Z = ops.gemm(A, w.contiguous(), bias=bias.contiguous(), relu=(l < 3))
        out = Z
        # ---- loss and seed
        dOut, parts = ops.lgp_mse_train(out, target, B, h, SEED_LD, LOSS_SCALE)
        loss = parts.sum()
        g = torch.zeros(self.n, device=dev, dtype=torch.float32)
        gv = lambda k: self.grad_view(g, k)

        def dweight(dPre: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
            """dW [N_out, K_in] = dPre^T . X, fp32 (both operands transposed so the contraction runs over rows)."""
            a, b = ops.transpose(dPre), ops.transpose(X)
            k = a.shape[1]
            if k % 32:                                    # tiny taps (2x2 pixels): zero-pad the contraction axis
                pad = (0, 32 - k % 32)
                a, b = torch.nn.functional.pad(a, pad).contiguous(), torch.nn.functional.pad(b, pad).contiguous()
            return ops.gemm(a, b, out_f32=True)

        # ---- backward.  dPre_l = gradient w.r.t. the pre-activation of Linear l (its ReLU mask applied)
        d4 = dOut[:, :OUT_PAD]                                                   # [M, 8], columns >= out_dim are zero
        gv("layers.12.weight").copy_(dweight(d4, As[3])[:self.out_dim])
        gv("layers.12.bias").copy_(ops.colsum(dOut[:, :OUT_PAD])[:self.out_dim])
        w4t = torch.nn.functional.pad(self.w16("layers.12.weight").t(), (0, SEED_LD - self.out_dim)).contiguous()
        dA = ops.gemm(dOut, w4t)                                                 # [M, 64]
        dZ = None
        for l in (3, 2, 1, 0):
            dg, db = ops.bn_param_grads(zs[l], dA, stats[l])
            gv(f"layers.{BNS[l]}.weight").copy_(dg)
            gv(f"layers.{BNS[l]}.bias").copy_(db)
            dZ = ops.bn_relu_bwd(zs[l], dA, 1, B, hw, stats[l], self.w16(f"layers.{BNS[l]}.weight"), True)
            gv(f"layers.{LIN[l]}.bias").copy_(ops.colsum(dZ))
            if l > 0:
                gv(f"layers.{LIN[l]}.weight").copy_(dweight(dZ, As[l - 1]))
                dA = ops.gemm(dZ, self.w16(f"layers.{LIN[l]}.weight").t().contiguous())
        # ---- layer 0: per-tap adjoint of the bilinear gather, then GEMMs at the taps' native resolutions
        gW0 = gv("layers.0.weight")
        off = 0
        for (F, s), C in zip(taps, self.tap_channels):
            dP = ops.lgp_layer0_scatter(dZ, B, h, s, self.H0)                    # [B*s*s, H0]
            gW0[:, off:off + C].copy_(dweight(dP, F))
            off += C
        gW0[:, self.E:].copy_(dweight(dZ, Ex)[:, :40])
        return loss, g

    # ------------------------------------------------------------------------------------------ collective
    def all_reduce(self, g: torch.Tensor, bucket_bytes: int = 15 << 20) -> torch.Tensor:
        """Average the flat gradient over the ranks: sketch2img_amd.dist.allreduce_mean_ (15 MB buckets like the
        reference's DDP).  No-op on one rank."""
        from .dist import allreduce_mean_
        return allreduce_mean_(g, bucket_bytes)

    # ------------------------------------------------------------------------------------------ optimizer
    def current_lr(self) -> float:
        """diffusers "constant_with_warmup": lr * min(1, step / warmup) (trainer.py:133-138)."""
        if self.warmup <= 0:
            return self.lr
        return self.lr * min(1.0, float(self.step_count) / float(max(1, self.warmup)))

    @torch.no_grad()
    def step(self, g: torch.Tensor):
        lr = self.current_lr()
        self.step_count += 1
        ops.adamw_step(self.p, g, self.m, self.v, self.p16, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                       self.step_count, 1.0 / LOSS_SCALE)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """The reference's 30-key checkpoint (fp32 master weights + running statistics)."""
        sd: Dict[str, torch.Tensor] = {}
        for i in (0, 2, 3, 5, 6, 8, 9, 11, 12):
            for n in ("weight", "bias"):
                o, shp = self.layout[f"layers.{i}.{n}"]
                sd[f"layers.{i}.{n}"] = self.p[o:o + shp.numel()].view(shp).clone()
            if i in BNS:
                l = BNS.index(i)
                sd[f"layers.{i}.running_mean"] = self.running_mean[l].clone()
                sd[f"layers.{i}.running_var"] = self.running_var[l].clone()
                sd[f"layers.{i}.num_batches_tracked"] = torch.tensor(self.num_batches_tracked[l])
        return sd


# ------------------------------------------------------------------------------------------------ the full step
def add_noise(latents: torch.Tensor, noise: torch.Tensor, timesteps: Sequence[int], alphas_cumprod: torch.Tensor):
    """DDPMScheduler.add_noise (trainer.py:215) and get_noise_level (trainer.py:196-204), host-side fp32 scalars per
    sample: returns (noisy latents, noise_level), both [B,4,h,h] fp32."""
    acp = alphas_cumprod.to(torch.float32)
    a = torch.stack([acp[int(t)] ** 0.5 for t in timesteps]).view(-1, 1, 1, 1).to(latents.device)
    s = torch.stack([(1 - acp[int(t)]) ** 0.5 for t in timesteps]).view(-1, 1, 1, 1).to(latents.device)
    return a * latents + s * noise, s * noise


@torch.no_grad()
def unet_taps(net, noisy_latents: torch.Tensor, timesteps: Sequence[int], ehs: torch.Tensor):
    """The nine hooked feature maps (modules/latent_predictor.py:47-81) of a frozen UNet forward, one sample at a
    time because every sample has its own timestep (trainer.py:212) and the time-embedding bias is folded into the
    conv epilogues per launch.  Returns [(fp16 [B*s*s, C], s)] x 9."""
    from .unet import CIN_PAD
    B, _, h, _ = noisy_latents.shape
    per_tap: List[List[torch.Tensor]] = [[] for _ in range(9)]
    sizes: List[int] = []
    for b in range(B):
        net.prepare_context(ehs[b:b + 1])
        x32 = ops.nchw_to_nhwc(noisy_latents[b:b + 1].to(net.dev, torch.float32).contiguous(), CIN_PAD)
        _, taps = net.forward(x32, int(timesteps[b]), 1, h, None, want_taps=True, want_eps=False)
        sizes = [s for _, s in taps]
        for i, (t, _) in enumerate(taps):
            per_tap[i].append(t)
    return [(torch.cat(ts).contiguous(), s) for ts, s in zip(per_tap, sizes)]


@torch.no_grad()
def train_step(trainer: HipLGPTrainer, net, latents: torch.Tensor, sketch_latents: torch.Tensor, ehs: torch.Tensor,
               timesteps: Sequence[int], noise: torch.Tensor, alphas_cumprod: torch.Tensor):
    """trainer.py:208-246 for one batch: noise the latents, frozen UNet forward, LGP loss + gradients, gradient
    all-reduce across ranks, AdamW.  Returns the loss (0-dim tensor)."""
    noisy, noise_level = add_noise(latents.to(trainer.dev, torch.float32), noise.to(trainer.dev, torch.float32),
                                   timesteps, alphas_cumprod)
    taps = unet_taps(net, noisy, timesteps, ehs)
    loss, g = trainer.loss_and_grads(taps, noise_level, sketch_latents)
    trainer.all_reduce(g)
    trainer.step(g)
    return loss
