#!/usr/bin/env python3
"""Time the HIP VAE decoder (SD config) on S latents of 64x64 -> 512x512 images; 2.51 TFLOP per image."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd.config import SD_VAE  # noqa: E402
from sketch2img_amd.vae import AutoencoderKL  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vae = AutoencoderKL(SD_VAE).to("cuda")
vae.hip.chunk = chunk
lat = 0.18215 * torch.randn(S, 4, 64, 64, generator=torch.Generator().manual_seed(0)) * 4
img = vae.decode_latents(lat)
torch.cuda.synchronize()
assert img.shape == (S, 512, 512, 3) and torch.isfinite(img).all()
t = time.time()
for _ in range(3):
    img = vae.decode_latents(lat)
torch.cuda.synchronize()
dt = (time.time() - t) / 3
print(f"{S} images, chunk {chunk}: {dt * 1e3:.1f} ms = {dt / S * 1e3:.1f} ms/image = {2.5145 * S / dt:.0f} TFLOP/s; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB; image mean {float(img.mean()):.3f}")
