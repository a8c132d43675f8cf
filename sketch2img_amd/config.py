"""Architecture description of the SD-style UNet the sampler drives (host-side data only).

Field meanings follow the diffusers ``UNet2DConditionModel`` config the reference loads through
``AntiGradientPipeline.from_pretrained`` (app.py:32-37); SD1.5 / SD2.1 values are the public model
configs (validated by exact parameter counts in tests/test_oracle.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)      # SD1.5: 8 heads per block; SD2.1: (5, 10, 20, 20)
    use_linear_projection: bool = False
    norm_groups: int = 32
    sample_size: int = 64

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


SD15 = UNetConfig()
SD21 = UNetConfig(cross_attention_dim=1024, num_heads=(5, 10, 20, 20), use_linear_projection=True,
                  sample_size=96)
TINY = UNetConfig(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64, num_heads=(2, 2, 4, 4),
                  norm_groups=8, sample_size=32)


def up_block_plan(cfg: UNetConfig) -> List[List[Tuple[int, int, int]]]:
    """Per up block, per resnet: (channels of h, channels of the skip, output channels)."""
    rev = tuple(reversed(cfg.block_out_channels))
    nb = len(rev)
    plan = []
    for i in range(nb):
        out_c = rev[i]
        prev = rev[i - 1] if i > 0 else rev[0]
        inp = rev[min(i + 1, nb - 1)]
        plan.append([(prev if j == 0 else out_c, inp if j == cfg.layers_per_block else out_c, out_c)
                     for j in range(cfg.layers_per_block + 1)])
    return plan


def tap_channels(cfg: UNetConfig) -> List[int]:
    b = cfg.block_out_channels
    rev = tuple(reversed(b))
    return [b[0], b[1], b[2], b[-1], b[-1], b[-1], rev[0], rev[1], rev[2]]


def tap_sizes(h: int) -> List[int]:
    return [h // 2, h // 4, h // 8, h // 8, h // 8, h // 8, h // 4, h // 2, h]


# ---------------------------------------------------------------------------------------- VAE decoder
@dataclass(frozen=True)
class VAEConfig:
    """AutoencoderKL decoder of Stable Diffusion (app.py:28-30 loads runwayml/stable-diffusion-v1-5 subfolder vae)."""
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215


SD_VAE = VAEConfig()
TINY_VAE = VAEConfig(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_groups=8)


def vae_up_plan(cfg: VAEConfig):
    """[(up block index, [(cin, cout) per resnet], has_upsampler)] in execution order (channels reversed,
    layers_per_block + 1 resnets per block, nearest-2x + conv after every block but the last)."""
    rev = list(reversed(cfg.block_out_channels))
    plan, prev = [], rev[0]
    for i, co in enumerate(rev):
        plan.append((i, [(prev if j == 0 else co, co) for j in range(cfg.layers_per_block + 1)], i != len(rev) - 1))
        prev = co
    return plan


# ---------------------------------------------------------------------------------------- CLIP vision tower
@dataclass(frozen=True)
class CLIPVisionConfig:
    """transformers CLIPVisionModel (modules/clip_guided_inf.py:49-54 loads openai/clip-vit-large-patch14)."""
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5

    @property
    def num_tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


VIT_L_14 = CLIPVisionConfig()
TINY_CLIP = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                             image_size=56, patch_size=14)


# ------------------------------------------------------------------------------------------- CLIP text encoder
@dataclass(frozen=True)
class CLIPTextConfig:
    """transformers CLIPTextModel, the ``text_encoder`` StableDiffusionPipeline._encode_prompt runs
    (modules/pipeline.py:55-57).  SD 1.x: CLIP ViT-L/14 text tower; SD 2.x: OpenCLIP ViT-H, 23 of 24 layers, gelu."""
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5


SD15_TEXT = CLIPTextConfig()
SD21_TEXT = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                           hidden_act="gelu")
TINY_TEXT = CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                           num_attention_heads=4)
