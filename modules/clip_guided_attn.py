"""Alias of sketch2img_amd.modules.clip_guided_attn (the reference imports `modules.clip_guided_attn`)."""
from sketch2img_amd.modules.clip_guided_attn import *  # noqa: F401,F403
from sketch2img_amd.modules import clip_guided_attn as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
