"""Alias of sketch2img_amd.modules.pipeline (the reference imports `modules.pipeline`)."""
from sketch2img_amd.modules.pipeline import *  # noqa: F401,F403
from sketch2img_amd.modules import pipeline as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
