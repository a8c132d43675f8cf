"""Alias of sketch2img_amd.modules.latent_predictor (the reference imports `modules.latent_predictor`)."""
from sketch2img_amd.modules.latent_predictor import *  # noqa: F401,F403
from sketch2img_amd.modules import latent_predictor as _m
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
