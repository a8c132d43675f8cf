"""Alias of sketch2img_amd.modules.sketch_encoder (the reference imports `modules.sketch_encoder`)."""
from sketch2img_amd.modules.sketch_encoder import *  # noqa: F401,F403
from sketch2img_amd.modules.sketch_encoder import SketchEncoder  # noqa: F401
