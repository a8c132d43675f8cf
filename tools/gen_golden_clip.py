#!/usr/bin/env python3
"""Golden vector (checked by tests/test_oracle.py and tests/test_gpu_api.py) from transformers' own CLIPVisionModel (the class the reference calls at
modules/clip_guided_inf.py:49-54,103).  Runs in the build container only (needs `transformers`); writes
tests/golden/clip_vision_tiny.npz = {weights (transformers key names), pixel_values, last_hidden_state}."""
import os
import sys

import numpy as np
import torch
from transformers import CLIPVisionConfig, CLIPVisionModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sketch2img_amd import synthetic  # noqa: E402  (seeded weight recipe only; no arithmetic of ours enters the vector)
from sketch2img_amd.config import TINY_CLIP  # noqa: E402

torch.set_num_threads(1)
cfg = TINY_CLIP
hf = CLIPVisionConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      image_size=cfg.image_size, patch_size=cfg.patch_size, layer_norm_eps=cfg.layer_norm_eps,
                      hidden_act="quick_gelu", attention_dropout=0.0)
model = CLIPVisionModel(hf).eval()
W = synthetic.clip_vision_state_dict(cfg)       # seeded values, loaded INTO the transformers model
sd = model.state_dict()
prefix = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""
missing = model.load_state_dict({prefix + k: v for k, v in W.items()}, strict=False)
assert not missing.unexpected_keys, missing
assert all("position_ids" in k for k in missing.missing_keys), missing
x = torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(5))
with torch.no_grad():
    out = model(x, output_hidden_states=True).last_hidden_state
print("transformers", getattr(__import__("transformers"), "__version__"), "last_hidden_state", tuple(out.shape))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_vision_tiny.npz"), pixel_values=x.numpy(),
                    last_hidden_state=out.numpy(), **{"w." + k: v.numpy() for k, v in W.items()})
