#!/usr/bin/env python3
"""Golden vectors (checked by tests/test_oracle.py and tests/test_gpu_api.py) from transformers' own CLIPTextModel - the
``text_encoder`` that StableDiffusionPipeline._encode_prompt runs for the reference (modules/pipeline.py:55-57).  Runs
in the build container only (needs `transformers`); writes tests/golden/clip_text_tiny.npz = {weights (transformers key
names), input_ids, last_hidden_state for hidden_act quick_gelu (SD 1.x) and gelu (SD 2.x)}."""
import dataclasses
import os
import sys

import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sketch2img_amd import synthetic  # noqa: E402  (seeded weight recipe only; no arithmetic of ours enters the vector)
from sketch2img_amd.config import TINY_TEXT  # noqa: E402

torch.set_num_threads(1)
cfg = TINY_TEXT
W = synthetic.clip_text_state_dict(cfg)         # seeded values, loaded INTO the transformers model
g = torch.Generator().manual_seed(7)
ids = torch.randint(0, cfg.vocab_size - 2, (3, cfg.max_position_embeddings), generator=g)
ids[:, 0] = cfg.vocab_size - 2                  # bos
for b, n in enumerate((5, 40, 76)):             # eos, then padding with the eos id (SD's tokenizer pads with eos)
    ids[b, n:] = cfg.vocab_size - 1
out = {}
for act in ("quick_gelu", "gelu"):
    hf = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        max_position_embeddings=cfg.max_position_embeddings, layer_norm_eps=cfg.layer_norm_eps,
                        hidden_act=act, attention_dropout=0.0, bos_token_id=cfg.vocab_size - 2,
                        eos_token_id=cfg.vocab_size - 1, pad_token_id=cfg.vocab_size - 1)
    model = CLIPTextModel(hf).eval()
    sd = model.state_dict()
    prefix = "text_model." if any(k.startswith("text_model.") for k in sd) else ""
    missing = model.load_state_dict({prefix + k: v for k, v in W.items()}, strict=False)
    assert not missing.unexpected_keys, missing
    assert all("position_ids" in k for k in missing.missing_keys), missing
    with torch.no_grad():
        out[act] = model(ids)[0]
    print("transformers", getattr(__import__("transformers"), "__version__"), act, tuple(out[act].shape))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_text_tiny.npz"), input_ids=ids.numpy(),
                    last_hidden_state_quick_gelu=out["quick_gelu"].numpy(), last_hidden_state_gelu=out["gelu"].numpy(),
                    **{"w." + k: v.numpy() for k, v in W.items()})
