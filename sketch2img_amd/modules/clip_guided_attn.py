"""Drop-in for the reference's modules/clip_guided_attn.py: SatMixin over CLIP image tokens.

``SatMixin(unet)`` creates one parameter block per BasicTransformerBlock with the reference's module names, so
``load_state_dict(torch.load("sketch_attn_model.pt"))`` (modules/clip_guided_inf.py:46-47) works unchanged;
``set_state`` (:29-31) hands the (rows, 257, 1024) token tensor - [zeros; clip hidden state],
modules/clip_guided_inf.py:107 - to the HIP injector, ``set_scale`` (:33-35) sets the strength.
"""
from ._sat_common import SatMixinBase


class SatMixin(SatMixinBase):
    variant = "clip"

    def set_state(self, hidden_state):
        self._injector().set_state(hidden_state)
