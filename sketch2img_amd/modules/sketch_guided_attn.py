"""Drop-in for the reference's modules/sketch_guided_attn.py: SatMixin over UNet residual samples.

``set_res_samples(res_samples)`` (:29-40) takes the per-down-block tuples of (rows, C, h, w) feature maps (what
modules/sketch_encoder.py returns) and routes them to the 16 blocks exactly as the reference does.
"""
from ._sat_common import SatMixinBase


class SatMixin(SatMixinBase):
    variant = "sketch"

    def set_res_samples(self, res_samples):
        self._injector().set_res_samples(res_samples)
