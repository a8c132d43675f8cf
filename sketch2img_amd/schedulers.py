"""Configuration holders with the constructor signatures the reference uses for its schedulers
(app.py:13-25, evaluation.py:21-32, modules/clip_guided_inf.py:15-26).  diffusers is not a dependency of this
package: these classes carry the configuration only; the arithmetic lives in sketch2img_amd.sampler
(host-side tables) and libskg.so (the latent update kernels).  A real diffusers scheduler object passed as
`scheduler=` works too - the pipeline reads its `.config`.
"""
from __future__ import annotations

from types import SimpleNamespace


class _ConfigScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, **kwargs):
        self.config = SimpleNamespace(**kwargs)

    def scale_model_input(self, sample, timestep=None):       # identity for both (modules/pipeline.py:86)
        return sample


class DDIMScheduler(_ConfigScheduler):
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=1, **kwargs):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError("scaled_linear betas (Stable Diffusion) only")
        if clip_sample:
            raise NotImplementedError("clip_sample=False (Stable Diffusion) only")
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                         beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                         steps_offset=steps_offset, **kwargs)


class DPMSolverMultistepScheduler(_ConfigScheduler):
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 trained_betas=None, solver_order=2, predict_epsilon=True, thresholding=False,
                 algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True, **kwargs):
        if beta_schedule != "scaled_linear" or trained_betas is not None:
            raise NotImplementedError("scaled_linear betas (Stable Diffusion) only")
        super().__init__(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                         beta_schedule=beta_schedule, trained_betas=trained_betas, solver_order=solver_order,
                         predict_epsilon=predict_epsilon, thresholding=thresholding, algorithm_type=algorithm_type,
                         solver_type=solver_type, lower_order_final=lower_order_final, **kwargs)
