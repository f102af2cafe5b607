"""DDIMScheduler restated from diffusers 0.35.1 for the SD-v1.5 scheduler_config.json:
beta_schedule=scaled_linear, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000,
clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type=epsilon,
timestep_spacing=leading.  eta=0 only."""
from dataclasses import dataclass
import numpy as np
import torch
from ..configuration_utils import FrozenDict


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, clip_sample=False):
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                         beta_end=beta_end, steps_offset=steps_offset,
                                         set_alpha_to_one=set_alpha_to_one, clip_sample=clip_sample,
                                         beta_schedule="scaled_linear", prediction_type="epsilon",
                                         timestep_spacing="leading")
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        assert eta == 0.0
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        pred_original_sample = (sample - beta_prod_t ** (0.5) * model_output) / alpha_prod_t ** (0.5)
        pred_epsilon = model_output
        std_dev_t = 0.0
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** (0.5) * pred_epsilon
        prev_sample = alpha_prod_t_prev ** (0.5) * pred_original_sample + pred_sample_direction
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)


class DDPMScheduler(DDIMScheduler):
    pass


class DPMSolverMultistepScheduler: pass
class EulerAncestralDiscreteScheduler: pass
class EulerDiscreteScheduler: pass
class LMSDiscreteScheduler: pass
class PNDMScheduler: pass


class FlowMatchEulerDiscreteScheduler:
    """diffusers 0.35.1 FlowMatchEulerDiscreteScheduler restated for the SD3 / SD3.5 scheduler_config.json (shift 3.0,
    use_dynamic_shifting False): the table construction and the Euler `step`.  Third-party stand-in, parity unpinned."""
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=3.0):
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=False)
        self.shift = shift
        s = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy() / num_train_timesteps
        s = shift * s / (1 + (shift - 1) * s)
        self.sigma_min, self.sigma_max = float(s[-1]), float(s[0])
        self.sigmas = torch.from_numpy(s)
        self.timesteps = self.sigmas * num_train_timesteps
        self._step_index = None

    @property
    def config(self):
        return self._internal_dict

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        T = self.config.num_train_timesteps
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max * T, self.sigma_min * T, num_inference_steps) / T
        sigmas = np.asarray(sigmas, dtype=np.float32)
        sigmas = (self.shift * sigmas / (1 + (self.shift - 1) * sigmas)).astype(np.float32)
        sig = torch.from_numpy(sigmas)
        self.timesteps = sig * T
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._step_index = None

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        if self._step_index is None:
            self._step_index = int((self.timesteps == timestep).nonzero()[0])
        sample = sample.to(torch.float32)
        prev = sample + (self.sigmas[self._step_index + 1] - self.sigmas[self._step_index]) * model_output
        self._step_index += 1
        return (prev.to(model_output.dtype),)
