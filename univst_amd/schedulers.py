"""DDIMScheduler with the diffusers 0.35.1 semantics the UniVST SD path relies on (SD-v1.5
scheduler_config.json: scaled_linear betas, steps_offset=1, leading spacing, set_alpha_to_one=False, eta=0).
Used when ``diffusers`` is not installed; the pipeline is duck-typed and accepts the real diffusers object too
(it only reads ``timesteps``, ``alphas_cumprod``, ``final_alpha_cumprod``, ``num_inference_steps`` and
``config.num_train_timesteps``).  The arithmetic of ``step`` itself runs in the axpby HIP kernel
(univst_amd.engine.ddim_step); this class only owns the schedule tables."""
import json
import os

import numpy as np
import torch


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, set_alpha_to_one=False, clip_sample=False, prediction_type="epsilon",
                 timestep_spacing="leading", **unused):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon" or timestep_spacing != "leading" or clip_sample:
            raise NotImplementedError("only the SD-v1.5 DDIM configuration is implemented")
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, steps_offset=steps_offset, set_alpha_to_one=set_alpha_to_one,
                           clip_sample=clip_sample, prediction_type=prediction_type, timestep_spacing=timestep_spacing)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_pretrained(cls, path=None, subfolder=None, **kw):
        cfg = {}
        if path is not None:
            p = os.path.join(path, subfolder or "", "scheduler_config.json")
            if os.path.isfile(p):
                with open(p) as f:
                    cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cfg.update(kw)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict: bool = True):
        """DDIMScheduler.step, eta = 0 (API completeness for callers that drive the scheduler themselves; the pipelines and
        engine.ddim_step fold the same update into one axpby kernel launch).  Tensor arithmetic on the inputs' device."""
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        prev = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * model_output
        if not return_dict:
            return (prev,)
        return _Cfg(prev_sample=prev, pred_original_sample=x0)


class FlowMatchEulerDiscreteScheduler:
    """The schedule tables of diffusers 0.35.1 ``FlowMatchEulerDiscreteScheduler`` for the SD3 / SD3.5 configuration (shift 3.0,
    no dynamic shifting) — THIRD-PARTY, restated from the published definition, parity unpinned (oracle/sd3_ref.flow_match_schedule is
    the same reading).  Used when ``diffusers`` is not installed; the SD3 pipeline mirror is duck-typed and takes the real object too
    (it reads ``timesteps``, ``sigmas``, ``config.num_train_timesteps`` and calls ``set_timesteps``).  The Euler update itself is
    folded into one three-term kernel launch by the pipeline (custom_pipeline.py mirror); ``step`` is here for API completeness."""
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=False, **unused):
        if use_dynamic_shifting:
            raise NotImplementedError("use_dynamic_shifting (the SD3 checkpoints ship with it off)")
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=False)
        self.shift = shift
        s = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy() / num_train_timesteps
        s = shift * s / (1 + (shift - 1) * s)
        self.sigma_min, self.sigma_max = float(s[-1]), float(s[0])
        self.timesteps = torch.from_numpy(s * num_train_timesteps)
        self.sigmas = torch.from_numpy(s)
        self.num_inference_steps = None
        self._step_index = None

    from_pretrained = classmethod(DDIMScheduler.from_pretrained.__func__)

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        self.num_inference_steps = num_inference_steps
        if sigmas is None:                   # linspace between the ALREADY shifted end points, then shifted again (as diffusers does)
            T = self.config.num_train_timesteps
            sigmas = np.linspace(self.sigma_max * T, self.sigma_min * T, num_inference_steps) / T
        sigmas = np.asarray(sigmas, dtype=np.float32)
        sigmas = (self.shift * sigmas / (1 + (self.shift - 1) * sigmas)).astype(np.float32)
        sig = torch.from_numpy(sigmas)
        self.timesteps = (sig * self.config.num_train_timesteps).to(device=device)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).to(device=device)
        self._step_index = None

    def step(self, model_output, timestep, sample, return_dict=True, **unused):
        """prev = sample + (sigma_next - sigma) * model_output in fp32, cast back (Euler step of the flow ODE)."""
        if self._step_index is None:
            self._step_index = int((self.timesteps == timestep).nonzero()[0])
        i = self._step_index
        prev = (sample.float() + float(self.sigmas[i + 1] - self.sigmas[i]) * model_output.float()).to(model_output.dtype)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return _Cfg(prev_sample=prev)
