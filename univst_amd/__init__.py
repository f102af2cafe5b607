"""univst_amd — MI355X (gfx950) native implementation of the UniVST SD-v1.5 denoising hot path.

Python here is host orchestration only: it mirrors the reference's plugin surface
(`backbones/video_diffusion_sd/{models,pipelines,pnp_utils}`, `inversion_tools.ddim_inversion`,
`src.{util,mask_propagation,cal_optica_flow}`) and forwards every hot operation to hand-written HIP
kernels in `lib/libunivst_hip.so` through the C ABI declared in `include/univst.h`.
There is no CPU / eager fallback: without the shared library (and a GPU) the hot methods raise.
"""
__version__ = "0.1.0"
