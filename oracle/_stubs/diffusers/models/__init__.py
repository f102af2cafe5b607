class AutoencoderKL:  # name only (stable_diffusion.py:19)
    pass


class AutoencoderKLTemporalDecoder:  # name only (run_*_sd.py)
    pass
