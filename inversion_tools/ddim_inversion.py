from univst_amd.inversion_tools.ddim_inversion import *  # noqa: F401,F403
