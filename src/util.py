from univst_amd.src.util import *  # noqa: F401,F403
