from univst_amd.src.cal_optica_flow import *  # noqa: F401,F403
