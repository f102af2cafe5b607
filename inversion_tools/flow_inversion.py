from univst_amd.inversion_tools.flow_inversion import *  # noqa: F401,F403
from univst_amd.inversion_tools.flow_inversion import rf_inversion, rf_solver, content_inversion_reconstruction, style_inversion_reconstruction  # noqa: F401
