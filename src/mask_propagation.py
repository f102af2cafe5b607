from univst_amd.src.mask_propagation import *  # noqa: F401,F403
from univst_amd.src.mask_propagation import build_parser, video_mask_propogation
if __name__ == "__main__":
    video_mask_propogation(build_parser().parse_args())
