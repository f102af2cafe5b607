"""The boundary the reference leaves to OpenCV — ``cv2.remap(..., INTER_LINEAR, BORDER_CONSTANT)`` on uint8 images
(cal_optica_flow.py:31-41) — pinned by VECTORS: tests/golden/remap_opencv_cases.json holds sample positions on a 3x3 image with
the value OpenCV 4.9's fixed-point bilinear path produces, each derived by hand from imgwarp.cpp (INTER_BITS = 5 sub-pixel grid
with round-half-to-even, 15-bit weights, + 2^14 rounding, zero border taps, short saturation of the integer part).  OpenCV is on
neither box, so this is what "cv2-exact" means for the oracle here and, in tests/test_gpu_unet.py, for the warp kernel."""
import json
import os

import numpy as np

from oracle import flow_ref

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    d = json.load(open(os.path.join(HERE, "golden", "remap_opencv_cases.json")))
    return [(np.array(d["image"], np.uint8), d["cases"]), (np.array(d["image_b"], np.uint8), d["cases_b"])]


def test_oracle_remap_matches_hand_computed_opencv_vectors():
    n = 0
    for img, cases in load_cases():
        src = np.repeat(img[:, :, None], 3, axis=2)
        mx = np.array([[c["x"] for c in cases]], np.float32)
        my = np.array([[c["y"] for c in cases]], np.float32)
        got = flow_ref.remap_bilinear_u8(src, mx, my)[0]
        for c, g in zip(cases, got):
            assert (g == c["want"]).all(), (c, g.tolist())
            n += 1
    assert n >= 20


def test_weights_sum_to_two_to_the_fifteen():
    """the property the fixed-point table relies on: for every (fx, fy) of the 32 x 32 sub-pixel grid the four integer weights
    sum to 2^15 exactly, so a constant image is reproduced exactly wherever all four taps are inside."""
    fx, fy = np.meshgrid(np.arange(32), np.arange(32))
    s = ((32 - fx) * (32 - fy) + fx * (32 - fy) + (32 - fx) * fy + fx * fy) * 32
    assert (s == 1 << 15).all()
    img = np.full((5, 5, 3), 173, np.uint8)
    rs = np.random.RandomState(0)
    mx = rs.uniform(0, 3.99, (7, 9)).astype(np.float32)
    my = rs.uniform(0, 3.99, (7, 9)).astype(np.float32)
    assert (flow_ref.remap_bilinear_u8(img, mx, my) == 173).all()
