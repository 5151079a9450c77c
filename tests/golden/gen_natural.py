"""Natural-content fixture for the image-level parity tests (Next 2b).

Decodes the reference's own screenshot (/root/reference/screenshot.png, 1850x1060 RGB8: GUI text, a sky gradient, foliage,
brick and metal texture — content the synthetic generator of fidelityfx-fsr_amd/frames.py does not have), crops the top-left
1477 x 831 pixels (the "Ultra Quality" render size for a 1080p target, sample/src/DX12/FSRSample.h:79-95) and stores the 8-bit
codes as tests/golden/natural_1477x831_rgb8.npz.  Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/gen_natural.py

The fixture is DATA (pixel codes of an image the reference ships as documentation); tests widen it with `natural_frame()` of
oracle/image_parity.py: code / 255 rounded to binary16, alpha = 1.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 1477, 831

if __name__ == "__main__":
    im = np.asarray(Image.open("/root/reference/screenshot.png").convert("RGB"))
    crop = np.ascontiguousarray(im[:H, :W, :])
    assert crop.shape == (H, W, 3) and crop.dtype == np.uint8
    out = os.path.join(HERE, "natural_%dx%d_rgb8.npz" % (W, H))
    np.savez_compressed(out, rgb8=crop)
    print(out, os.path.getsize(out), "bytes; mean code %.2f" % crop.mean())
