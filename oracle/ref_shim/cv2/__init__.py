"""cv2 stand-in — TEST INFRASTRUCTURE ONLY (OpenCV is not installed here and there is no network).

Lets the reference's `exps/data/data_augment_flip.py` import unmodified so that its own `preproc` / `_mirror` /
`DoubleTrainTransform` can mint golden vectors for the device input pipeline (oracle/make_golden_input.py).
Only `resize` is provided, and only for the two ratios whose result does not depend on OpenCV's fixed-point
interpolation tables:
  * same size  -> copy;
  * exact 2x decimation in both axes -> cv::resize switches INTER_LINEAR to the INTER_AREA fast path when
    iscale_x == iscale_y == 2 (modules/imgproc/src/resize.cpp, `is_area_fast`), whose uint8 kernel is
    (a + b + c + d + 2) >> 2 over each 2x2 block (ResizeAreaFastVec).  Restated from the published source —
    PARITY UNPINNED for this case: no OpenCV binary is available to check it against.
Anything else raises, so a golden vector can never silently contain a guessed interpolation."""
import numpy as np

INTER_LINEAR = 1
COLOR_BGR2HSV = 40
COLOR_HSV2BGR = 54


def resize(img, dsize, interpolation=INTER_LINEAR):
    w, h = int(dsize[0]), int(dsize[1])
    H, W = img.shape[:2]
    if (h, w) == (H, W):
        return img.copy()
    if interpolation == INTER_LINEAR and img.dtype == np.uint8 and H == 2 * h and W == 2 * w:
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    raise NotImplementedError("cv2 stand-in: resize %dx%d -> %dx%d needs OpenCV's interpolation tables" % (H, W, h, w))
