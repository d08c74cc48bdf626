"""cv2.resize(image, (w, h)) for uint8 images, restated ([3P] OpenCV, absent from this image).

The reference preprocesses with `cv2.resize(image, (size, size))` (modules/models/__init__.py:64), i.e. INTER_LINEAR on uint8 data:
bilinear taps at pixel centres, NO antialiasing when down-scaling (PIL's BILINEAR does antialias, which is why it is not a stand-in),
11-bit fixed-point coefficients.  Published algorithm (opencv/modules/imgproc/src/resize.cpp, portable path):
  * source position of destination pixel d:  f = (d + 0.5) * (src / dst) - 0.5 in float32, s = floor(f), f -= s;
    s < 0 -> (s, f) = (0, 0);  s >= src - 1 -> (s, f) = (src - 1, 0)  (both taps on the edge pixel);
  * coefficients a0 = rint((1 - f) * 2048), a1 = rint(f * 2048) as int16;
  * horizontal pass (int32):   H = S[s] * a0 + S[s + 1] * a1;
  * vertical pass:             D = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
  * an exact 2 x 2 down-scale is computed as the 2 x 2 box mean (sum + 2) >> 2 (cv::resize switches INTER_LINEAR to its area path there).
Builds of OpenCV that dispatch to a vendor primitive library may differ from the portable path by one grey level on some pixels."""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _taps(src: int, dst: int):
    scale = np.float64(src) / np.float64(dst)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= src - 1
    s[hi], f[hi] = src - 1, 0.0
    a0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int32)
    a1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    return s, np.minimum(s + 1, src - 1), a0, a1


def resize_linear_u8(image: np.ndarray, size) -> np.ndarray:
    """image (H, W) or (H, W, C) uint8 -> (size[1], size[0][, C]) uint8; size = (width, height) like cv2.resize."""
    assert image.dtype == np.uint8 and image.ndim in (2, 3)
    dw, dh = int(size[0]), int(size[1])
    sh, sw = image.shape[:2]
    if (sh, sw) == (dh, dw):
        return image.copy()
    img = image.astype(np.int32)
    if sw == 2 * dw and sh == 2 * dh:                                   # exact 2 x 2: box mean
        return ((img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, ax0, ax1 = _taps(sw, dw)
    y0, y1, by0, by1 = _taps(sh, dh)
    shape = (1, dw) + (1,) * (image.ndim - 2)
    H = img[:, x0] * ax0.reshape(shape) + img[:, x1] * ax1.reshape(shape)            # (sh, dw[, C]), scale 2^11
    shape = (dh, 1) + (1,) * (image.ndim - 2)
    out = (((by0.reshape(shape) * (H[y0] >> 4)) >> 16) + ((by1.reshape(shape) * (H[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
