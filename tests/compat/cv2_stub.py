"""A `cv2` provider for the reference's tools, sufficient for their inference path (SURVEY.md section 0 lists the
surface: tools/test.py:105,270-273,285-294,325-326,392,494,525; tools/demo.py:39-42,60-62).

  * resize / warpAffine   -> oracle/cv_ops.py (restatement of OpenCV 3.4's INTER_LINEAR paths; OpenCV rounding
                             "parity unpinned", DESIGN.md 5.5 -- cv2 itself is not installable here)
  * findContours / contourArea / minAreaRect / boxPoints / boundingRect -> small scipy/numpy implementations that are
    adequate for "largest blob -> rotated rectangle" (tools/test.py:285-294); NOT OpenCV-exact and not on the path
    under test (they only shape state['ploygon'])
  * GUI calls are no-ops; selectROI returns `SELECT_ROI` (set by the harness)."""
import time

import numpy as np

from oracle import cv_ops

__version__ = "3.4.3"          # requirements.txt pins opencv-python==3.4.3.18; test.py:285 reads __version__[-5]
INTER_LINEAR, BORDER_CONSTANT = 1, 0
RETR_EXTERNAL, CHAIN_APPROX_NONE = 0, 1
WND_PROP_FULLSCREEN, WINDOW_FULLSCREEN = 0, 1
FONT_HERSHEY_SIMPLEX = 0
SELECT_ROI = (300, 110, 165, 250)   # the box demo.py's user would draw on data/tennis/00000.jpg (harness may override)
CALLS = {}                          # name -> count, for the tests


def _count(name):
    CALLS[name] = CALLS.get(name, 0) + 1


def imread(path, flags=None):
    from PIL import Image
    _count("imread")
    rgb = np.asarray(Image.open(path).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])          # BGR like OpenCV


def imwrite(path, img):
    _count("imwrite")
    return True


def resize(src, dsize, *a, **k):
    _count("resize")
    return cv_ops.cv_resize_linear_u8(np.ascontiguousarray(src), (int(dsize[0]), int(dsize[1])))


def warpAffine(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0):
    _count("warpAffine")
    assert flags == INTER_LINEAR and borderMode == BORDER_CONSTANT
    return cv_ops.cv_warp_affine_linear_f32(np.ascontiguousarray(src, dtype=np.float32), np.asarray(M, dtype=np.float64),
                                            (int(dsize[0]), int(dsize[1])), float(borderValue))


def findContours(mask, mode, method):
    """external boundaries of the 8-connected blobs, as [(N,1,2) int32 (x, y)]; OpenCV-3 style 3-tuple"""
    from scipy import ndimage
    _count("findContours")
    m = np.asarray(mask) > 0
    lab, n = ndimage.label(m, structure=np.ones((3, 3), int))
    contours = []
    er = ndimage.binary_erosion(m, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool), border_value=0)
    edge = m & ~er
    for i in range(1, n + 1):
        ys, xs = np.nonzero(edge & (lab == i))
        contours.append(np.stack([xs, ys], 1).astype(np.int32).reshape(-1, 1, 2))
    return mask, contours, None


def _hull(pts):
    pts = np.unique(np.asarray(pts, dtype=np.float64).reshape(-1, 2), axis=0)
    if len(pts) < 3:
        return pts
    from scipy.spatial import ConvexHull, QhullError
    try:
        return pts[ConvexHull(pts).vertices]
    except QhullError:
        return pts


def contourArea(cnt):
    """area of the blob's convex hull (shoelace) -- an upper bound of OpenCV's polygon area; used only for
    "largest blob" / "> 100" decisions (tools/test.py:289-291)"""
    h = _hull(cnt)
    if len(h) < 3:
        return 0.0
    x, y = h[:, 0], h[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def minAreaRect(points):
    """rotating calipers over the convex hull -> ((cx, cy), (w, h), angle_degrees)"""
    h = _hull(points)
    if len(h) == 0:
        return (0.0, 0.0), (0.0, 0.0), 0.0
    if len(h) < 3:
        c = h.mean(axis=0)
        d = h[-1] - h[0]
        return (float(c[0]), float(c[1])), (float(np.hypot(*d)), 0.0), float(np.degrees(np.arctan2(d[1], d[0])))
    best = None
    for i in range(len(h)):
        e = h[(i + 1) % len(h)] - h[i]
        n = np.hypot(*e)
        if n == 0:
            continue
        u = e / n
        v = np.array([-u[1], u[0]])
        pu, pv = h @ u, h @ v
        w, hh = pu.max() - pu.min(), pv.max() - pv.min()
        if best is None or w * hh < best[0]:
            c = u * (pu.max() + pu.min()) / 2 + v * (pv.max() + pv.min()) / 2
            best = (w * hh, c, w, hh, np.degrees(np.arctan2(u[1], u[0])))
    _, c, w, hh, ang = best
    return (float(c[0]), float(c[1])), (float(w), float(hh)), float(ang)


def boxPoints(rect):
    (cx, cy), (w, h), ang = rect
    a = np.radians(ang)
    u = np.array([np.cos(a), np.sin(a)]) * w / 2
    v = np.array([-np.sin(a), np.cos(a)]) * h / 2
    c = np.array([cx, cy])
    return np.stack([c - u + v, c - u - v, c + u - v, c + u + v]).astype(np.float32)


def boundingRect(pts):
    p = np.asarray(pts).reshape(-1, 2)
    x0, y0 = p.min(axis=0)
    x1, y1 = p.max(axis=0)
    return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)


def getTickCount():
    return time.perf_counter_ns()


def getTickFrequency():
    return 1e9


def selectROI(*a, **k):
    _count("selectROI")
    return SELECT_ROI


def _noop(*a, **k):
    return None


def waitKey(*a, **k):
    _count("waitKey")
    return 0


def imshow(*a, **k):
    _count("imshow")


namedWindow = setWindowProperty = destroyAllWindows = polylines = rectangle = putText = addWeighted = _noop
