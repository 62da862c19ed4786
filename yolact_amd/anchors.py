"""Prior-box (`priors`) construction for the Detect hot path.

Host-side, vectorised numpy equivalent of the reference's
``Anchor._generate_anchors`` + ``Anchor.get_anchors``
(``data/anchor.py:23-68``) and of the feature-map size arithmetic in
``yolact.py:75-88``.  Output layout is what ``Detect`` consumes:
``[A,4]`` float32 ``[ymin, xmin, ymax, xmax]`` in pixels.
"""
from __future__ import annotations

import math

import numpy as np

DEFAULT_SCALES = (24, 48, 96, 192, 384)      # train.py:65-68
DEFAULT_ASPECT_RATIOS = (1.0, 0.5, 2.0)


def _conv_out(n: int) -> int:
    # 7x7/2 pad 3, 3x3/2 pad 1 and the stride-2 1x1 convs of keras ResNet50 all
    # map n -> floor((n-1)/2)+1 (backbone/resnet.py:173-183,262).
    return (n - 1) // 2 + 1


def feature_map_sizes(img_h: int, img_w: int):
    """P3..P7 sizes as derived in yolact.py:75-86 (C3,C4,C5 then two ceil(/2))."""
    sizes = []
    h, w = _conv_out(_conv_out(img_h)), _conv_out(_conv_out(img_w))   # stem: /4
    for _ in range(3):                                                # C3, C4, C5
        h, w = _conv_out(h), _conv_out(w)
        sizes.append((h, w))
    for _ in range(2):                                                # P6, P7
        h, w = math.ceil(h / 2), math.ceil(w / 2)
        sizes.append((h, w))
    return sizes


def proto_size(img_h: int, img_w: int):
    """Prototype map = 2 x P3 (yolact.py:88)."""
    p3 = feature_map_sizes(img_h, img_w)[0]
    return 2 * p3[0], 2 * p3[1]


def make_priors(img_h: int, img_w: int, feature_maps=None, aspect_ratios=DEFAULT_ASPECT_RATIOS,
                scales=DEFAULT_SCALES) -> np.ndarray:
    """[A,4] float32 yxyx pixel priors, same anchor order as data/anchor.py:33-46
    (level, row j, column i, aspect ratio)."""
    if feature_maps is None:
        feature_maps = feature_map_sizes(img_h, img_w)
    chunks = []
    ars = np.sqrt(np.asarray(aspect_ratios, dtype=np.float64))
    for (fh, fw), s in zip(feature_maps, scales):
        fh, fw = int(fh), int(fw)
        x = (np.arange(fw, dtype=np.float64) + 0.5) / fw * img_w
        y = (np.arange(fh, dtype=np.float64) + 0.5) / fh * img_h
        yy, xx = np.meshgrid(y, x, indexing="ij")                     # [fh,fw]
        n_ar = ars.shape[0]
        cx = np.repeat(xx.reshape(-1), n_ar)
        cy = np.repeat(yy.reshape(-1), n_ar)
        w = np.tile(s * ars, fh * fw)
        h = np.tile(s / ars, fh * fw)
        chunks.append(np.stack([cx, cy, w, h], axis=-1))
    cxcywh = np.concatenate(chunks, axis=0).astype(np.float32)        # f64 -> f32 (anchor.py:52-53)
    two = np.float32(2)
    w, h = cxcywh[:, 2], cxcywh[:, 3]
    return np.stack([cxcywh[:, 1] - h / two, cxcywh[:, 0] - w / two,
                     cxcywh[:, 1] + h / two, cxcywh[:, 0] + w / two], axis=-1).astype(np.float32)


_DEVICE_PRIORS = {}


def make_priors_device(img_h: int, img_w: int, feature_maps=None, aspect_ratios=DEFAULT_ASPECT_RATIOS,
                       scales=DEFAULT_SCALES, device=0):
    """The same [A,4] priors built on the GPU by ``yp_make_priors`` (csrc/priors.hip), cached per
    (shape, scales, ratios, device) like the reference's one-time ``Anchor`` object (yolact.py:98-104).
    Returns a torch CUDA tensor; raises without the library or a HIP device."""
    import ctypes

    import torch

    from . import _lib
    if feature_maps is None:
        feature_maps = feature_map_sizes(img_h, img_w)
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    fm = tuple((int(h), int(w)) for h, w in feature_maps)
    sc = tuple(float(x) for x in scales)[:len(fm)]
    ar = tuple(float(x) for x in aspect_ratios)
    key = (int(img_h), int(img_w), fm, sc, ar, dev.index)
    hit = _DEVICE_PRIORS.get(key)
    if hit is not None:
        return hit
    L = _lib.load()
    c_fm = (ctypes.c_int32 * (2 * len(fm)))(*[v for hw in fm for v in hw])
    c_sc = (ctypes.c_double * len(sc))(*sc)
    c_ar = (ctypes.c_double * len(ar))(*ar)
    n = L.yp_make_priors(img_h, img_w, len(fm), c_fm, c_sc, len(ar), c_ar, None, 0, dev.index or 0, None)
    if n < 0:
        _lib.check(int(n))
    out = torch.empty((int(n), 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = L.yp_make_priors(img_h, img_w, len(fm), c_fm, c_sc, len(ar), c_ar, out.data_ptr(), int(n), dev.index or 0,
                          ctypes.c_void_p(stream))
    if rc < 0:
        _lib.check(int(rc))
    _DEVICE_PRIORS[key] = out
    return out
