"""The C-ABI boundary without a GPU: the library loads, exports every symbol the header declares,
reports errors the way the reference's constructor does, and the Python mirror refuses to fall back."""
import ctypes
import os
import re

import numpy as np
import pytest

from yolact_amd import _lib
from yolact_amd.detect import Detect

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "yolact_post.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yp_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    syms = _header_symbols()
    assert set(syms) == set(_lib.EXPORTS), (syms, _lib.EXPORTS)
    for s in syms:
        assert hasattr(L, s), s
    assert L.yp_version() == 210


def test_config_struct_matches_header():
    text = open(os.path.join(ROOT, "include", "yolact_post.h")).read()
    body = re.search(r"typedef struct yp_config \{(.*?)\} yp_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip() for n in decl.split(None, 1)[1].split(",")]
    assert names == [f[0] for f in _lib.YpConfig._fields_]
    assert ctypes.sizeof(_lib.YpConfig) == 4 * len(names)


def _cfg(**kw):
    base = dict(num_anchors=240, num_classes=5, mask_dim=32, proto_h=12, proto_w=12, img_h=64, img_w=64,
                conf_thresh=0.05, nms_thresh=0.5, nms_mode=0, top_k=15, max_dets=15, use_cropped_mask=1,
                zero_pad_masks=0, mask_threshold=-1.0, max_batch=2, device_id=0)
    base.update(kw)
    return _lib.YpConfig(**base)


@pytest.mark.parametrize("kw,needle", [
    (dict(nms_thresh=0.0), "nms_threshold must be non negative."),      # detection.py:17-18
    (dict(top_k=0), "top_k"), (dict(top_k=257), "top_k"), (dict(mask_dim=16), "mask_dim"),
    (dict(num_classes=1), "num_classes"), (dict(nms_mode=7), "nms_mode"), (dict(max_batch=0), "max_batch"),
    (dict(nms_mode=1, max_dets=300), "max_dets"),
    (dict(proto_h=40000, proto_w=40000), "mask rows"),                  # 32-bit row offsets in mask_assemble
    (dict(proto_h=2048, proto_w=2048, max_dets=300, top_k=15), "mask rows"),
    (dict(num_anchors=1 << 24, max_batch=128), "max_batch*num_anchors"),
])
def test_create_rejects_bad_config(kw, needle):
    L = _lib.load()
    h = ctypes.c_void_p()
    c = _cfg(**kw)
    rc = L.yp_create(ctypes.byref(c), ctypes.byref(h))
    assert rc == -1 and not h.value
    assert needle in L.yp_last_error().decode()


def test_create_bounded_rejects_a_bound_below_one():
    import ctypes
    from yolact_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.yp_create_bounded(None, 0, ctypes.byref(h)) == -1
    assert b"keys_per_anchor" in L.yp_last_error()


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    h = ctypes.c_void_p()
    c = _cfg()
    rc = L.yp_create(ctypes.byref(c), ctypes.byref(h))
    assert rc == -2 and not h.value                                       # YP_ENODEV: no CPU backend
    assert "HIP device" in L.yp_last_error().decode()


def test_detect_mirrors_reference_constructor():
    with pytest.raises(ValueError, match="nms_threshold must be non negative."):
        Detect(81, 300, 100, 0.05, 0.0)
    d = Detect(81, max_output_size=17, per_class_max_output_size=5, conf_thresh=0.05, nms_thresh=0.5)
    assert (d.max_output_size, d.per_class_max_output_size, d.use_fast_nms) == (300, 100, False)   # detection.py:20-22
    assert (d.nms_mode, d.top_k, d.iou_thresh) == ("cc", 15, 0.5)                                   # detection.py:83,181
    with pytest.raises(NotImplementedError):
        d({}, [1, 550, 550, 3], trad_nms=True)


def test_detect_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d = Detect(5, 300, 100, 0.05, 0.5)
    net = {"pred_offset": np.zeros((1, 4, 4), np.float32), "pred_cls": np.zeros((1, 4, 5), np.float32),
           "pred_mask_coef": np.zeros((1, 4, 32), np.float32), "priors": np.zeros((4, 4), np.float32),
           "proto_out": np.zeros((1, 4, 4, 32), np.float32)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d(net, [1, 64, 64, 3])


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "yolact_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_make_priors_size_query_and_errors():
    """yp_make_priors (data/anchor.py:23-68 on the device): the size query and argument checks need no GPU."""
    import ctypes
    from yolact_amd import _lib
    from yolact_amd.anchors import feature_map_sizes
    L = _lib.load()
    fm = feature_map_sizes(550, 550)
    c_fm = (ctypes.c_int32 * 10)(*[v for hw in fm for v in hw])
    c_sc = (ctypes.c_double * 5)(24, 48, 96, 192, 384)
    c_ar = (ctypes.c_double * 3)(1.0, 0.5, 2.0)
    assert L.yp_make_priors(550, 550, 5, c_fm, c_sc, 3, c_ar, None, 0, 0, None) == 19248
    assert L.yp_make_priors(550, 550, 9, c_fm, c_sc, 3, c_ar, None, 0, 0, None) < 0
    assert b"levels" in L.yp_last_error()
    c_bad = (ctypes.c_double * 3)(1.0, -0.5, 2.0)
    assert L.yp_make_priors(550, 550, 5, c_fm, c_sc, 3, c_bad, None, 0, 0, None) < 0
    assert L.yp_make_priors(0, 550, 5, c_fm, c_sc, 3, c_ar, None, 0, 0, None) < 0


def test_pair_enumeration_used_by_the_nms_kernels():
    """csrc/nms.hip pair_sweep(): column j of the Fast NMS triangle has the j rows i < j; columns f and
    k-1-f are folded into one of k-1 rows, cut into G parts of L rows, one (folded column, part) per
    thread.  Restated here with the same integer arithmetic: every pair (i, j), i < j, exactly once, for
    every k the kernels can meet (top_k <= 256), for both workgroup sizes and for the shares of a workgroup that
    one class gets when the classes of a group are swept side by side."""
    for threads in (7, 32, 102, 512, 1024):     # the whole workgroup, or one class's share of it
        for k in range(0, 258):
            seen = set()
            if k >= 2:
                nf = (k + 1) >> 1
                G = max(1, threads // nf)
                L = (k - 1 + G - 1) // G
                for s in range(nf * G):
                    part, f = divmod(s, nf)
                    jb = k - 1 - f
                    rows = f + (jb if jb != f else 0)
                    for v in range(part * L, min(part * L + L, rows)):
                        pr = (v, f) if v < f else (v - f, jb)
                        assert pr[0] < pr[1] and pr not in seen, (k, s, v)
                        seen.add(pr)
            assert len(seen) == k * (k - 1) // 2, (threads, k)
