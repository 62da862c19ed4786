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
    assert L.yp_version() == 310


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


def _disassemble_kernel(name_part):
    """The gfx950 ISA of one kernel of the built library (llvm-objdump on a copy: it unpacks the offload bundles
    next to the file it is given)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    from yolact_amd import _lib
    with tempfile.TemporaryDirectory() as d:
        so = shutil.copy(_lib.LIB_PATH, d)
        subprocess.run([objdump, "--offloading", so], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(os.path.join(d, "*gfx950"))):
            dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
            lines, take = [], False
            for ln in dis.splitlines():
                if ln.endswith(">:"):
                    take = name_part in ln
                elif take and "\t" in ln:
                    lines.append(ln.split("//")[0].strip())
            if lines:
                return lines
    raise AssertionError(f"no kernel matching {name_part} in the gfx950 code objects")


def test_cross_workgroup_handoff_is_write_through_in_the_emitted_isa():
    """csrc/nms.hip hands the class groups' pooled survivors to the image's last-arriving workgroup INSIDE one
    launch: write-through stores, a drained wave, an agent-scope atomic on the arrival counter, and an invalidate
    on the consumer's side.  The C++ memory model promises nothing for relaxed stores without a release, so what
    the compiler emits is pinned here: a toolchain that lowers it differently fails the build
    check instead of corrupting detections now and then."""
    for inst in ("group_nms_kernelILi512ELi6E", "group_nms_kernelILi512ELi4E"):   # beside a fill / launched alone
        _check_handoff_isa(_disassemble_kernel(inst))


def _check_handoff_isa(isa):
    wt8 = [l for l in isa if l.startswith("global_store_dwordx2") and " sc1" in l]
    wt4 = [l for l in isa if l.startswith("global_store_dword ") and " sc1" in l]
    assert len(wt8) >= 4, "the pool's keys and boxes (8-byte halves) must be stored with sc1 (write-through)"
    assert len(wt4) >= 1, "pool_count must be stored with sc1 (write-through)"
    plain8 = [l for l in isa if l.startswith("global_store_dwordx2") and " sc1" not in l]
    assert not plain8, f"an 8-byte store of the class-group kernel lost its sc1: {plain8[:2]}"
    at = [i for i, l in enumerate(isa) if l.startswith("global_atomic_add")]
    assert len(at) == 1, "exactly one atomic: the arrival counter"
    assert " sc0" in isa[at[0]], "the arrival add must return the old value"
    before = isa[max(0, at[0] - 400):at[0]]
    assert any(l.startswith("s_waitcnt vmcnt(0)") for l in before) and any(l.startswith("s_barrier") for l in before), \
        "the stores are drained and the workgroup meets before it arrives"
    after = isa[at[0]:at[0] + 60]
    assert any(l.startswith("buffer_inv") and "sc1" in l for l in after), "the last arriver invalidates (acquire) before it reads the pools"


def test_score_filter_buffer_loads_keep_their_offset_inside_the_range_check():
    """the staging loads of score_filter go through a buffer descriptor of exactly the tile and
    rely on the hardware's range check to fetch nothing past it.  gfx9-family hardware checks VGPR offset +
    immediate only - an SGPR offset is outside the check - so every staging load must carry its whole offset in
    the VGPR / immediate part: soffset 0, `offen`."""
    import re
    n = 0
    for mode in ("ILi0E", "ILi1E"):                      # cross-class and per-class instantiations
        isa = _disassemble_kernel("score_filter_kernel" + mode + "Li11ELb1E")
        loads = [l for l in isa if l.startswith("buffer_load_dwordx4")]
        assert len(loads) >= 11
        for l in loads:
            assert re.search(r"s\[\d+:\d+\], 0 offen", l), f"staging load with an SGPR offset: {l}"
        n += len(loads)
    assert n >= 22
