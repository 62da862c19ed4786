"""Known-answer tests of the consumer-step oracle (oracle/consumer_oracle.py) and of the host-side RLE
string encoder of the library.  cv2 and pycocotools are not importable here and the reference holds
no fixture of their outputs (PARITY UNPINNED): the expected values below are worked out by hand from
the published algorithms, except where the reference states the algorithm itself
(coco_eval.py:91-98 binary_mask_to_rle)."""
import numpy as np
import pytest

from oracle import consumer_oracle as co


def test_counts_follow_the_reference_binary_mask_to_rle():
    rng = np.random.default_rng(0)
    for shape in ((1, 1), (3, 5), (7, 4), (16, 16)):
        for p in (0.0, 0.2, 0.5, 1.0):
            m = (rng.random(shape) < p).astype(np.uint8)
            assert co.binary_mask_to_rle(m)["counts"] == co.rle_counts(m).tolist()
            assert np.array_equal(co.rle_decode(co.rle_counts(m), *shape), m)


def test_rle_known_answers():
    z = np.zeros((3, 4), np.uint8)
    assert co.rle_counts(z).tolist() == [12]                      # all zero: one run
    assert co.rle_counts(1 - z).tolist() == [0, 12]               # all one: leading 0 (coco_eval.py:95-96)
    m = z.copy(); m[1, 2] = 1                                     # column-major position 2*3+1 = 7
    assert co.rle_counts(m).tolist() == [7, 1, 4]
    cb = (np.add.outer(np.arange(2), np.arange(2)) % 2).astype(np.uint8)   # [[0,1],[1,0]] -> F order 0,1,1,0
    assert co.rle_counts(cb).tolist() == [1, 2, 1]


def test_rle_string_known_answers_and_round_trip():
    # 5 bits per character, +48; bit 5 (0x20) = more characters follow; a group whose bit 4 is set ends
    # the number only if what is left is -1 (sign), otherwise only if what is left is 0
    assert co.rle_to_string([0]) == "0" and co.rle_to_string([7]) == "7" and co.rle_to_string([12]) == "<"
    assert co.rle_to_string([16]) == "`0"      # 16 = 0b10000: bit 4 set, rest 0 != -1 -> 16|32+48 = '`', then 0 -> '0'
    assert co.rle_to_string([32]) == "P1"      # low group 0 with more -> 32+48 = 'P', then 1 -> '1'
    # from the fourth count on: difference to the count two places before (7-1 = 6 -> '6'; 1-4 = -3 -> 29+48 = 'M')
    assert co.rle_to_string([7, 1, 4, 7, 1]) == "7146M"
    rng = np.random.default_rng(1)
    for _ in range(50):
        c = rng.integers(0, 100000, rng.integers(1, 40)).tolist()
        assert co.rle_from_string(co.rle_to_string(c)) == c


def test_library_rle_string_matches(tmp_path):
    """yp_rle_string is host code: checked here, without a GPU."""
    from yolact_amd.consumer import rle_string
    rng = np.random.default_rng(2)
    for _ in range(100):
        c = rng.integers(0, 2 ** rng.integers(1, 31), rng.integers(1, 60)).astype(np.uint32)
        assert rle_string(c) == co.rle_to_string(c)
    assert rle_string(np.array([7, 1, 4, 7, 1], np.uint32)) == "7146M"


def test_resize_known_answers():
    # identity size: a copy
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(co.resize_linear_f32(a, 3, 4), a)
    # constant image stays constant
    assert np.array_equal(co.resize_linear_f32(np.full((5, 7), 0.25, np.float32), 11, 13), np.full((11, 13), 0.25, np.float32))
    # 1x2 -> 1x4, half-pixel centres: fx = (dx+.5)/2-.5 = -.25, .25, .75, 1.25 -> clamp, .25, .75, clamp
    r = co.resize_linear_f32(np.array([[0.0, 1.0]], np.float32), 1, 4)
    assert np.allclose(r, [[0.0, 0.25, 0.75, 1.0]], atol=0) 
    # 2x upsampling of a step along y
    r = co.resize_linear_f32(np.array([[0.0], [1.0]], np.float32), 4, 1)
    assert np.allclose(r[:, 0], [0.0, 0.25, 0.75, 1.0], atol=0)
    # bytes: 0/1 step, 1x2 -> 1x4: (b0*(R0>>4)>>16 + b1*(R1>>4)>>16 + 2)>>2 with R = s0*a0 + s1*a1 in 1/2048:
    # R = [0, 512, 1536, 2048] -> R>>4 = [0, 32, 96, 128]; b0 = 2048: (2048*v)>>16 = [0, 1, 3, 4]; (+2)>>2 = [0, 0, 1, 1]
    r8 = co.resize_linear_u8(np.array([[0, 1]], np.uint8), 1, 4)
    assert r8.tolist() == [[0, 0, 1, 1]]
    # shrinking is two-tap INTER_LINEAR as well: 1x4 -> 1x3, scale 4/3: fx = (dx+.5)*4/3-.5 = 1/6, 1.5, 17/6 -> taps
    # (0,1) w 1/6, (1,2) w .5, (2,3) w 5/6
    r = co.resize_linear_f32(np.array([[0.0, 3.0, 6.0, 9.0]], np.float32), 1, 3)
    assert np.allclose(r, [[0.5, 4.5, 8.5]], atol=1e-6)
    # ... except exactly half in both axes, which cv::resize hands to INTER_AREA: not restated, refused
    with pytest.raises(ValueError):
        co.resize_linear_f32(np.ones((8, 8), np.float32), 4, 4)
    assert co.resize_linear_f32(np.ones((8, 8), np.float32), 4, 8).shape == (4, 8)        # one axis only: still linear


def test_threshold_orders_differ_only_near_edges():
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:24, 0:24]
    m = (1.0 / (1.0 + np.exp(-(8.0 - np.hypot(yy - 11.3, xx - 12.1))))).astype(np.float32)   # a soft disc
    a = co.mask_to_bitmap(m, 96, 96, 0.5, co.RESIZE_THEN_THRESHOLD)
    b = co.mask_to_bitmap(m, 96, 96, 0.5, co.THRESHOLD_THEN_RESIZE)
    assert a.shape == b.shape == (96, 96) and set(np.unique(a)) <= {0, 1} and set(np.unique(b)) <= {0, 1}
    assert 0 < (a != b).mean() < 0.05 and abs(int(a.sum()) - int(b.sum())) < 0.05 * a.sum()
    _ = rng


def test_loss_mask_backward_oracle_against_finite_differences():
    """The F4 backward oracle (no reference code exists for it: TF autodiff) is pinned to the forward oracle
    by central differences in float64, and the C restatement (kernel summation order, float32) to the float64
    analytic form."""
    from oracle import c_oracle, detect_oracle as po
    rng = np.random.default_rng(3)
    ph, pw, K, n = 14, 17, 32, 6
    proto = np.maximum(0, rng.normal(0, 1, (ph, pw, K)))
    coef = 0.3 * np.tanh(rng.normal(0, 1, (n, K)))
    box = np.stack([rng.uniform(0, 30, n), rng.uniform(0, 30, n), rng.uniform(40, 100, n), rng.uniform(40, 100, n)], -1)
    g = rng.normal(0, 1, (n, ph, pw))
    fwd = lambda pr, cf: po.loss_mask_assembly(pr, cf, box, 100.0, 100.0)
    m = fwd(proto, coef)
    gp, gc = po.loss_mask_assembly_backward(proto, coef, m, g)
    eps = 1e-6
    for _ in range(12):
        i = tuple(int(rng.integers(0, s)) for s in coef.shape)
        d = np.zeros_like(coef); d[i] = eps
        fd = ((fwd(proto, coef + d) - fwd(proto, coef - d)) * g).sum() / (2 * eps)
        assert abs(fd - gc[i]) <= 1e-6 * max(1.0, abs(fd))
        j = tuple(int(rng.integers(0, s)) for s in proto.shape)
        d = np.zeros_like(proto); d[j] = eps
        fd = ((fwd(proto + d, coef) - fwd(proto - d, coef)) * g).sum() / (2 * eps)
        assert abs(fd - gp[j]) <= 1e-6 * max(1.0, abs(fd))
    f4 = [x.astype(np.float32) for x in (proto, coef, m, g)]
    cp, cc = c_oracle.loss_masks_backward(*f4)
    ap, ac = po.loss_mask_assembly_backward(*[x.astype(np.float64) for x in f4])
    assert np.abs(cp - ap).max() <= 1e-5 and np.abs(cc - ac).max() <= 1e-4


def test_batched_rle_strings_match_the_oracle_and_the_single_call():
    """yp_rle_strings (host code of the library, no device needed): many count lists in one call give the
    strings of pycocotools' rleToString restatement, list by list; empty lists give empty strings; a buffer
    that is too small is reported by the returned size."""
    import ctypes
    from oracle import consumer_oracle as co
    from yolact_amd import _lib, consumer
    rng = np.random.default_rng(9)
    n, stride = 40, 64
    lens = rng.integers(0, stride + 1, n).astype(np.int32); lens[3] = 0; lens[7] = stride
    counts = rng.integers(0, 400000, (n, stride)).astype(np.uint32)
    counts[5, :4] = [0, 1, 2**31, 5]                         # large differences, both signs
    got = consumer.rle_strings(counts, lens)
    for i in range(n):
        assert got[i] == co.rle_to_string(counts[i, :lens[i]]), i
        assert got[i] == consumer.rle_string(counts[i, :lens[i]])
    L = _lib.load()
    need = sum(len(x) + 1 for x in got)
    buf = ctypes.create_string_buffer(16); offs = np.zeros(n + 1, np.int64)
    starts = np.arange(n, dtype=np.int64) * stride
    assert L.yp_rle_strings(counts.ctypes.data, starts.ctypes.data, lens.ctypes.data, n, counts.size, ctypes.addressof(buf), 16, offs.ctypes.data) == need
    bad = lens.copy(); bad[n - 1] = stride + 1                 # would run past the end of the array
    assert L.yp_rle_strings(counts.ctypes.data, starts.ctypes.data, bad.ctypes.data, n, counts.size, ctypes.addressof(buf), 16, offs.ctypes.data) < 0
    # packed back to back
    flat = np.concatenate([counts[i, :lens[i]] for i in range(n)])
    pst = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    assert consumer.rle_strings(flat, lens, pst) == got


def test_batched_rle_strings_on_several_threads():
    """A batch's worth of lists (more than 65536 counts) is encoded by several host threads: the strings, their
    offsets and the truncation of a buffer that is too small are those of the one-list call."""
    import ctypes
    from yolact_amd import _lib, consumer
    rng = np.random.default_rng(11)
    n = 700
    lens = rng.integers(0, 400, n).astype(np.int32); lens[::50] = 0; lens[13] = 5000        # ragged, one long list
    starts = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    flat = rng.integers(0, 90000, int(lens.sum())).astype(np.uint32)
    assert flat.size > 65536
    got = consumer.rle_strings(flat, lens, starts)
    for i in range(0, n, 7):
        assert got[i] == consumer.rle_string(flat[starts[i]:starts[i] + lens[i]]), i
    assert got[13] == consumer.rle_string(flat[starts[13]:starts[13] + 5000])
    L = _lib.load()
    need = sum(len(x) + 1 for x in got)
    cap = need // 2
    buf = ctypes.create_string_buffer(cap); offs = np.zeros(n + 1, np.int64)
    assert L.yp_rle_strings(flat.ctypes.data, starts.ctypes.data, lens.ctypes.data, n, flat.size, ctypes.addressof(buf), cap, offs.ctypes.data) == need
    raw = buf.raw
    whole = [i for i in range(n) if offs[i + 1] <= cap]          # the strings that fit are complete, in their places
    assert len(whole) > 100
    for i in whole:
        assert raw[offs[i]:offs[i + 1] - 1].decode("ascii") == got[i]
