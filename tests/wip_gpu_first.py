import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from yolact_amd import synth, PostProcessor, PostConfig
from oracle import c_oracle as co
from tests.util import to_numpy, compare
def run(net, shp, mode, top_k, max_dets, conf=0.05, exact=True):
    B, A, C = net['pred_cls'].shape
    ph, pw = net['proto_out'].shape[1:3]
    pp = PostProcessor(PostConfig(A, C, ph, pw, int(shp[1]), int(shp[2]), conf_thresh=conf, nms_mode=mode, top_k=top_k, max_dets=max_dets, max_batch=B))
    dev = {k: torch.from_numpy(v).cuda() for k, v in net.items()}
    out = pp.run(dev['pred_offset'], dev['pred_cls'], dev['pred_mask_coef'], dev['priors'], dev['proto_out'])
    pp.sync()
    got = to_numpy(out)
    ref = co.detect(net, shp, conf_thresh=conf, top_k=top_k, nms_mode=mode, max_dets=max_dets, nthreads=8)
    print(mode, top_k, 'num', got['num_detections'][:8], ref['num_detections'][:8], 'surv', pp.last_survivors(B)[:8], ref['num_survivors'][:8], flush=True)
    assert np.array_equal(pp.last_survivors(B), ref['num_survivors'])
    compare(got, ref, exact_values=exact, label=f'{mode}/{top_k}')
    print('  OK bit-exact' if exact else '  OK tol', flush=True)
    return pp, dev
net, shp = synth.make_toy(3)
for mode, tk, md in (('cc', 15, 15), ('cc', 200, 100), ('per_class', 200, 100)):
    run(net, shp, mode, tk, md)
cfg = synth.CONFIGS[2]
net, shp = synth.make_batch(cfg, batch=4)
for mode, tk, md in (('cc', 15, 300), ('cc', 200, 100), ('per_class', 200, 100)):
    pp, dev = run(net, shp, mode, tk, md)
print('ALL OK')
