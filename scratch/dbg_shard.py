import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ookiedokie_amd as ok
from ookiedokie_amd import distributed as okd
from tests.helpers import GOLDEN, golden_path, iq_from_rle
g = json.load(open(os.path.join(GOLDEN, "vectors.json")))["G1"]
iq = iq_from_rle(g["i_rle"], g["num_samples"])
rng = np.random.default_rng(21)
iq = (iq + rng.integers(-40, 41, size=iq.size)).astype(np.int16)
n = iq.size // 2
flt = ok.Filter.load(golden_path("filters", "fs128_fs16_dec4"))
dev = ok.Device.load(golden_path("devices", "p3l-nexa2012"), 750000)
b = okd.shard_bounds(n, 2, 8192, 4)
print("bounds", b)
for rounds in (False, True):
    outs = []
    rxs = []
    for r in range(2):
        lo, hi = b[r], b[r+1]
        t = torch.from_numpy(iq[2*lo:2*hi].copy()).cuda()
        rx = ok.Receiver(flt, dev, max_samples=hi-lo, samples_per_buffer=8192, fsm_rounds=rounds)
        H = rx.halo_samples
        halo = iq[2*(lo-H):2*lo] if r > 0 else None
        res, out = rx.shard_begin(t.data_ptr(), hi-lo, halo, r == 1, None)
        print(rounds, "rank", r, "begin msgs", res.msg_samples, "path", res.stats["fsm_path"], res.stats["fsm_fallback_reason"], "out", out.state, out.num_bits, out.k, out.prev_bit)
        outs.append(out); rxs.append((rx, t))
    res, out = rxs[1][0].shard_refine(outs[0])
    print(rounds, "rank 1 refine msgs", res.msg_samples, "path", res.stats["fsm_path"], res.stats["fsm_fallback_reason"], "edges", res.stats["num_edges"])
