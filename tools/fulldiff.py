import sys, numpy as np
sys.path.insert(0, '.')
import mfsgd_amd as mf
from mfsgd_amd import _lib
w = mf.synth.workload("cfg2_ml20m", float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
res = []
for flags in (_lib.FLAG_DEVICE_INGEST | _lib.FLAG_HOST_PACK, _lib.FLAG_DEVICE_INGEST, _lib.FLAG_DEVICE_INGEST):
    with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16, flags=flags) as m:
        m.set_ratings(w["u"], w["i"], w["r"])
        info = m.schedule_info()
        res.append((m.order()[0], m.debug_schedule(), info))
        print(flags, info["device_ingest"], info["total_steps"], info["total_rows"], info["lds_bytes"], flush=True)
ref = res[0]
for k, other in enumerate(res[1:]):
    print("run", k, "order equal", np.array_equal(other[0], ref[0]))
    for a, b, name in zip(other[1], ref[1], ("cells", "rows", "subs", "entries")):
        eq = np.array_equal(a, b)
        print("  ", name, eq, a.shape, b.shape)
        if not eq and a.shape == b.shape:
            bad = np.flatnonzero((a != b).reshape(a.shape[0], -1).any(axis=1))
            print("     first differing rows", bad[:10], "count", bad.size)
            if name == "entries":
                # which cells?
                cells = ref[1][0]
                ent_off = cells[:, 1].astype(np.int64) * info["slots"]
                c = np.searchsorted(ent_off[:info["blocks"]**2], bad[:5], side="right") - 1
                print("     cells", c, "entry offsets", ent_off[c], "n_steps", cells[c, 2] & 0x7FFFFFFF, "nu|ni", cells[c,3] & 0xFFFF, cells[c,3] >> 16)
                for x in bad[:3]:
                    print("     dev", a[x], "host", b[x])
