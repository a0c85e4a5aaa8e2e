"""Diagnostic: per-phase shader cycles of one persistent epoch (mfsgd_debug_epoch_profile).

    python tools/phase_profile.py WORKLOAD SCALE [BLOCKS [WAVES [FLAGS]]]

MFSGD_EMU=N in the environment: profile ONE partition of the N-rank weak-scaling job of bench.py (N times the
items, the global plan's partition 0 as a problem of its own).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mfsgd_amd as mf  # noqa: E402

name, scale = sys.argv[1], float(sys.argv[2])
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 0
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
emu = int(os.environ.get("MFSGD_EMU", "1"))
w = mf.synth.workload(name, scale, item_mult=emu)
if emu > 1:
    _, ip = mf.dsgd_plan(np.ones(1, np.int64), np.bincount(w["i"], minlength=w["I"]), emu)
    keep = ip[w["i"]] == 0
    new_id = np.cumsum(ip == 0) - 1
    w["u"], w["i"], w["r"] = w["u"][keep], new_id[w["i"][keep]].astype(np.int32), w["r"][keep]
    w["I"] = int((ip == 0).sum())
    print(f"partition 0 of {emu}: {keep.sum()} ratings, {w['I']} items, heaviest item {np.bincount(w['i']).max()}")
with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, blocks=blocks, waves=waves, flags=flags) as m:
    m.set_ratings(w["u"], w["i"], w["r"])
    m.init_factors()
    info = m.schedule_info()
    m.fit(1, rmse=False)
    ms, _ = m.train_timed(5)
    prof = m.debug_epoch_profile().astype(np.float64)
    slowest = m.last_slowest_cell.astype(np.float64)
    slowpass = m.last_slowest_pass.astype(np.float64)
names = ["drain+issue", "tile wait", "barrier", "tile gather", "ratings", "publish", "own store"]
tot = prof.sum(axis=1)
nnz = info['nnz']
print(f"{name} x{scale}: B={info['blocks']} W={info['waves']} lds={info['lds_bytes']} chunks={info['chunks']} "
      f"split={info['split_cells']} workgroups={prof.shape[0]} epoch={ms / 5:.3f} ms "
      f"rate={nnz / (ms / 5) / 1e3:.0f} M/s")
print("  phase cycles (mean over workgroups, share of the mean total %.0f):" % tot.mean())
for k, nm in enumerate(names):
    print(f"    {nm:12s} {prof[:, k].mean():12.0f}  {100 * prof[:, k].mean() / tot.mean():5.1f} %")
print(f"  slowest cell's ratings phase per workgroup: mean {slowest.mean():.0f} max {slowest.max():.0f} cycles; "
      f"max_cell_steps {info['max_cell_steps']} -> {slowest.mean() / max(1, info['max_cell_steps']):.1f} cycles per step-equivalent; "
      f"sum of phases {tot.mean():.0f} cycles per epoch -> {tot.mean() / (ms / 5) / 1e6:.2f} GHz")
print("  the pass with the slowest cell, per workgroup (mean cycles): " + ", ".join(f"{nm} {slowpass[:, k].mean():.0f}" for k, nm in enumerate(names))
      + f"; total {slowpass.sum(axis=1).mean():.0f}")
