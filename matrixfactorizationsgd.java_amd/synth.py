"""Synthetic rating matrices of the shapes BASELINE.json's configs name.

The reference publishes no data and no generator (/root/reference/README.md:1-2),
and there is no network for MovieLens/Netflix, so every workload is synthetic:

* ground truth: rank-``k_true`` factors ``P*, Q* ~ U(0,1) * sqrt(12/k_true)``
  (mean rating 3), ``r = dot(P*[u], Q*[i]) + N(0, noise^2)``;
* ``dense``: every (u, i) pair once (config 0);
* ``uniform``: nnz distinct pairs uniformly at random (config 1);
* ``zm`` (Zipf-Mandelbrot): users and items drawn independently with weight
  ``1/(rank + q)^s``, pairs de-duplicated, ids randomly permuted so that
  popularity is not correlated with the index.  The offsets are calibrated so
  that the heaviest user / item carry the share of ratings they carry in the
  published MovieLens-20M statistics (heaviest item 67,310 of 20,000,263
  ratings = 0.34 %, heaviest user 9,254 = 0.046 %, lightest user 20).
  SURVEY.md section 8d proposed a pure Zipf(s=1, q=0); de-duplicated it
  saturates -- its 18 heaviest items are rated by EVERY user and its heaviest
  users rate EVERY item -- which no rating dataset resembles, so the offset
  form replaces it.  ``zipf_q=0`` reproduces the pure form.

Generator: numpy PCG64 seeded per workload (same stream here and on the GPU
box: same image, same numpy).
"""
import math

import numpy as np

# name -> generator arguments.  BASELINE.json configs[0..4].
WORKLOADS = {
    "cfg0_dense100x80": dict(U=100, I=80, nnz=8000, k=8, dist="dense", seed=1),
    "cfg1_ml100k": dict(U=943, I=1682, nnz=100_000, k=32, dist="uniform", seed=2),
    # offsets fitted on the GENERATED extremes (de-duplication flattens the head: the expected share of
    # the heaviest item at q_item = 14 is 0.9 %, what survives is 0.34 %): heaviest item 67,853 ratings
    # (MovieLens-20M: 67,310), heaviest user 9,215 (9,254), lightest user 11 (20)
    "cfg2_ml20m": dict(U=138_493, I=26_744, nnz=20_000_000, k=64, dist="zm", seed=3,
                       q_user=140.0, q_item=14.0),
    # SURVEY.md 8d's own proposal, pure Zipf(s = 1) de-duplicated (saturated head; bench / DESIGN only)
    "cfg2_zipf": dict(U=138_493, I=26_744, nnz=20_000_000, k=64, dist="zm", seed=3, q_user=0.0, q_item=0.0),
    # the round-1 calibration (heaviest item 40,395 = 0.20 %), kept so that round-1 numbers stay comparable
    "cfg2_r1": dict(U=138_493, I=26_744, nnz=20_000_000, k=64, dist="zm", seed=3, q_user=370.0, q_item=40.0),
    # same shape as cfg2 with uniform popularity: no long per-row chains (throughput-bound regime)
    "cfg2_uniform": dict(U=138_493, I=26_744, nnz=20_000_000, k=64, dist="uniform", seed=13),
    # fitted on the generated extremes like cfg2: heaviest item 232,809 ratings (Netflix Prize: 232,944),
    # heaviest user 16,761 (17,653), lightest user 17
    "cfg3_netflix": dict(U=480_189, I=17_770, nnz=100_000_000, k=128, dist="zm", seed=4,
                         q_user=100.0, q_item=24.0),
    "cfg4_powerlaw": dict(U=10_000_000, I=1_000_000, nnz=1_000_000_000, k=256, dist="zm", seed=5,
                          s_user=1.1, s_item=1.1, q_user=2000.0, q_item=200.0),
}


def zm_weights(n, s, q):
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64) + q, s)
    return w / w.sum()


def _draw(rng, cdf, m):
    x = np.searchsorted(cdf, rng.random(m), side="right")
    np.minimum(x, cdf.size - 1, out=x)
    return x


def make_pairs(U, I, nnz, dist, rng, s_user=1.0, s_item=1.0, q_user=0.0, q_item=0.0):
    """Returns (u, i) int32 arrays of nnz distinct pairs in random order."""
    if dist == "dense":
        if nnz != U * I:
            raise ValueError("dense needs nnz == U*I")
        key = np.arange(U * I, dtype=np.int64)
    elif dist == "uniform":
        if nnz > U * I:
            raise ValueError("nnz exceeds U*I")
        if U * I <= 50_000_000:
            key = rng.choice(U * I, size=nnz, replace=False).astype(np.int64)
        else:
            key = np.empty(0, np.int64)
            while key.size < nnz:
                m = int((nnz - key.size) * 1.1) + 1024
                cand = rng.integers(0, U, m, dtype=np.int64) * I + rng.integers(0, I, m, dtype=np.int64)
                key = np.unique(np.concatenate([key, cand]))
            key = rng.permutation(key)[:nnz]
    elif dist == "zm":
        cu = np.cumsum(zm_weights(U, s_user, q_user))
        ci = np.cumsum(zm_weights(I, s_item, q_item))
        key = np.empty(0, np.int64)
        while key.size < nnz:
            m = int((nnz - key.size) * 1.25) + 1024
            cand = _draw(rng, cu, m).astype(np.int64) * I + _draw(rng, ci, m)
            key = np.unique(np.concatenate([key, cand]))
        key = rng.permutation(key)[:nnz]
        # popularity rank -> random id
        pu = rng.permutation(U).astype(np.int64)
        pi = rng.permutation(I).astype(np.int64)
        key = pu[key // I] * I + pi[key % I]
    else:
        raise ValueError(f"unknown dist {dist!r}")
    key = rng.permutation(key)
    return (key // I).astype(np.int32), (key % I).astype(np.int32)


def make_ratings(U, I, nnz, k, dist="uniform", seed=0, k_true=16, noise=0.1, **kw):
    """(u, i, r) for one synthetic workload.  `k` is unused by the data (the truth has
    rank k_true); it is accepted so WORKLOADS entries can be splatted."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u, i = make_pairs(U, I, nnz, dist, rng, **kw)
    scale = math.sqrt(12.0 / k_true)
    Pt = (rng.random((U, k_true), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    Qt = (rng.random((I, k_true), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    r = np.empty(nnz, np.float32)
    step = 2_000_000
    for a in range(0, nnz, step):
        b = min(a + step, nnz)
        r[a:b] = np.einsum("nk,nk->n", Pt[u[a:b]], Qt[i[a:b]])
        r[a:b] += rng.normal(0.0, noise, b - a).astype(np.float32)
    return u, i, r


def make_ratings_device(U, I, nnz, k, dist="zm", seed=0, k_true=16, noise=0.1, s_user=1.0, s_item=1.0, q_user=0.0,
                        q_item=0.0, device="cuda", user_range=None, log=None):
    """The `zm` generator of make_ratings on the GPU (torch ops: plumbing for device memory, not the product):
    the same construction -- Zipf-Mandelbrot draws by inverse CDF, pairs de-duplicated, popularity rank -> random
    id, rank-k_true ground truth plus noise -- with torch's generator instead of numpy's, so the SAME distribution
    but a DIFFERENT sample than the host generator (the workload dict says which one made it).  It exists for the
    sizes the host generator needs half an hour for (BASELINE configs[4]: 1 B ratings): seconds here.
    user_range = (lo, hi), or a function (deg_user, deg_item) -> (lo, hi) called with the GLOBAL rating counts per
    row (numpy int64): keep only the ratings of users lo <= u < hi -- a DSGD rank's shard of the global set (every
    rank generates the same global set from the same seed on its own GPU, derives the same plan from the same
    degrees and keeps its range; nothing is exchanged and the global set never exists on a host).
    Returns host numpy arrays (u, i, r): the C-ABI takes host pointers."""
    import torch

    if dist != "zm":
        raise ValueError("the device generator implements the zm distribution only")
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) * 7919 + 13)
    cu = torch.cumsum(torch.from_numpy(zm_weights(U, s_user, q_user)).to(dev), 0)
    ci = torch.cumsum(torch.from_numpy(zm_weights(I, s_item, q_item)).to(dev), 0)

    def draw(cdf, m):
        x = torch.searchsorted(cdf, torch.rand(m, generator=g, device=dev, dtype=torch.float64), right=True)
        return x.clamp_(max=cdf.numel() - 1)

    key = torch.empty(0, dtype=torch.int64, device=dev)
    chunk = 250_000_000  # candidates per pass: bounds the sort's temporaries (~3 x 8 bytes each)
    while key.numel() < nnz:
        m = min(chunk, int((nnz - key.numel()) * 1.25) + 1024)
        cand = draw(cu, m) * I + draw(ci, m)
        key = torch.unique(torch.cat([key, cand]))
        del cand
        if log:
            log(f"  device generator: {key.numel()} distinct pairs of {nnz}")
    key = key[torch.randperm(key.numel(), generator=g, device=dev)[:nnz]]  # a random subset, in random order
    pu = torch.randperm(U, generator=g, device=dev)  # popularity rank -> random id
    pi = torch.randperm(I, generator=g, device=dev)
    u = pu[key // I]
    i = pi[key % I]
    del key, pu, pi, cu, ci
    if user_range is not None:
        if callable(user_range):
            user_range = user_range(torch.bincount(u, minlength=U).cpu().numpy(), torch.bincount(i, minlength=I).cpu().numpy())
        keep = (u >= user_range[0]) & (u < user_range[1])
        u, i = u[keep], i[keep]
        del keep
    scale = math.sqrt(12.0 / k_true)
    Pt = torch.rand((U, k_true), generator=g, device=dev, dtype=torch.float32) * scale
    Qt = torch.rand((I, k_true), generator=g, device=dev, dtype=torch.float32) * scale
    n = u.numel()
    r = torch.empty(n, dtype=torch.float32, device=dev)
    step = 50_000_000
    for a in range(0, n, step):
        b = min(a + step, n)
        r[a:b] = (Pt[u[a:b]] * Qt[i[a:b]]).sum(1) + torch.randn(b - a, generator=g, device=dev) * noise
    out = (u.to(torch.int32).cpu().numpy(), i.to(torch.int32).cpu().numpy(), r.cpu().numpy())
    del u, i, r, Pt, Qt
    torch.cuda.empty_cache()
    return out


def workload(name, scale=1.0, seed_offset=0, item_mult=1, generator="host", user_range=None, log=None):
    """Ratings of a named workload; scale < 1 shrinks U, I and nnz together
    (for parity tests at sizes the oracle finishes in seconds); seed_offset
    gives each DSGD rank its own users and ratings; item_mult widens the item
    catalogue (weak scaling over N GPUs: N times the items, same ratings per rank).
    generator = "device": make_ratings_device (zm workloads; another sample of the same distribution)."""
    w = dict(WORKLOADS[name])
    w["seed"] = w["seed"] + seed_offset
    if item_mult != 1:
        w["I"] = w["I"] * item_mult
        if "q_item" in w:
            w["q_item"] = w["q_item"] * item_mult
    if scale != 1.0:
        if w["dist"] == "dense":
            w["U"] = max(2, int(w["U"] * math.sqrt(scale)))
            w["I"] = max(2, int(w["I"] * math.sqrt(scale)))
            w["nnz"] = w["U"] * w["I"]
        else:
            w["U"] = max(8, int(w["U"] * scale))
            w["I"] = max(8, int(w["I"] * scale))
            w["nnz"] = max(16, min(int(w["nnz"] * scale), w["U"] * w["I"] // 2))
            for q in ("q_user", "q_item"):
                if q in w:
                    w[q] = w[q] * scale
    k = w["k"]
    args = {x: w[x] for x in w if x not in ("k",)}
    if generator == "device":
        u, i, r = make_ratings_device(k=k, user_range=user_range, log=log, **args)
    else:
        u, i, r = make_ratings(k=k, **args)
        if user_range is not None:
            if callable(user_range):
                user_range = user_range(np.bincount(u, minlength=w["U"]).astype(np.int64), np.bincount(i, minlength=w["I"]).astype(np.int64))
            keep = (u >= user_range[0]) & (u < user_range[1])
            u, i, r = u[keep], i[keep], r[keep]
    return dict(U=w["U"], I=w["I"], nnz=int(u.size), k=k, u=u, i=i, r=r, name=name, dist=w["dist"], generator=generator)
