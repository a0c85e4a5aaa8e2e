#!/usr/bin/env python3
"""bench.py -- rating-updates/sec of the MF-SGD hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one epoch: one pass of the hot path (dot, error, rank-1 update of
P[u] and Q[i]) over every rating of the workload.  N = 1 runs BASELINE.json's
configs[2] -- the configuration its metric is quoted on: MovieLens-20M shape
(138,493 x 26,744, 20M ratings), k = 64, fp32 -- on synthetic data.

N > 1 is DSGD: one process per GPU, the item factors cut into N (x parts-per-rank)
blocks that rotate between the ranks over RCCL send/recv under the C-ABI
(csrc/dsgd.cpp).  `python bench.py --gpus N` starts the N ranks itself (fresh
child processes, before anything in this process touches the GPU) and relays
rank 0's JSON line; under `python -m torch.distributed.run` (RANK / WORLD_SIZE in
the environment) it is one of the ranks.  Two problem definitions:

  --scaling weak    every rank brings the users and ratings of the N = 1 run and
                    the item catalogue is N times larger (default for cfg2_*);
  --scaling strong  ONE global rating set (BASELINE configs 3 and 4 as stated: the
                    Netflix shape and the power-law shape cut over the N GPUs by
                    mfsgd_dsgd_plan); default for cfg3_netflix / cfg4_powerlaw.

value = ratings processed by all ranks / max-over-ranks time.  Ratings, schedules
and factors are resident in HBM before the timed region; the timed region
contains the training passes only (no RMSE pass, no host<->device copies).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (algorithmic bytes of
the dominant kernel / its average launch duration vs the 8 TB/s HBM peak) and
"cpu_baseline" (the oracle's multithreaded CPU path on the host cores, N = 1
only -- a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
LR, LAM, SEED = 0.01, 0.05, 3
# generator revision of the named workloads (synth.py WORKLOADS): bumped whenever a calibration changes, so that
# two bench lines of the same workload name are comparable only if this matches (r1: cfg2 head 40 K; r2: 67.9 K)
WORKLOAD_REV = 2


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(16, n))  # a 1-GPU box gives this job 16 cores


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


# ---- N > 1 without a launcher: this process starts the ranks ---------------------------------------------
def spawn_ranks(n, argv):
    """Starts `n` fresh copies of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays
    rank 0's stdout (the one JSON line) and returns the worst exit code.  Nothing here touches the GPU: the children
    are new processes, not re-executions of one that has initialised HIP."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rk in range(n):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL))
    rc, alive = 0, set(range(n))
    try:
        while alive:
            for rk in sorted(alive):
                r = procs[rk].poll()
                if r is None:
                    continue
                alive.discard(rk)
                if r != 0 and rc == 0:
                    rc = r if r > 0 else 1
                    log(f"rank {rk} exited with {r}; stopping the others")
                    for other in alive:  # exactly the processes started above
                        procs[other].terminate()
            if alive:
                time.sleep(0.1)
    except BaseException:
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise
    out = procs[0].stdout.read().decode()
    if rc == 0:
        lines = [x for x in out.splitlines() if x.strip()]
        if len(lines) != 1:
            log(f"rank 0 printed {len(lines)} lines instead of one")
            rc = 1
        sys.stdout.write(out)
        sys.stdout.flush()
    return rc


def cpu_baseline(w, m, budget_s=12.0):
    """Times the oracle's multithreaded block-schedule epoch (kind "port": this
    repository's CPU restatement; the reference has no runnable CPU path) on the
    same ratings and the same schedule, for about `budget_s` seconds -- the bit-exact
    checker's arithmetic (lane-tree dot) first, then the plain left-to-right fp32 loop a
    Java maintainer would write (`textbook`), same schedule and threads, with the RMSE
    gap between the two after the same number of epochs."""
    from tests.oracle_bind import Oracle

    orc = Oracle()
    info = m.schedule_info()
    order, cell_ptr = m.order()
    threads = host_threads()

    def run(fn, budget):
        P, Q = orc.init_factors(w["U"], w["I"], w["k"], SEED)
        t_used, epochs, rm = 0.0, 0, []
        # whole epochs; stop once the budget is used (at least one)
        while epochs < 1 or (t_used < budget and t_used / epochs * (epochs + 1) < budget * 1.5):
            t0 = time.perf_counter()
            fn(P, Q, w["u"], w["i"], w["r"], order, cell_ptr, info["rounds"], info["blocks"], LR, LAM, threads)
            t_used += time.perf_counter() - t0
            epochs += 1
            if epochs <= 2:  # the RMSE gap between the two arithmetics: two epochs are enough (a single-threaded pass each)
                rm.append(orc.rmse(P, Q, w["u"], w["i"], w["r"]))
        return t_used, epochs, rm

    t_c, e_c, rm_c = run(orc.sgd_epoch_mt, budget_s)
    t_t, e_t, rm_t = run(orc.textbook_epoch_mt, budget_s * 0.75)
    n = min(len(rm_c), len(rm_t))
    return {
        "value": w["nnz"] * e_c / t_c,
        "unit": "updates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{e_c} full epoch(s) of the same {w['nnz']} ratings and block schedule, "
                  f"oracle/mfsgd_oracle.c mfo_sgd_epoch_mt (the bit-exact checker's arithmetic), {t_c:.1f} s",
        "textbook": {
            "value": w["nnz"] * e_t / t_t, "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"{e_t} full epoch(s), mfo_textbook_epoch_mt: plain left-to-right fp32 loop p += lr*(e*q - lambda*p) "
                      f"on the same block schedule and threads, {t_t:.1f} s",
            "rmse_gap_to_contract": max(abs(a - b) for a, b in zip(rm_c[:n], rm_t[:n])),
            "epochs_compared": n,
        },
    }


def read_traffic(workload, k, nnz):
    """HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 PMC passes of THIS round's build, tools/prof.sh)
    from the committed summary of the same workload, if there is one.  It is a profile of the same command
    taken under rocprofv3, not a counter read inside this run (PMC collection needs the profiler)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_*.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("workload") == workload and t.get("k") == k and t.get("nnz") == nnz:
                return t.get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2_ml20m")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debugging only)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N > 1: weak = every rank brings the N = 1 problem (items x N); strong = one global rating set cut "
                         "over the ranks by mfsgd_dsgd_plan; auto = strong for cfg3_netflix / cfg4_powerlaw, weak otherwise")
    ap.add_argument("--generator", default="auto", choices=["auto", "host", "device"], help="where the synthetic ratings are made: host "
                    "(numpy; the sample every committed number of cfg0..cfg3 is quoted on), device (torch on the GPU: the same "
                    "distribution, another sample, seconds instead of half an hour at 1 B ratings); auto = device for cfg4_powerlaw "
                    "above 100 M ratings, host otherwise")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--parts-per-rank", type=int, default=0, help="DSGD: item partitions a rank holds at a time (0 = 1)")
    ap.add_argument("--plan-crit", type=float, default=0.0, help="DSGD item plan: 0 = longest-processing-time-first (default), "
                    "> 0 = chain-aware packing with this criticality threshold (mfsgd_dsgd_plan_ex; not for a ring, DESIGN.md section 6)")
    ap.add_argument("--emulate-world", type=int, default=0, help="debugging: run ONE rank of an N-GPU DSGD job on one "
                    "GPU without communication (per-rank compute time of that job)")
    ap.add_argument("--emulate-rank", type=int, default=0, help="which rank --emulate-world runs (strong scaling: its user range)")
    ap.add_argument("--selftest-native-ring", type=int, default=0, metavar="PARTS", help="debugging, one GPU: run the N > 1 "
                    "code path of this file (global plan, ring under the C-ABI, its timing) with PARTS item partitions and an "
                    "RCCL self-ring on this rank alone")
    ap.add_argument("--round-launch", action="store_true", help="one kernel per round instead of the persistent epoch kernel")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="debugging: the N-rank code path of this file with "
                    "EVERY rank on GPU (local_rank mod device count) -- the ring under the C-ABI moves its blocks through the "
                    "driver's shared-memory rehearsal transport (RCCL refuses two ranks on one GPU), round launches (the "
                    "ranks' persistent kernels would not be co-resident).  Not a measurement.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver calls it: this process is the launcher
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE line, the JSON.  Native libraries write there too (gloo announces its peers on stdout):
    # from here on file descriptor 1 is stderr, and the line goes to a copy of the real stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.rehearse_on_one_gpu:
        # the rehearsal transport is not in the product library: this rank loads libmfsgd_rehearsal.so instead
        os.environ["MFSGD_DSGD_TRANSPORT"] = "shm"  # read by mfsgd_dsgd_unique_id on rank 0
        os.environ.setdefault("MFSGD_LIBRARY", os.path.join(ROOT, "matrixfactorizationsgd.java_amd", "lib", "libmfsgd_rehearsal.so"))
        args.round_launch = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    emu = args.emulate_world if args.emulate_world > 1 and world == 1 else 0
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch

    import mfsgd_amd
    from mfsgd_amd import _lib, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmfsgd has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        # control plane only (the RCCL id, the item histogram, the max of the times): the blocks travel under the C-ABI
        dist.init_process_group("gloo", rank=rank, world_size=world)

    generator = args.generator
    if generator == "auto":
        big = args.workload == "cfg4_powerlaw" and synth.WORKLOADS[args.workload]["nnz"] * args.scale > 100e6
        generator = "device" if big else "host"
    vworld = emu if emu else world  # ranks the problem is sized for
    vrank = (args.emulate_rank % emu) if emu else rank
    scaling = args.scaling
    if scaling == "auto":
        scaling = "strong" if args.workload in ("cfg3_netflix", "cfg4_powerlaw") else "weak"
    if vworld == 1:
        scaling = "weak"  # one rank: the two definitions coincide

    # partitions of one rank share its users, so they are trained one after another; more than
    # one per rank only makes the schedules smaller and lets the shifts hide behind training
    ppr = args.parts_per_rank if args.parts_per_rank > 0 else 1
    selftest = args.selftest_native_ring if world == 1 and not emu and args.selftest_native_ring > 1 else 0
    if selftest:
        ppr = selftest
    n_parts = vworld * ppr if (vworld > 1 or selftest) else 0
    native = world > 1 or bool(selftest)

    # ---- workload --------------------------------------------------------------------------------------------
    t0 = time.time()
    u_offset, u_total, plan_info = 0, None, None
    if scaling == "strong":
        # ONE global rating set; every rank derives the same plan from the same degrees and keeps its user range.  With
        # the device generator the global set exists only on each rank's GPU while it is being made (1 B ratings for
        # cfg4_powerlaw: seconds there, half an hour on a host) and only the rank's shard comes down to the host arrays.
        plan = {}

        def shard(deg_u, deg_i):
            ub, ip, pinfo = mfsgd_amd.dsgd_plan_ex(deg_u, deg_i, vworld, ppr, synth.WORKLOADS[args.workload]["k"], args.plan_crit)
            plan.update(ub=ub, ip=ip, info=pinfo, nnz=int(deg_i.sum()), mi=int(deg_i.max()), mu=int(deg_u.max()))
            return int(ub[vrank]), int(ub[vrank + 1])

        w = synth.workload(args.workload, args.scale, generator=generator, user_range=shard, log=log if rank == 0 else None)
        nnz_global, max_item_degree, max_user_degree = plan["nnz"], plan["mi"], plan["mu"]
        item_part, plan_info = plan["ip"], plan["info"]
        lo, hi = int(plan["ub"][vrank]), int(plan["ub"][vrank + 1])
        u_total, u_offset = w["U"], lo
        w = dict(w, U=hi - lo, u=(w["u"] - lo).astype(np.int32), U_global=u_total)
    else:
        # weak scaling: every rank brings its own 138,493 users and 20 M ratings, and the item
        # catalogue grows with the rank count (N x 26,744 items), so that the longest per-row
        # dependency chain a rank has to serialise stays what it is at N = 1
        w = synth.workload(args.workload, args.scale, seed_offset=1000 * vrank, item_mult=vworld, generator=generator)
        nnz_global = w["nnz"] * vworld
        u_total, u_offset = w["U"] * vworld, vrank * w["U"]
        # the longest chain of dependent updates on one row: what a sequentially consistent epoch cannot go below
        max_item_degree = int(np.bincount(w["i"], minlength=w["I"]).max())
        max_user_degree = int(np.bincount(w["u"], minlength=w["U"]).max())
        item_part = None
        if n_parts:
            # the global partitioner: item partitions balanced by the GLOBAL rating counts (users are already
            # dealt out evenly: every rank brought the same number of ratings)
            deg_i = torch.from_numpy(np.bincount(w["i"], minlength=w["I"]).astype(np.int64))
            if dist is not None:
                dist.all_reduce(deg_i)
            elif emu:
                deg_i = deg_i * emu  # one rank of the job: the other ranks' histograms look like this one's
            _, item_part, plan_info = mfsgd_amd.dsgd_plan_ex(np.ones(vworld, np.int64), deg_i.numpy(), vworld, ppr, w["k"], args.plan_crit)
    if rank == 0:
        log(f"generated {w['nnz']} ratings ({w['U']} x {w['I']}, {w['dist']}, scaling {scaling}) in {time.time() - t0:.1f} s")
    k, nnz = w["k"], w["nnz"]
    flags = (_lib.FLAG_NO_GRAPH if args.no_graph else 0) | (_lib.FLAG_ROUND_LAUNCH if args.round_launch else 0)

    m = mfsgd_amd.MatrixFactorizationSGD(w["U"], w["I"], k, LR, LAM, SEED, device=local_rank, blocks=args.blocks,
                                         waves=args.waves, n_parts=n_parts,
                                         host_threads=host_threads(), flags=flags)
    t0 = time.time()
    if item_part is not None:
        m.set_item_partition(item_part)
    m.set_ratings(w["u"], w["i"], w["r"])
    infos = [m.schedule_info(p) for p in range(max(1, n_parts))]
    if rank == 0:
        i0 = infos[0]
        log(f"schedule built in {time.time() - t0:.1f} s: B={i0['blocks']} W={i0['waves']} G={i0['slots']} "
            f"lds={i0['lds_bytes']} steps={sum(i['total_steps'] for i in infos)} rows={sum(i['total_rows'] for i in infos)}")

    launches_per_epoch = sum((i["rounds"] if args.round_launch else 1) for i in infos if i["nnz"] > 0)

    def _barrier():
        if dist is not None:
            dist.barrier()

    ring_stats = emulation = None
    if vworld == 1 and not selftest:
        m.init_factors(SEED)
        rmse0 = m.rmse()  # also moves everything to the device
        for _ in range(args.warmup):
            m.fit(1, rmse=False)
        torch.cuda.synchronize()
        t_wall0 = time.perf_counter()
        dev_ms, launches = m.train_timed(args.steps)
        torch.cuda.synchronize()
        wall_s = time.perf_counter() - t_wall0
        elapsed_s = max(wall_s, dev_ms / 1e3)
        rmse1 = m.rmse()
    elif native:
        # the ring under the C-ABI (csrc/dsgd.cpp): RCCL send / recv on its own stream.  It comes up on every rank or the
        # bench fails -- a line produced by another transport would not measure the product
        from mfsgd_amd.dsgd import NativeDSGD

        obj = [NativeDSGD.unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(obj, src=0)
        m.init_p_offset(SEED, u_offset)
        d = NativeDSGD(m, rank, world, obj[0])
        d.init_q(SEED, u_total)
        rmse0 = d.rmse()
        d.train(args.warmup, rmse=False)
        torch.cuda.synchronize()
        _barrier()
        torch.cuda.synchronize()
        t_wall0 = time.perf_counter()
        d.train(args.steps, rmse=False)  # returns when the compute and the communication stream are idle
        torch.cuda.synchronize()
        _barrier()
        torch.cuda.synchronize()
        wall_s = time.perf_counter() - t_wall0
        tmax = torch.tensor([wall_s], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_s = float(tmax[0])
        dev_ms = elapsed_s * 1e3
        launches = launches_per_epoch * args.steps
        rmse1 = d.rmse()
        ring_stats = d.stats()
        d.close()
    else:
        # --emulate-world: ONE rank of an `emu`-rank job on this GPU, no communication.  Every item partition of the rank
        # is timed on its own (HIP events round `steps` launches on the stream they are launched on).  Two figures come
        # out of it: the SUM over the partitions -- what this rank computes per epoch -- and emu x the SLOWEST partition:
        # the ring's pace.  A Q block is trained by one rank after the other, so block p needs emu x t_p per epoch however
        # the ranks overlap, and with one block per rank every sub-epoch lasts as long as its slowest partition.  The
        # line reports the ring's pace (`value`, ms_per_step); the sum is in "emulation".
        m.init_p_offset(SEED, u_offset)
        stream = torch.cuda.current_stream(dev)
        part_ms = []
        sse0 = sse1 = 0.0
        for p in range(n_parts):
            blk = torch.from_numpy(m.part_init_q(p, SEED, u_total)).to(dev)
            if infos[p]["nnz"] == 0:
                part_ms.append(0.0)
                continue
            sse0 += m.part_sse(p, blk.data_ptr(), stream.cuda_stream)
            for _ in range(args.warmup):
                m.part_train(p, blk.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.steps):
                m.part_train(p, blk.data_ptr(), stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            part_ms.append(e0.elapsed_time(e1) / args.steps)
            sse1 += m.part_sse(p, blk.data_ptr(), stream.cuda_stream)
            del blk
        rmse0, rmse1 = (sse0 / max(1, nnz)) ** 0.5, (sse1 / max(1, nnz)) ** 0.5  # this rank's ratings only (partitions trained apart)
        group_ms = [sum(part_ms[g * ppr:(g + 1) * ppr]) for g in range(vworld)]  # a rank holds a group of ppr partitions at a time
        ring_ms = vworld * max(group_ms)
        emulation = {"partition_ms": part_ms, "sum_ms": sum(part_ms), "slowest_group_ms": max(group_ms), "ring_pace_ms": ring_ms,
                     "note": "ring_pace_ms = world x the slowest group of partitions: what an epoch takes when every block has to be "
                             "trained by one rank after the other (no communication time included); sum_ms = this rank's own compute"}
        wall_s = elapsed_s = ring_ms * args.steps / 1e3
        dev_ms = elapsed_s * 1e3
        launches = launches_per_epoch * args.steps

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        m.close()
        return

    # whole-job updates: all ranks' ratings (an emulated rank reports its own share only)
    total_updates = (nnz if emu else (nnz_global if world > 1 else nnz)) * args.steps
    value = total_updates / elapsed_s
    # ---- roofline of the dominant kernel ------------------------------------------------------------------------
    # algorithmic bytes per update: 12 (COO triple) + 4 rows x 4k bytes, no reuse credited
    bytes_per_update = 16 * k + 12
    # N > 1: one launch per partition and epoch, one after another (an emulated rank: its own launches, not the ring's pace)
    avg_launch_s = ((emulation["sum_ms"] * args.steps if emulation else dev_ms) / 1e3) / max(1, launches)
    units_per_launch = nnz * args.steps / max(1, launches)  # this rank's
    achieved_gbs = units_per_launch * bytes_per_update / avg_launch_s / 1e9  # per GPU
    traffic, traffic_source = read_traffic(args.workload, k, nnz) if (world == 1 and not emu and not selftest) else (None, None)
    roofline = {
        # "hbm" is the roofline this path is priced against (gather + axpy, 0.74 flop/B).  `achieved` is
        # ALGORITHMIC bytes / time, a throughput yardstick: what actually limits a skewed epoch is the longest
        # chain of dependent updates on one row (`limiter`), and the bytes really moved are `traffic`.
        "bound": "hbm",
        "limiter": "dependency chain of the heaviest row (%d sequential updates; sum_round_steps %d)"
                   % (max(max_item_degree, max_user_degree), sum(i["sum_round_steps"] for i in infos)),
        "kernel": ("mfsgd::cell_kernel<L,W,train> (one launch per round)" if args.round_launch
                   else "mfsgd::epoch_kernel<L,W> (persistent: one launch per epoch and item partition)"),
        "achieved": achieved_gbs,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved_gbs / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "bytes_per_update": bytes_per_update,
        "updates_per_launch": units_per_launch,
        "avg_launch_us": avg_launch_s * 1e6,
        "launches": launches,
    }
    if world == 1:
        par = f"selftest-1rank-{selftest}parts-native-rccl-self-ring" if selftest else (f"one-rank-of-dsgd{emu}-no-communication" if emu else "single")
    else:
        par = f"dsgd{world}" + ("-native-shm-ring-one-gpu-rehearsal" if args.rehearse_on_one_gpu else "-native-rccl-ring")
    out = {
        "metric": "rating-updates/sec",
        "value": value,
        "unit": "updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_s * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload} (MovieLens-20M shape, Zipf-Mandelbrot degrees)" if args.workload == "cfg2_ml20m" else args.workload,
            "workload_rev": WORKLOAD_REV,
            "generator": dict({x: v for x, v in synth.WORKLOADS[args.workload].items() if x not in ("U", "I", "nnz", "k")}, made_on=generator),
            "users_per_gpu": w["U"], "items": w["I"], "nnz_per_gpu": nnz, "nnz_global": nnz_global, "users_global": u_total, "k": k,
            "lr": LR, "lambda": LAM, "scale": args.scale,
            "max_item_degree": max_item_degree, "max_user_degree": max_user_degree,
            "sum_round_steps": sum(i["sum_round_steps"] for i in infos),
            "blocks": infos[0]["blocks"], "waves": infos[0]["waves"],
            "parts_per_rank": ppr, "emulated_world": emu, "emulated_rank": vrank if emu else None,
            "parallelism": par,
            # the global plan: sum over the partitions of their heaviest item's (global) rating count -- what a rank's
            # epoch serialises, times 1 / world --, chain-critical items, partitions they were packed into, threshold
            "plan": plan_info,
        },
        "rmse_before": rmse0,
        "rmse_after": rmse1,
        "device_ms": dev_ms,
        "wall_ms": wall_s * 1e3,
        "roofline": roofline,
    }
    if ring_stats is not None:
        out["ring"] = ring_stats
    if emulation is not None:
        out["emulation"] = emulation
    if world == 1 and not emu and not selftest:
        # What a caller of train(u, i, r, 10) with HOST arrays sees end to end on a fresh handle: hashing and
        # uploading the triples, building the schedule (device ingest + device packer), seeding the factors,
        # 10 epochs AND the RMSE pass after each (the Java train() returns per-epoch RMSE).  Never `value`.
        e2e = []
        for _ in range(2):  # the first call also pays for one-time kernel loading; the second is the steady state
            with mfsgd_amd.MatrixFactorizationSGD(w["U"], w["I"], k, LR, LAM, SEED, device=local_rank, blocks=args.blocks,
                                                  waves=args.waves, host_threads=host_threads(), flags=flags) as m2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m2.train(w["u"], w["i"], w["r"], 10)
                e2e.append(time.perf_counter() - t0)
        out["end_to_end"] = {
            "value": nnz * 10 / e2e[1], "unit": "updates/s", "epochs": 10, "seconds": e2e[1], "first_call_seconds": e2e[0],
            "includes": "host arrays -> set_ratings (hash, upload, device ingest + packer) -> init_factors -> 10 x (epoch + RMSE pass)",
        }
    if world == 1 and not emu and not selftest and not args.no_cpu_baseline:
        log("timing the CPU baselines (oracle, multithreaded: the checker's arithmetic, then the textbook loop) ...")
        out["cpu_baseline"] = cpu_baseline(w, m)
    m.close()
    print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
