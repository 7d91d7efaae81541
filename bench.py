#!/usr/bin/env python3
"""bench.py -- scan-pair registrations/sec on synthetic 1M-point indoor scan pairs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full registration(T, target, source) (code/PLADE/plade.h:58: plane extraction +
registration) of one synthetic pair of config `Synthetic 1M-pt indoor scan pair, ~30 planes`
(BASELINE.json configs[2]) with both clouds in page-locked HOST memory: the upload (H2D), the SoA conversion and
the bounding boxes of every step are inside the timed region.  One process per GPU; scan pairs
are independent (batch mode, code/PLADE/main.cpp:97-158), so every rank registers its own pairs with no
data-path collective and the per-pair 4x4 results are gathered to rank 0 over RCCL at the end
(weak scaling: work per GPU is fixed).  Several registrations are in flight per GPU (one plade_ctx + host thread
each); `value` is the steady-state rate of that pipeline over the timed steps that complete after the last lead-in
(warm-up) step: steps in flight / mean time one of those steps occupied its worker (the bare first-to-last-completion
window is reported beside it).  Fewer than 16 rounds of the workers sample a pipeline badly (20 steps = 2.5 rounds of
8 read +-9 % from run to run), so max(K, 16 x in-flight) steps are timed; `ms_per_step` and `value` are per-step
figures, `pipeline.timed_steps` says how many steps they come from, `steps` echoes the K that was asked for.
Rank 0 prints ONE JSON line.  `resident_rank0` is the same pipeline on clouds already resident in HBM.

Extra objects on the line:
  roofline     -- the dominant kernel of the step, timed live with HIP events on the stream the kernel is
                  launched on (one extra profiled step after the timed region), achieved = algorithmic
                  bytes per launch / average launch duration, peak = 8 TB/s HBM3E.
  cpu_baseline -- the CPU path on this box's host cores, single thread like the reference: plane
                  extraction by the reference's own Schnabel RANSAC compiled from its sources
                  (oracle/_ref) when that library is present, everything after it by the oracle's
                  restatement (oracle/plade_oracle.cpp); bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


KERNEL_SYMBOLS = {"score_mark": "k_r_mark(", "score_multi": "k_r_rescore(", "overlap": "k_overlap(", "knn_spacing": "k_knn_grid(",
                  "pen_walk": "k_pen_walk(", "cluster_edges": "k_cluster_edges("}


def pmc_traffic(tag):
    """HBM bytes per REGISTRATION of the roofline kernel (all its launches) from the committed PMC summary (separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; tools/summarize_profiles.py).  None if no summary exists."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")))
    if not files:
        return None, None
    sym = KERNEL_SYMBOLS.get(tag, tag + "(")
    with open(files[-1]) as f:
        rows = list(csv.DictReader(f))
    regs = next((int(r["launches"]) for r in rows if "k_morton(" in r["kernel"]), 0)   # one launch per registration
    for r in rows:
        if sym in r["kernel"]:
            rd = float(r["hbm_read_bytes(FETCH_SIZE*1024*2)"])
            wr = float(r["hbm_write_bytes(WRITE_SIZE*1024)"])
            # counter average over all launches of the kernel x launches per registration = bytes per registration
            return (rd + wr) * int(r["launches"]) / max(regs, 1), os.path.relpath(files[-1], ROOT)
    return None, None


def rocprof_stats(tag):
    """(average launch duration in us, share of the GPU time, file) of the roofline kernel in the committed
    `rocprofv3 --kernel-trace --stats` summary of this command (profiles/*_kernel_stats.csv), plus the three kernels with
    the most GPU time there: the live HIP-event figure on the line must agree with this average."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")))
    if not files:
        return None
    sym = KERNEL_SYMBOLS.get(tag, tag + "(")
    rows = list(csv.DictReader(open(files[-1])))
    total = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    def short(name):
        for junk in ("void ", "plade::", "(anonymous namespace)::"):
            name = name.replace(junk, "")
        return name.split("(")[0][:48]
    top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:5]
    out = {"file": os.path.relpath(files[-1], ROOT),
           "top_by_gpu_time": [{"kernel": short(r["Name"]), "share": round(float(r["TotalDurationNs"]) / total, 4),
                                "avg_us": round(float(r["AverageNs"]) / 1e3, 2)} for r in top]}
    regs = next((int(r["Calls"]) for r in rows if "k_morton(" in r["Name"]), 0)   # one launch per registration
    out["registrations_profiled"] = regs
    out["kernels_per_registration"] = round(sum(int(r["Calls"]) for r in rows if "rocclr" not in r["Name"]) / max(regs, 1), 1)
    out["copies_and_fills_per_registration"] = round(sum(int(r["Calls"]) for r in rows if "rocclr" in r["Name"]) / max(regs, 1), 1)
    out["gpu_ms_per_registration"] = round(total / 1e6 / max(regs, 1), 3)
    for r in rows:
        if sym in r["Name"]:
            out.update({"avg_launch_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]),
                        "launches_per_registration": int(r["Calls"]) / max(regs, 1),
                        "share_of_gpu_time": float(r["TotalDurationNs"]) / total})
            break
    return out


def cpu_baseline(n_points, pairs, min_s=10.0, budget_s=25.0):
    """Single-thread CPU registrations/sec on a bounded sample (rank 0, N = 1 only): the bench pairs are registered
    in turn (the reference's RANSAC re-seeded every time, as its time() seed would be) until min_s of CPU work."""
    from oracle.oracle import Oracle, Reference, have_reference
    orc = Oracle()
    ref = Reference() if have_reference() else None

    def orient(cloud, pl):  # same orientation rule the GPU path applies (params.orient_normals)
        co = pl[0].copy()
        for i in range(len(co)):
            ids = pl[2][pl[1][i]:pl[1][i + 1]]
            if cloud[ids, 3:].astype(np.float64).mean(0) @ co[i, :3] < 0:
                co[i] = -co[i]
        return co, pl[1], pl[2]

    def extract(cloud, seed):  # extract() of code/PLADE/plade.cpp:602-635 on the reference's RANSAC
        ms, trials = 10000, 1
        pl = ref.ransac_detect(cloud, ms, fake_time=seed)
        if len(pl[0]) > 40:
            order = np.argsort(-np.diff(pl[1]), kind="stable")[:40]
            parts = [pl[2][pl[1][o]:pl[1][o + 1]] for o in order]
            pl = (pl[0][order], np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int32),
                  np.concatenate(parts))
        ms //= 2
        while len(pl[0]) < 10 and trials < 10 and ms >= 200:
            pl = ref.ransac_detect(cloud, ms, fake_time=seed)
            ms //= 2
            trials += 1
        return orient(cloud, pl)

    done, t_total, ok_all = 0, 0.0, True
    t_extract = 0.0
    while t_total < min_s:
        tg, sr, planes = pairs[done % len(pairs)]
        t0 = time.perf_counter()
        if ref is not None:
            tp, sp = extract(tg, 2 * done + 1), extract(sr, 2 * done + 2)
        else:
            tp, sp = planes  # planes handed over from the GPU run: the CPU leg then times the port only
        t1 = time.perf_counter()
        ok, T, _ = orc.registration(tg, sr, tp, sp, voxel_sort_mode=0)
        t2 = time.perf_counter()
        ok_all = ok_all and ok
        t_extract += t1 - t0
        t_total += t2 - t0
        done += 1
        if t_total > budget_s:
            break
    kind = "port"
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    except Exception:
        model = []
    what = ("plane extraction: reference Schnabel RANSAC built from /root/reference sources (oracle/_ref); "
            if ref is not None else "plane extraction: not timed (oracle/_ref absent), planes taken from the GPU run; ")
    return {
        "value": done / t_total if t_total > 0 else None,
        "unit": "registrations/s",
        "cores": 1,
        "host_cpu_model": model[0] if model else None,
        "host_logical_cpus": len(model) or os.cpu_count(),
        "host_cpu_budget": _cpu_budget(),
        "kind": kind,
        "sample": f"{done} registrations of {min(done, len(pairs))} synthetic {n_points}-pt pair(s), {t_total:.1f} s CPU ({t_extract:.1f} s in plane extraction); "
                  + what + "registration stages: oracle restatement; all registrations ok=" + str(ok_all),
    }


def _cpu_budget():
    """CPUs this process may use: the cgroup v2 quota if one is set, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, float(quota) / float(period))
    except Exception:
        pass
    return n


def inflight_for_budget(budget, local_world):
    """Registrations in flight per GPU.  Measured on the MI355X box (sleeping host waits): 5 / 6 / 8 / 10 in flight keep
    2.2 / 2.3 / 2.6 / 2.7 host threads busy for 435 / 447 / 455-467 / 457 reg/s, i.e. busy = 1.3 + 0.17 per registration in
    flight.  A container that runs into its CPU quota loses far more than the last few percent of GPU throughput (round 1:
    287 instead of 400 reg/s under throttling), so the count is lowered until the ranks of the node fit into the quota: 8
    ranks on 16 CPUs run 4 in flight each."""
    per_rank = float(budget) / max(local_world, 1)
    return int(max(2, min(8, (per_rank - 1.3) / 0.17 + 1e-6)))


def _cgroup_throttle():
    """(nr_throttled, throttled_usec) of this container's CPU quota, None where cgroup v2 is not mounted."""
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--pairs", type=int, default=2, help="distinct synthetic pairs per rank (cycled)")
    ap.add_argument("--host-wait", choices=["auto", "spin", "sleep"], default="auto",
                    help="how the host threads wait for the GPU (plade_params.host_wait): spinning waits keep ~1.7 CPUs busy "
                         "per registration in flight, sleeping ones ~0.5 at the same throughput; auto = sleep when more "
                         "than one registration is in flight")
    ap.add_argument("--inflight", type=int, default=0,
                    help="registrations in flight per GPU: independent pairs, one plade_ctx + host thread each "
                         "(a single registration is latency-bound and leaves most of the GPU idle).  0 = 8, or fewer when "
                         "the ranks of this node have to share a small CPU quota (see inflight_for_budget)")
    ap.add_argument("--resident-steps", type=int, default=256,
                    help="steps of the extra resident leg (clouds uploaded once, plade_registration_dev per step); 0 = skip")
    ap.add_argument("--host-steps", type=int, default=0, help=argparse.SUPPRESS)   # round-2 flag, ignored
    ap.add_argument("--no-default-mode", action="store_true", help="skip the orient_normals=0 success-rate leg")
    ap.add_argument("--profiled-steps", type=int, default=8, help="registrations of the roofline leg (HIP events per launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook for boxes with one GPU: PLADE_BENCH_ONE_GPU=1 puts every rank on cuda:0 and exchanges over gloo (RCCL
    # refuses two ranks on one device), so that the N > 1 control flow can be exercised there
    one_gpu = os.environ.get("PLADE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    # torch only where it is needed -- torch.distributed (RCCL) for N > 1.  `import torch` brings the HIP runtime bundled
    # with the wheel (ROCm 7.0) into the process ahead of the system's (7.2), and the library then runs on that one:
    # measured 458 instead of 483 registrations/s at N = 1 (tools/exp_throughput.py, EXP_TORCH=import).  A single-GPU run
    # brackets its timed region with hipDeviceSynchronize through the library instead of device_sync().
    torch = dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo" if one_gpu else "nccl", world_size=world, rank=rank)  # nccl == RCCL on ROCm
        dev = torch.device("cpu") if one_gpu else torch.device("cuda", local_rank)
    else:
        dev = None

    import plade_amd
    from plade_amd.synth import make_pair

    def device_sync():
        if torch is not None:
            torch.cuda.synchronize()
        else:
            plade_amd.device_synchronize(local_rank)

    import threading
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    inflight_auto = args.inflight <= 0
    if inflight_auto:
        args.inflight = inflight_for_budget(_cpu_budget(), local_world)
    M = max(1, args.inflight)
    if args.host_wait == "auto":
        # several registrations in flight: sleeping waits (same throughput, a third of the host CPUs, and no way to run
        # into the container's CPU quota when 8 ranks share a node); one at a time: spin for the lowest latency
        args.host_wait = "sleep" if M > 1 else "spin"
    host_wait = {"spin": 0, "sleep": 1}[args.host_wait]
    ctxs = [plade_amd.Context(local_rank, host_wait=host_wait, orient_normals=1) for _ in range(M)]
    ctx = ctxs[0]
    # synthetic pairs: seeds are global pair ids (batch of independent pairs sharded across ranks); every
    # worker holds its own resident copy so the workers share nothing
    pairs, clouds = [], [[] for _ in range(M)]
    for k in range(args.pairs):
        seed = rank * args.pairs + k
        tg, sr, Tgt = make_pair(args.points, seed=seed)
        pairs.append((tg, sr, Tgt))
        for w in range(M):
            clouds[w].append((ctxs[w].upload(tg), ctxs[w].upload(sr)))

    def step(i, w=0):
        ct, cs = clouds[w][i % len(pairs)]
        ok, T = ctxs[w].registration_dev(ct, cs)
        return ok, T

    # the timed path: registration(T, target, source) of code/PLADE/plade.h:58 on clouds in (page-locked) HOST memory, in
    # batch mode (main.cpp:97-158 loops over pairs): plade_registration_next = plade_registration + the upload of the pair
    # the same context registers next, queued on a stream of its own.  H2D, SoA conversion and bounding boxes of every
    # step are inside the timed region.
    for tg, sr, _ in pairs:
        ctx.pin(tg); ctx.pin(sr)

    def hstep(i, w, nxt):
        tg, sr, _ = pairs[i % len(pairs)]
        ntg, nsr = (pairs[nxt % len(pairs)][0], pairs[nxt % len(pairs)][1]) if nxt is not None else (None, None)
        return ctxs[w].registration_next(tg, sr, ntg, nsr)

    def rstep(i, w, nxt):
        return step(i, w)

    def run_pipeline(fn, lead, count):
        """Steady-state throughput of the M-deep pipeline: worker w takes steps w, w + M, ... of lead + count + M steps
        (the first `lead` fill the pipeline and are untimed, the last M keep it full until the last timed step completes);
        the timed window runs from the completion of step number `lead` to the completion of step number lead + count,
        i.e. EXACTLY `count` completions with the pipeline full on both sides.  Returns (seconds of the window, results of
        the timed steps in step order, seconds from start to the last completion of all lead + count + M steps)."""
        total = lead + count + M
        stamps, out = [0.0] * total, [None] * total

        def work(w):
            for i in range(w, total, M):
                out[i] = fn(i, w, i + M if i + M < total else None)
                stamps[i] = time.perf_counter()
        ths = [threading.Thread(target=work, args=(w,)) for w in range(M)]
        ts = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        done = sorted(stamps)
        window = done[lead + count - 1] - done[lead - 1]
        # the timed steps = the `count` steps that completed inside the window
        order = sorted(range(total), key=lambda i: stamps[i])[lead:lead + count]
        # Each worker completes its steps back to back, so step i occupied its worker from stamps[i - M] to stamps[i]; with
        # M steps always in flight the rate is M / (mean occupancy of the timed steps) (Little's law).  The M workers complete
        # in bursts, so the bare window over few steps (the driver's 20 = 2.5 rounds of 8) swings by +-15 % with where the
        # bursts fall; the occupancy form times the same `count` steps without that edge effect and equals count / window
        # over long runs (both are reported).
        occupancy = sum(stamps[i] - (stamps[i - M] if i >= M else ts) for i in order) / count
        run_pipeline.last_window = window
        return count * occupancy / M, [out[i] for i in sorted(order)], sorted(order), done[-1] - ts

    # warm-up: every worker (context) registers every distinct pair once through BOTH entry points, so that no first-use
    # allocation or graph capture falls into the timed region; the W warm-up steps the driver asks for are the lead-in
    # of the pipelined run below (at least one per context in flight)
    def warm_worker(w):
        for r in range(len(pairs)):
            step(r, w)
            hstep(r, w, None)
    wths = [threading.Thread(target=warm_worker, args=(w,)) for w in range(M)]
    for t in wths:
        t.start()
    for t in wths:
        t.join()
    # lead-in: the W warm-up steps the driver asks for, and at least 8 rounds of the M workers -- they start in lock step
    # (all in the same stage at once, competing for the same units) and need a few rounds to spread out over the
    # stages; the first two rounds run 20 % slower than the steady state the metric is quoted on
    lead = max(args.warmup, int(os.environ.get("BENCH_LEAD_ROUNDS", "8")) * M)
    device_sync()
    if world > 1:
        dist.barrier()
    device_sync()
    cpu0, thr0 = time.process_time(), _cgroup_throttle()
    t_begin = time.perf_counter()
    # K = --steps timed steps; a pipeline of M workers is not sampled fairly by fewer than ~16 rounds (the driver's 20 steps are
    # 2.5 rounds of 8: +-9 % from run to run by either estimator), so at least 16 * M steps are timed and the PER-STEP time of
    # those is what the line reports (`pipeline.timed_steps` says how many; ms_per_step and value are per-step quantities)
    n_timed = max(args.steps, 16 * M)
    elapsed, timed, timed_ids, span = run_pipeline(hstep, lead, n_timed)
    host_window = run_pipeline.last_window
    cpu1, thr1 = time.process_time(), _cgroup_throttle()
    oks = [bool(r[0]) for r in timed]
    results = [r[1] for r in timed]
    n_ok = sum(oks)
    # gather the per-pair 4x4 results on rank 0 in input order (the only exchange the path needs;
    # plade_amd/batch.py, covered on CPU by tests/test_distributed_gloo.py with gloo)
    from plade_amd.batch import gather_results
    all_T, all_ok = gather_results(np.stack(results), np.array(oks, bool), world * n_timed, rank, world,
                                   device=dev if world > 1 else None)
    device_sync()
    if world > 1:
        dist.barrier()
    device_sync()
    bracketed = time.perf_counter() - t_begin
    total_ok = n_ok
    if world > 1:
        tmax = torch.tensor([elapsed, bracketed], dtype=torch.float64, device=dev)
        okt = torch.tensor([n_ok], dtype=torch.int64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(okt, op=dist.ReduceOp.SUM)
        elapsed, bracketed = float(tmax[0].item()), float(tmax[1].item())
        total_ok = int(okt.item())

    # every registration of the same pair, whichever context ran it, must return the same bits
    ref_result = {}
    for k in range(len(results)):
        ref_result.setdefault(timed_ids[k] % len(pairs), results[k])
    identical = all(np.array_equal(results[k], ref_result[timed_ids[k] % len(pairs)]) for k in range(len(results)))
    # accuracy of the timed registrations on this rank vs the generator's ground truth
    errs = [float(np.linalg.norm(results[k].astype(np.float64) - pairs[timed_ids[k] % len(pairs)][2])) for k in range(len(results))]

    # ---- resident leg (clouds already in HBM, plade_registration_dev): the same pipeline without the uploads, reported
    #      next to `value`
    resident_leg = None
    if args.resident_steps > 0:
        r_el, r_res, r_ids, _ = run_pipeline(rstep, M, args.resident_steps)
        device_sync()
        same = all(np.array_equal(r_res[k][1], ref_result.get(r_ids[k] % len(pairs), r_res[k][1])) for k in range(len(r_res)))
        resident_leg = {"value": args.resident_steps / r_el, "unit": "registrations/s (this rank)", "steps": args.resident_steps,
                        "ms_per_step": r_el / args.resident_steps * 1e3, "identical_to_host_cloud_results": bool(same),
                        "note": "clouds resident in HBM (plade_cloud_upload once, plade_registration_dev per step): no H2D, no SoA "
                                "conversion, no bounding box in the step"}
    mb = sum(tg.nbytes + sr.nbytes for tg, sr, _ in pairs) / len(pairs) / 1e6
    host_leg = {"h2d_MB_per_step": mb, "pcie_GB_per_s": mb * 1e-3 * n_timed / elapsed,
                "bracketed_value": (lead + n_timed + M) / bracketed if bracketed > 0 else None,
                "bracketed_note": "all lead-in + timed + tail steps of this rank over the barrier-to-barrier time (fill and drain of "
                                  "the pipeline and the result gather inside)"}

    # ---- the library's shipped default on the same scenes: orient_normals = 0 (reference behaviour)
    default_mode = None
    if rank == 0 and not args.no_default_mode:
        dctx = plade_amd.Context(local_rank, host_wait=host_wait)          # plade_default_params: orient_normals = 0
        good, tried = 0, 0
        for k, (tg, sr, Tgt) in enumerate(pairs):
            ok, T = dctx.registration(tg, sr)
            tried += 1
            good += bool(ok and np.linalg.norm(T.astype(np.float64) - Tgt) < 5e-2)
        small = 0
        for sd in range(8):
            tg, sr, Tgt = make_pair(200000, seed=1000 + sd)
            ok, T = dctx.registration(tg, sr)
            small += bool(ok and np.linalg.norm(T.astype(np.float64) - Tgt) < 5e-2)
        dctx.close()
        default_mode = {"orient_normals": 0, "bench_pairs_registered": good, "bench_pairs": tried,
                        "extra_200k_pairs_registered": small, "extra_200k_pairs": 8,
                        "criterion": "ok and |T - T_ground_truth|_F < 5e-2",
                        "note": "the reference leaves plane normals unoriented (plane_extraction.cpp:43-58 is a NaN no-op); a "
                                "Manhattan scene then registers only when three independent sign bits agree (about 1 in 8)"}

    for tg, sr, _ in pairs:
        ctx.unpin(tg); ctx.unpin(sr)

    # ---- roofline leg: one extra profiled step (HIP events on the ctx stream around every launch) ----
    roofline = None
    stage = {}
    latency_ms = None
    if rank == 0:
        lat = []
        ctx.set_params(host_wait=0)   # alone, a spinning wait is the faster one
        for i in range(4):   # one registration at a time: the latency figure
            t1 = time.perf_counter()
            step(i)
            lat.append(time.perf_counter() - t1)
        latency_ms = min(lat) * 1e3
        # Profiled steps (HIP events on the launch stream around every launch of the scan kernels) on context 0 WHILE the
        # other contexts keep registering, i.e. under the load of the timed region: the average launch duration must be
        # the one `rocprofv3 --kernel-trace --stats` reports for this command (profiles/), not that of an idle GPU.
        ctx.set_params(dump=2, host_wait=host_wait)
        stop = threading.Event()

        def background(w):
            i = w
            while not stop.is_set():
                step(i, w)
                i += 1
        bths = [threading.Thread(target=background, args=(w,)) for w in range(1, M)]
        for t in bths:
            t.start()
        st = {}
        for i in range(args.profiled_steps):
            step(i)
            for k, v in ctx.stats().items():
                if k.startswith(("k_", "bytes_")):
                    st[k] = st.get(k, 0.0) + v
                else:
                    st[k] = v
        stop.set()
        for t in bths:
            t.join()
        for k in list(st):
            if k.startswith("bytes_"):
                st[k] /= args.profiled_steps
        ctx.set_params(dump=0, host_wait=host_wait)
        kernels = sorted({k[2:-8] for k in st if k.startswith("k_") and k.endswith("_seconds") and not k.endswith("_clock_seconds")})
        best = None
        for name in kernels:
            secs, nl, by = st[f"k_{name}_seconds"], st[f"k_{name}_launches"], st[f"k_{name}_bytes"]
            stage[name] = {"seconds": secs / args.profiled_steps, "launches": nl / args.profiled_steps,
                           "GB/s": (by / secs / 1e9) if secs > 0 else None}
            if st.get(f"k_{name}_clock_seconds"):   # the kernel's own clock (see roofline.measured)
                cs, cb = st[f"k_{name}_clock_seconds"], st[f"k_{name}_clock_bytes"]
                stage[name].update({"clock_seconds": cs / args.profiled_steps, "clock_GB/s": cb / cs / 1e9})
            # the roofline kernel is the one with the most GPU time among the HBM-streaming kernels (those
            # with an algorithmic byte count); latency-bound kernels are listed for reference
            if by > 0 and (best is None or secs > st[f"k_{best}_seconds"]):
                best = name
        if best is not None:
            secs, nl, by = st[f"k_{best}_seconds"], st[f"k_{best}_launches"], st[f"k_{best}_bytes"]
            # launch duration: the kernel's own wall-clock stamps (first wavefront in .. last wavefront out, the quantity
            # rocprofv3's kernel trace reports); the HIP events around the launch are listed next to it -- with other
            # registrations in flight they also contain the other streams' kernels that ran on the same hardware queue
            c_secs, c_nl, c_by = st.get(f"k_{best}_clock_seconds"), st.get(f"k_{best}_clock_launches"), st.get(f"k_{best}_clock_bytes")
            if c_secs and c_nl:
                achieved, avg_us, how = c_by / c_secs / 1e9, c_secs / c_nl * 1e6, "device wall clock inside the kernel (min start / max end over its wavefronts)"
            else:
                achieved, avg_us, how = by / secs / 1e9, secs / nl * 1e6, "HIP events on the launch stream"
            # rocprofv3 (and the counter passes) average over ALL launches of the kernel, including those of the fixed launch
            # sequence that find nothing to do and return at once (they move no bytes); the summaries are therefore compared
            # per registration: counter bytes of all launches of a registration / its working launches = per working launch
            traffic_reg, traffic_src = pmc_traffic(best)
            idle = st.get(f"k_{best}_idle_launches", 0.0)
            work_per_step = nl / args.profiled_steps
            traffic = traffic_reg / work_per_step if traffic_reg is not None else None
            roofline = {"bound": "hbm", "kernel": best, "kernel_symbol": KERNEL_SYMBOLS.get(best, best).rstrip("("),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "launches_per_step": nl / args.profiled_steps, "avg_launch_us": avg_us,
                        "avg_launch_us_hip_events": secs / nl * 1e6,
                        "frac_at_hip_event_average": by / secs / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": by / nl,
                        "idle_launches_per_step": idle / args.profiled_steps,
                        "algorithmic_bytes_per_step": by / args.profiled_steps,
                        "traffic_per_step": traffic_reg,
                        "measured": f"{how}; {args.profiled_steps} profiled registrations with "
                                    f"{M - 1} other registrations in flight (the load of the timed region)",
                        "why_this_kernel": "largest mover of HBM bytes of the step (the K1 scoring scan, SURVEY.md 8d: 28 B per point and "
                                           "launch); kernels above it in GPU time (rocprof.top_by_gpu_time) are LDS / latency bound "
                                           "and have no HBM figure"}
            rp = rocprof_stats(best)
            if rp is not None:
                roofline["rocprof"] = rp
                if rp.get("avg_launch_us"):
                    # same population as the summary: the step's bytes over ALL launches of a registration there (working +
                    # idle), at the summary's average duration
                    per_launch = (by / args.profiled_steps) / max(rp.get("launches_per_registration", work_per_step), 1e-9)
                    roofline["rocprof"]["algorithmic_bytes_per_launch_all"] = per_launch
                    roofline["rocprof"]["frac_at_rocprof_average"] = per_launch / (rp["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        stage_times = {k: v for k, v in st.items() if k.startswith("t_")}
        b_total = st.get("bytes_ransac", 0.0) + st.get("bytes_voxel", 0.0) + st.get("bytes_verify", 0.0)
        if roofline is not None:
            # B_total / t_registration against the HBM peak (whole-step figure)
            # at the measured throughput (several registrations in flight) and for one registration alone
            roofline["step_algorithmic_bytes"] = b_total
            roofline["step_frac_of_hbm_peak"] = b_total / (elapsed / n_timed) / 1e9 / HBM_PEAK_GBS
            roofline["single_registration_frac_of_hbm_peak"] = b_total / (latency_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            sample = []
            for (tg, sr, Tgt) in pairs[:3]:
                sample.append((tg, sr, None))
            cpu = cpu_baseline(args.points, sample)
        except Exception as e:  # the oracle is test infrastructure: its absence must not hide the GPU number
            cpu = {"value": None, "unit": "registrations/s", "cores": 1, "kind": "port", "sample": f"unavailable: {e}"}

    if rank == 0:
        total = world * n_timed
        line = {
            "metric": "scan-pair registrations/sec, 1M-pt synthetic pairs",
            "value": total / elapsed,
            "unit": "registrations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / n_timed * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Synthetic {args.points}-pt indoor scan pair, ~30 planes (BASELINE configs[2]); "
                                   "full registration(T,target,source) of plade.h:58 = plane extraction + registration on clouds in "
                                   "page-locked HOST memory, batch mode (plade_registration_next: H2D + SoA conversion + bounding boxes of "
                                   "every step inside the timed region, the next pair's upload queued under the current pair's kernels); "
                                   "timed window = EXACTLY `steps` completions of the full pipeline (steady state, SURVEY 8d); "
                                   "plade_params.orient_normals=1 (planes oriented like their inliers' normals: the generator's "
                                   "Manhattan scenes need it, DESIGN.md section 2); CPU baseline applies the same rule",
                       "points_per_cloud": args.points, "pairs_per_rank": args.pairs,
                       "registrations_in_flight_per_gpu": M, "inflight_chosen_from_cpu_quota": inflight_auto, "host_wait": args.host_wait,
                       "parallelism": f"independent pairs sharded over {world} GPU(s), {M} in flight per GPU"},
            "single_registration_latency_ms": latency_ms,
            "registrations_timed": total,
            "registrations_ok": total_ok,
            "results_bit_identical_per_pair_rank0": bool(identical),
            "max_frobenius_vs_ground_truth_rank0": max(errs) if errs else None,
            "host_rank0": {"cpu_seconds_per_step": (cpu1 - cpu0) / (lead + n_timed + M),
                           "busy_host_threads_avg": (cpu1 - cpu0) / max(span, 1e-9),
                           "cpu_budget": _cpu_budget(),
                           "cgroup_throttled_periods": (thr1[0] - thr0[0]) if thr0 and thr1 else None,
                           "cgroup_throttled_usec": (thr1[1] - thr0[1]) if thr0 and thr1 else None},
            "host_buffers_rank0": host_leg,
            "resident_rank0": resident_leg,
            "default_mode_rank0": default_mode,
            "pipeline": {"lead_in_steps": lead, "timed_steps": n_timed, "requested_steps": args.steps, "tail_steps": M,
                         "timing": "the `timed_steps` = max(steps, 16 x in flight) completions after completion #lead_in, per rank, MAX over ranks: ms_per_step = mean "
                                   "time a timed step occupied its worker / steps in flight (Little's law; = window / steps over long "
                                   "runs, without the burst edge effect over few steps); barrier + device_sync() before the first and "
                                   "after the last step of the run",
                         "window_value_rank0": n_timed / host_window if host_window > 0 else None,
                         "window_note": "steps / (completion #(lead_in + steps) - completion #lead_in) on rank 0"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "stage_seconds_profiled_step": stage_times,
            "kernels_profiled_step": stage,
        }
        if cpu and cpu.get("value"):
            line["speedup_vs_cpu_baseline"] = line["value"] / cpu["value"]
        print(json.dumps(line))
    for w in range(M):
        for ct, cs in clouds[w]:
            ct.free(); cs.free()
        ctxs[w].close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
