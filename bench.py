#!/usr/bin/env python3
"""bench.py -- scan-pair registrations/sec on synthetic 1M-point indoor scan pairs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full registration(T, target, source) (code/PLADE/plade.h:58: plane extraction +
registration) of one synthetic pair of config `Synthetic 1M-pt indoor scan pair, ~30 planes`
(BASELINE.json configs[2]) with both clouds in page-locked HOST memory: the upload (H2D), the SoA conversion and
the bounding boxes of every step are inside the timed region.  One process per GPU; scan pairs
are independent (batch mode, code/PLADE/main.cpp:97-158), so every rank registers its own pairs with no
data-path collective and the per-pair 4x4 results are gathered to rank 0 over the ranks' loopback rendezvous at the end
(weak scaling: work per GPU is fixed).  Several GROUPS of pairs are in flight per GPU (one plade_ctx + host thread per
group, --group consecutive pairs of the batch per plade_registration_pairs call: the plane extraction of a group's clouds
is one launch sequence); `value` = timed steps / the window between the completion of the last lead-in group and the
completion of the last timed group, i.e. exactly `steps` registrations complete inside the window with the pipeline full
on both sides.  A pipeline of M x S registrations in flight is not sampled fairly by a handful of steps (the driver's 20
are less than one round of 32), so max(K, 32 x registrations in flight) steps are timed, in whole groups: `steps`
on the line is the number really timed, `requested_steps` echoes K; the occupancy estimator of round 3 is reported
beside it (`pipeline.occupancy_value`).
Rank 0 prints ONE JSON line.  `resident_rank0` is the same pipeline on clouds already resident in HBM.

Extra objects on the line:
  roofline     -- the dominant kernel of the step, timed live after the timed region under the same load, in two legs:
                  HIP events on the launch stream around every launch (the loop launched kernel by kernel), and
                  the kernel's own device clock with the loop launched as in the timed region (one hipGraph per
                  iteration; `launch_path`): achieved = algorithmic bytes per launch / average
                  launch duration of the graph leg, peak = 8 TB/s HBM3E.
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


KERNEL_SYMBOLS = {"score_mark": "k_r_mark(", "score_multi": "k_r_rescore(", "overlap": "k_overlap", "knn_spacing": "k_knn_grid",
                  "pen_walk": "k_pen_walk", "cluster_edges": "k_cluster_edges", "sort_pass": "k_rs_pass", "sort_hist": "k_rs_histogram"}


def _profile_registrations(csv_path, rows, name_key, calls_key):
    """Registrations behind a committed rocprofv3 summary: the count the profiled command printed (tools/summarize_profiles.py keeps
    it beside the summary, <round>_profile_meta.json); older summaries: one k_overlap launch per registration."""
    meta = csv_path.rsplit("_", 2)[0] + "_profile_meta.json"
    try:
        return int(json.load(open(meta))["registrations"])
    except Exception:
        return next((int(r[calls_key]) for r in rows if "k_overlap(" in r[name_key]), 0)


def pmc_traffic(tag, group=8):
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
    # registrations in the profile = launches of the verification kernel (one per registration, whatever the size of its group:
    # the profiled command also registers a few pairs alone)
    regs = _profile_registrations(files[-1], rows, "kernel", "launches") or group * next((int(r["launches"]) for r in rows if "k_morton(" in r["kernel"]), 0)
    tot, hit = 0.0, False
    for r in rows:      # (a kernel of the lock-step stages appears once per merged-launch width: summed)
        if sym in r["kernel"]:
            rd = float(r["hbm_read_bytes(FETCH_SIZE*1024*2)"])
            wr = float(r["hbm_write_bytes(WRITE_SIZE*1024)"])
            # counter average over all launches of the kernel x launches per registration = bytes per registration
            tot += (rd + wr) * int(r["launches"]) / max(regs, 1)
            hit = True
    return (tot, os.path.relpath(files[-1], ROOT)) if hit else (None, None)


def rocprof_stats(tag, group=8):
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
        import re
        m = re.search(r"k_batch.*?(k_[a-z0-9_]+)", name)     # a kernel in its lock-step form (launch.h): k_batch<&k_xxx, ...>
        if m:
            return "k_batch<" + m.group(1) + ">"
        for junk in ("void ", "plade::", "(anonymous namespace)::"):
            name = name.replace(junk, "")
        return name.split("(")[0][:48]
    top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:5]
    out = {"file": os.path.relpath(files[-1], ROOT),
           "top_by_gpu_time": [{"kernel": short(r["Name"]), "share": round(float(r["TotalDurationNs"]) / total, 4),
                                "avg_us": round(float(r["AverageNs"]) / 1e3, 2)} for r in top]}
    regs = _profile_registrations(files[-1], rows, "Name", "Calls") or group * next((int(r["Calls"]) for r in rows if "k_morton(" in r["Name"]), 0)
    out["registrations_profiled"] = regs
    out["kernels_per_registration"] = round(sum(int(r["Calls"]) for r in rows if "rocclr" not in r["Name"]) / max(regs, 1), 1)
    out["copies_and_fills_per_registration"] = round(sum(int(r["Calls"]) for r in rows if "rocclr" in r["Name"]) / max(regs, 1), 1)
    out["gpu_ms_per_registration"] = round(total / 1e6 / max(regs, 1), 3)
    calls = sum(int(r["Calls"]) for r in rows if sym in r["Name"])
    if calls:
        dur = sum(float(r["TotalDurationNs"]) for r in rows if sym in r["Name"])
        out.update({"avg_launch_us": dur / calls / 1e3, "calls": calls, "launches_per_registration": calls / max(regs, 1),
                    "share_of_gpu_time": dur / total})
    return out


def _gen_pair_worker(job):
    n, seed, path = job
    from plade_amd.synth import make_pair
    tg, sr, T = make_pair(n, seed=seed)
    np.save(path + "_t.npy", tg); np.save(path + "_s.npy", sr); np.save(path + "_T.npy", T)
    return seed


def generate_pairs(n_points, seeds, procs):
    """The synthetic pairs of this rank (plade_amd/synth.py: one scene per seed), generated by `procs` worker processes -- the
    generator is ~4 s of numpy per 1M-point pair, and the batch of BASELINE configs[3] holds 64 of them."""
    from plade_amd.synth import make_pair
    if len(seeds) <= 2 or procs <= 1:
        return [make_pair(n_points, seed=sd) for sd in seeds]
    import multiprocessing as mp
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="plade_pairs_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        with mp.get_context("spawn").Pool(min(procs, len(seeds))) as pool:
            pool.map(_gen_pair_worker, [(n_points, sd, os.path.join(d, str(sd))) for sd in seeds], chunksize=1)
        return [(np.load(os.path.join(d, f"{sd}_t.npy")), np.load(os.path.join(d, f"{sd}_s.npy")), np.load(os.path.join(d, f"{sd}_T.npy")))
                for sd in seeds]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _match_set(d):
    q = np.repeat(np.arange(len(d["match_offsets"]) - 1), np.diff(d["match_offsets"]))
    return set(zip(q.tolist(), d["match_nbr"].tolist()))


def _parity_worker(job):
    """One pair of the parity leg in a process of its own (spawned: no HIP in it): the oracle in its default mode -- the
    reference's fp32 SVD solves -- on the planes the GPU extracted, against what the GPU returned for the same pair."""
    path, modes = job
    import numpy as np
    from oracle.oracle import Oracle
    orc = Oracle()
    z = np.load(path)
    tp = (z["tp_coef"], z["tp_off"], z["tp_idx"])
    sp = (z["sp_coef"], z["sp_off"], z["sp_idx"])
    gpu_set = set(zip(z["m_q"].tolist(), z["m_nbr"].tolist()))
    out = {}
    for mode in modes:
        orc.set_closest_point_mode(mode)
        ok, T, d = orc.registration(z["tg"], z["sr"], tp, sp, voxel_sort_mode=1)
        q = np.repeat(np.arange(len(d["match_offsets"]) - 1), np.diff(d["match_offsets"]))
        o_set = set(zip(q.tolist(), d["match_nbr"].tolist()))
        out[mode] = {"ok": bool(ok), "flips": len(gpu_set ^ o_set), "matches": int(len(d["match_nbr"])),
                     "dT": float(np.linalg.norm(np.asarray(T, np.float64) - z["T"].astype(np.float64))),
                     "T_bits_equal": bool(np.array_equal(np.asarray(T, np.float32), z["T"])),
                     "overlap_counts_equal": bool(np.array_equal(d["overlap_counts"], z["overlap_counts"]))}
    orc.reset_closest_point_mode()
    return out


def parity_vs_reference_solver(device, pairs, seeds, procs, n_generic=2):
    """CHECKER leg (outside every timed region; the oracle is test infrastructure): are the timed results the reference's
    arithmetic?  EVERY pair of the timed cycle, as generated (axis-aligned rooms: what `value` times), is registered once
    more on the GPU in the timed mode (the library's default: closest_point_mode = 1, the reference's fp32 cv::solve,
    util.cpp:1183-1226, 1467-1497, k_svd.h) with the intermediates kept, and the oracle -- same mode, restated from OpenCV's
    lapack.cpp:533-812 -- registers the same clouds on the planes the GPU extracted (the reference's RANSAC is time-seeded).
    flips = descriptor matches (query, neighbour) in one set and not in the other; dT = Frobenius norm of the difference of the
    final 4 x 4; the integer overlap counts of every verified candidate are compared bit for bit.  The first n_generic pairs
    are also checked turned into a generic orientation, and in the opt-in closed-form mode against both oracle modes."""
    import multiprocessing as mp
    import shutil
    import tempfile
    import plade_amd
    q = np.random.default_rng(4).normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R0 = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def turned(c):
        o = np.empty_like(c)
        o[:, :3] = (c[:, :3].astype(np.float64) @ R0.T).astype(np.float32)
        o[:, 3:] = (c[:, 3:].astype(np.float64) @ R0.T).astype(np.float32)
        return o

    c = plade_amd.Context(device, dump=1, orient_normals=1)
    d_ = tempfile.mkdtemp(prefix="plade_parity_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    jobs, tags, gpu_T = [], [], {}

    def stage(tag, a, b, mode, oracle_modes):
        c.set_params(closest_point_mode=mode)
        ok, T = c.registration(a, b)
        d = c.dump()
        path = os.path.join(d_, f"{len(jobs)}.npz")
        mq = np.repeat(np.arange(len(d["match_offsets"]) - 1), np.diff(d["match_offsets"]))
        np.savez(path, tg=a, sr=b, T=np.asarray(T, np.float32), overlap_counts=d["overlap_counts"], m_q=mq, m_nbr=d["match_nbr"],
                 tp_coef=d["tgt_planes"].reshape(-1, 4), tp_off=d["tgt_plane_offsets"], tp_idx=d["tgt_plane_idx"],
                 sp_coef=d["src_planes"].reshape(-1, 4), sp_off=d["src_plane_offsets"], sp_idx=d["src_plane_idx"])
        jobs.append((path, oracle_modes))
        tags.append((tag, bool(ok)))
        gpu_T[tag] = np.asarray(T, np.float64)

    try:
        for k, (tg, sr, _) in enumerate(pairs):
            stage(("timed_mode", k), tg, sr, 1, (1,))
        for k, (tg, sr, _) in enumerate(pairs[:n_generic]):
            stage(("timed_mode_generic", k), turned(tg), turned(sr), 1, (1,))
            stage(("closed_form", k), tg, sr, 0, (0, 1))
            stage(("closed_form_generic", k), turned(tg), turned(sr), 0, (0, 1))
        c.close()
        with mp.get_context("spawn").Pool(max(1, min(int(procs), len(jobs), 16))) as pool:
            res = pool.map(_parity_worker, jobs, chunksize=1)
    finally:
        shutil.rmtree(d_, ignore_errors=True)

    def agg(rows):
        return {"pairs": len(rows), "all_ok": all(r["ok"] for r in rows), "flips": max(r["flips"] for r in rows),
                "matches_min": min(r["matches"] for r in rows), "matches_total": sum(r["matches"] for r in rows),
                "dT": max(r["dT"] for r in rows), "T_bits_equal": all(r["T_bits_equal"] for r in rows),
                "overlap_counts_equal": all(r["overlap_counts_equal"] for r in rows)}
    by = {}
    for (tag, ok), r in zip(tags, res):
        for mode, v in r.items():
            by.setdefault((tag[0], mode), []).append(dict(v, ok=v["ok"] and ok))
    timed = agg(by[("timed_mode", 1)])
    out = {"timed_mode": "closest_point_mode = 1 (the library's default): the reference's fp32 cv::solve(DECOMP_SVD) arithmetic",
           "pairs_checked_seeds": [int(seeds[0]), int(seeds[len(pairs) - 1])],
           "flips": timed["flips"], "dT": timed["dT"],
           "as_generated_all_timed_pairs": timed,
           "generic_orientation": agg(by[("timed_mode_generic", 1)]) if ("timed_mode_generic", 1) in by else None,
           "closed_form_opt_in": {
               "as_generated_vs_oracle_closed_form": agg(by[("closed_form", 0)]) if ("closed_form", 0) in by else None,
               "as_generated_vs_oracle_reference_solver": agg(by[("closed_form", 1)]) if ("closed_form", 1) in by else None,
               "generic_vs_oracle_closed_form": agg(by[("closed_form_generic", 0)]) if ("closed_form_generic", 0) in by else None,
               "generic_vs_oracle_reference_solver": agg(by[("closed_form_generic", 1)]) if ("closed_form_generic", 1) in by else None},
           "moved_closed_form_vs_timed_mode_frobenius": max(float(np.linalg.norm(gpu_T[("closed_form", k)] - gpu_T[("timed_mode", k)]))
                                                            for k in range(min(n_generic, len(pairs)))) if n_generic else None,
           "note": "worst case over the checked pairs; `value` times the DEFAULT mode = the reference's solver arithmetic, checked on every "
                   "pair of the timed cycle against the oracle's restatement of it (flips 0, dT 0, integer overlap counts equal = bit "
                   "parity); closest_point_mode = 0 (fp64 closed form) is an opt-in deviation whose rate is `closed_form_mode_rank0`"}
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


def _cpu_batch_worker(wid, files, n_regs, barrier, q):
    """One single-threaded CPU worker of cpu_baseline_batch (a process of its own, like one CLI process of the reference)."""
    import numpy as np
    from oracle.oracle import Oracle, Reference, have_reference
    orc = Oracle()
    ref = Reference() if have_reference() else None
    clouds = [(np.load(t), np.load(s_)) for t, s_ in files]
    barrier.wait()
    t0 = time.perf_counter()
    ok_all = True
    for r in range(n_regs):
        tg, sr = clouds[(wid + r) % len(clouds)]
        planes = []
        for cloud, seed in ((tg, 2 * (wid * n_regs + r) + 1), (sr, 2 * (wid * n_regs + r) + 2)):
            pl = ref.ransac_detect(cloud, 10000, fake_time=seed)
            co = pl[0].copy()
            for i in range(len(co)):      # params.orient_normals = 1 (the rule the GPU path applies)
                ids = pl[2][pl[1][i]:pl[1][i + 1]]
                if cloud[ids, 3:].astype(np.float64).mean(0) @ co[i, :3] < 0:
                    co[i] = -co[i]
            planes.append((co, pl[1], pl[2]))
        ok, T, _ = orc.registration(tg, sr, planes[0], planes[1], voxel_sort_mode=0)
        ok_all = ok_all and bool(ok)
    q.put((wid, t0, time.perf_counter(), ok_all))


def cpu_baseline_batch(pairs, n_pairs_batch=64):
    """for the batch config the CPU analogue of sharding pairs over GPUs is min(#cores, #pairs) independent
    single-threaded worker PROCESSES.  Every worker registers one pair (reference RANSAC from oracle/_ref + the oracle's
    restatement); rate = registrations / (last finish - first start).  Skipped where oracle/_ref is absent."""
    import multiprocessing as mp
    import tempfile
    from oracle.oracle import have_reference
    if not have_reference():
        return {"value": None, "sample": "unavailable: oracle/_ref absent"}
    workers = int(max(1, min(_cpu_budget(), n_pairs_batch)))
    ctx = mp.get_context("spawn")
    d = tempfile.mkdtemp(prefix="plade_cpu_batch_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    files = []
    try:
        for k, (tg, sr, _) in enumerate(pairs):
            ft, fs = os.path.join(d, f"t{k}.npy"), os.path.join(d, f"s{k}.npy")
            np.save(ft, tg); np.save(fs, sr)
            files.append((ft, fs))
        barrier, q = ctx.Barrier(workers), ctx.Queue()
        procs = [ctx.Process(target=_cpu_batch_worker, args=(w, files, 1, barrier, q)) for w in range(workers)]
        for p_ in procs:
            p_.start()
        res = [q.get(timeout=300) for _ in procs]
        for p_ in procs:
            p_.join(timeout=60)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    t_first, t_last = min(r[1] for r in res), max(r[2] for r in res)
    return {"value": workers / (t_last - t_first), "unit": "registrations/s", "processes": workers, "cores": workers,
            "registrations": workers, "all_ok": all(r[3] for r in res), "seconds": t_last - t_first,
            "sample": f"{workers} single-threaded worker processes (min(CPU budget, 64 pairs)), one 1M-pt registration each, started "
                      "together: reference Schnabel RANSAC (oracle/_ref) + oracle restatement; rate = registrations / (last finish - "
                      "first start)"}


def cli_end_to_end(pairs, n_pairs=64, inflight=None, group=None):
    """the CLI end to end (code/PLADE/main.cpp:97-158 batch mode), PLY parse and process start-up included: a
    file_pairs.txt of n_pairs lines pairs over the bench pairs' PLY files (binary little endian, float x y z nx ny nz),
    `PLADE file_pairs.txt result.txt` as a user would run it."""
    import subprocess
    import tempfile
    from plade_amd.plyio import write_ply
    cli = os.path.join(ROOT, "plade_amd", "PLADE")
    if not os.path.exists(cli):
        return {"value": None, "note": "plade_amd/PLADE not built"}
    d = tempfile.mkdtemp(prefix="plade_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        names = []
        for k, (tg, sr, _) in enumerate(pairs):
            ft, fs = os.path.join(d, f"t{k}.ply"), os.path.join(d, f"s{k}.ply")
            write_ply(ft, tg); write_ply(fs, sr)
            names.append((ft, fs))
        lst, out = os.path.join(d, "file_pairs.txt"), os.path.join(d, "result.txt")
        with open(lst, "w") as f:
            for i in range(n_pairs):
                f.write(f"{names[i % len(names)][0]}\n{names[i % len(names)][1]}\n")
        lst_long = os.path.join(d, "file_pairs_512.txt")
        with open(lst_long, "w") as f:
            for i in range(512):
                f.write(f"{names[i % len(names)][0]}\n{names[i % len(names)][1]}\n")
        env = dict(os.environ, PLADE_ORIENT_NORMALS="1")   # GPUs, workers and group size: the CLI's own defaults
        if inflight:
            env["PLADE_INFLIGHT"] = str(inflight)
        if group:
            env["PLADE_GROUP"] = str(group)
        runs = []
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([cli, lst, out], capture_output=True, text=True, timeout=600, env=env)
            runs.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"value": None, "note": "CLI failed: " + r.stderr[-300:]}
        blocks = open(out).read().count("transformation:")
        failed = open(out).read().count("registration failed, an identity")
        tl = time.perf_counter()
        rl = subprocess.run([cli, lst_long, out], capture_output=True, text=True, timeout=900, env=env)
        long_s = time.perf_counter() - tl
        long_blocks = open(out).read().count("transformation:") if rl.returncode == 0 else 0
        t1 = subprocess.run([cli, names[0][0], names[0][1], out], capture_output=True, text=True, timeout=600, env=env)
        ts = time.perf_counter()
        subprocess.run([cli, names[0][0], names[0][1], out], capture_output=True, text=True, timeout=600, env=env)
        single = time.perf_counter() - ts
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    best = min(runs)
    return {"value": n_pairs / best, "unit": "registrations/s", "pairs": n_pairs, "distinct_pairs": min(n_pairs, len(names)),
            "registered": blocks, "failed": failed, "seconds": best,
            "seconds_all_runs": runs, "single_pair_process_seconds": single,
            "list_of_512": {"value": 512 / long_s if long_blocks == 512 else None, "seconds": long_s, "blocks": long_blocks,
                            "note": "a 512-line list over the same files: the CLI's defaults for long lists (4 workers x groups of 8)"}, "workers": inflight or "CLI default (2 below 512 pairs, else 4)", "pairs_per_group": group or "CLI default (4 below 512 pairs, else 8)",
            "note": "wall time of the whole `PLADE file_pairs.txt result.txt` process: HIP start-up (~0.3 s), PLY parse of 2 x 24 MB "
                    "per pair (files in the page cache), registration, ordered result file"}


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


# busy host threads per rank, measured on the MI355X box in round 4 (sleeping host waits, groups of 8 pairs,
# host clouds): groups in flight -> busy threads (registrations/s): see profiles/r5_experiments.md
BUSY_THREADS_BY_GROUPS = {1: 1.02, 2: 1.5, 3: 1.52, 4: 1.78}   # r6 (reference arithmetic, profiles/r6_experiments.md 5): 3 groups 680-693 registrations/s at 1.46-1.52 threads, 4 groups 733-734 at 1.75-1.78; 1 / 2 groups: r5 (457 / 665)


def inflight_for_budget(budget, local_world):
    """Groups (of 8 pairs) in flight per GPU.  A container that runs into its CPU quota loses far more than the last few
    percent of GPU throughput (round 1: 287 instead of 400 reg/s under throttling), so the count is the largest one whose
    measured host load (BUSY_THREADS_BY_GROUPS), times the ranks of this node, still fits the quota with 10 % to spare."""
    per_rank = 0.9 * float(budget) / max(local_world, 1)
    best = 1
    for g in sorted(BUSY_THREADS_BY_GROUPS):
        if g <= 4 and BUSY_THREADS_BY_GROUPS[g] <= per_rank:
            best = g
    return best


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
    ap.add_argument("--pairs", type=int, default=0,
                    help="distinct synthetic pairs per rank, seeds rank * pairs ... (cycled in input order); 0 = the batch of BASELINE "
                         "configs[3]: 64 pairs on one GPU, max(16, 64 / ranks) per rank on several")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity leg (every timed pair on the GPU vs the oracle, both with the reference's fp32 SVD solver)")
    ap.add_argument("--closed-form-steps", type=int, default=256, help="steps of the extra leg with closest_point_mode = 0 (opt-in closed form); 0 = skip")
    ap.add_argument("--host-wait", choices=["auto", "spin", "sleep"], default="auto",
                    help="how the host threads wait for the GPU (plade_params.host_wait): spinning waits keep ~1.7 CPUs busy "
                         "per registration in flight, sleeping ones ~0.5 at the same throughput; auto = sleep when more "
                         "than one registration is in flight")
    ap.add_argument("--inflight", type=int, default=0,
                    help="GROUPS in flight per GPU: one plade_ctx + host thread each, every call registers --group consecutive "
                         "pairs of the batch (a single registration is latency-bound and leaves most of the GPU idle).  0 = 4, or "
                         "fewer when the ranks of this node have to share a small CPU quota (see inflight_for_budget)")
    ap.add_argument("--group", type=int, default=8,
                    help="pairs per group (plade_registration_pairs, 1..8): the plane extraction of all clouds of a group is one "
                         "launch sequence; 1 = one pair per call (round 3's scheme)")
    ap.add_argument("--resident-steps", type=int, default=256,
                    help="steps of the extra resident leg (clouds uploaded once, plade_registration_dev per step); 0 = skip")
    ap.add_argument("--host-steps", type=int, default=0, help=argparse.SUPPRESS)   # round-2 flag, ignored
    ap.add_argument("--no-default-mode", action="store_true", help="skip the orient_normals=0 success-rate leg")
    ap.add_argument("--profiled-steps", type=int, default=16, help="registrations of the roofline leg (HIP events per launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cli", action="store_true", help="skip the CLI end-to-end leg (64-pair file_pairs.txt through plade_amd/PLADE)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook for boxes with one GPU: PLADE_BENCH_ONE_GPU=1 puts every rank on cuda:0 and exchanges over gloo (RCCL
    # refuses two ranks on one device), so that the N > 1 control flow can be exercised there
    one_gpu = os.environ.get("PLADE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    # No torch in the ranks.  `import torch` brings the HIP runtime bundled with the wheel (ROCm 7.0) into the process ahead
    # of the system's (7.2), and the library then runs on that one: measured 458 instead of 483 registrations/s at N = 1
    # (round 3's A/B: the same pipeline with and without `import torch` in the process) -- a penalty every rank would pay.  The path shards whole pairs and has
    # no data-path collective; what the ranks exchange -- the barriers around the timed region, three numbers to reduce and
    # 68 bytes of result per pair -- goes over their loopback rendezvous (plade_amd/rendezvous.py; the ranks of
    # `torch.distributed.run --nnodes=1` share one host).  The timed region is bracketed with hipDeviceSynchronize through
    # the library.  PLADE_BENCH_TORCH=1 restores the torch.distributed exchange (RCCL; gloo with PLADE_BENCH_ONE_GPU).
    torch = dist = comm = dev = boot = None
    exchange_note = "single process"
    if world > 1 and os.environ.get("PLADE_BENCH_TORCH") == "1":
        exchange_note = "torch.distributed"
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo" if one_gpu else "nccl", world_size=world, rank=rank)  # nccl == RCCL on ROCm
        dev = torch.device("cpu") if one_gpu else torch.device("cuda", local_rank)
    elif world > 1:
        # Default: RCCL over xGMI, bound by the library itself (dlopen, no torch in the process): plade_amd/rccl_comm.py.  The
        # loopback rendezvous carries the 128-byte unique id and is the fallback where RCCL cannot serve (two ranks on one
        # device -- the one-GPU test hook --, no librccl, an initialisation that does not finish): the ranks agree on it.
        from plade_amd.rendezvous import Rendezvous
        boot = Rendezvous.from_env()
        comm, exchange_note = boot, "loopback rendezvous (plade_amd/rendezvous.py)"
        if os.environ.get("PLADE_BENCH_NO_RCCL") != "1":
            from plade_amd import rccl_comm
            rc_comm, why = rccl_comm.connect(rank, world, local_rank, boot)
            if rc_comm is not None:
                comm, exchange_note = rc_comm, "rccl (ncclAllGather over xGMI, librccl opened by libplade_hip.so; bootstrap: loopback rendezvous)"
            else:
                exchange_note += f"; rccl not used: {why}"

    import plade_amd
    from plade_amd.synth import make_pair

    def device_sync():
        if torch is not None:
            torch.cuda.synchronize()
        else:
            plade_amd.device_synchronize(local_rank)

    def barrier():
        if comm is not None:
            comm.barrier()
        elif dist is not None:
            dist.barrier()

    import threading
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    inflight_auto = args.inflight <= 0
    if inflight_auto:
        args.inflight = inflight_for_budget(_cpu_budget(), local_world)
    M = max(1, args.inflight)                  # groups in flight
    S = max(1, min(8, args.group))             # pairs per group (PLADE_GROUP_MAX)
    RIF = M * S                                # registrations in flight
    if args.host_wait == "auto":
        # several registrations in flight: sleeping waits (same throughput, a third of the host CPUs, and no way to run
        # into the container's CPU quota when 8 ranks share a node); one at a time: spin for the lowest latency
        args.host_wait = "sleep" if RIF > 1 else "spin"
    host_wait = {"spin": 0, "sleep": 1}[args.host_wait]
    ctxs = [plade_amd.Context(local_rank, host_wait=host_wait, orient_normals=1) for _ in range(M)]
    ctx = ctxs[0]
    # synthetic pairs: seeds are global pair ids (batch of independent pairs sharded across ranks); every
    # worker holds its own resident copy so the workers share nothing
    if args.pairs <= 0:
        args.pairs = 64 if world == 1 else max(16, 64 // world)
    seeds = [rank * args.pairs + k for k in range(args.pairs)]
    t_gen = time.perf_counter()
    pairs = generate_pairs(args.points, seeds, max(1, int(_cpu_budget() / max(local_world, 1))))
    t_gen = time.perf_counter() - t_gen
    NP = len(pairs)
    # resident copies (plade_cloud_upload): context 0 holds every pair (the pair-alone comparison), the others the first NR
    # (the resident leg and the background load of the roofline leg cycle over those)
    NR = min(NP, 16)
    clouds = [[] for _ in range(M)]
    for k, (tg, sr, _) in enumerate(pairs):
        for w in range(M):
            if w == 0 or k < NR:
                clouds[w].append((ctxs[w].upload(tg), ctxs[w].upload(sr)))

    executed = [0] * (M + 1)      # registrations this process ran, per worker (profiles divide the kernel statistics by their sum)

    def step(i, w=0):       # one pair alone on resident clouds (latency figure, default-mode leg)
        ct, cs = clouds[w][i % (NP if w == 0 else NR)]
        executed[M] += 1
        return ctxs[w].registration_dev(ct, cs)

    # Step number i registers pair i % NP; group number j holds the steps j*S .. j*S + S - 1 (consecutive pairs of the batch).
    def members(j):
        return [j * S + q for q in range(S)]

    # the timed path: registration(T, target, source) of code/PLADE/plade.h:58 on clouds in (page-locked) HOST memory, in
    # batch mode (main.cpp:97-158 loops over pairs): plade_registration_pairs = S x plade_registration with the plane
    # extraction of the group's clouds in one launch sequence + the upload of the group the same context registers next,
    # queued on a stream of its own.  H2D, SoA conversion and bounding boxes of every step are inside the timed region.
    for tg, sr, _ in pairs:
        ctx.pin(tg); ctx.pin(sr)

    def hgroup(j, w, nxt):
        cur = [(pairs[i % NP][0], pairs[i % NP][1]) for i in members(j)]
        nx = [(pairs[i % NP][0], pairs[i % NP][1]) for i in members(nxt)] if nxt is not None else None
        executed[w] += len(cur)
        return ctxs[w].registration_pairs(cur, nx)

    def rgroup(j, w, nxt):
        executed[w] += S
        return ctxs[w].registration_pairs_dev([clouds[w][i % NR] for i in members(j)])

    def run_pipeline(fn, lead_groups, count_groups):
        """Steady-state throughput of the pipeline of M groups in flight: worker w takes groups w, w + M, ... of lead + count + M
        groups (the first `lead` fill the pipeline and are untimed, the last M keep it full until the last timed group
        completes); the timed window runs from the completion of group number `lead` to the completion of group number
        lead + count, i.e. EXACTLY count x S registrations complete inside it with the pipeline full on both sides.
        Returns (seconds of the window, [(step, ok, T)] of the timed steps, occupancy estimate of the same steps in seconds,
        seconds from start to the last completion)."""
        total = lead_groups + count_groups + M
        stamps, out = [0.0] * total, [None] * total

        def work(w):
            for j in range(w, total, M):
                out[j] = fn(j, w, j + M if j + M < total else None)
                stamps[j] = time.perf_counter()
        ths = [threading.Thread(target=work, args=(w,)) for w in range(M)]
        ts = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        done = sorted(stamps)
        window = done[lead_groups + count_groups - 1] - done[lead_groups - 1]
        order = sorted(range(total), key=lambda j: stamps[j])[lead_groups:lead_groups + count_groups]
        # secondary estimator (Little's law): a group occupied its worker from stamps[j - M] to stamps[j]; with M groups
        # always in flight the rate is M * S / (mean occupancy of the timed groups)
        occ = {j: stamps[j] - (stamps[j - M] if j >= M else ts) for j in order}
        occupancy = sum(occ.values()) / count_groups
        steps = []
        for j in sorted(order):
            for q, i in enumerate(members(j)):
                steps.append((i, bool(out[j][q][0]), out[j][q][1]))
        run_pipeline.group_occupancy = occ      # seconds a timed group occupied its worker, by group number
        run_pipeline.completions = done         # completion stamps of all groups, ascending
        return window, steps, count_groups * occupancy / M, done[-1] - ts

    # warm-up: every worker (context) registers every group composition once through BOTH entry points, so that no
    # first-use allocation or graph capture falls into the timed region; the W warm-up steps the driver asks for are part
    # of the lead-in of the pipelined run below
    n_comp = NP // S if NP % S == 0 else NP          # distinct group compositions of the cycle
    n_comp_r = NR // S if NR % S == 0 else NR

    def warm_worker(w):
        for j in range(max(n_comp_r, 1)):
            rgroup(j, w, None)
        for j in range(max(n_comp, 1)):
            hgroup(j, w, None)
    wths = [threading.Thread(target=warm_worker, args=(w,)) for w in range(M)]
    for t in wths:
        t.start()
    for t in wths:
        t.join()
    # lead-in: the W warm-up steps the driver asks for, and at least 8 rounds of the M workers -- they start in lock step
    # (all in the same stage at once, competing for the same units) and need a few rounds to spread out over the stages
    lead_groups = max((args.warmup + S - 1) // S, int(os.environ.get("BENCH_LEAD_ROUNDS", "8")) * M)
    device_sync()
    barrier()
    device_sync()
    cpu0, thr0 = time.process_time(), _cgroup_throttle()
    t_begin = time.perf_counter()
    # K = --steps; a pipeline of M x S registrations in flight is not sampled fairly by fewer than ~32 rounds of it (the driver's
    # 20 steps are little more than ONE round of 16), so at least 32 * M * S steps are timed, in whole groups; `steps` on the
    # line is the number of steps really timed, `requested_steps` echoes K
    timed_groups = (max(args.steps, int(os.environ.get("BENCH_MIN_ROUNDS", "32")) * RIF) + S - 1) // S   # (BENCH_MIN_ROUNDS: profiling runs)
    n_timed = timed_groups * S
    window, timed, occ_elapsed, span = run_pipeline(hgroup, lead_groups, timed_groups)
    host_occ = dict(run_pipeline.group_occupancy)
    # the figure for exactly the K steps the driver asked for (in whole groups): the first ceil(K / S) completions of the window
    req_groups = max(1, min(timed_groups, (args.steps + S - 1) // S))
    req_window = run_pipeline.completions[lead_groups + req_groups - 1] - run_pipeline.completions[lead_groups - 1]
    elapsed = window
    cpu1, thr1 = time.process_time(), _cgroup_throttle()
    timed_ids = [t[0] for t in timed]
    oks = [t[1] for t in timed]
    results = [t[2] for t in timed]
    n_ok = sum(oks)
    # gather the per-pair 4x4 results on rank 0 in input order (the only exchange the path needs;
    # plade_amd/batch.py, covered on CPU by tests/test_distributed_gloo.py with gloo)
    from plade_amd.batch import gather_results
    all_T, all_ok = gather_results(np.stack(results), np.array(oks, bool), world * n_timed, rank, world,
                                   device=dev, comm=comm)
    device_sync()
    barrier()
    device_sync()
    bracketed = time.perf_counter() - t_begin
    total_ok = n_ok
    if comm is not None:
        elapsed, bracketed, occ_elapsed, req_window = comm.all_reduce_max([elapsed, bracketed, occ_elapsed, req_window])
        total_ok = int(comm.all_reduce_sum([n_ok])[0])
    elif world > 1:
        tmax = torch.tensor([elapsed, bracketed, occ_elapsed, req_window], dtype=torch.float64, device=dev)
        okt = torch.tensor([n_ok], dtype=torch.int64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(okt, op=dist.ReduceOp.SUM)
        elapsed, bracketed, occ_elapsed, req_window = (float(tmax[k].item()) for k in range(4))
        total_ok = int(okt.item())

    # every registration of the same pair, whichever context ran it and whatever its partners in the group were, must
    # return the same bits -- and the bits of the pair registered ALONE (one plade_registration_dev per distinct pair)
    alone, alone_ms, work_by_seed = [], [], []
    for k in range(NP):
        t1 = time.perf_counter()
        alone.append(step(k))
        alone_ms.append((time.perf_counter() - t1) * 1e3)
        st_k = ctx.stats()
        work_by_seed.append({q: st_k.get(q) for q in ("ransac_iterations", "n_planes_tgt", "n_planes_src", "n_matches", "n_clusters",
                                                        "n_candidates_verified")})
    # service time by group composition: a group of S consecutive pairs occupied one of the M workers for occ seconds, i.e. it
    # cost the pipeline occ / M of wall time = occ / (M * S) per registration (Little's law, as `occupancy_value`)
    by_comp = {}
    for j, o in host_occ.items():
        by_comp.setdefault(j % max(n_comp, 1), []).append(o / (M * S) * 1e3)
    comp_ms = sorted(float(np.mean(v)) for v in by_comp.values())

    def mmm(v):
        v = sorted(float(x) for x in v if x is not None)
        return {"min": v[0], "median": v[len(v) // 2], "max": v[-1]} if v else None
    value_by_seed = {
        "distinct_pairs_this_rank": NP, "seeds": [seeds[0], seeds[-1]],
        "ms_per_registration_by_group_composition": mmm(comp_ms), "group_compositions": len(comp_ms),
        "registrations_per_s_by_group_composition": ({"min": 1e3 / comp_ms[-1], "median": 1e3 / comp_ms[len(comp_ms) // 2], "max": 1e3 / comp_ms[0]}
                                                     if comp_ms else None),
        "pair_alone_latency_ms": mmm(alone_ms),
        "ransac_iterations": mmm([w_["ransac_iterations"] for w_ in work_by_seed]),
        "planes_target": mmm([w_["n_planes_tgt"] for w_ in work_by_seed]), "planes_source": mmm([w_["n_planes_src"] for w_ in work_by_seed]),
        "descriptor_matches": mmm([w_["n_matches"] for w_ in work_by_seed]), "clusters": mmm([w_["n_clusters"] for w_ in work_by_seed]),
        "candidates_verified": mmm([w_["n_candidates_verified"] for w_ in work_by_seed]),
        "note": "every pair of the cycle is a different scene (seed); the timed steps run through them in input order, so `value` is the "
                "batch average; by group composition = mean time a group of consecutive pairs occupied its worker / (groups in flight "
                "x pairs per group); pair alone = one plade_registration_dev with sleeping host waits"}
    identical = all(np.array_equal(results[k], alone[timed_ids[k] % NP][1]) and oks[k] == bool(alone[timed_ids[k] % NP][0])
                    for k in range(len(results)))
    ref_result = {k: alone[k][1] for k in range(NP)}
    # accuracy of the timed registrations on this rank vs the generator's ground truth
    errs = [float(np.linalg.norm(results[k].astype(np.float64) - pairs[timed_ids[k] % NP][2])) for k in range(len(results))]

    # ---- resident leg (clouds already in HBM, plade_registration_pairs_dev): the same pipeline without the uploads,
    #      reported next to `value`
    resident_leg = None
    if args.resident_steps > 0:
        r_groups = (args.resident_steps + S - 1) // S
        r_win, r_res, _, _ = run_pipeline(rgroup, 2 * M, r_groups)
        device_sync()
        same = all(np.array_equal(T, ref_result[i % NR]) for (i, ok, T) in r_res)
        resident_leg = {"value": r_groups * S / r_win, "unit": "registrations/s (this rank)", "steps": r_groups * S,
                        "ms_per_step": r_win / (r_groups * S) * 1e3, "identical_to_host_cloud_results": bool(same),
                        "note": "clouds resident in HBM (plade_cloud_upload once, plade_registration_pairs_dev per group): no H2D, no SoA "
                                "conversion, no bounding box in the step"}
    # ---- the same host-cloud pipeline with the opt-in closed-form closest points (closest_point_mode = 0: fp64 formula instead
    #      of the reference's fp32 SVD solves): what the parity with the reference's solver costs
    closed_leg = None
    if args.closed_form_steps > 0:
        for c_ in ctxs:
            c_.set_params(closest_point_mode=0)
        s_groups = (args.closed_form_steps + S - 1) // S
        s_win, s_res, _, _ = run_pipeline(hgroup, 2 * M, s_groups)
        device_sync()
        for c_ in ctxs:
            c_.set_params(closest_point_mode=1)
        s_err = [float(np.linalg.norm(T.astype(np.float64) - pairs[i % NP][2])) for (i, ok, T) in s_res]
        closed_leg = {"value": s_groups * S / s_win, "unit": "registrations/s (this rank)", "steps": s_groups * S,
                      "ms_per_step": s_win / (s_groups * S) * 1e3, "all_ok": all(ok for (_, ok, _) in s_res),
                      "max_frobenius_vs_ground_truth": max(s_err) if s_err else None,
                      "moved_vs_timed_mode_max_frobenius": max(float(np.linalg.norm(T.astype(np.float64) - ref_result[i % NP].astype(np.float64)))
                                                               for (i, ok, T) in s_res),
                      "note": "plade_params.closest_point_mode = 0 on host clouds, same pipeline as `value`: NOT the reference's arithmetic (on "
                              "these axis-aligned scenes the reference's fp32 solves are ill-conditioned and the two modes part, DESIGN.md section 2)"}
    parity = None
    if rank == 0 and world == 1 and not args.no_parity and not args.no_cpu_baseline:
        try:
            parity = parity_vs_reference_solver(local_rank, pairs, seeds, _cpu_budget())
        except Exception as e:   # the oracle is test infrastructure: its absence must not hide the GPU number
            parity = {"note": f"unavailable: {e}"}
    mb = sum(tg.nbytes + sr.nbytes for tg, sr, _ in pairs) / NP / 1e6
    host_leg = {"h2d_MB_per_step": mb, "pcie_GB_per_s": mb * 1e-3 * n_timed / elapsed,
                "bracketed_value": (lead_groups + timed_groups + M) * S / bracketed if bracketed > 0 else None,
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
    roofline = roofline_sort = dominant = None
    stage = {}
    latency_ms = latency_by_scene = None
    if rank == 0:
        lat = []
        ctx.set_params(host_wait=0)   # alone, a spinning wait is the faster one
        for i in range(NP):   # one registration at a time, every scene of the cycle once: the latency figure is the MEDIAN over the scenes
            t1 = time.perf_counter()
            step(i)
            lat.append(time.perf_counter() - t1)
        latency_ms = sorted(lat)[len(lat) // 2] * 1e3
        latency_by_scene = {"min": min(lat) * 1e3, "median": latency_ms, "max": max(lat) * 1e3, "scenes": len(lat),
                            "note": "one plade_registration_dev at a time on resident clouds, spinning host waits, every scene of the cycle once"}
        # Profiled steps (HIP events on the launch stream around every launch of the scan kernels) on context 0 WHILE the
        # other contexts keep registering, i.e. under the load of the timed region: the average launch duration must be
        # the one `rocprofv3 --kernel-trace --stats` reports for this command (profiles/), not that of an idle GPU.
        stop = threading.Event()

        def background(w):
            j = w
            while not stop.is_set():
                rgroup(j, w, None)
                j += M
        bths = [threading.Thread(target=background, args=(w,)) for w in range(1, M)]
        for t in bths:
            t.start()
        # a profiled "step" is a GROUP of S pairs on context 0, as in the timed region (the scan launches of its extraction
        # cover the 2 x S clouds of the group); every per-step figure below is per REGISTRATION: / (groups x S)
        prof_groups = max(1, (args.profiled_steps + S - 1) // S)

        def profiled(mode):
            acc = {}
            ctx.set_params(dump=mode, host_wait=host_wait)
            for j in range(prof_groups):
                rgroup(j, 0, None)
                for q in range(S):
                    for k, v in ctx.stats(pair=q).items():
                        if k.startswith(("k_", "bytes_")):
                            acc[k] = acc.get(k, 0.0) + v
                        elif q == 0:
                            acc[k] = v
            for k in list(acc):
                if k.startswith("bytes_"):
                    acc[k] /= prof_groups * S
            return acc
        # leg 1: HIP events on the launch stream around every profiled launch (every stage; the extraction loop is launched
        # kernel by kernel for it); leg 2: the extraction loop as the timed region launches it -- one captured hipGraph per
        # iteration -- with the scan kernels stamping the device's wall clock themselves (events cannot sit between the
        # nodes of a graph launch).  The roofline figure is leg 2's.
        st = profiled(2)
        st_graph = profiled(6)
        stop.set()
        for t in bths:
            t.join()
        args.profiled_steps = prof_groups * S
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
            # the roofline kernel is the largest mover of algorithmic HBM bytes of the step: the K1 scan of the
            # plane extraction (0.76 GB per registration; the verification kernel, next, moves ~15 MB); latency-bound kernels
            # are listed for reference
            if by > 0 and (best is None or by > st[f"k_{best}_bytes"]):
                best = name
        if best is not None:
            secs, nl, by = st[f"k_{best}_seconds"], st[f"k_{best}_launches"], st[f"k_{best}_bytes"]
            # launch duration: the kernel's own wall-clock stamps (first wavefront in .. last wavefront out, the quantity
            # rocprofv3's kernel trace reports); the HIP events around the launch are listed next to it -- with other
            # registrations in flight they also contain the other streams' kernels that ran on the same hardware queue
            c_secs, c_nl, c_by = st.get(f"k_{best}_clock_seconds"), st.get(f"k_{best}_clock_launches"), st.get(f"k_{best}_clock_bytes")
            g_secs, g_nl, g_by = (st_graph.get(f"k_{best}_clock_seconds"), st_graph.get(f"k_{best}_clock_launches"),
                                  st_graph.get(f"k_{best}_clock_bytes"))
            direct = ({"achieved": c_by / c_secs / 1e9, "avg_launch_us": c_secs / c_nl * 1e6, "frac": c_by / c_secs / 1e9 / HBM_PEAK_GBS,
                       "launches_per_step": c_nl / args.profiled_steps} if c_secs and c_nl else None)
            if g_secs and g_nl:
                achieved, avg_us, how = (g_by / g_secs / 1e9, g_secs / g_nl * 1e6,
                                         "device wall clock inside the kernel (min start / max end over its wavefronts), launched as "
                                         "a node of the iteration's hipGraph exactly as in the timed region")
            elif c_secs and c_nl:
                achieved, avg_us, how = c_by / c_secs / 1e9, c_secs / c_nl * 1e6, "device wall clock inside the kernel (min start / max end over its wavefronts)"
            else:
                achieved, avg_us, how = by / secs / 1e9, secs / nl * 1e6, "HIP events on the launch stream"
            # rocprofv3 (and the counter passes) average over ALL launches of the kernel, including those of the fixed launch
            # sequence that find nothing to do and return at once (they move no bytes); the summaries are therefore compared
            # per registration: counter bytes of all launches of a registration / its working launches = per working launch
            traffic_reg, traffic_src = pmc_traffic(best, S)
            idle = st.get(f"k_{best}_idle_launches", 0.0)
            work_per_step = nl / args.profiled_steps
            traffic = traffic_reg / work_per_step if traffic_reg is not None else None
            roofline = {"bound": "hbm", "kernel": best, "kernel_symbol": KERNEL_SYMBOLS.get(best, best).rstrip("("),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "launches_per_step": nl / args.profiled_steps, "avg_launch_us": avg_us,
                        "launch_path": "hipGraph" if (g_secs and g_nl) else "direct",
                        "direct_launch_leg": direct,
                        "avg_launch_us_hip_events": secs / nl * 1e6,
                        "frac_at_hip_event_average": by / secs / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": by / nl,
                        "idle_launches_per_step": idle / args.profiled_steps,
                        "algorithmic_bytes_per_step": by / args.profiled_steps,
                        "traffic_per_step": traffic_reg,
                        "measured": f"{how}; {args.profiled_steps} profiled registrations with "
                                    f"{(M - 1) * S} other registrations in flight (the load of the timed region); launches are those of groups of {S} pairs, as timed",
                        "why_this_kernel": "dominant by BYTES: the largest mover of algorithmic HBM bytes of the step (the K1 scoring scan, SURVEY.md "
                                           "8d: 28 B per point and launch); the kernels with more GPU TIME are on the line as "
                                           "`dominant_by_gpu_time`, each with its own fraction"}
            # effective vs physical: `achieved` counts the ALGORITHMIC bytes of the launch (every point of the clouds it serves,
            # SURVEY.md 8d), the counters what the launch really moved -- tile-box culling and the compacted scan view skip
            # most of the rest, so the HBM is far less busy than `frac` reads
            if traffic is not None and avg_us:
                roofline["hbm_counter_GBps"] = traffic / (avg_us * 1e-6) / 1e9
                roofline["hbm_counter_frac"] = roofline["hbm_counter_GBps"] / HBM_PEAK_GBS
                roofline["bytes_skipped_frac"] = max(0.0, 1.0 - traffic / (by / nl))
                roofline["effective_vs_physical"] = ("frac = algorithmic bytes / duration (effective bandwidth: work the kernel was asked to do); "
                                                     "hbm_counter_frac = FETCH_SIZE/WRITE_SIZE bytes / duration (physical HBM utilisation); "
                                                     "bytes_skipped_frac = share of the algorithmic bytes never fetched (culled tiles, compacted view)")
            rp = rocprof_stats(best, S)
            if rp is not None:
                roofline["rocprof"] = rp
                if rp.get("avg_launch_us"):
                    # same population as the summary: the step's bytes over ALL launches of a registration there (working +
                    # idle), at the summary's average duration
                    per_launch = (by / args.profiled_steps) / max(rp.get("launches_per_registration", work_per_step), 1e-9)
                    roofline["rocprof"]["algorithmic_bytes_per_launch_all"] = per_launch
                    roofline["rocprof"]["frac_at_rocprof_average"] = per_launch / (rp["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        if st.get("k_sort_pass_seconds"):
            secs, nl, by = st["k_sort_pass_seconds"], st["k_sort_pass_launches"], st["k_sort_pass_bytes"]
            roofline_sort = {"bound": "hbm", "kernel": "sort_pass", "kernel_symbol": "k_rs_pass", "achieved": by / secs / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / secs / 1e9 / HBM_PEAK_GBS,
                             "avg_launch_us": secs / nl * 1e6, "launches_per_step": nl / args.profiled_steps,
                             "algorithmic_bytes_per_launch": by / nl, "algorithmic_bytes_per_step": by / args.profiled_steps,
                             "seconds_per_step": secs / args.profiled_steps,
                             "measured": "HIP events on the launch stream around every pass of the stable LSD radix sort (keys + values read "
                                         "once and written once per pass: 2 n (sizeof key + 4) B), kernel-by-kernel leg under the load of the "
                                         "other groups; in the timed region the sorts of a group's pairs behind the extraction are merged launches",
                             "why": "the kernel family with the most GPU time after the RANSAC chain (rocprof.top_by_gpu_time)"}
            traffic_reg, traffic_src = pmc_traffic("sort_pass", S)
            if traffic_reg is not None:
                roofline_sort["traffic_per_step"] = traffic_reg
                roofline_sort["traffic_source"] = traffic_src
            rp = rocprof_stats("sort_pass", S)
            if rp and rp.get("avg_launch_us"):
                roofline_sort["rocprof"] = {k: rp[k] for k in ("file", "avg_launch_us", "calls", "launches_per_registration", "share_of_gpu_time") if k in rp}
        # the kernel (family) with the most GPU time in the committed kernel trace, with ITS roofline fraction from this run's
        # HIP-event leg: `roofline.kernel` is chosen by bytes moved, this one by time spent
        dominant = None
        rp_any = rocprof_stats(best, S) if best is not None else None
        if rp_any and rp_any.get("top_by_gpu_time"):
            rows = []
            for t_ in rp_any["top_by_gpu_time"][:3]:
                tag = next((k for k, sym in KERNEL_SYMBOLS.items() if sym.rstrip("(") in t_["kernel"]), None)
                row = dict(t_, stage=tag)
                if tag and tag in stage and stage[tag].get("GB/s"):
                    row["algorithmic_GBps"] = stage[tag]["GB/s"]
                    row["frac_of_hbm_peak"] = stage[tag]["GB/s"] / HBM_PEAK_GBS
                    row["seconds_per_step"] = stage[tag]["seconds"]
                tr_, _src = pmc_traffic(tag, S) if tag else (None, None)
                if tr_ is not None and tag in stage and stage[tag].get("seconds"):
                    row["hbm_counter_bytes_per_step"] = tr_
                    row["hbm_counter_frac"] = tr_ / stage[tag]["seconds"] / 1e9 / HBM_PEAK_GBS
                rows.append(row)
            dominant = {"source": rp_any["file"], "kernels": rows,
                        "note": "share / avg_us: committed rocprofv3 kernel trace of this command; frac_of_hbm_peak: algorithmic bytes / HIP-event "
                                "duration of the same kernel in this run's profiled leg (under the load of the other groups)"}
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
            for (tg, sr, Tgt) in pairs[:6]:       # >= 4 distinct seeds
                sample.append((tg, sr, None))
            cpu = cpu_baseline(args.points, sample)
        except Exception as e:  # the oracle is test infrastructure: its absence must not hide the GPU number
            cpu = {"value": None, "unit": "registrations/s", "cores": 1, "kind": "port", "sample": f"unavailable: {e}"}

    cpu_batch = cli_e2e = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_batch = cpu_baseline_batch(pairs[:16])
        except Exception as e:
            cpu_batch = {"value": None, "sample": f"unavailable: {e}"}
    if rank == 0 and world == 1 and not args.no_cli:
        try:
            cli_e2e = cli_end_to_end(pairs[:64])
        except Exception as e:
            cli_e2e = {"value": None, "note": f"unavailable: {e}"}

    if rank == 0:
        print(f"[bench] registrations executed by rank 0: {sum(executed)}", file=sys.stderr)
        total = world * n_timed
        line = {
            "metric": "scan-pair registrations/sec, 1M-pt synthetic pairs",
            "value": total / elapsed,
            "unit": "registrations/s",
            "n_gpus": world,
            "steps": n_timed,
            "warmup": args.warmup,
            "requested_steps": args.steps,
            "value_at_requested_steps": {"value": world * req_groups * S / req_window if req_window > 0 else None, "steps": req_groups * S,
                                         "note": "the first ceil(requested_steps / pairs_per_group) group completions of the same timed window: "
                                                 "with fewer steps than registrations in flight this samples less than one round of the pipeline"},
            "ms_per_step": elapsed / n_timed * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "rank_exchange": exchange_note,
            "torch_in_process": "torch" in sys.modules,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Synthetic {args.points}-pt indoor scan pairs, ~30 planes (BASELINE configs[2]), a batch of {args.pairs} DISTINCT pairs per GPU "
                                   "cycled in input order (configs[3]: 64 pairs, seeds 0..63, at N = 1); "
                                   "full registration(T,target,source) of plade.h:58 = plane extraction + registration on clouds in "
                                   "page-locked HOST memory, batch mode (plade_registration_pairs: consecutive pairs of the batch in groups whose plane "
                                   "extraction is one launch sequence; H2D + SoA conversion + bounding boxes of every step inside the timed "
                                   "region, the next group's upload queued under the current group's kernels); "
                                   "timed window = EXACTLY `steps` completions of the full pipeline (steady state, SURVEY 8d); "
                                   "plade_params.orient_normals=1 (planes oriented like their inliers' normals: the generator's "
                                   "Manhattan scenes need it, DESIGN.md section 2); CPU baseline applies the same rule",
                       "points_per_cloud": args.points, "pairs_per_rank": args.pairs, "distinct_pairs_total": args.pairs * world,
                       "lock_step": os.environ.get("PLADE_NO_LOCKSTEP") is None,
                       "groups_in_flight_per_gpu": M, "pairs_per_group": S, "registrations_in_flight_per_gpu": RIF,
                       "inflight_chosen_from_cpu_quota": inflight_auto, "host_wait": args.host_wait,
                       "inflight_for_local_world_8": inflight_for_budget(_cpu_budget(), 8),
                       "parallelism": f"independent pairs sharded over {world} GPU(s), {M} groups of {S} pairs in flight per GPU"},
            "single_registration_latency_ms": latency_ms,
            "single_registration_latency_by_scene_ms": latency_by_scene,
            "registrations_timed": total,
            "registrations_ok": total_ok,
            "results_bit_identical_to_the_pair_alone_rank0": bool(identical),
            "value_by_seed": value_by_seed,
            "parity_vs_reference_solver": parity,
            "closed_form_mode_rank0": closed_leg,
            "pair_generation_seconds": t_gen,
            "max_frobenius_vs_ground_truth_rank0": max(errs) if errs else None,
            "host_rank0": {"cpu_seconds_per_step": (cpu1 - cpu0) / ((lead_groups + timed_groups + M) * S),
                           "busy_host_threads_avg": (cpu1 - cpu0) / max(span, 1e-9),
                           "cpu_budget": _cpu_budget(),
                           "cgroup_throttled_periods": (thr1[0] - thr0[0]) if thr0 and thr1 else None,
                           "cgroup_throttled_usec": (thr1[1] - thr0[1]) if thr0 and thr1 else None},
            "host_buffers_rank0": host_leg,
            "resident_rank0": resident_leg,
            "default_mode_rank0": default_mode,
            "pipeline": {"lead_in_steps": lead_groups * S, "timed_steps": n_timed, "requested_steps": args.steps, "tail_steps": M * S,
                         "groups_in_flight": M, "pairs_per_group": S,
                         "timing": "value = timed_steps / (completion of group #(lead_in + timed) - completion of group #lead_in), per rank, MAX of "
                                   "the window over ranks: exactly `steps` registrations complete inside the window with the pipeline full on both "
                                   "sides; barrier + device_sync() before the first and after the last step of the run.  steps = max(requested, "
                                   "32 x registrations in flight) in whole groups",
                         "occupancy_value": world * n_timed / occ_elapsed if occ_elapsed > 0 else None,
                         "occupancy_note": "secondary estimator (round 3's `value`): groups in flight x pairs per group / mean time a timed "
                                           "group occupied its worker (Little's law)"},
            "roofline": roofline,
            "roofline_sort": roofline_sort,
            "dominant_by_gpu_time": dominant,
            "cpu_baseline": cpu,
            "cpu_baseline_batch": cpu_batch,
            "cli_end_to_end": cli_e2e,
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
    if comm is not None:
        comm.close()
    if boot is not None and boot is not comm:
        boot.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
