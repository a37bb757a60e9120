#!/usr/bin/env python3
"""bench.py — ICP iterations/s (+ occupancy cells ray-cast/s) on N MI355X GPUs.

A *step* is one pass of the hot path over one batch of synthetic 2 048-beam scan
pairs resident in HBM: per pair exactly what the reference's ``ICP()`` call does
(voxel filter of both scans, target normals, point-to-line ICP to convergence
with the config.yaml parameters of BASELINE config 2), ``--pairs-per-gpu`` pairs
per GPU (the batched loop-closure shape of config 5), followed by the all_gather
of the result records.  The main line is weak scaling (every rank owns its own
pairs); BASELINE config 5 as written — 512 candidate pairs in all — rides on the
same line as ``config5_512`` (one GPU) / ``strong_512`` (N GPUs, 512/N pairs each).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3      # starts 8 child ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "iterative-closest-point-avmi_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md; 6 290 GB/s measured copy)
FP64_VALU_PEAK_TOPS = 39.3       # 78.6 TFLOP/s vector FP64 counts an FMA as 2; the NN loop issues plain ops
ICP_KW = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n_gpus, argv, port):
    """The child command of a plain `python bench.py --gpus N` (N > 1): one rank per GPU under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """Started without WORLD_SIZE but asked for several GPUs: start the ranks as CHILD processes — before this process
    has imported torch or touched the GPU, and never by exec — relay rank 0's JSON line and return the children's code."""
    import subprocess
    cmd = launch_command(n_gpus, argv, free_port())
    if os.environ.get("ICPMI_BENCH_DRYLAUNCH") == "1":          # CPU test hook: show the decision, start nothing
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    env = dict(os.environ)
    # The image exports HSA_ENABLE_IPC_MODE_LEGACY=0 (its host driver only supports dmabuf IPC, without which RCCL's
    # cross-process buffer sharing fails with hipIpcGetMemHandle: invalid argument — the task environment's statement, NOT
    # verified by this repo: no multi-rank RCCL run has been recorded yet); kept when set, defaulted when not.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for ln in proc.stdout:                                       # the JSON line to stdout, everything else to stderr
        (sys.stdout if ln.lstrip().startswith('{"metric"') else sys.stderr).write(ln)
        sys.stdout.flush()
    return proc.wait()


def csrc_signature():
    """sha256 over the kernel sources the loaded library was built from (names + bytes, sorted): what a committed PMC
    summary must carry to be quoted as this build's `roofline.traffic`."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "iterative-closest-point-avmi_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def latest_profile(suffix):
    """The newest committed profiles/rNN_<suffix> (rounds sort by name)."""
    d = os.path.join(REPO, "profiles")
    names = sorted(f for f in os.listdir(d) if f.endswith(suffix) and f[:1] == "r" and f[1:3].isdigit() and f[3] == "_") if os.path.isdir(d) else []
    return os.path.join(d, names[-1]) if names else os.path.join(d, "r00_" + suffix)


PMC_TRAFFIC = latest_profile("pmc_traffic.json")
PMC_MIX = latest_profile("pmc_instruction_mix.json")


def batch_launches(B):
    """kernel name -> workgroups of the launches one batch of B pairs makes (csrc/icp2.hip launch_icp2)."""
    grids = {"icp2_fused_kernel": B, "icp2_far_kernel": min(B, 256)}          # (the continuation of pairs that start far: none here, it looks and leaves)
    if B >= 1024:
        grids.update({"icp2_resume_kernel": max(256, B // 8), "icp2_resume_rest_kernel": 256, "icp2_wide_kernel": min(B, max(256, B // 32))})
    return grids


def pmc_issue(workgroups):
    """Vector-issue roofline of a batch from the committed SQ-counter summary: the share of the chip's vector issue cycles
    the launches of a batch used, weighted by their busy cycles; (None, why) when the summary is stale or absent."""
    try:
        doc = json.load(open(PMC_MIX))
    except (OSError, ValueError):
        return None, "no committed SQ-counter summary"
    if doc.get("csrc_sha256") != csrc_signature():
        return None, (f"profiles/{os.path.basename(PMC_MIX)} was collected on csrc {doc.get('csrc_sha256')}, this build is "
                      f"{csrc_signature()}: stale, not quoted")
    act = busy = 0.0
    per = {}
    for name, grid in batch_launches(workgroups).items():
        for k, v in doc.get("kernels", {}).items():
            if (name + "<" in k or name + " [" in k) and f"[{grid} workgroups]" in k and v.get("SQ_BUSY_CYCLES"):
                act += v["SQ_ACTIVE_INST_VALU"]
                busy += v["SQ_BUSY_CYCLES"]
                per[k.replace("void icpmi::", "")] = {"valu_issue_frac": v["derived"].get("valu_issue_frac"),
                                                       "active_lanes": v["derived"].get("active_lanes_per_valu_instruction")}
    if not busy:
        return None, "kernel/grid not in the committed SQ-counter summary"
    return {"frac": round(act * 4.0 / (1024.0 * busy / 32.0), 4), "per_launch": per,
            "what": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines) over the launches of a "
                    "batch: the share of the chip's vector-instruction issue cycles used (1.0 = a vector instruction issued on "
                    "every SIMD in every cycle the kernel was busy)"}, f"profiles/{os.path.basename(PMC_MIX)} (same csrc signature)"


def pmc_traffic(workgroups):
    """(bytes per launch, source note) from the committed rocprofv3 --pmc summary, or (None, why) when that summary was
    collected on other kernel sources than the ones this library was built from (a stale figure is refused)."""
    try:
        doc = json.load(open(PMC_TRAFFIC))
    except (OSError, ValueError):
        return None, "no committed PMC summary"
    if doc.get("csrc_sha256") != csrc_signature():
        return None, (f"profiles/{os.path.basename(PMC_TRAFFIC)} was collected on csrc {doc.get('csrc_sha256')}, this build is "
                      f"{csrc_signature()}: stale, not quoted")
    # A batch is up to five launches of the fused ICP code (csrc/icp2.hip: every pair up to 12 iterations on `workgroups`
    # workgroups; the pairs still running, one per workgroup of an eighth as many, and whatever that leaves on 256 more; the
    # pairs with wide clouds, on a thirty-second): the bytes of the batch are their sum.  Below 1 024 pairs: one launch.
    grids = batch_launches(workgroups)
    key = [k for k in doc.get("kernels", {}) for name, grid in grids.items()
           if (name + "<" in k or name + "(" in k) and f"[{grid} workgroups]" in k]
    if not key:
        return None, "kernel/grid not in the committed PMC summary"
    total = sum(doc["kernels"][k]["hbm_bytes_per_launch"] for k in key)
    return total, (f"profiles/{os.path.basename(PMC_TRAFFIC)} (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, summed over the "
                   f"{len(key)} launch(es) of a batch; same csrc signature)")


class Leg:
    """One timed leg of the hot path: a resident IcpBatch of B pairs per rank, K steps, barrier + synchronize on both
    sides, MAX over ranks; every step = voxel x2 + prepare + fused ICP + (world > 1) the all_gather of the results."""

    def __init__(self, torch, dist, IcpBatch, gather_results, synth, B, rank, world, red_dev, seed0):
        self.torch, self.dist, self.gather, self.B, self.rank, self.world, self.red_dev = torch, dist, gather_results, B, rank, world, red_dev
        self.srcs, self.tgts = synth.loop_closure_batch(B, seed0=seed0 + 100003 * rank)
        self.batch = IcpBatch(self.srcs + self.tgts, np.arange(B), np.arange(B, 2 * B), **ICP_KW)
        self.n_total = B * world

    def step(self, ev=None):
        res = self.batch.run(events=ev)
        return self.gather(res[:self.B], self.n_total, self.rank, self.world) if self.world > 1 else res

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup):
        torch = self.torch
        for _ in range(warmup):
            self.step()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        self.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            out = self.step(events[k])
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=self.red_dev)
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        from icpmi import _lib
        res = out.cpu().numpy()                               # all pairs of all ranks (gathered) or the local batch
        self.iters_per_step = float(res[:self.n_total, _lib.RES_ITERS].sum())
        settled = res[:self.n_total, _lib.RES_STATUS] == _lib.ST_CONVERGED
        self.settled_pairs = int(settled.sum())
        self.settled_iters = float(res[:self.n_total, _lib.RES_ITERS][settled].sum())
        self.elapsed, self.steps = elapsed, steps
        self.value = self.iters_per_step * steps / elapsed
        self.k_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        return self

    def roofline(self):
        """Dominant kernel (fused ICP) of rank 0's launch: algorithmic bytes / live HIP-event time."""
        from icpmi import _lib
        b, B = self.batch, self.B
        local = b.results.cpu().numpy()[:B]
        cnt = b.vox.cnt.cpu().numpy()
        self.N = cnt[b.pair_src_host].astype(np.float64)
        self.M = cnt[b.pair_tgt_host].astype(np.float64)
        it = local[:, _lib.RES_ITERS]
        alg_bytes = float((it * (28.0 * self.N + 16.0 * self.M)).sum())   # 16N+16M read + 12N written per pair-iteration
        achieved = alg_bytes / (self.k_ms * 1e-3) / 1e9
        name = ("icp2_fused_kernel (+ icp2_resume_kernel / icp2_wide_kernel: the launches of one batch; fused ICP, "
                "sorted-sweep search)") if b.fast else "icp_fused_kernel"
        # `bound`: what limits the kernel — its vector instruction issue, not HBM (the pair stays on chip for all its
        # iterations).  achieved / peak / frac price the ALGORITHMIC bytes against HBM as SURVEY 8d defines them (the figure
        # BASELINE's target is stated in); `issue` is the roofline of the actual limiter, from the committed SQ counters.
        r = {"kernel": name, "bound": "valu_issue" if b.fast else "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "kernel_ms": round(self.k_ms, 4),
             "algorithmic_bytes_per_launch": alg_bytes, "pair_iterations_per_launch": float(it.sum()),
             "note": "algorithmic bytes = sum over pairs of iterations x (28 N + 16 M), N/M rows after voxel filtering; the "
                     "kernel keeps a pair on chip for all its iterations, so it is bound by its instruction stream "
                     "(VALU + divergence), not by HBM: see `issue`, `traffic` and the committed SQ-counter summary under profiles/"}
        if b.fast:
            r["issue"], r["issue_source"] = pmc_issue(B)
            r["traffic"], r["traffic_source"] = pmc_traffic(B)
            if r["traffic"]:
                r["traffic_GBps"] = round(r["traffic"] / (self.k_ms * 1e-3) / 1e9, 1)
                r["traffic_frac"] = round(r["traffic"] / (self.k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            r["limiter"] = ("vector instruction issue (`issue`): `frac` prices ALGORITHMIC bytes, most "
                            "of which never cross HBM — the pair stays on chip for all its iterations; `traffic_frac` is the measured "
                            "HBM utilisation")
        return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=16384,
                    help="scan pairs resident per GPU (weak scaling); the few pairs that run to max_iterations leave most "
                         "CUs idle at the end of a launch, so throughput grows with the batch")
    ap.add_argument("--pairs-total", type=int, default=0,
                    help="strong scaling as the MAIN line: this many pairs in all, split over the GPUs (BASELINE config 5 "
                         "uses 512; the default line carries that case as config5_512 / strong_512 anyway)")
    ap.add_argument("--raycast-scans", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raycast", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the main line (no single-pair, submap, ... legs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from icpmi import _lib, synth
    from icpmi.batch import IcpBatch
    from icpmi.dist import gather_results

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # ICPMI_BENCH_REHEARSE=1: every rank on cuda:0 with gloo (collectives staged through the host) — only to
    # rehearse the multi-rank code path on a one-GPU box; its numbers say nothing about xGMI scaling
    rehearse = os.environ.get("ICPMI_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if rehearse else dev       # where the scalars of an all_reduce live
    backend = None
    if world > 1:
        backend = "gloo" if rehearse else "nccl"
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ── main line: distinct pairs per rank, inputs resident in HBM before timing ──
    B = args.pairs_per_gpu
    if args.pairs_total:
        if args.pairs_total % world:
            raise SystemExit("--pairs-total must be a multiple of the number of GPUs")
        B = args.pairs_total // world
    mk = lambda b, seed0: Leg(torch, dist, IcpBatch, gather_results, synth, b, rank, world, red_dev, seed0)
    leg = mk(B, 1000).run(args.steps, args.warmup)
    roofline = leg.roofline()
    srcs, tgts = leg.srcs, leg.tgts

    line = {"metric": "icp_iterations_per_sec", "value": round(leg.value, 1), "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(leg.elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if args.pairs_total else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config 2 scan pairs (2048-beam room scans, point_to_line ICP, voxel 0.04, "
                                   "normal_k 12, thr 1e-10) batched as in config 5; throughput batch of "
                                   f"{B} resident pairs per GPU (config 5 itself, 512 pairs: see config5_512 / strong_512 and "
                                   "config5_run_icp_pair_512)",
                       "candidate_poses": "target pose = source pose + (d cos a, d sin a, yaw): d ~ U(0, 0.6 m), a ~ U(-pi, pi), "
                                          "yaw ~ U(-6, 6 deg) — offsets ICP converges from without pre-alignment "
                                          "(icpmi.synth.loop_closure_batch defaults); SURVEY 8d's 3 m / 20 deg candidates are "
                                          "the config5_run_icp_pair_512 leg, which pre-aligns as the reference does",
                       "pairs_per_gpu": B, "pairs_total": leg.n_total,
                       "mean_points_after_voxel": [round(float(leg.N.mean()), 1), round(float(leg.M.mean()), 1)],
                       "iterations_per_step": leg.iters_per_step, "parallelism": f"pairs sharded over {world} GPU(s)",
                       "includes": "voxel_downsample x2 + estimate_normals_2d + ICP loop + result all_gather"},
            "pairs_per_sec": round(leg.n_total * args.steps / leg.elapsed, 1),
            "converging_pairs_only": {"pairs": leg.settled_pairs, "of": leg.n_total,
                                      "iterations_per_sec": round(leg.settled_iters * args.steps / leg.elapsed, 1),
                                      "note": "iterations of the pairs that met the error threshold, over the same wall time; the "
                                              "others run to max_iterations (limit cycles of the point_to_line step, in the "
                                              "reference too) and make up the rest of `value`"},
            "roofline": roofline}
    if world > 1:
        line["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                              "library": "RCCL over xGMI" if backend == "nccl" else "gloo (rehearsal)",
                              "what": "one all_gather_into_tensor of 128-B result records per step, inside the timed region",
                              "status": "multi-rank RCCL path: exercised by gloo world-2 CPU tests and a world-of-one RCCL run on "
                                        "one GPU (tests); no multi-GPU run had been recorded when this bench was written"}

    # ── BASELINE config 5 as written: 512 candidate pairs in all (one GPU: one batch; N GPUs: 512/N pairs each +
    #    the all_gather) — strong scaling, same K steps, same barriers ──
    if not args.pairs_total and 512 % world == 0 and not (args.no_extras and world == 1):
        c5 = mk(512 // world, 5000).run(args.steps, args.warmup)
        r5 = c5.roofline()
        obj = {"workload": "BASELINE config 5: 512 candidate scan pairs in all, point_to_line ICP as config 2 (ICP only; "
                           "candidate poses as the main line)",
               "pairs_total": 512, "pairs_per_gpu": 512 // world, "value": round(c5.value, 1), "unit": "iterations/s",
               "ms_per_step": round(c5.elapsed / args.steps * 1e3, 4), "iterations_per_step": c5.iters_per_step,
               "scaling": "strong", "roofline": r5}
        obj["pairs_per_sec"] = round(512 * args.steps / c5.elapsed, 1)
        line["config5_512" if world == 1 else "strong_512"] = obj
        # ... and as the reference runs it: _run_icp_pair per candidate = rotation search + ICP (slam.py:575-579 -> 53-98)
        line["config5_run_icp_pair_512" if world == 1 else "strong_run_icp_pair_512"] = bench_run_icp_pair(
            torch, dist, synth, rank, world, red_dev, args.steps, args.warmup, c5, 0.6, 6.0)
        # ... and on SURVEY 8d's candidate geometry (config.yaml:70): within 3 m and 20 degrees
        line["config5_run_icp_pair_512_3m_20deg" if world == 1 else "strong_run_icp_pair_512_3m_20deg"] = bench_run_icp_pair(
            torch, dist, synth, rank, world, red_dev, max(3, args.steps // 4), 1, c5, 3.0, 20.0)
        if world > 1:
            line["weak"] = {"pairs_per_gpu": B, "value": line["value"], "ms_per_step": line["ms_per_step"]}

    if rank == 0 and world == 1 and not args.no_extras:
        # single-pair latency (config 2 exactly as the reference calls it, one pair)
        one = IcpBatch([srcs[0], tgts[0]], [0], [1], **ICP_KW)
        lat = timed(torch, one.run)
        it1 = float(one.results.cpu().numpy()[0, _lib.RES_ITERS])
        line["single_pair"] = {"ms_per_icp": round(lat * 1e3, 4), "iterations": it1,
                               "iterations_per_sec": round(it1 / lat, 1)}
        line["pipelined"] = bench_pipelined(torch, IcpBatch, _lib, srcs, tgts, max(args.steps, 8))
        line["nn_exhaustive"] = bench_nn_exhaustive(torch, leg.batch, _lib)
        line["submap"] = bench_submap(torch, synth, not args.no_cpu_baseline)
        line["scan_pair_host_api"] = bench_host_api(srcs[0], tgts[0], not args.no_cpu_baseline)
        if not args.no_raycast:
            line["raycast"] = bench_raycast(torch, synth, args.raycast_scans, not args.no_cpu_baseline)
        line["pose_graph"] = bench_pose_graph(torch, not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(srcs, tgts)
    if world > 1 and not args.no_raycast:
        line["map_replay_sharded"] = bench_replay_sharded(torch, dist, synth, args.raycast_scans, rank, world, dev, red_dev)
    if rehearse:
        line["data"] = "synthetic (REHEARSAL: all ranks share cuda:0 over gloo; not a scaling measurement)"
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_run_icp_pair(torch, dist, synth, rank, world, red_dev, steps, warmup, icp_only, max_offset, max_yaw_deg):
    """BASELINE config 5 as slam.py runs it: 512 loop-closure candidates of ONE current scan, each matched by
    _run_icp_pair (slam.py:53-98) = rotation_search (config.yaml:37-39: voxel 0.15, 1.5 / 0.1 degree steps) + point_to_line
    ICP from its result.  Candidate poses within max_offset metres / max_yaw_deg degrees of the current scan: (0.6, 6) = the
    poses of the ICP-only legs (the like-for-like line); (3, 20) = SURVEY 8d / config.yaml:70.
    512 / world pairs per rank, the result all_gather inside the timed region."""
    from icpmi import _lib
    from icpmi.dist import RunIcpPairSharded
    n_total = 512
    B = n_total // world
    srcs, tgts = synth.loop_closure_batch(n_total, seed0=7000, shared_source=True, max_offset=max_offset, max_yaw_deg=max_yaw_deg)
    # the product's entry point for this path: candidate i on rank i mod world, one all_gather of the records (icpmi/dist.py)
    feat = dict(rotation_voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1)
    job = RunIcpPairSharded(srcs[0], tgts, ICP_KW, feat, max_rows_hint=1024)
    b = job.batch

    def step(ev=None):
        return job.run(events=ev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        out = step(events[k])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    res = out.cpu().numpy()[:n_total]
    rec = b.search.records.cpu().numpy()[:B]
    iters = float(res[:, _lib.RES_ITERS].sum())
    search_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    icp_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))         # prepare + fused ICP of the ICP half
    n_s, n_t = rec[:, 0], rec[:, 1]
    n_coarse, exact = len(b.search.tables.coarse), rec[:, 12] + rec[:, 13]
    # algorithmic bytes of the search (SURVEY 8d, K1 per NN pass: 28 N + 16 M): over the passes the launch MAKES (the
    # angles it scores exactly) for `frac`; over the passes the REFERENCE makes (every coarse angle and the winner's fine
    # grid) as `reference_passes` — the launch proves most of those cannot win and never makes them, that is its point
    alg = float((exact * (28.0 * n_s + 16.0 * n_t)).sum())
    alg_ref = float(((n_coarse + rec[:, 8]) * (28.0 * n_s + 16.0 * n_t)).sum())
    ms_step = elapsed / steps * 1e3
    note = ("same candidate poses as config5_512: the like-for-like cost of adding the pre-alignment" if max_offset < 1.0 else
            "SURVEY 8d poses; the rotation search (the reference's algorithm, its result reproduced bit for bit) leaves "
            "`registered_fraction` of these candidates within ICP's reach — the others start metres off and never settle.  The "
            "first launch hands such a pair (mean squared error of its first step above 1 m^2) to "
            "icp2_far_kernel, whose searches give up a long walk for a box hierarchy over the sort order (same matches, same "
            "bits); one pair that circles for 150 iterations with a quarter of its rows searching sets this leg's time")
    return {"workload": "BASELINE config 5 as slam.py:575-579 runs it: 512 candidates of one current scan (target pose within "
                        f"{max_offset} m / {max_yaw_deg} deg: d ~ U(0, {max_offset}), direction uniform, yaw ~ U(-{max_yaw_deg}, {max_yaw_deg})), "
                        "each _run_icp_pair = rotation_search (voxel 0.15, 240 coarse + ~31 fine angles) "
                        "+ point_to_line ICP from its R, t; one chain of launches, no host round trip",
            "note": note,
            "pairs_total": n_total, "pairs_per_gpu": B, "ms_per_step": round(ms_step, 4),
            "pairs_per_sec": round(n_total * steps / elapsed, 1), "icp_iterations_per_step": iters,
            "icp_iterations_per_sec": round(iters * steps / elapsed, 1),
            "registered_fraction": round(float((res[:, _lib.RES_ERR] < 0.05).mean()), 4),
            "rotation_search_ms": round(search_ms, 4), "icp_ms": round(icp_ms, 4),
            "vs_icp_only_step": round(ms_step / (icp_only.elapsed / icp_only.steps * 1e3), 3),
            "rotation_search": {"kernel": "rotation_search_batch_kernel (+ voxel filter, means, target order)",
                                "filtered_points": [round(float(n_s.mean()), 1), round(float(n_t.mean()), 1)],
                                "coarse_angles": n_coarse, "fine_angles": int(rec[:, 8].max()),
                                "angles_scored_exactly_mean": round(float(exact.mean()), 1),
                                "angles_scored_exactly_max": int(exact.max()),
                                "roofline": {"bound": "valu_issue", "achieved": round(alg / (search_ms * 1e-3) / 1e9, 3),
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(alg / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                             "algorithmic_bytes_per_launch": alg,
                                             "reference_passes": {"algorithmic_bytes": alg_ref,
                                                                  "equivalent_GBps": round(alg_ref / (search_ms * 1e-3) / 1e9, 1)},
                                             "note": "bytes of the nearest-neighbour passes the launch makes (one per angle it "
                                                     "scores exactly); the reference makes one per angle of both sweeps "
                                                     "(`reference_passes`) — a distance field of the target proves most angles "
                                                     "cannot win and they are never scored.  Bound by its instruction stream and "
                                                     "LDS latency, not by HBM (the pair lives in LDS)"}},
            "reference_python_ms_per_pair": {"run_icp_pair": 149.0, "note": "survey container, 1 core (BASELINE.md section 2)"}}


def timed(torch, fn, warm_s=0.1, min_s=0.15):
    """Seconds per call of fn (which enqueues GPU work): at least warm_s of untimed calls first — the short legs run
    after seconds of host-side setup, and a GPU that has idled takes a while to clock up again — then at least
    min_s of timed calls, synchronised at both ends."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= min_s:
                break
    return (time.perf_counter() - t0) / n


def bench_pipelined(torch, IcpBatch, _lib, srcs, tgts, steps):
    """Small batches (the 512 candidates of BASELINE config 5) one at a time and with 2 / 3 in flight on separate HIP
    streams: the few pairs of a batch that run to max_iterations (alone on their CUs for half of the kernel) then
    overlap the bulk of the next batch.  Not `value`: that line times one large batch at a time."""
    B = min(512, len(srcs))
    out = {"note": f"distinct resident batches of {B} pairs alternating over HIP streams; every step is the full hot path"}
    batches = []
    for depth in (1, 2, 3):
        while len(batches) < depth:
            lo = (len(batches) * B) % max(len(srcs) - B + 1, 1)
            batches.append(IcpBatch(srcs[lo:lo + B] + tgts[lo:lo + B], np.arange(B), np.arange(B, 2 * B), **ICP_KW))
        streams = [torch.cuda.Stream() for _ in range(depth)]
        for k in range(depth):                                   # warm-up, also gives each batch's iteration count
            with torch.cuda.stream(streams[k]):
                batches[k].run()
        torch.cuda.synchronize()
        iters = [float(b.results.cpu().numpy()[:B, _lib.RES_ITERS].sum()) for b in batches[:depth]]
        t0 = time.perf_counter()
        for k in range(steps * depth):
            with torch.cuda.stream(streams[k % depth]):
                batches[k % depth].run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"in_flight_{depth}"] = {"iterations_per_sec": round(sum(iters) * steps / dt, 1),
                                     "ms_per_step": round(dt / (steps * depth) * 1e3, 4)}
    return out


def bench_nn_exhaustive(torch, batch, _lib):
    """K1, the LDS-tiled exhaustive NN kernel north_star names, on the same voxel-filtered pairs (one NN pass each)."""
    import ctypes as C
    from icpmi.batch import _ptr, _stream
    L = _lib.lib()
    B, stride = batch.B, batch.max_src_n
    idx = torch.empty((B, stride), dtype=torch.int32, device=batch.vox.pts.device)
    dist = torch.empty((B, stride), dtype=torch.float64, device=batch.vox.pts.device)

    def go():
        _lib.check(L.icpmi_nn_batch(_ptr(batch.vox.pts), _ptr(batch.vox.off), _ptr(batch.vox.cnt), _ptr(batch.pair_src),
                                    _ptr(batch.pair_tgt), B, stride, 2, _ptr(idx), _ptr(dist), stride, _stream()), "nn")
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    cnt = batch.vox.cnt.cpu().numpy()
    N = cnt[batch.pair_src_host].astype(np.float64)
    M = cnt[batch.pair_tgt_host].astype(np.float64)
    byts, evals = float((28.0 * N + 16.0 * M).sum()), float((N * M).sum())
    gbs, tops = byts / (ms * 1e-3) / 1e9, evals * 6 / (ms * 1e-3) / 1e12
    return {"kernel": "nn_batch_kernel<2,S>", "pairs": B, "kernel_ms": round(ms, 4),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": byts},
            "valu": {"distance_evals_per_launch": evals, "fp64_ops_per_eval": 6, "achieved_Tops": round(tops, 3),
                     "peak_Tops": FP64_VALU_PEAK_TOPS, "frac": round(tops / FP64_VALU_PEAK_TOPS, 4)}}


def bench_host_api(src, tgt, with_cpu):
    """What slam.py's _run_icp_pair does per loop-closure candidate (slam.py:53-98, config.yaml:19-39), through the
    drop-in NumPy-in / NumPy-out functions: rotation_search (240 + ~30 angles) then point_to_line ICP with its
    result as the initial guess.  PCIe and host round trips included — never the headline value."""
    from utilities import features, icp as uicp
    uicp.VERBOSE = features.VERBOSE = False
    kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, method="point_to_line", normal_k=12)

    def pair():
        R0, t0, _ = features.rotation_search(src, tgt, voxel_size=0.15, angle_step_coarse=1.5, angle_step_fine=0.1)
        return uicp.ICP(src, tgt, R_init=R0, t_init=t0, **kw)

    import torch
    rs_ms = timed(torch, lambda: features.rotation_search(src, tgt, voxel_size=0.15, angle_step_coarse=1.5,
                                                          angle_step_fine=0.1)) * 1e3
    pair_ms = timed(torch, pair) * 1e3
    out = {"rotation_search_ms": round(rs_ms, 3), "run_icp_pair_ms": round(pair_ms, 3),
           "reference_python_ms": {"rotation_search": 73.0, "run_icp_pair": 149.0,
                                   "note": "survey container, 1 core (BASELINE.md section 2), not this box"}}
    if with_cpu:
        import oracle
        t0 = time.perf_counter()
        for _ in range(5):
            R0, tt0, _ = oracle.rotation_search(src, tgt, 0.15, 1.5, 0.1)
            oracle.icp(src, tgt, R_init=R0, t_init=tt0, **kw)
        out["cpu_baseline_run_icp_pair_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    return out


def bench_submap(torch, synth, with_cpu):
    """BASELINE config 3: one 2 048-beam scan against a ~10 k-point rolling submap (point_to_point, max_corr_dist 1.5,
    initial guess = truth perturbed by (0.05, -0.04, 1 deg)), plus the _build_submap voxel filter of 90 x 2 048 points."""
    from icpmi.batch import CloudSet, IcpBatch, voxel_downsample_set
    segs = synth.maze_segments()
    poses = synth.trajectory(90, step=0.35)
    allpts = np.vstack([synth.to_world(synth.scan(p, 500 + i, segs=segs), p) for i, p in enumerate(poses)])
    cs = CloudSet.from_numpy([allpts])
    out = voxel_downsample_set(cs, 0.04)
    vox_ms = timed(torch, lambda: voxel_downsample_set(cs, 0.04, out=out)) * 1e3
    sub = out.to_numpy()[0]
    pose = poses[-1]
    cur = synth.scan(pose, 999, segs=segs)
    th = pose[2] + np.deg2rad(1.0)
    R0 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t0v = np.array([pose[0] + 0.05, pose[1] - 0.04])
    kw = dict(error_threshold=1e-10, max_iterations=150, voxel_size=0.04, R_init=R0, t_init=t0v,
              method="point_to_point", max_corr_dist=1.5)
    b = IcpBatch([cur, sub], [0], [1], **kw)
    ms = timed(torch, b.run) * 1e3
    it = float(b.results.cpu().numpy()[0, 14])
    res = {"workload": f"config 3: 2048-beam scan vs {len(sub)}-point submap, point_to_point, max_corr_dist 1.5",
           "ms_per_icp": round(ms, 4), "iterations": it, "iterations_per_sec": round(it / (ms * 1e-3), 1),
           "build_submap_voxel_ms": round(vox_ms, 4), "build_submap_points_in": int(len(allpts))}
    if with_cpu:
        import oracle
        t0 = time.perf_counter()
        io = oracle.icp(cur, sub, 1e-10, 150, 0.04, R_init=R0, t_init=t0v, method="point_to_point", max_corr_dist=1.5)[3]
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.voxel_downsample(allpts, 0.04)
        dv = time.perf_counter() - t0
        res["cpu_baseline"] = {"ms_per_icp": round(dt * 1e3, 3), "iterations": io["iters"], "build_submap_voxel_ms": round(dv * 1e3, 3),
                               "cores": 1, "kind": "port"}
    return res


def raycast_workload(synth, n_scans):
    """BASELINE config 4: 2 242 x 2 402 grid @0.05 m, 2 048-beam scans along a short drive -> (grid, origins, hits, cell updates)."""
    from utilities.mapping import OccupancyGrid2D
    p0 = (0.3, -0.2, np.deg2rad(10.0))
    first = synth.to_world(synth.scan(p0, 2), p0)
    b = (first[:, 0].min() - 50, first[:, 0].max() + 50, first[:, 1].min() - 50, first[:, 1].max() + 50)
    g = OccupancyGrid2D(*b, resolution=0.05, p_hit=0.85, p_miss=0.42, log_odds_min=-8.0, log_odds_max=8.0)
    poses = [(0.3 + 0.02 * i, -0.2 + 0.01 * i, np.deg2rad(10.0 + 0.5 * i)) for i in range(n_scans)]
    hits = [synth.to_world(synth.scan(p, 2 + i), p) for i, p in enumerate(poses)]
    org = np.array([[p[0], p[1]] for p in poses])
    # cell updates = in-bounds hit adds + free cells; every ray lies inside this grid (50 m margin)
    cells = 0
    for o, h in zip(org, hits):
        ox, oy = np.floor((o[0] - g.min_x) / 0.05), np.floor((o[1] - g.min_y) / 0.05)
        hx, hy = np.floor((h[:, 0] - g.min_x) / 0.05), np.floor((h[:, 1] - g.min_y) / 0.05)
        cells += int(np.maximum(np.abs(hx - ox), np.abs(hy - oy)).sum()) + len(h)
    return g, org, hits, cells


def bench_replay_sharded(torch, dist, synth, n_scans, rank, world, dev, red_dev):
    """SURVEY §8e map rebuild: every rank replays the whole history into its own band of rows, then one
    all_gather of the bands.  Band boundaries are planned once (untimed); replay + gather are timed."""
    from icpmi import dist as idist
    g, org, hits, cells = raycast_workload(synth, n_scans)
    d_org = torch.from_numpy(org).to(dev)
    d_hits = [torch.from_numpy(h).to(dev) for h in hits]
    bands = idist.row_bands(g.ny, world, idist.row_costs(g.ny, g.min_y, g.resolution, d_org, d_hits))
    idist.replay_scans_sharded(g, d_org, d_hits, bands=bands)     # warm-up (also of the collective)
    reps = 5
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.reset()
        idist.replay_scans_sharded(g, d_org, d_hits, bands=bands)
    dist.barrier(); torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=red_dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    checksum = torch.tensor([float(g.device_log_odds.double().sum().item())], dtype=torch.float64, device=red_dev)
    lo, hi = checksum.clone(), checksum.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return {"workload": f"config 4 grid {g.ny}x{g.nx}, {n_scans} scans x 2048 beams, rows split over {world} GPUs",
            "bands": bands, "cell_updates": cells, "cells_per_sec": round(cells / dt, 1), "ms_per_replay": round(dt * 1e3, 4),
            "includes": "grid reset + band replay on every rank + all_gather of the bands",
            "grids_identical_on_all_ranks": bool(lo.item() == hi.item())}


def bench_pose_graph(torch, with_cpu):
    """utilities/pose_graph.py:83-134 on an odometry chain of 1 000 poses with 20 loop closures (the shape slam.py
    builds): every Gauss-Newton iteration in one launch."""
    from utilities import pose_graph as upg
    upg.VERBOSE = False
    rng = np.random.default_rng(0)
    n, k = 1000, 20
    th = np.cumsum(rng.normal(0.0, 0.05, n)) + np.linspace(0, 4 * np.pi, n)
    xy = np.cumsum(np.stack([0.3 * np.cos(th), 0.3 * np.sin(th)], axis=1), axis=0)
    T = [upg.pose_vec_to_matrix(v) for v in np.column_stack([xy, upg.normalize_angle(th)])]
    est, edges = [T[0]], []
    for i in range(1, n):
        z = upg.relative_transform_vec(T[i - 1], T[i]) + rng.normal(0.0, 0.005, 3)
        est.append(est[-1] @ upg.pose_vec_to_matrix(z))
        edges.append((i - 1, i, z, np.eye(3) * rng.uniform(100, 1e4)))
    for _ in range(k):
        a_, b_ = int(rng.integers(n // 2, n)), int(rng.integers(0, n // 3))
        edges.append((a_, b_, upg.relative_transform_vec(T[a_], T[b_]) + rng.normal(0.0, 0.002, 3), np.eye(3) * 2e4))
    nodes = np.array([upg.pose_matrix_to_vec(t) for t in est])

    def once():
        pg = upg.PoseGraph2D()
        for v in nodes:
            pg.add_node(v)
        for e in edges:
            pg.add_edge(*e)
        t0 = time.perf_counter()
        pg.optimize()
        return time.perf_counter() - t0, pg

    once()
    dt, pg = min((once() for _ in range(3)), key=lambda r: r[0])
    out = {"workload": f"{n} poses, {n - 1} odometry + {k} closure edges", "ms_per_optimize": round(dt * 1e3, 3),
           "iterations": pg.last_info["iterations"], "status": pg.last_info["status"],
           "device_us_per_phase": {a: round(v, 1) for a, v in pg.last_info["phase_us"].items()}}
    if with_cpu:
        from oracle import pose_graph as opg
        ei, ej = np.array([e[0] for e in edges]), np.array([e[1] for e in edges])
        zz, om = np.array([e[2] for e in edges]), np.array([e[3] for e in edges])
        t0 = time.perf_counter()
        ref, it, st, _ = opg.optimize(nodes, ei, ej, zz, om)
        out["cpu_baseline"] = {"ms_per_optimize": round((time.perf_counter() - t0) * 1e3, 1), "iterations": it,
                               "kind": "port", "note": "dense NumPy restatement of the reference (LAPACK on all host cores)",
                               "max_abs_pose_difference": float(np.abs(np.array(pg.nodes) - ref).max())}
    return out


def bench_raycast(torch, synth, n_scans, with_cpu):
    """BASELINE config 4: 2 048-beam scans replayed in order on one GPU (slam.py:271-277 shape)."""
    g, org, hits, cells = raycast_workload(synth, n_scans)
    d_org = torch.from_numpy(org).cuda()
    d_hits = torch.from_numpy(np.concatenate(hits)).cuda()
    off = np.zeros(n_scans + 1, dtype=np.int32)
    np.cumsum([len(h) for h in hits], out=off[1:])
    box = g._cell_box(d_org, d_hits)                             # cell bounds of all rays (one read-back, outside the timing)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.1:                       # warm-up replays (see timed())
        g._apply(d_org, d_hits, off, box=box)
        g.reset()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    g._apply(d_org, d_hits, off, box=box)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    out = {"workload": f"config 4 grid {g.ny}x{g.nx}, {n_scans} scans x 2048 beams replayed in order",
           "cell_updates": cells, "cells_per_sec": round(cells / wall, 1), "ms_per_scan": round(wall / n_scans * 1e3, 5),
           "device_ms_per_scan": round(dev_ms / n_scans, 5),
           "counter_workspace_bytes": int(g._ws.numel()), "grid_bytes": int(g.ny * g.nx * 4),
           "roofline": {"bound": "hbm", "achieved": round((8.0 * cells + 16.0 * 2048 * n_scans) / wall / 1e9, 3),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round((8.0 * cells + 16.0 * 2048 * n_scans) / wall / 1e9 / HBM_PEAK_GBS, 6)}}
    # BASELINE config 4 as written: ONE live update_scan of 2 048 beams (slam.py:552-557) — count + finalise launch
    g.reset()
    one_org, one_hits = d_org[:1].contiguous(), torch.from_numpy(hits[0]).cuda()
    one_off = np.array([0, len(hits[0])], dtype=np.int32)
    ox, oy = np.floor((org[0, 0] - g.min_x) / 0.05), np.floor((org[0, 1] - g.min_y) / 0.05)
    hx, hy = np.floor((hits[0][:, 0] - g.min_x) / 0.05), np.floor((hits[0][:, 1] - g.min_y) / 0.05)
    one_cells = int(np.maximum(np.abs(hx - ox), np.abs(hy - oy)).sum()) + len(hits[0])
    one_box = g._cell_box(one_org, one_hits)
    one_s = timed(torch, lambda: g._apply(one_org, one_hits, one_off, box=one_box))
    one_bytes = 8.0 * one_cells + 16.0 * len(hits[0])
    out["single_scan"] = {"workload": "config 4: one update_scan, 2048 beams, data resident in HBM", "us_per_scan": round(one_s * 1e6, 2),
                          "cell_updates": one_cells, "cells_per_sec": round(one_cells / one_s, 1),
                          "roofline": {"bound": "hbm", "achieved": round(one_bytes / one_s / 1e9, 3), "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": round(one_bytes / one_s / 1e9 / HBM_PEAK_GBS, 6),
                                       "note": "ONE launch (ray_owner_kernel: a workgroup per 64 x 64 tile of the scan's box takes the beams that cross it, "
                                               "counts in LDS and applies hit / miss replay and clip to its own cells; no counter region): "
                                               "bound by the origin tiles' workgroups (every beam crosses them), ~14 us of kernel"}}
    # ... and the call slam.py:557 makes: NumPy origin and hits in, through OccupancyGrid2D.update_scan (upload included)
    g.reset()
    host_s = timed(torch, lambda: g.update_scan(org[0], hits[0]))
    out["host_api_update_scan_us"] = round(host_s * 1e6, 2)
    g.reset()
    if with_cpu:
        import oracle
        ref = np.zeros((g.ny, g.nx), dtype=np.float32)
        t0 = time.perf_counter()
        n = 0
        for o, h in list(zip(org, hits))[:20]:
            n += oracle.grid_update_scan(ref, g.min_x, g.min_y, 0.05, o, h, g.l_hit, g.l_miss, -8.0, 8.0)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n / dt, 1), "unit": "cells/s", "cores": 1, "kind": "port",
                               "sample": "first 20 scans, C oracle (includes the reference's whole-grid clip per scan)"}
    return out


def cpu_baseline(srcs, tgts):
    """The C oracle (k-d tree search, same algorithm as the reference's ICP) on this box's host, one core."""
    import oracle
    oracle.icp(srcs[0], tgts[0], **ICP_KW)
    t0 = time.perf_counter()
    iters = n = 0
    while time.perf_counter() - t0 < 12.0:                  # a bounded ~12 s sample: the pair list, repeated
        for s, t in zip(srcs, tgts):
            iters += oracle.icp(s, t, **ICP_KW)[3]["iters"]
            n += 1
            if time.perf_counter() - t0 > 12.0:
                break
    dt = time.perf_counter() - t0
    out = {"value": round(iters / dt, 1), "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": f"{n} ICP calls on the same scan pairs, {iters} iterations, {dt:.1f} s, oracle/icp_oracle.c (k-d tree NN)",
           "host_cpus": os.cpu_count()}
    # the same pairs over all host cores: independent pairs on a thread pool — ctypes drops the GIL
    # for the duration of each C call, so threads scale like processes without copying the scans
    from concurrent.futures import ThreadPoolExecutor
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                                               # the box's CPU share for one GPU
    pairs = [(srcs[i % len(srcs)], tgts[i % len(tgts)]) for i in range(cores * 16)]
    t0 = time.perf_counter()
    it_all = todo = 0
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < 8.0:                                    # rounds of 16 pairs per thread, about 8 s
            it_all += sum(ex.map(lambda p: oracle.icp(p[0], p[1], **ICP_KW)[3]["iters"], pairs))
            todo += len(pairs)
    dt_all = time.perf_counter() - t0
    out["all_cores"] = {"value": round(it_all / dt_all, 1), "unit": "iterations/s", "cores": cores, "kind": "port",
                        "sample": f"{todo} ICP calls on a {cores}-thread pool, {it_all} iterations, {dt_all:.1f} s"}
    return out


if __name__ == "__main__":
    main()
