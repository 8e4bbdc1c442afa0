#!/usr/bin/env python3
"""Headline benchmark: upsampled query-points/s on the BASELINE.json config
"Synthetic sphere 5000 pts, 4x upsample, M=48 T=4, 1xMI355X".

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1: one process per GPU over RCCL.  Started by ``torch.distributed.run`` (RANK / WORLD_SIZE in the environment) the
ranks run directly; started as a plain command, this process spawns ``python -m torch.distributed.run --nproc-per-node N``
of itself BEFORE touching the GPU and exits with the child's code.  On a box with fewer than N GPUs the run is a
REHEARSAL (every rank on cuda:0, gloo): it exercises sharding, barriers, the all-gather and the max-over-ranks timing;
the line says "rehearsal": true and its numbers mean nothing.

A step = ONE pass of the hot path over one resident batch of B=4096 query points per GPU:
outer kNN + gather/centre -> fn forward -> normalise -> gather/rotate -> fd forward -> displace
(+ for N > 1 the all-gather of the refined points).  Inputs (cloud, queries, weights) are in HBM
before the timed region.  Weak scaling (default, the BASELINE workload): every rank refines its own 4096 queries per
step.  --scaling strong: a step = the whole 385 582-seed cloud of the same sphere (its real dense-grid seeds), sharded
contiguously over the ranks, one all-gather of the refined cloud per step; the default run also times one such pass as
the extra object "strong_scaling" so that the driver's N = 1, 2, 4, 8 runs give a strong-scaling curve too.

Extra objects on the JSON line:
  roofline      the dominant kernel (fn_edge_chain_kernel<512, 12, .>: block 3 of fn, its whole per-edge chain fn/snn_coder.py:355-389
                in one launch), launched alone through its C-ABI entry between two events on the current stream: achieved =
                issued f16 MFMA TFLOP/s against the 2.5 PFLOP/s dense peak; beside it SURVEY.md 8(d)'s fp32-MFMA model figure,
                the VALU neuron-loop rate against its measured floor, HBM bytes per launch and per step from the committed
                --pmc passes (null when they were not taken on these kernel sources).
  cpu_baseline  the oracle (our CPU restatement, torch-CPU, all host threads) timed on a bounded
                sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
M_PTS = 48
T_STEPS = 4
N_CLOUD = 5000
FN_KW = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=T_STEPS, num_heads=8)
FD_KW = dict(k=32, emb_dims=768, time_steps_enc=T_STEPS, num_heads=8, k_scales=[8, 16, 32, 48])
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md, Chip-level parameters (spec; 6.3 TB/s measured copy)
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense f16 MFMA (spec)
PEAK_F32_MFMA_TFLOPS = 157.3    # f32-input MFMA = the vector rate (MI355X_MICROARCH.md, Matrix cores)


def build_models(dev):
    import sapcu_amd
    from sapcu_amd import testing as T
    gold = os.path.join(ROOT, "tests", "golden")
    bn = {}
    for kind in ("fn", "fd"):
        p = os.path.join(gold, "bn_calib_%s.npz" % kind)
        bn[kind] = dict(np.load(p)) if os.path.exists(p) else None
    fn = sapcu_amd.ImprovedSNNNormalEstimation(**FN_KW)
    fd = sapcu_amd.EnhancedSNNDistanceEstimation(**FD_KW)
    sdn = T.conditioned_state_dict(fn.state_dict(), 0, bn_stats=bn["fn"])
    sdd = T.conditioned_state_dict(fd.state_dict(), 0, bn_stats=bn["fd"])
    fn.load_state_dict(sdn)
    fd.load_state_dict(sdd)
    fn, fd = fn.to(dev), fd.to(dev)
    fn.knn_cache_mode = "fresh"       # every batch computes its own in-patch neighbours (no replay shortcut)
    return fn, fd, sdn, sdd


def csrc_sha256():
    """Hash of the kernel sources (the same function as profiles/pmc_to_json.py): a PMC file is only quoted when it was taken
    on exactly these kernels."""
    import hashlib
    d = os.path.join(ROOT, "c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def recorded_pmc():
    """Newest profiles/r*_pmc.json taken on the current kernel sources, or (None, reason)."""
    cands = sorted([f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json")], reverse=True)
    mine = csrc_sha256()
    for cand in cands:
        try:
            recs = json.load(open(os.path.join(ROOT, "profiles", cand)))
        except (OSError, ValueError):
            continue
        if recs.get("_meta", {}).get("csrc_sha256") == mine:
            return recs, "profiles/" + cand
    return None, ("no profiles/*_pmc.json was taken on these kernel sources (csrc sha256 %s...): re-run profiles/run_profiles.sh" % mine[:12])


# algorithmic HBM bytes per query = 2 x 576 B patches in + 16 B out + ~410 B outer-kNN share
ALGO_BYTES_PER_QUERY = 2 * 576 + 16 + 410
DOMINANT_KERNEL = "fn_edge_chain_kernel<512, 12, true>"      # rocprofv3's name (third argument: 32-bit gather offsets)
NEURON_FLOOR_NS_PER_1000 = 1.51       # profiles/micro/lif_rate.hip on MI355X: the 4-step LIF loop alone, 2 waves per SIMD


def roofline_leg(dev, reps=5):
    """Time the dominant kernel alone: fn_edge_chain_kernel<512, 12, .> (csrc/fn_edge_chain.hip; 40 % of the step's GPU time) — block 3
    of fn, the whole per-edge chain (fn/snn_coder.py:355-389: pe1 -> fc_delta2 -> attn_in -> fc_gamma -> fc_gamma2 ->
    softmax-aggregate) of one 4096-patch batch in ONE launch — through its C-ABI entry, on the current stream between two events.

    Roofline: the matrix pipe.  achieved = ISSUED f16 MFMA flops (3 split-f16 products per algorithmic MAC: 3 x 3 GEMMs x
    2 r d^2) / launch time against the 2.5 PFLOP/s dense f16 peak; SURVEY.md 8(d)'s model (algorithmic flops against the
    157.3 TFLOP/s fp32-MFMA rate) and the VALU side (the kernel's real bound: the 4-step neuron loops, against the measured
    floor of the bare loop) are reported beside it; `traffic` = HBM bytes per launch from the separate --pmc passes."""
    from sapcu_amd import _lib
    lib = _lib.load()
    d, kk, heads, T = 512, FN_KW["k_values"][2], FN_KW["num_heads"], 4
    P = B_PER_GPU * M_PTS
    r = P * kk
    g = torch.Generator(device="cpu").manual_seed(0)
    patch = (torch.randn((B_PER_GPU, M_PTS, 3), generator=g) * 0.05).to(dev)
    idx = torch.stack([torch.randperm(M_PTS, generator=g)[:kk] for _ in range(P)]).to(torch.int32).to(dev)   # any neighbours
    qkv = torch.rand((P, 3 * d), generator=g).to(dev)

    def lin(n, k, gain):
        return ((torch.rand((n, k), generator=g) * 2 - 1) * gain / k ** 0.5).to(dev), (torch.randn(n, generator=g) * 0.4 + 0.6).to(dev)

    def lif():
        return torch.stack([torch.full((d,), 0.9), torch.full((d,), 0.01), torch.full((d,), 0.5), torch.ones(d)]).to(dev)

    wd, bd = lin(d, 3, 20.0)
    (w1, b1), (w2, b2), (w3, b3) = lin(d, d, 2.0), lin(d, d, 2.0), lin(d, d, 4.0)
    ld, l1, l2 = lif(), lif(), lif()
    need = lib.sapcu_fn_edge_chain_workspace_bytes(P, d, kk)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    res = torch.empty((P, d), device=dev)
    ops = [patch.view(P, 3), idx.view(-1), qkv, wd, bd, ld, w1, b1, l1, w2, b2, l2, w3, b3]

    def launch():
        _lib.check(lib.sapcu_fn_edge_chain_f32(_lib.ptr(ops[0]), _lib.ptr(ops[1]), P, M_PTS, d, kk, *[_lib.ptr(t) for t in ops[2:]],
                                               heads, T, _lib.ptr(res), _lib.ptr(ws), need, _lib.current_stream()))

    launch()
    torch.cuda.synchronize()
    assert torch.isfinite(res).all()
    # the entry point also runs seven tiny helper kernels (edge records, weight split / pack: ~60 us against ~18 ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                       # torch's current stream IS the stream the kernels are launched on
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / reps
    flop = 3 * 2.0 * r * d * d                         # algorithmic: three d x d contractions per edge row
    issued = 3 * flop                                  # a_lo.w_hi + a_hi.w_lo + a_hi.w_hi
    elems = 3.0 * r * d                                # neuron elements (4 steps each): pe1, pe, g
    algo_bytes = r * 24.0 + P * 3 * d * 4.0 + P * d * 4.0      # edge records + q|k|v rows in, res out
    recs, src = recorded_pmc()
    traffic, busy, step_traffic = None, None, None
    if recs is not None:
        rec = recs.get(DOMINANT_KERNEL)
        if rec:
            traffic, busy = float(rec["hbm_bytes_per_launch"]), rec.get("mfma_busy_frac")
        st = recs.get("_step")
        if st:
            algo_step = float(ALGO_BYTES_PER_QUERY * B_PER_GPU)
            step_traffic = {"hbm_bytes_per_step": st["hbm_bytes_per_step"], "algorithmic_bytes_per_step": algo_step,
                            "traffic_ratio": round(st["hbm_bytes_per_step"] / algo_step, 1),
                            "note": "counter bytes of one 4096-query step / SURVEY.md 8(d) algorithmic bytes (patches in, results out, "
                                    "kNN share); the step also streams 33 MB of weights"}
    ns_per_1000 = avg_s * 1e9 / (elems / 1000.0)
    return {"bound": "mfma", "achieved": round(issued / avg_s / 1e12, 1), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(issued / avg_s / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
            "traffic": traffic, "traffic_source": src, "kernel": DOMINANT_KERNEL,
            "avg_launch_ms": round(avg_s * 1e3, 4), "launches_timed": reps,
            "flops_per_launch": {"algorithmic": flop, "issued_f16": issued},
            "algorithmic_tflops": round(flop / avg_s / 1e12, 2),
            "fp32_mfma_model": {"peak_tflops": PEAK_F32_MFMA_TFLOPS, "frac": round(flop / avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                "note": "SURVEY.md 8(d): algorithmic flops against the fp32-MFMA rate (the path issues 3 f16 MFMAs per "
                                        "f32-quality product, so this is a model figure, not a utilisation)"},
            "mfma_busy_frac_pmc": busy,
            "algorithmic_bytes_per_launch": algo_bytes,
            "valu": {"neuron_elements_per_launch": elems, "ns_per_1000_elements": round(ns_per_1000, 3),
                     "floor_ns_per_1000_elements": NEURON_FLOOR_NS_PER_1000, "frac_of_neuron_floor": round(NEURON_FLOOR_NS_PER_1000 / ns_per_1000, 4),
                     "note": "the kernel's physical bound is VALU issue of the 4-step neuron loops (12 packed + 6 transcendental "
                             "instructions per pair and step); floor = the bare loop, profiles/micro/lif_rate.hip"},
            "step_traffic": step_traffic}


def knn_leg(dev, cloud, seeds, reps=20):
    """The outer kNN kernel alone on the bench workload against SURVEY.md 8(d)'s streamed-bytes model: every query streams
    the cloud once at 12 B per point (fp32-equivalent), B * N * 12 bytes per launch against the 8 TB/s HBM peak.  The
    kernel tiles the f64 cloud (120 KB) through LDS, so its real HBM traffic is ~5 MB per launch; its physical bound is
    f64 VALU issue.  `frac_24B` is the same with the 24 B per point the kernel really reads."""
    from sapcu_amd import generation as gen
    gen.knn_gather(cloud, seeds, M_PTS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gen.knn_gather(cloud, seeds, M_PTS)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    streamed = float(seeds.shape[0]) * cloud.shape[0] * 12.0
    return {"kernel": "knn_outer_kernel", "us": round(t * 1e6, 1),
            "model": "SURVEY.md 8(d): B*N*12 bytes streamed per launch (fp32-equivalent cloud per query)",
            "achieved": round(streamed / t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(streamed / t / 1e9 / PEAK_HBM_GBS, 4), "frac_24B": round(2 * streamed / t / 1e9 / PEAK_HBM_GBS, 4),
            "pair_distances_per_s": round(float(seeds.shape[0]) * cloud.shape[0] / t, 0)}


def m100_leg(fn, fd, dev, cloud, seeds, steps=3):
    """The same step at the reference's DEFAULT patch size (generation.py:68 k_neighbors=100; BASELINE's M=48 is a benchmark
    choice): B=4096 queries, M=100, T=4 — a secondary figure, not the headline metric."""
    import sapcu_amd
    gen100 = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=100, batch_size=B_PER_GPU)
    with torch.no_grad():
        gen100.refine(cloud, seeds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, _, _ = gen100.refine(cloud, seeds)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out).all()
    return {"workload": "as config.workload with M=100 neighbours (the reference's default k_neighbors)", "value": round(B_PER_GPU / dt, 2),
            "unit": "query-points/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps}


def ref_default_leg(dev, cloud, seeds, steps=2):
    """The configuration `generate.py` + `config/*.yaml` of the reference really run: k_neighbors = 100 (generation.py:68; generate.py:135
    keeps it), fn time_steps_enc = 6 (config/fn.yaml:41), fd time_steps_enc = 7 (config/fd.yaml:47) — BASELINE's M = 48, T = 4 is a benchmark
    choice.  Own model handles (the temporal-integration weights have shape [T]); same cloud, same 4096 queries, conditioned weights seed 0."""
    import sapcu_amd
    from sapcu_amd import testing as T
    gold = os.path.join(ROOT, "tests", "golden")
    fn = sapcu_amd.ImprovedSNNNormalEstimation(**dict(FN_KW, time_steps_enc=6))
    fd = sapcu_amd.EnhancedSNNDistanceEstimation(**dict(FD_KW, time_steps_enc=7))
    for m, kind in ((fn, "fn"), (fd, "fd")):
        p = os.path.join(gold, "bn_calib_%s.npz" % kind)
        m.load_state_dict(T.conditioned_state_dict(m.state_dict(), 0, bn_stats=dict(np.load(p)) if os.path.exists(p) else None))
    fn, fd = fn.to(dev), fd.to(dev)
    fn.knn_cache_mode = "fresh"
    g = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=100, batch_size=B_PER_GPU)
    with torch.no_grad():
        g.refine(cloud, seeds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, _, _ = g.refine(cloud, seeds)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out).all()
    g.check_numeric_guards()
    return {"workload": "as config.workload with the reference's defaults: M=100 neighbours (generation.py:68), fn T=6 (config/fn.yaml:41), "
                        "fd T=7 (config/fd.yaml:47)", "value": round(B_PER_GPU / dt, 2), "unit": "query-points/s",
            "ms_per_step": round(dt * 1e3, 3), "steps": steps}


def cpu_baseline(sdn, sdd, sample=256):
    """Oracle on `sample` of the same queries (kNN + fn + rotate + fd + displace), all host threads.  256 = ONE full chunk of the
    reference's own batching (generate.py:135 batch_size=256), ~45 s on the GPU box's 16 host cores."""
    from sapcu_amd import testing as T
    from oracle import geom_path as G, snn_path as O
    fn_hp = dict(FN_KW)
    fd_hp = dict(FD_KW)
    cloud, q = T.sphere_cloud(N_CLOUD, 0), T.grid_queries(sample, 0)

    def fn_fwd(patch, pre):
        with torch.no_grad():
            return O.fn_forward(sdn, patch, fn_hp, knn_idx=pre), None

    def fd_fwd(patch):
        with torch.no_grad():
            return O.fd_forward(sdd, patch, fd_hp)

    # the GPU box gives one GPU's share of the host (16 cores); never oversubscribe a cgroup-limited box
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    G.upsample_core(cloud, q, fn_fwd, fd_fwd, M_PTS, sample, "fresh")
    dt = time.perf_counter() - t0
    return {"value": round(sample / dt, 3), "unit": "query-points/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d queries = one reference-sized chunk (generate.py:135 batch_size=256), oracle (torch-CPU restatement), %.1f s" % (sample, B_PER_GPU, dt)}


def _count_gpus_in_child():
    """Fallback of spawn_ranks when sysfs has no KFD topology: the runtime's own device count, taken in a child process so
    that THIS process still never loads HIP."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise SystemExit("bench.py: cannot count the GPUs (no /sys/class/kfd topology, and the runtime query failed: %s)" % r.stderr[-500:])


def spawn_ranks(args):
    """`python bench.py --gpus N` as a plain command: start one rank per GPU through torch.distributed.run as a CHILD
    process and return its exit code.  This parent is GPU-free BY CONSTRUCTION: the devices are counted from the KFD topology
    in sysfs + the visibility variables (sapcu_amd.dist.visible_gpu_count), never through the HIP runtime."""
    import socket
    import subprocess
    from sapcu_amd.dist import visible_gpu_count
    n_dev = visible_gpu_count()
    if n_dev is None:                 # no KFD topology in sysfs: ask the runtime, but in a throw-away CHILD process
        n_dev = _count_gpus_in_child()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < args.gpus:
        if args.gpus > 6:
            raise SystemExit("--gpus %d on a box with %d GPU(s): a rehearsal keeps every rank on cuda:0 and at most 6 "
                             "processes may share it" % (args.gpus, n_dev))
        env["SAPCU_BENCH_REHEARSE"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


STRONG_CLOUD_SPACING = 0.004        # generation.py:69 default dense_spacing -> 385 582 seeds on the N=5000 sphere


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-strong-leg", action="store_true", help="skip the extra whole-cloud pass of the weak mode")
    ap.add_argument("--no-m100", action="store_true", help="skip the secondary M=100 figure")
    ap.add_argument("--no-ref-default", action="store_true", help="skip the figure at the reference's default configuration (M=100, fn T=6, fd T=7)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            raise SystemExit(spawn_ranks(args))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    # SAPCU_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend — exercises the N > 1 code path (sharding, barriers,
    # all-gather, max-over-ranks timing) on a one-GPU box; the numbers of such a run mean nothing.
    rehearse = os.environ.get("SAPCU_BENCH_REHEARSE") == "1"
    dev = torch.device("cuda", 0 if rehearse else local)
    torch.cuda.set_device(dev)
    # SAPCU_BENCH_NCCL1=1: the N = 1 run also builds the process group of the N > 1 runs (nccl = RCCL, device_id) and sends its
    # barrier / all-gather / all-reduce through it — the RCCL contact a one-GPU box allows (tests/test_gpu_parity.py)
    use_pg = world > 1 or os.environ.get("SAPCU_BENCH_NCCL1") == "1"
    if use_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import sapcu_amd
    from sapcu_amd import testing as T
    from sapcu_amd import dist as sdist
    from sapcu_amd import generation as sgen
    fn, fd, sdn, sdd = build_models(dev)
    gen = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=M_PTS, batch_size=B_PER_GPU)
    cloud_host = T.sphere_cloud(N_CLOUD, 0)
    cloud = torch.as_tensor(cloud_host, device=dev)
    seeds = torch.as_tensor(T.grid_queries(B_PER_GPU * world, 0)[rank * B_PER_GPU:(rank + 1) * B_PER_GPU], device=dev)
    strong = args.scaling == "strong"
    want_strong_leg = strong or not args.no_strong_leg
    all_seeds = None
    if want_strong_leg:       # the cloud's real seeds (in-process dense grid flood, same on every rank), resident before timing
        # every rank floods the same cloud before the timed region: share the host's cores between the ranks of the node
        os.environ.setdefault("SAPCU_SEED_THREADS", str(min(16, max(1, (os.cpu_count() or 16) // max(1, world)))))
        all_seeds = torch.as_tensor(sgen.dense_seeds(cloud_host, STRONG_CLOUD_SPACING), device=dev)

    def weak_step():
        with torch.no_grad():
            refined, _, _ = gen.refine(cloud, seeds)
            if use_pg:
                refined = sdist.gather_refined(refined, B_PER_GPU * world)   # the one collective of the path
        return refined

    def strong_step():
        n = all_seeds.shape[0]
        s, e = sdist.shard_range(n, rank, world)
        with torch.no_grad():
            refined, _, _ = gen.refine(cloud, all_seeds[s:e])
            if use_pg:
                refined = sdist.gather_refined(refined, n)
        return refined

    step = strong_step if strong else weak_step

    def barrier():
        if use_pg:
            import torch.distributed as dist
            dist.barrier()

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    def timed(fn_step, k):
        """k steps between barrier + synchronize on both sides; MAX over ranks (seconds).  Also returns (min, max) over the ranks
        of each rank's OWN time from the first barrier to its last step's completion — before it waits for the others — so that
        a skewed rank shows in the record."""
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            out = fn_step()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        barrier()
        dt = time.perf_counter() - t0
        assert torch.isfinite(out).all()
        spread = (own, own)
        if use_pg:
            import torch.distributed as dist
            cdev = torch.device("cpu") if rehearse else dev
            tt = torch.tensor([dt, own, -own], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, spread = float(tt[0].item()), (-float(tt[2].item()), float(tt[1].item()))
        return dt, out, spread

    def allgather_alone(n_total, reps=10):
        """The path's one collective by itself: all-gather of [ceil(n/G), 3] f64 slabs (what gather_refined sends), `reps` calls
        between two events on the current stream (RCCL's own stream is ordered against it by torch on both sides); gloo
        rehearsals: wall clock.  Returns microseconds per call, max over the ranks."""
        import torch.distributed as dist
        s, e = sdist.shard_range(n_total, rank, world)
        slab = torch.zeros((e - s, 3), dtype=torch.float64, device=dev)
        for _ in range(2):
            sdist.gather_refined(slab, n_total)
        torch.cuda.synchronize()
        barrier()
        if rehearse:
            t0 = time.perf_counter()
            for _ in range(reps):
                sdist.gather_refined(slab, n_total)
            us = (time.perf_counter() - t0) * 1e6 / reps
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                sdist.gather_refined(slab, n_total)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
        tt = torch.tensor([us], dtype=torch.float64, device=torch.device("cpu") if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return round(float(tt.item()), 1)

    log("models and inputs resident; warmup x%d" % args.warmup)
    for i in range(args.warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log("warmup step %d: %.1f ms" % (i, (time.perf_counter() - tw) * 1e3))
    dt, out, spread = timed(step, args.steps)
    per_step = all_seeds.shape[0] if strong else B_PER_GPU * world
    log("timed %d steps: %.1f ms/step (per rank, own time: %.1f .. %.1f ms/step)"
        % (args.steps, dt / args.steps * 1e3, spread[0] / args.steps * 1e3, spread[1] / args.steps * 1e3))
    allgather_us = allgather_alone(per_step) if use_pg else None

    strong_leg = None
    if want_strong_leg and not strong:
        # one whole-cloud pass (385 582 seeds sharded over the ranks + the all-gather), after one untimed pass at N > 1 so
        # that RCCL's buffers for this message size exist
        if use_pg:
            strong_step()
        dts, outs, sspread = timed(strong_step, 1)
        strong_leg = {"seeds": int(all_seeds.shape[0]), "n_gpus": world, "ms_per_pass": round(dts * 1e3, 2),
                      "value": round(all_seeds.shape[0] / dts, 2), "unit": "query-points/s", "scaling": "strong",
                      "collective": "one all-gather of the refined [n,3] f64 cloud per pass",
                      "rank_ms_per_pass": {"min": round(sspread[0] * 1e3, 2), "max": round(sspread[1] * 1e3, 2)}}
        if use_pg:
            strong_leg["allgather_us"] = allgather_alone(int(all_seeds.shape[0]))
        log("strong-scaling leg: %s" % strong_leg)

    line = None
    if rank == 0:
        total = per_step * args.steps
        line = {
            "metric": "upsampled query-points/sec (kNN + fn fwd + rotate + fd fwd + displace), 4x scale, M=48 T=4",
            "value": round(total / dt, 2), "unit": "query-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("f32 (GEMMs on the exact-f32 MFMA kernels, SAPCU_GEMM=f32; outer kNN f64)" if os.environ.get("SAPCU_GEMM") == "f32"
                      else "f32 (GEMMs as 3 x f16 MFMA with f32 accumulation; outer kNN f64)"), "data": "synthetic",
            "config": {"workload": ("synthetic sphere N=%d (seed 0), %s, M=%d, T=%d, "
                                    "fn k=[24,18,12] emb 640, fd k=32 scales [8,16,32,48] emb 768, conditioned-random "
                                    "weights seed 0, in-patch kNN recomputed every batch"
                                    % (N_CLOUD, ("all %d dense-grid seeds of the cloud (spacing %.3f) sharded over the GPUs per step"
                                                 % (per_step, STRONG_CLOUD_SPACING)) if strong else
                                       "B=%d grid queries per GPU per step" % B_PER_GPU, M_PTS, T_STEPS)),
                       "queries_per_step": per_step, "cloud_points": N_CLOUD, "neighbours": M_PTS,
                       "time_steps": T_STEPS, "outer_knn": "f64 brute force", "parallelism": "query shards x%d" % world},
            "per_gpu": round(total / dt / world, 2),
            # canonical (dead-stage-eliminated, EdgeConv-factored) algorithmic work = 1.850 GFLOP per query
            "algorithmic_tflops": round(total / dt * 1.850e9 / 1e12, 2),
        }
        if rehearse:
            line["rehearsal"] = True
        if use_pg:
            import torch.distributed as dist
            line["collective_backend"] = dist.get_backend()
            # the record explains itself: the collective alone, and each rank's own step time (before it waits for the others)
            line["allgather_us"] = allgather_us
            line["rank_ms_per_step"] = {"min": round(spread[0] / args.steps * 1e3, 3), "max": round(spread[1] / args.steps * 1e3, 3)}
        if strong_leg:
            line["strong_scaling"] = strong_leg
        if not args.no_roofline:
            line["roofline"] = roofline_leg(dev)
            log("roofline leg done: %s" % line["roofline"])
            line["knn_kernel"] = knn_leg(dev, cloud, seeds)
            if not args.no_m100:
                line["m100"] = m100_leg(fn, fd, dev, cloud, seeds)
                log("M=100 leg: %s" % line["m100"])
            if not args.no_ref_default:
                line["ref_default"] = ref_default_leg(dev, cloud, seeds)
                log("reference-default leg: %s" % line["ref_default"])
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sdn, sdd)
            log("cpu baseline done: %s" % line["cpu_baseline"])
    barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_pg:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
