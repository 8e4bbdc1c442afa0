"""Multi-GPU: shard the query points, one process per GPU, one all-gather of the refined cloud.

The path shards embarrassingly: every seed is independent given the replicated
input cloud (<= 120 KB) and weights (33 MB).  Rank r refines the contiguous range
``seeds[r*ceil(n/G) : (r+1)*ceil(n/G)]``; the only exchange is ONE all-gather of the refined
points per cloud (``[ceil(n/G), 3]`` f64 per rank, last rank padded, trimmed after).  On the
fully connected xGMI mesh that is ~1 MB per rank — latency bound; it is issued once, not per batch.
``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors
(tests).

fn's shape-keyed neighbour cache (fn/snn_coder.py:47-59) makes reference-mode results depend on
which batch a process sees first, so a sharded run cannot reproduce a single-process reference run
bit for bit; sharded runs therefore use ``knn_cache_mode='fresh'`` (stated in DESIGN.md).
"""
import os

import torch
import torch.distributed as dist


def visible_gpu_count(sysfs_root="/sys/class/kfd/kfd/topology/nodes", dev_root="/dev/dri", environ=None):
    """Number of GPUs a process started from here would see — WITHOUT loading the HIP runtime (a launcher parent must stay
    GPU-free so that it may start ranks; touching HIP and then exec'ing / forking rank processes is what takes boxes down).
    Counts the KFD topology nodes that are GPUs (``simd_count`` > 0; CPU nodes have 0) and whose render node
    ``/dev/dri/renderD<drm_render_minor>`` this process can open (a container sees every node of the host in sysfs but only its
    own device files), then applies the runtime's visibility lists the way ROCm does: ROCR_VISIBLE_DEVICES filters the
    physical list, HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES index into the result (an empty string hides everything; an
    entry the list cannot resolve ends it, as in the runtime).  Returns None when there is no KFD topology to read (not a ROCm
    box) — the caller decides what that means."""
    env = os.environ if environ is None else environ
    try:
        nodes = sorted(os.listdir(sysfs_root), key=lambda s: (len(s), s))
    except OSError:
        return None
    gpus = []
    for nd in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(sysfs_root, nd, "properties")) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        node = os.path.join(dev_root, "renderD%d" % minor)
        if minor < 0 or not os.access(node, os.R_OK | os.W_OK):
            continue
        gpus.append(nd)
    count = len(gpus)

    def apply(var, n):
        if var not in env:
            return n
        entries = [e.strip() for e in env[var].split(",")] if env[var].strip() else []
        k = 0
        for e in entries:
            if e.isdigit():
                if int(e) >= n:
                    break
            elif not e.upper().startswith("GPU-"):          # neither an index nor a UUID: the runtime stops here
                break
            k += 1
        return min(k, n)

    count = apply("ROCR_VISIBLE_DEVICES", count)
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        count = apply(var, count)
    return count


def shard_range(n, rank, world):
    """Contiguous [start, end) of rank's share of n items; equal ceil(n/world) slabs, last ones short."""
    per = -(-n // world) if world > 0 else n
    s = min(n, rank * per)
    return s, min(n, s + per)


def gather_refined(local, n_total, group=None):
    """All-gather row slabs of a [n_local, C] tensor sharded by ``shard_range`` -> [n_total, C] on every rank."""
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()               # rehearsals of the N > 1 path on one GPU: gloo moves host memory
    padded = local.new_zeros((per,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return out[:n_total].to(dev)


def upsample_sharded(generator, cloud_dev, seeds_dev, group=None):
    """Refine this rank's shard with ``generator.refine`` and all-gather the refined cloud.
    Returns (refined [n,3] f64 on every rank, (start, end) of the local shard)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = seeds_dev.shape[0]
    s, e = shard_range(n, rank, world)
    model = generator.model1
    old = getattr(model, "knn_cache_mode", None)
    if old is not None:
        model.knn_cache_mode = "fresh"
    try:
        if e > s:
            local, _, _ = generator.refine(cloud_dev, seeds_dev[s:e])
        else:
            local = seeds_dev.new_empty((0, 3))
    finally:
        if old is not None:
            model.knn_cache_mode = old
    return gather_refined(local, n, group), (s, e)
