"""Worker of tests/test_gpu_parity.py::test_sharded_upsample_two_ranks_on_one_gpu — run under torch.distributed.run with
2 ranks, BOTH on cuda:0, gloo backend (a one-GPU box cannot host two RCCL ranks): the real Generator3D6 + upsample_sharded
+ gather_refined; every rank checks the gathered cloud against its own single-rank refine of all seeds, bit for bit.

`dist_rehearsal.py nccl` (test_rccl_world1_sharded_upsample_on_device_tensors): ONE rank with the nccl (= RCCL) backend and
device_id=cuda:0 — the branch of dist.gather_refined that keeps the f64 slabs on the device and the
init_process_group("nccl", device_id=...) call of bench.py, which no gloo rehearsal reaches."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import sapcu_amd
    from sapcu_amd import dist as sdist, testing as T
    from conftest import FD_KW, FN_KW, GOLDEN
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        fn = sapcu_amd.ImprovedSNNNormalEstimation(**FN_KW)
        fd = sapcu_amd.EnhancedSNNDistanceEstimation(**FD_KW)
        fn.load_state_dict(T.conditioned_state_dict(fn.state_dict(), 0, bn_stats=dict(np.load(os.path.join(GOLDEN, "bn_calib_fn.npz")))))
        fd.load_state_dict(T.conditioned_state_dict(fd.state_dict(), 0, bn_stats=dict(np.load(os.path.join(GOLDEN, "bn_calib_fd.npz")))))
        fn, fd = fn.to(dev), fd.to(dev)
        gen = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=48, batch_size=64)
        cloud = torch.as_tensor(T.sphere_cloud(1024, 0), device=dev)
        seeds = torch.as_tensor(T.grid_queries(333, 9), device=dev)
        with torch.no_grad():
            gathered, (s, e) = sdist.upsample_sharded(gen, cloud, seeds)
            assert fn.knn_cache_mode == "reference"
            fn.knn_cache_mode = "fresh"
            single, _, _ = gen.refine(cloud, seeds)
        assert gathered.shape == single.shape and gathered.is_cuda and gathered.dtype == torch.float64
        if backend == "nccl":
            assert dist.get_backend() == "nccl"
            # the collective itself on device tensors, uneven slabs (last rank short): every rank's rows come back in place
            probe = torch.arange(7 * 3, dtype=torch.float64, device=dev).view(7, 3) + 100.0 * rank
            s0, e0 = sdist.shard_range(7 * world - 2, rank, world)
            got = sdist.gather_refined(probe[: e0 - s0], 7 * world - 2)
            assert got.is_cuda and torch.equal(got[s0:e0], probe[: e0 - s0])
        assert torch.equal(gathered, single), "rank %d: gathered cloud differs from the single-rank refine" % rank
        assert (s, e) == sdist.shard_range(333, rank, world)
        dist.barrier()
        if rank == 0:
            print("REHEARSAL_OK ranks=%d backend=%s seeds=333 shard0=%d..%d" % (world, dist.get_backend(), s, e), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
