"""Worker of tests/test_gpu_parity.py::test_sharded_upsample_two_ranks_on_one_gpu — run under torch.distributed.run with
2 ranks, BOTH on cuda:0, gloo backend (a one-GPU box cannot host two RCCL ranks): the real Generator3D6 + upsample_sharded
+ gather_refined; every rank checks the gathered cloud against its own single-rank refine of all seeds, bit for bit."""
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
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
        assert gathered.shape == single.shape and gathered.is_cuda
        assert torch.equal(gathered, single), "rank %d: gathered cloud differs from the single-rank refine" % rank
        assert (s, e) == sdist.shard_range(333, rank, world)
        dist.barrier()
        if rank == 0:
            print("REHEARSAL_OK ranks=%d seeds=333 shard0=%d..%d" % (world, s, e), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
