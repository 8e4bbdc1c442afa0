"""CPU suite, part 2: host logic, the C-ABI library surface (no compute calls), drop-in boundary,
and the N > 1 sharding path under gloo (world_size 2)."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import FD_KW, FN_KW, ROOT, golden

import sapcu_amd
from sapcu_amd import _lib, packing, testing
from sapcu_amd import generation as gen
from sapcu_amd import dist as sdist


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "sapcu.h")).read()
    declared = set(re.findall(r"\b(sapcu_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libsapcu_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.load().sapcu_abi_version() == _lib.ABI_VERSION == 2


def test_argument_errors_come_back_as_codes_not_crashes():
    lib = _lib.load()
    rc = lib.sapcu_knn_gather_f64(None, 10, None, 1, 4, None, None, None, None)
    assert rc == -1 and b"null" in lib.sapcu_last_error()
    with pytest.raises(_lib.SapcuError) as ei:
        _lib.check(lib.sapcu_workspace_bytes(None, 1, 48))
    assert ei.value.args[0] == -1


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.SapcuLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_state_dict_layout_matches_reference_dump():
    g = golden("state_dict_layout.npz")
    for kind, cls, kw in (("fn", sapcu_amd.ImprovedSNNNormalEstimation, FN_KW), ("fd", sapcu_amd.EnhancedSNNDistanceEstimation, FD_KW)):
        sd = cls(**kw).state_dict()
        assert list(sd) == list(g[kind + "_keys"])
        assert [str(tuple(v.shape)) for v in sd.values()] == list(g[kind + "_shapes"])
    assert sum(v.numel() for v in sapcu_amd.ImprovedSNNNormalEstimation(**FN_KW).state_dict().values()) == 6796652


def test_constructor_and_forward_error_conventions():
    with pytest.raises(NotImplementedError):
        sapcu_amd.ImprovedSNNNormalEstimation(use_snn_decoder=True)
    with pytest.raises(NotImplementedError):
        sapcu_amd.EnhancedSNNDistanceEstimation(use_snn_decoder=True)
    m = sapcu_amd.ImprovedSNNNormalEstimation(**FN_KW)
    assert not m.training
    with pytest.raises(ValueError):
        m(torch.zeros(4, 48))                    # bad rank
    with pytest.raises(ValueError):
        m(torch.zeros(2, 48, 4))                 # not xyz
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 48, 3))                 # CPU tensors: no CPU path
    m.train()                                    # fn has a training path (row f-4) ...
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 48, 3))                 # ... which has no CPU path either
    m.reset_states()
    m.eval()
    assert not m.training
    with pytest.raises(NotImplementedError):
        sapcu_amd.EnhancedSNNDistanceEstimation(**FD_KW).train()     # fd is inference-only


def test_packer_slot_tables(weights):
    blob, d = packing.pack_fn(weights("fn"))
    assert d.size == packing.FN_SLOTS == 81 and blob.dtype == np.float32 and (d % 4 == 0).all()
    blob, d = packing.pack_fd(weights("fd"), 4)
    assert d.size == packing.FD_SLOTS == 42 and (d % 4 == 0).all() and d[-1] < blob.size


def test_bn_fold_is_exact_in_float64():
    rng = np.random.default_rng(0)
    sd = {"l.weight": torch.from_numpy(rng.normal(size=(5, 7)).astype(np.float32)), "l.bias": torch.from_numpy(rng.normal(size=5).astype(np.float32)),
          "b.weight": torch.from_numpy(rng.uniform(0.5, 1.5, 5).astype(np.float32)), "b.bias": torch.from_numpy(rng.normal(size=5).astype(np.float32)),
          "b.running_mean": torch.from_numpy(rng.normal(size=5).astype(np.float32)), "b.running_var": torch.from_numpy(rng.uniform(0.1, 1, 5).astype(np.float32))}
    w, b = packing.fold_bn(sd, "l", "b")
    x = rng.normal(size=(3, 7))
    ref = torch.nn.functional.batch_norm(torch.nn.functional.linear(torch.from_numpy(x), sd["l.weight"].double(), sd["l.bias"].double()),
                                         sd["b.running_mean"].double(), sd["b.running_var"].double(), sd["b.weight"].double(), sd["b.bias"].double(), False, 0.0, 1e-5)
    np.testing.assert_allclose(x @ w.T + b, ref.numpy(), rtol=1e-12, atol=1e-12)


def test_split_batches_is_array_split():
    for n in (1, 63, 64, 65, 901, 4096):
        for bs in (64, 256, 400):
            ref = np.array_split(np.arange(n), max(1, n // bs))
            assert [(int(c[0]), int(c[-1]) + 1) for c in ref] == gen.split_batches(n, bs)


def test_conditioned_weights_are_deterministic(weights):
    a, b = weights("fd"), weights("fd")
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert float(a["encoder.snn_blocks.0.delta_T"].min()) >= 0.9 - 1e-6


def test_shard_ranges_cover_and_are_contiguous():
    for n in (0, 1, 7, 8, 4096, 385123):
        for w in (1, 2, 4, 8):
            r = [sdist.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def _gloo_worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seeds = torch.arange(n * 3, dtype=torch.float64).view(n, 3)

        class FakeGen:                      # stands in for the GPU refine: the gather logic is what is under test
            model1 = type("M", (), {"knn_cache_mode": "reference"})()

            def refine(self, cloud, s):
                assert self.model1.knn_cache_mode == "fresh"
                return s * 2.0 + 1.0, None, None

        g = FakeGen()
        out, (s, e) = sdist.upsample_sharded(g, None, seeds)
        assert g.model1.knn_cache_mode == "reference"
        q.put((rank, bool(torch.equal(out, seeds * 2.0 + 1.0)), (s, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7, 1])
def test_sharded_upsample_all_gather_gloo_world2(n):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2][0] == 0 and res[1][2][1] == n and res[0][2][1] == res[1][2][0]


def test_visible_gpu_count_reads_the_kfd_topology_not_the_runtime(tmp_path):
    """sapcu_amd.dist.visible_gpu_count on a fake sysfs / dev tree: CPU nodes (simd_count 0) are not GPUs, a GPU whose render node
    this process cannot open does not count (containers see the host's whole topology), and the visibility variables apply the
    way the runtime applies them.  No torch.cuda / HIP call is involved (bench.py's launcher parent relies on that)."""
    import inspect
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor) in enumerate([(0, -1), (0, -1), (1024, 128), (1024, 129), (1024, 130), (1024, 131)]):
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (0 if simd else 64, simd, minor))
    for minor in (128, 129, 130):                    # renderD131 is missing: that GPU belongs to another container
        (dri / ("renderD%d" % minor)).write_text("")
    cnt = lambda env: sdist.visible_gpu_count(str(nodes), str(dri), env)
    assert cnt({}) == 3
    assert cnt({"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert cnt({"HIP_VISIBLE_DEVICES": ""}) == 0
    assert cnt({"ROCR_VISIBLE_DEVICES": "1,2", "HIP_VISIBLE_DEVICES": "0,1,2"}) == 2     # HIP indices beyond the ROCR list end it
    assert cnt({"CUDA_VISIBLE_DEVICES": "0,7,1"}) == 1                                    # an unresolvable entry ends the list
    assert cnt({"ROCR_VISIBLE_DEVICES": "GPU-abcdef0123456789"}) == 1
    assert sdist.visible_gpu_count(str(tmp_path / "absent"), str(dri), {}) is None
    import ast
    tree = ast.parse(inspect.getsource(sdist.visible_gpu_count))
    names = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)} | {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute)}
    assert not names & {"torch", "cuda", "device_count", "is_available", "ctypes", "CDLL"}, names


@pytest.mark.parametrize("n", [385, 13, 5, 1])
def test_sharded_upsample_all_gather_gloo_world8(n):
    """world_size 8 (the driver's node): n not divisible by 8 — the last slabs short, and for n < 8 whole ranks EMPTY
    (n = 5: ceil = 1, ranks 5..7 refine nothing and still join the all-gather; n = 13: ceil = 2, rank 6 has one seed, rank 7 none)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 8, port, n, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    ranges = [r[2] for r in res]
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(ranges[i][1] == ranges[i + 1][0] for i in range(7))
    per = -(-n // 8)
    assert all(e - s0 == max(0, min(per, n - i * per)) for i, (s0, e) in enumerate(ranges))
    if n < 8:
        assert sum(1 for s0, e in ranges if e == s0) == 8 - n


def test_inprocess_seed_generator_matches_reference_dense_bit_for_bit():
    """csrc/dense_seeds.cpp against outputs of the reference's own dense.cpp (tests/golden/dense_seeds.npz,
    made by oracle/_ref/dense): same seeds, same ORDER, same 6-decimal values; full-size case by SHA-256."""
    import hashlib
    from sapcu_amd import testing as T
    g = golden("dense_seeds.npz")
    cases = (("sphere2048_c030", T.sphere_cloud(2048, 0)), ("torus2048_c020", T.analytic_cloud("torus", 2048, 1)),
             ("cube300_c050", T.analytic_cloud("cube", 300, 2)), ("tiny7_c050", T.sphere_cloud(7, 3)))
    for name, cloud in cases:
        s = gen.dense_seeds(cloud, float(g[name + "_cell"]))
        assert np.array_equal(s, g[name].reshape(-1, 3)), name
    big = gen.dense_seeds(T.sphere_cloud(5000, 0), 0.004)
    assert big.shape[0] == int(g["sphere5000_c004_count"]) == 385582
    assert hashlib.sha256(big.tobytes()).hexdigest() == str(g["sphere5000_c004_sha256"])
    assert np.array_equal(big[:64], g["sphere5000_c004_head"]) and np.array_equal(big[-64:], g["sphere5000_c004_tail"])
    assert np.array_equal(gen.dense_seeds(T.sphere_cloud(2048, 0), 0.03), golden("e2e_upsample.npz")["seeds"])


def test_pipeline_normalisation_and_fps_argument_checks():
    """generate.py:43-54 mirror is bit-identical to the reference run; FPS has no CPU path and checks its arguments."""
    from sapcu_amd import pipeline
    g = golden("fps.npz")
    c, loc, scale = pipeline.normalize_pointcloud(g["norm_in"])
    np.testing.assert_array_equal(c, g["norm_cloud"])
    np.testing.assert_array_equal(loc, g["norm_loc"])
    assert scale == float(g["norm_scale"])
    c, loc, scale = pipeline.normalize_pointcloud(g["flat_in"])
    np.testing.assert_array_equal(c, g["flat_cloud"])
    assert scale == 0.0
    with pytest.raises(ValueError):
        pipeline.farthest_point_sample_device(torch.zeros(8, 3), 4)          # CPU tensor: no CPU path
    lib = _lib.load()
    assert lib.sapcu_fps_workspace_bytes(100) >= 100 * 8 + 8
    assert lib.sapcu_fps_f32(None, 10, 11, None, None, 0, None) == -1       # npoint > n
    assert lib.sapcu_fps_f32(None, 10, 4, None, None, 0, None) == -1        # null pointers
    assert lib.sapcu_fps_f32(None, 10, 0, None, None, 0, None) == 0         # nothing to sample


# ---------------------------------------------------------------------------------------------
# Multi-rank path on the CPU: the REAL Generator3D6.refine batching / fusing logic and the real upsample_sharded +
# gather_refined under a gloo process group.  Only the per-pass device work is stood in for: numpy geometry from the
# oracle and two deterministic per-patch stand-in "models" (no GPU in this container).
# ---------------------------------------------------------------------------------------------
class _StubModel(object):
    def __init__(self):
        self.knn_cache_mode = "reference"
        self._knn_cache = {}
        self.calls = []

    def tiled_knn_tables(self, size, k, times):      # presence enables the fused passes of refine()
        raise AssertionError("sharded runs are 'fresh': no cached tables may be replayed")


def _host_generator(batch_size, fuse):
    from oracle import geom_path as G

    class HostGen(gen.Generator3D6):
        def __init__(self):                            # the real constructor insists on a ROCm device
            self.model1, self.model2 = _StubModel(), _StubModel()
            self.k_neighbors, self.batch_size, self.fuse_queries = 8, batch_size, fuse
            self.passes = []

        def _refine_pass(self, cloud, q, knn_in=None):
            assert knn_in is None
            self.passes.append(int(q.shape[0]))
            c, qq = cloud.numpy(), q.numpy()
            idx = G.knn_bruteforce(c, qq, self.k_neighbors)
            patch = G.gather_centre(c, qq, idx)
            n = patch.mean(axis=1) + np.array([0.3, -0.2, 0.9])          # stand-in for fn: any per-patch function
            n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
            rot = G.rotate_patches(patch, n)
            d = np.abs(rot[:, :, 0]).mean(axis=1).astype(np.float32)     # stand-in for fd
            out = G.displace(qq, n, d)
            return torch.from_numpy(out), torch.from_numpy(n), torch.from_numpy(d)

    return HostGen()


def _gloo_refine_worker(rank, world, port, n, batch_size, fuse, q):
    import torch.distributed as dist
    from sapcu_amd import testing as T
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cloud = torch.from_numpy(T.sphere_cloud(200, 1))
        seeds = torch.from_numpy(T.grid_queries(n, 3))
        g = _host_generator(batch_size, fuse)
        out, (s, e) = sdist.upsample_sharded(g, cloud, seeds)
        assert g.model1.knn_cache_mode == "reference"          # restored
        shard_passes = list(g.passes)
        single = _host_generator(batch_size, fuse)
        single.model1.knn_cache_mode = "fresh"
        ref, _, _ = single.refine(cloud, seeds)
        q.put((rank, bool(torch.equal(out, ref)), (s, e), shard_passes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,batch_size,fuse", [(301, 64, 0), (301, 64, 200), (37, 400, 4096), (2, 64, 4096)])
def test_sharded_refine_with_real_batching_logic_gloo_world2(n, batch_size, fuse):
    """world_size 2: each rank refines its contiguous shard through Generator3D6.refine (reference batch boundaries inside
    the shard, fused passes when enabled) and the all-gathered cloud equals a single-process refine of all seeds bit for bit."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_refine_worker, args=(r, 2, port, n, batch_size, fuse, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _, _ in res), res
    assert res[0][2][0] == 0 and res[1][2][1] == n and res[0][2][1] == res[1][2][0]
    for _, _, (s0, e0), passes in res:
        assert sum(passes) == e0 - s0
        if fuse == 0 and e0 - s0 >= batch_size:                 # one pass per reference batch of the shard
            assert passes == [b - a for a, b in gen.split_batches(e0 - s0, batch_size)]


def test_bench_spawns_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus N` as a plain command re-launches itself through torch.distributed.run as a child process;
    the parent must not initialise the GPU first (checked statically here: spawn_ranks only counts devices)."""
    import ast
    import inspect
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    src = inspect.getsource(bench.spawn_ranks)
    assert "torch.distributed.run" in src and "subprocess.call" in src and "visible_gpu_count" in src
    for forbidden in ("is_available", "set_device", "os.exec", "execv", "device_count", "torch.cuda"):     # no HIP runtime in the parent
        assert forbidden not in src, forbidden
    assert "subprocess.run" in inspect.getsource(bench._count_gpus_in_child)                                # the fallback asks a child
    tree = ast.parse(inspect.getsource(bench.main))
    calls = [n.lineno for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "spawn_ranks"]
    sets = [n.lineno for n in ast.walk(tree) if isinstance(n, ast.Attribute) and n.attr == "set_device"]
    assert calls and sets and max(calls) < min(sets)


def test_shape_suite_seeds_match_the_reference_runs():
    """BASELINE config 3 / 4 stand-ins: the in-process seed generator reproduces the seeds the reference's dense.cpp gave its
    own Generator3D6.upsample runs (tests/golden/shape_suite.npz, scale16.npz) — same seeds, same order."""
    from sapcu_amd import pipeline
    g = golden("shape_suite.npz")
    for name, _, _, spacing in testing.SHAPE_SUITE:
        assert float(g[name + "_spacing"]) == spacing
        cloud = testing.suite_cloud(name, g)
        assert cloud.shape == (testing.SHAPE_SUITE_N, 3)
        assert np.array_equal(gen.dense_seeds(cloud, spacing), g[name + "_seeds"]), name
        assert 150 <= g[name + "_seeds"].shape[0] <= 1000
        assert g[name + "_unfiltered"].shape == g[name + "_seeds"].shape
    s16 = golden("scale16.npz")
    cloud, loc, scale = pipeline.normalize_pointcloud(testing.scale16_cloud())
    assert np.array_equal(cloud, s16["norm_cloud"])
    assert np.array_equal(gen.dense_seeds(cloud, testing.SCALE16_CASE["spacing"]), s16["seeds"])
    assert s16["filtered"].shape[0] >= testing.SCALE16_CASE["ratio"] * testing.SCALE16_CASE["n"] == s16["fps_idx"].shape[0]


def test_neuron_parameter_clamp_matches_the_reference_loop():
    """fn_trainer.clamp_neuron_parameters = trainfd.py:305-313 (membrane_decay [0.1, 0.99], threshold_adapt [0.001, 0.1],
    refractory_decay [0.1, 0.95]; threshold_base and everything else untouched)."""
    from sapcu_amd import fn_trainer
    m = sapcu_amd.ImprovedSNNNormalEstimation(**FN_KW)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if any(k in name for k in ("membrane_decay", "threshold_adapt", "refractory_decay", "threshold_base")):
                prm.copy_(torch.linspace(-1.0, 2.0, prm.numel()).view_as(prm))
    before = {n: q.detach().clone() for n, q in m.named_parameters()}
    fn_trainer.clamp_neuron_parameters(m)
    for name, prm in m.named_parameters():
        if "membrane_decay" in name:
            assert float(prm.min()) == pytest.approx(0.1) and float(prm.max()) == pytest.approx(0.99)
        elif "threshold_adapt" in name:
            assert float(prm.min()) == pytest.approx(0.001) and float(prm.max()) == pytest.approx(0.1)
        elif "refractory_decay" in name:
            assert float(prm.min()) == pytest.approx(0.1) and float(prm.max()) == pytest.approx(0.95)
        else:
            assert torch.equal(prm, before[name]), name
