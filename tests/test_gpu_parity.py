"""GPU suite (-m gpu): the HIP path, called through the C ABI / drop-in shells, against the oracle
and the committed golden vectors.  Bars (BASELINE.json north_star): kNN indices bit-exact;
normals / distances within 1e-4 abs.

fd's blocks 1-3 pick neighbours in 64/128/256-d soft-spike space by fp32 scores whose summation
order is unspecified in the reference (oneDNN sgemm): a near-tie at rank k can flip between two
correct fp32 implementations.  Protocol (DESIGN.md "kNN flips"): read back the neighbour tables the
device chose, evaluate the ORACLE on exactly those tables -> 1e-4 for every patch; separately count
rows whose table differs from the oracle's own and require them rare and genuine near-ties.
"""
import os

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import geom_path as G
from oracle import snn_path as O
import gpu_utils as U

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def models(weights):
    return U.build_gpu_models(weights)


def _dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=U.dev())
    return t.to(dtype) if dtype is not None else t


def _decode_split_rows(t, n):
    """[rows, n] f32 container holding split rows (csrc/gemm_epi.h) -> f32 values hi + lo.  n % 32 == 0: groups of 32 elements,
    each one 128-byte line = 32 hi halves then 32 lo halves; otherwise hi halves at [0, n), lo halves at [n, 2n)."""
    rows = t.shape[0]
    halves = t.contiguous().view(torch.float16).view(rows, 2 * n).float()
    if n % 32 == 0:
        g = halves.view(rows, n // 32, 2, 32)
        return (g[:, :, 0, :] + g[:, :, 1, :]).reshape(rows, n)
    return halves[:, :n] + halves[:, n:]


# ------------------------------------------------------------------------------- outer kNN
def test_outer_knn_bit_exact_golden_and_patches():
    from sapcu_amd import testing as T
    from sapcu_amd import generation as gen
    g = golden("outer_knn.npz")
    cloud, q = T.sphere_cloud(5000, 0), T.grid_queries(4096, 0)
    idx, dist, patch = gen.knn_gather(_dev(cloud), _dev(q), 48, want_dist=True)
    idx = idx.cpu().numpy()
    assert np.array_equal(idx, g["idx_n5000_k48"].astype(np.int64))                 # == sklearn KDTree
    assert np.array_equal(idx, G.knn_bruteforce(cloud, q, 48))                      # == oracle
    assert np.array_equal(dist.cpu().numpy()[:64], g["dist_head"])
    assert np.array_equal(patch.cpu().numpy(), G.gather_centre(cloud, q, idx).astype(np.float32))
    cloud2, q2 = T.sphere_cloud(2048, 0), T.grid_queries(256, 3)
    idx2, _, _ = gen.knn_gather(_dev(cloud2), _dev(q2), 100)                         # two slots per lane
    assert np.array_equal(idx2.cpu().numpy(), g["idx_n2048_k100"].astype(np.int64))


@pytest.mark.parametrize("n,b,k", [(1, 3, 1), (5, 7, 5), (64, 1, 64), (65, 130, 33), (1024, 5, 128), (1025, 9, 48), (3000, 2, 100)])
def test_outer_knn_edge_shapes(n, b, k):
    from sapcu_amd import generation as gen
    rng = np.random.default_rng(n * 1000 + b)
    cloud = np.round(rng.uniform(-0.5, 0.5, (n, 3)), 6)
    q = np.round(rng.uniform(-0.5, 0.5, (b, 3)), 6)
    if n > 2:
        cloud[n // 2] = cloud[0]                  # duplicate point -> exact tie, ascending index wins
        q[0] = cloud[n - 1]                       # query on a cloud point -> distance 0
    idx, dist, patch = gen.knn_gather(_dev(cloud), _dev(q), k, want_dist=True)
    ref = G.knn_bruteforce(cloud, q, k)
    assert np.array_equal(idx.cpu().numpy(), ref)
    d = dist.cpu().numpy()
    assert (np.diff(d, axis=1) >= 0).all()
    assert np.array_equal(patch.cpu().numpy(), G.gather_centre(cloud, q, ref).astype(np.float32))


def test_outer_knn_full_size_properties():
    """B=4096, N=5000, k=48: sortedness, idempotence, invariance to a cloud permutation."""
    from sapcu_amd import testing as T
    from sapcu_amd import generation as gen
    cloud, q = T.sphere_cloud(5000, 0), T.grid_queries(4096, 0)
    c, qq = _dev(cloud), _dev(q)
    i1, d1, _ = gen.knn_gather(c, qq, 48, want_dist=True)
    i2, d2, _ = gen.knn_gather(c, qq, 48, want_dist=True)
    assert torch.equal(i1, i2) and torch.equal(d1, d2)
    assert bool((d1[:, 1:] >= d1[:, :-1]).all())
    perm = np.random.default_rng(0).permutation(5000)
    i3, d3, _ = gen.knn_gather(_dev(cloud[perm]), qq, 48, want_dist=True)
    assert torch.equal(d3, d1)                                   # same distances
    back = torch.as_tensor(perm, device=U.dev())[i3]
    assert torch.equal(back.sort(1)[0], i1.sort(1)[0])           # same neighbour sets
    # checksum of checksums against the oracle
    assert int(i1.sum().item()) == int(G.knn_bruteforce(cloud, q, 48).sum())


# ------------------------------------------------------------------------------- rotation / displacement
def test_rotation_and_displacement_exact():
    from sapcu_amd import testing as T
    from sapcu_amd import generation as gen
    g = golden("rotation.npz")
    cloud, q = T.sphere_cloud(5000, 0), T.grid_queries(64, 0)
    idx = G.knn_bruteforce(cloud, q, 48)
    nrm = g["normals"]
    rot = gen.gather_rotate(_dev(cloud), _dev(q), _dev(idx), _dev(nrm)).cpu().numpy()
    ref = G.rotate_patches(G.gather_centre(cloud, q, idx), nrm).astype(np.float32)
    assert np.array_equal(ref, g["rotated_f32"])                 # oracle == reference run
    mism = int((rot != ref).sum())
    assert mism <= 2, "%d of %d rotated coordinates differ in the last f32 ulp" % (mism, rot.size)
    np.testing.assert_allclose(rot, ref, rtol=0, atol=1e-9)
    # rows 0/1 are n = +x / -x: identity (reference quirk for the antiparallel case)
    plain = gen.gather_rotate(_dev(cloud), _dev(q), _dev(idx), None).cpu().numpy()
    assert np.array_equal(rot[0], plain[0]) and np.array_equal(rot[1], plain[1])
    d = np.random.default_rng(1).uniform(0, 0.05, 64).astype(np.float32)
    out = gen.displace(_dev(q), _dev(nrm), _dev(d)).cpu().numpy()
    assert np.array_equal(out, G.displace(q, nrm, d))


# ------------------------------------------------------------------------------- neuron unit
def test_neuron_unit_against_reference_vectors():
    from sapcu_amd import _lib
    g = golden("neuron_unit.npz")
    lib = _lib.load()
    x, raw = _dev(g["x"]), [_dev(r) for r in g["raw_params"]]
    rows, ch = g["x"].shape
    for kind in ("lif", "eif"):
        for T in (1, 4, 7):
            outs = [torch.empty_like(x) for _ in range(4)]
            dT, rh = (raw[4], raw[5]) if kind == "eif" else (None, None)
            _lib.check(lib.sapcu_neuron_selfloop(_lib.ptr(x), rows, ch, T, _lib.ptr(raw[0]), _lib.ptr(raw[1]), _lib.ptr(raw[2]),
                                                 _lib.ptr(raw[3]), _lib.ptr(dT), _lib.ptr(rh), *[_lib.ptr(o) for o in outs],
                                                 _lib.current_stream()))
            for o, key in zip(outs, ("spikes", "membrane", "threshold", "refractory")):
                # 1e-6 abs on O(1) values; EIF membranes reach dT*e^5 ~ 700, where 1-2 ulp of exp is ~1e-4 abs
                np.testing.assert_allclose(o.cpu().numpy(), g["%s_T%d_%s" % (kind, T, key)], rtol=2e-5, atol=1e-6,
                                           err_msg="%s T=%d %s" % (kind, T, key))


def test_neuron_stepping_form_far_outside_the_spike_clamp():
    """neuron_wide.npz (the reference's fd neurons out to |x| = 1e4, every step's spikes): the packed stepping form that fd's
    kernels run (NeuronStep2: two rows of a channel; NeuronStep2V: two channels of a row), input at step 0 only.  The default
    build leaves the +-10 clamp of the spike function out, so below x - theta = -13.2 its spike is exactly 0 where the
    reference's is 3.85e-23; the refractory state is floored at that minimum (common.h SPIKE_FLOOR), so the kernels' gate test
    must report 0 open gates — as the reference's own gate does on these inputs — and r must stay > 0.  Also the self-loop
    form (fn's production loop) at T = 7 against the last step."""
    from sapcu_amd import _lib
    g = golden("neuron_wide.npz")
    lib = _lib.load()
    x, raw = _dev(g["x"]), [_dev(r) for r in g["raw_params"]]
    rows, ch = g["x"].shape
    assert int(g["lif_gate_open"]) == 0 and int(g["eif_gate_open"]) == 0
    for kind in ("lif", "eif"):
        ref = g[kind + "_spikes"]
        T = ref.shape[0]
        dT, rh = (raw[4], raw[5]) if kind == "eif" else (None, None)
        for pairv in (0, 1):
            spk = torch.full((T, rows, ch), float("nan"), device=U.dev())
            st = [torch.full((rows, ch), float("nan"), device=U.dev()) for _ in range(3)]
            gate = torch.zeros(1, dtype=torch.int32, device=U.dev())
            _lib.check(lib.sapcu_neuron_drive(_lib.ptr(x), rows, ch, T, _lib.ptr(raw[0]), _lib.ptr(raw[1]), _lib.ptr(raw[2]), _lib.ptr(raw[3]),
                                              _lib.ptr(dT), _lib.ptr(rh), pairv, _lib.ptr(spk), *[_lib.ptr(s) for s in st], _lib.ptr(gate),
                                              _lib.current_stream()))
            tag = "%s pairv=%d" % (kind, pairv)
            np.testing.assert_allclose(spk.cpu().numpy(), ref, rtol=2e-5, atol=1e-6, err_msg=tag)
            assert int(gate.item()) == 0, tag + ": %d open gates on inputs where the reference's gate is closed" % int(gate.item())
            r = st[2].cpu().numpy()
            assert (r > 0).all(), tag
            # the floor keeps r at or above the reference's own value wherever the reference sits at its minimum
            np.testing.assert_allclose(r, g[kind + "_refractory"], rtol=2e-5, atol=1e-6, err_msg=tag)
            # (the EIF membrane is m * (1 - s) of values up to dT e^5 ~ 700 with s near 1: the product keeps the ABSOLUTE error of
            #  a 1-2 ulp hardware exp, ~1e-5, on a small result)
            np.testing.assert_allclose(st[0].cpu().numpy(), g[kind + "_membrane"], rtol=2e-5, atol=5e-5 if kind == "eif" else 1e-6, err_msg=tag)
            np.testing.assert_allclose(st[1].cpu().numpy(), g[kind + "_threshold"], rtol=2e-5, atol=1e-6, err_msg=tag)
        out = torch.empty_like(x)
        _lib.check(lib.sapcu_neuron_selfloop(_lib.ptr(x), rows, ch, T, _lib.ptr(raw[0]), _lib.ptr(raw[1]), _lib.ptr(raw[2]), _lib.ptr(raw[3]),
                                             _lib.ptr(dT), _lib.ptr(rh), _lib.ptr(out), None, None, None, _lib.current_stream()))
        np.testing.assert_allclose(out.cpu().numpy(), ref[T - 1], rtol=2e-5, atol=1e-6, err_msg=kind + " self-loop")
        # many silent steps: with the floor at step 0 only, r * rdecay^t (rdecay clamps down to 0.1) would underflow to 0 after ~15 steps
        # and the gate test would fire; the reference's r is >= 3.85e-23 at EVERY step
        TL = 48
        spk = torch.empty((TL, rows, ch), device=U.dev())
        rr = torch.empty((rows, ch), device=U.dev())
        gate = torch.zeros(1, dtype=torch.int32, device=U.dev())
        _lib.check(lib.sapcu_neuron_drive(_lib.ptr(x), rows, ch, TL, _lib.ptr(raw[0]), _lib.ptr(raw[1]), _lib.ptr(raw[2]), _lib.ptr(raw[3]),
                                          _lib.ptr(dT), _lib.ptr(rh), 0, _lib.ptr(spk), None, None, _lib.ptr(rr), _lib.ptr(gate), _lib.current_stream()))
        assert int(gate.item()) == 0 and bool((rr > 0).all()), kind + ": gate test fired within %d steps" % TL
        np.testing.assert_allclose(spk[:T].cpu().numpy(), ref, rtol=2e-5, atol=1e-6, err_msg=kind + " first steps of the long run")


# ------------------------------------------------------------------------------- in-patch kNN
def test_patch_knn_xyz_exact_and_feature_space_flips():
    """In-patch kNN against the reference's own knn() run (indices AND the score matrix it ranked, patch_knn.npz).  xyz
    (c = 3) tables are exact.  In feature space two correct fp32 evaluations of -|a|^2 + 2ab - |b|^2 may order a near-tie
    differently: rows whose neighbour SET differs must be rare (<= 1 % of rows per case) and each must be a tie of the
    REFERENCE's scores to within a few ulps of the score magnitude at the k-th / (k+1)-th boundary."""
    from sapcu_amd import _lib
    g = golden("patch_knn.npz")
    lib = _lib.load()
    total_rows = total_flips = 0
    worst_ulps = 0.0
    for c in (3, 64, 128, 256):
        f = g["feat_c%d" % c]                                    # [b,c,m]
        feat = _dev(np.ascontiguousarray(f.transpose(0, 2, 1)))  # [b,m,c]
        b, m = feat.shape[0], feat.shape[1]
        scores = g["score_c%d" % c]                              # the matrix the reference handed to topk
        # (the oracle's own scores on THIS host differ from them by up to ~30 ulp of the score magnitude: torch's sgemm sums in
        # a CPU-dependent order — which is why flips are judged against the stored reference scores, not recomputed ones)
        ulp = float(np.spacing(np.float32(np.abs(scores).max())))
        for k in (8, 12, 16, 18, 24, 32, 48):
            out = torch.empty((b, m, k), dtype=torch.int32, device=U.dev())
            _lib.check(lib.sapcu_patch_knn(_lib.ptr(feat), b, m, c, c, k, _lib.ptr(out), _lib.current_stream()))
            got, ref = out.cpu().numpy().astype(np.int64), g["idx_c%d_k%d" % (c, k)].astype(np.int64)
            if c == 3:
                assert np.array_equal(got, ref), "xyz kNN must be bit-exact (k=%d)" % k
                continue
            same = np.sort(got, -1) == np.sort(ref, -1)
            bad_rows = np.argwhere(~same.all(-1))
            total_rows += b * m
            total_flips += len(bad_rows)
            assert len(bad_rows) <= max(1, (b * m) // 100), "too many neighbour-set flips: %d of %d rows (c=%d k=%d)" % (len(bad_rows), b * m, c, k)
            for bi, ri in bad_rows:                              # each flip must be a genuine near-tie of the reference's scores
                srow = scores[bi, ri]
                sym = np.setxor1d(got[bi, ri], ref[bi, ri])
                gap_ulps = float(np.ptp(srow[sym])) / ulp
                worst_ulps = max(worst_ulps, gap_ulps)
                assert gap_ulps <= 16.0, "row (%d,%d) c=%d k=%d: swapped neighbours differ by %.1f ulp of the score" % (bi, ri, c, k, gap_ulps)
    print("in-patch kNN (feature space): %d of %d rows with a different neighbour set; widest swapped gap %.1f ulp of the score magnitude"
          % (total_flips, total_rows, worst_ulps))


def test_patch_knn_exact_ties_resolve_by_ascending_index():
    """Rows with equal scores: the kernel ranks with 32-bit compares of the score keys first and redoes a row whose ranks are not a
    permutation with (score, index) pairs — include/sapcu.h: "descending score, equal scores by ascending index".  xyz patches
    with duplicated points (whole columns of the score matrix equal), m = 48 (one column per lane) and m = 100 (two), against a
    stable sort of the oracle's scores (c = 3: the device scores equal torch's bit for bit)."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for m, k in ((48, 24), (48, 48), (100, 32), (7, 7)):
        b = 6
        pts = rng.normal(size=(b, m, 3)).astype(np.float32)
        for p in range(b):                                   # p duplicated pairs / triples per patch (patch 0: none)
            for q in range(p):
                src, dst = rng.integers(0, m, 2)
                pts[p, dst] = pts[p, src]
            if p == b - 1 and m > 3:
                pts[p, 1] = pts[p, 0]
                pts[p, m - 1] = pts[p, 0]                    # a triple
        sc = O.inpatch_knn_scores(torch.from_numpy(np.ascontiguousarray(pts.transpose(0, 2, 1)))).numpy()     # [b,m,m]
        want = np.argsort(-sc, axis=-1, kind="stable")[..., :k]
        out = torch.empty((b, m, k), dtype=torch.int32, device=U.dev())
        feat = _dev(pts)
        _lib.check(lib.sapcu_patch_knn(_lib.ptr(feat), b, m, 3, 3, k, _lib.ptr(out), _lib.current_stream()))
        assert np.array_equal(out.cpu().numpy().astype(np.int64), want), (m, k)


# ------------------------------------------------------------------------------- GEMM
def _run_gemm(a, w, bias, split):
    """split: False = exact-f32 MFMA kernel; True = split-f16 with f32 A; "ring" = split-f16 ring kernel (A as split rows)."""
    from sapcu_amd import _lib
    lib = _lib.load()
    r, k = a.shape
    n = w.shape[0]
    A, W, Bv = _dev(a), _dev(w), _dev(bias)
    C = torch.full((r, n), float("nan"), device=U.dev())
    ws = torch.zeros(4 * n * k + 16, dtype=torch.uint8, device=U.dev()) if split else None
    a_split = 0
    if split == "ring":
        As = torch.empty_like(A)
        _lib.check(lib.sapcu_to_split_rows(_lib.ptr(A), r, k, k, _lib.ptr(As), k, _lib.current_stream()))
        A, a_split = As, 1
    _lib.check(lib.sapcu_gemm_f32(_lib.ptr(A), r, k, k, _lib.ptr(W), n, _lib.ptr(Bv), None, 0, _lib.ptr(C), n,
                                  _lib.ptr(ws), a_split, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    ovf = int(ws[-16:].view(torch.int32)[0].item()) if split else 0
    return C.cpu().numpy(), ovf


@pytest.mark.parametrize("split", [False, True, "ring"])
@pytest.mark.parametrize("r,k,n", [(1, 32, 1), (130, 64, 3), (257, 192, 640), (1000, 960, 768), (4096, 512, 512), (40000, 128, 128)])
def test_gemm_against_float64(r, k, n, split):
    rng = np.random.default_rng(r + k + n)
    a, w, bias = rng.normal(size=(r, k)).astype(np.float32), rng.normal(size=(n, k)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    c, ovf = _run_gemm(a, w, bias, split)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias
    err = np.abs(c - ref).max()
    assert ovf == 0 and err <= 2e-6 * np.sqrt(k) * 4, err


@pytest.mark.parametrize("r,k,n,csplit", [(1, 32, 4, 0), (300, 96, 132, 1), (257, 128, 128, 1), (5000, 256, 260, 0), (4099, 512, 512, 1)])
def test_ring_gemm_neuron_epilogue_and_split_row_output(r, k, n, csplit):
    """The ring kernel's row-layout consumer end to end through the C ABI: A in split rows, T = 4 neuron self-loop on
    every output (per-column parameters), output as f32 or as split rows; ragged row/column tiles (r, n not multiples
    of 128).  Oracle: float64 GEMM + the oracle's neuron step (the reference's arithmetic)."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(r * 7 + n)
    a = rng.normal(size=(r, k)).astype(np.float32)
    w = (rng.normal(size=(n, k)) / np.sqrt(k)).astype(np.float32)
    bias = rng.normal(size=n).astype(np.float32)
    raw = np.stack([rng.uniform(0.05, 1.1, n), rng.uniform(0.0, 0.2, n), rng.uniform(0.05, 1.0, n), rng.normal(0.5, 0.3, n)]).astype(np.float32)
    A, W, Bv, L = _dev(a), _dev(w), _dev(bias), _dev(raw)
    As = torch.empty_like(A)
    _lib.check(lib.sapcu_to_split_rows(_lib.ptr(A), r, k, k, _lib.ptr(As), k, _lib.current_stream()))
    C = torch.full((r, n), float("nan"), device=U.dev())
    ws = torch.zeros(4 * n * k + 16, dtype=torch.uint8, device=U.dev())
    _lib.check(lib.sapcu_gemm_f32(_lib.ptr(As), r, k, k, _lib.ptr(W), n, _lib.ptr(Bv), _lib.ptr(L), 4, _lib.ptr(C), n,
                                  _lib.ptr(ws), 1, csplit, _lib.current_stream()))
    torch.cuda.synchronize()
    got = C.cpu()
    if csplit:
        got = _decode_split_rows(got, n)
    pre = torch.from_numpy((a.astype(np.float64) @ w.astype(np.float64).T + bias).astype(np.float32))
    names = ["membrane_decay", "threshold_adapt", "refractory_decay", "threshold_base"]
    prm = O.neuron_params({"n." + names[i]: torch.from_numpy(raw[i]) for i in range(4)}, "n")
    v, st = pre, None
    for _ in range(4):
        v, st = O.neuron_step(v, st, prm)
    err = (got - v).abs().max().item()
    assert err <= 2e-5, err                      # pre-activation error 1e-6*sqrt(k) through a slope <= 2.6 per step


@pytest.mark.parametrize("mode", ["f32", "sf16", "ring"])
@pytest.mark.parametrize("b,m,kk,d", [(3, 5, 4, 64), (7, 48, 12, 128), (2, 48, 24, 256)])
def test_posenc_gemm_attention_epilogue(mode, b, m, kk, d):
    """sapcu_posenc_gemm_f32 (fn/snn_coder.py:360-368) through the C ABI in its three forms: exact-f32 MFMA, split-f16
    with f32 operands, and the production ring form (pe1 and attn_in as split rows): pe = LIF x4 (W pe1 + b),
    attn_in = q[point] - k[neighbour] + pe, against float64 GEMM + the oracle's neuron step."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(b * 1000 + d)
    r = b * m * kk
    pe1 = rng.random((r, d)).astype(np.float32)
    w = (rng.normal(size=(d, d)) / np.sqrt(d)).astype(np.float32)
    bias = rng.normal(size=d).astype(np.float32)
    raw = np.stack([rng.uniform(0.05, 1.1, d), rng.uniform(0.0, 0.2, d), rng.uniform(0.05, 1.0, d), rng.normal(0.5, 0.3, d)]).astype(np.float32)
    qkv = rng.random((b * m, 3 * d)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, m, kk)).astype(np.int32)
    P1, W, Bv, L, Q, I = _dev(pe1), _dev(w), _dev(bias), _dev(raw), _dev(qkv), _dev(idx)
    split = 1 if mode == "ring" else 0
    if split:
        tmp = torch.empty_like(P1)
        _lib.check(lib.sapcu_to_split_rows(_lib.ptr(P1), r, d, d, _lib.ptr(tmp), d, _lib.current_stream()))
        P1 = tmp
    pe = torch.full((r, d), float("nan"), device=U.dev())
    att = torch.full((r, d), float("nan"), device=U.dev())
    tab = torch.empty((r, 2), dtype=torch.int32, device=U.dev())
    ws = None if mode == "f32" else torch.zeros(4 * d * d + 16, dtype=torch.uint8, device=U.dev())
    _lib.check(lib.sapcu_posenc_gemm_f32(_lib.ptr(P1), r, d, _lib.ptr(W), _lib.ptr(Bv), _lib.ptr(L), 4, _lib.ptr(Q), _lib.ptr(I), kk, m,
                                         _lib.ptr(pe), _lib.ptr(att), _lib.ptr(tab), _lib.ptr(ws), split, _lib.current_stream()))
    torch.cuda.synchronize()
    pe, att = pe.cpu(), att.cpu()
    if split:                                    # attn_in leaves as split rows (pe stays f32)
        att = _decode_split_rows(att, d)
    pre = torch.from_numpy((pe1.astype(np.float64) @ w.astype(np.float64).T + bias).astype(np.float32))
    names = ["membrane_decay", "threshold_adapt", "refractory_decay", "threshold_base"]
    prm = O.neuron_params({"n." + names[i]: torch.from_numpy(raw[i]) for i in range(4)}, "n")
    v, st = pre, None
    for _ in range(4):
        v, st = O.neuron_step(v, st, prm)
    rows = np.arange(r)
    pt = rows // kk
    nbr = (pt // m) * m + idx.reshape(-1)
    want_att = torch.from_numpy(qkv[pt, :d] - qkv[nbr, d:2 * d]) + v
    assert (pe - v).abs().max().item() <= 2e-5
    assert (att - want_att).abs().max().item() <= 2e-5 + (2e-7 if split else 0.0)


def test_split_f16_gemm_error_bound_on_mixed_magnitudes():
    """Spikes down to 1e-6, weights spanning 1e-4..10, activations up to 2e4.  Bound: f32-level relative error
    on sum|a||w| plus the f16 subnormal quantum (2^-25 per activation below 0.25, 2^-29 per weight below
    2^-6) — an ABSOLUTE term that is negligible against O(1) pre-activations."""
    rng = np.random.default_rng(5)
    k, n, r = 512, 256, 1024
    a = (rng.random((r, k)) * np.power(10.0, rng.uniform(-6, 0, (r, 1)))).astype(np.float32)
    a[:8] *= 2e4
    w = (rng.normal(size=(n, k)) * np.power(10.0, rng.uniform(-4, 1, (n, 1)))).astype(np.float32)
    bias = np.zeros(n, np.float32)
    A64, W64 = np.abs(a).astype(np.float64), np.abs(w).astype(np.float64)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    bound = 4e-7 * (A64 @ W64.T) + 2.0 ** -25 * W64.sum(1)[None, :] + 2.0 ** -29 * A64.sum(1)[:, None]
    c32, _ = _run_gemm(a, w, bias, False)
    c16, ovf = _run_gemm(a, w, bias, True)
    cring, _ = _run_gemm(a, w, bias, "ring")
    r32, r16, rr = np.abs(c32 - ref) / bound, np.abs(c16 - ref) / bound, np.abs(cring - ref) / bound
    print("error / bound: f32 MFMA max %.2f, split-f16 max %.2f, ring max %.2f" % (r32.max(), r16.max(), rr.max()))
    assert ovf == 0 and r16.max() <= 1.0 and rr.max() <= 1.0
    a[0, 0] = 7e4                                                             # beyond f16: must be reported
    _, ovf = _run_gemm(a, w, bias, True)
    assert ovf >= 1


# ------------------------------------------------------------------------------- fn
def _fn_taps(b, m=48, emb=640):
    z = lambda *s: torch.empty(s, dtype=torch.float32, device=U.dev())
    return {"stem": z(b, m, 64), "block1": z(b, m, 64), "block2": z(b, m, 64), "block3": z(b, m, 64), "pooled": z(b, emb),
            "enc": z(b, 2048), "logits": z(b, 3)}


def test_fn_stage_taps_against_reference_vectors(models):
    fn, _, sdn, _ = models
    g = golden("fn_taps.npz")
    fn.knn_cache_mode = "fresh"
    taps = _fn_taps(4)
    n = fn(_dev(g["patch"]), taps=taps)
    torch.cuda.synchronize()
    for name in ("stem", "block1", "block2", "block3", "enc", "logits"):
        ref = g[name]
        err = np.abs(taps[name].cpu().numpy() - ref).max()
        assert err <= TOL * max(1.0, np.abs(ref).max()), "%s: %g" % (name, err)
    np.testing.assert_allclose(n.cpu().numpy(), g["normals"], rtol=0, atol=TOL)


def test_fn_forward_64_patches_vs_oracle_and_knn_tables(models):
    fn, _, sdn, _ = models
    patch = U.sphere_patches(64, 48, skip=100)
    fn.knn_cache_mode = "reference"
    fn._knn_cache.clear()
    n = fn(patch.to(U.dev())).cpu()
    taps = {}
    with torch.no_grad():
        ref = O.fn_forward(sdn, patch, U.FN_HP, taps=taps)
    for got, want in zip(fn.knn_tables(64, 48), taps["knn_idx"]):
        assert torch.equal(got.cpu().long(), want), "in-patch xyz neighbour tables must be bit-exact"
    err = (n - ref).abs()
    print("fn normals: max %.3g  p99.9 %.3g" % (err.max(), np.quantile(err.numpy(), 0.999)))
    assert err.max() <= TOL
    assert ref.std(0).max() > 1e-2 and (n.norm(dim=1) - 1).abs().max() < 1e-5


def test_fn_stale_cache_quirk_matches_reference(models):
    fn, _, _, _ = models
    g = golden("fn_cache_pair.npz")
    fn.knn_cache_mode = "reference"
    fn._knn_cache.clear()
    na = fn(_dev(g["patch_a"]))
    fn.reset_states()                                             # must NOT clear the cache (fn:726-738)
    nb = fn(_dev(g["patch_b"]))
    np.testing.assert_allclose(na.cpu().numpy(), g["normals_a"], rtol=0, atol=TOL)
    np.testing.assert_allclose(nb.cpu().numpy(), g["normals_b_stale"], rtol=0, atol=TOL)
    fn.knn_cache_mode = "fresh"
    np.testing.assert_allclose(fn(_dev(g["patch_b"])).cpu().numpy(), g["normals_b_fresh"], rtol=0, atol=TOL)


def test_fn_input_layouts(models):
    fn, _, _, _ = models
    fn.knn_cache_mode = "fresh"
    p = U.sphere_patches(6, 48, skip=300).to(U.dev())
    a = fn(p)
    assert torch.equal(fn(p.permute(0, 2, 1).contiguous()), a)            # [B,3,M]
    assert torch.equal(fn(p.view(2, 3, 48, 3)), a.view(2, 3, 3))           # [B,N,M,3]
    assert fn(p[:0]).shape == (0, 3)


# ------------------------------------------------------------------------------- fd
def test_fd_stage_taps_against_reference_vectors(models):
    _, fd, _, sdd = models
    g = golden("fd_taps.npz")
    b, m, T = 4, 48, 4
    z = lambda *s: torch.empty(s, dtype=torch.float32, device=U.dev())
    taps = {"fused0": z(b, m, 64), "spikes": z(T, b, m, 960), "knn": torch.empty((3, b, m, 32), dtype=torch.int32, device=U.dev()),
            "pooled": z(T, b, 768), "enc": z(b, 768)}
    # (1) the comparison with the reference's own stage outputs ALWAYS happens: the feature-space neighbour tables the
    #     reference itself chose (fixture knn1..3) are forced, so a near-tie flip cannot skip it
    force = torch.from_numpy(np.stack([g["knn%d" % i].astype(np.int32) for i in (1, 2, 3)])).to(U.dev())
    d = fd(_dev(g["patch"]), taps=taps, knn_force=force)
    torch.cuda.synchronize()
    assert fd.gate_violations() == 0
    assert torch.equal(taps["knn"], force)
    np.testing.assert_allclose(taps["fused0"].cpu().numpy(), g["fused0"], rtol=0, atol=TOL)
    np.testing.assert_allclose(taps["spikes"][0].cpu().numpy(), g["spikes_t0"], rtol=0, atol=TOL)
    np.testing.assert_allclose(taps["spikes"][T - 1].cpu().numpy(), g["spikes_tlast"], rtol=0, atol=TOL)
    np.testing.assert_allclose(taps["pooled"].cpu().numpy(), g["pooled"], rtol=0, atol=TOL * 4)
    np.testing.assert_allclose(taps["enc"].cpu().numpy(), g["enc"], rtol=0, atol=TOL)
    np.testing.assert_allclose(d.cpu().numpy(), g["dist"], rtol=0, atol=TOL)
    # (2) free-running: the tables the device chooses itself, counted against the reference's
    free = torch.empty_like(force)
    d_free = fd(_dev(g["patch"]), taps={"knn": free})
    knn = free.cpu().numpy()
    flips = sum(int((np.sort(knn[i], -1) != np.sort(g["knn%d" % (i + 1)].astype(np.int64), -1)).any(-1).sum()) for i in range(3))
    print("fd taps: neighbour-set flips in %d of %d rows (free-running)" % (flips, 3 * b * m))
    assert flips <= 3
    if flips == 0:
        np.testing.assert_allclose(d_free.cpu().numpy(), g["dist"], rtol=0, atol=TOL)


def test_fd_forward_256_patches_forced_neighbour_protocol(models):
    _, fd, _, sdd = models
    patch = U.sphere_patches(256, 48, skip=200)
    d_gpu, d_forced, d_free, flips, _ = U.fd_forward_forced(fd, sdd, patch)
    assert fd.gate_violations() == 0
    err = (d_gpu - d_forced).abs()
    flip_patches = (flips[0] | flips[1] | flips[2]).any(-1)
    print("fd: max |gpu - oracle(forced)| %.3g; p99.9 %.3g; patches with a neighbour flip %d / 256; "
          "max |gpu - oracle(free)| on flip-free patches %.3g" %
          (err.max(), np.quantile(err.numpy(), 0.999), int(flip_patches.sum()),
           (d_gpu - d_free).abs()[~flip_patches].max()))
    assert err.max() <= TOL
    assert (d_gpu - d_free).abs()[~flip_patches].max() <= TOL
    assert int(flip_patches.sum()) <= 16            # measured 9-10 of 256 (round 1); a 2x regression fails
    flip_rows = sum(int(f.sum()) for f in flips)
    print("fd: rows with a flipped neighbour set %d of %d" % (flip_rows, 3 * 256 * 48))
    assert flip_rows <= 3 * 256 * 48 // 100
    assert d_free.std() > 1e-2


def _far_from_threshold_fd_weights(weights):
    """The conditioned fd weights with BatchNorm biases moved so that every neuron stage sees pre-activations far outside the
    spike function's +-10 clamp on both sides: channels 0, 7, 14, .. of scale_fusion and of the three EdgeConv blocks get
    beta = -110 (LeakyReLU: x0 ~ -22, x0 - theta ~ -23) and channels 3, 10, .. beta = +22."""
    sd = {k: v.clone() for k, v in weights("fd").items()}
    for name in ["encoder.scale_fusion.1.bias"] + ["encoder.conv_blocks.%d.1.bias" % i for i in range(3)]:
        sd[name][0::7] = -110.0
        sd[name][3::7] = 22.0
    return sd


def test_fd_preactivations_far_outside_the_spike_clamp(weights, monkeypatch):
    """the default build's packed spike function has no +-10 clamp, so for x0 - theta below about -13.2 its
    spike is exactly 0 where the reference's is 3.85e-23.  The reference's refractory gate is closed there (r > 0) and it runs
    normally; so must this build: distances equal to the oracle's under forced neighbour tables (1e-4), NO gate violation
    reported by either path (fused encoder, per-stage kernels), and Generator3D6.upsample_seeds does not raise."""
    import sapcu_amd
    from conftest import FD_KW
    from sapcu_amd import generation as gen, testing as T
    sdd = _far_from_threshold_fd_weights(weights)
    fds = []
    for env in ({"SAPCU_FD_FUSED": "1"}, {"SAPCU_FD_FUSED": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fd = sapcu_amd.EnhancedSNNDistanceEstimation(**FD_KW)
        fd.load_state_dict(sdd, strict=True)
        fd = fd.to(U.dev())
        fd._engine()
        for k in env:
            monkeypatch.delenv(k, raising=False)
        fds.append(fd)
    assert fds[0].fused_blocks(48) == 1 and fds[1].fused_blocks(48) == 0
    patch = U.sphere_patches(24, 48, skip=2100)
    x0 = torch.full((24, 48, 960), float("nan"), device=U.dev())
    fds[0](patch.to(U.dev()), taps={"x0": x0})
    x0 = x0.cpu()
    theta = torch.cat([sdd["encoder.snn_blocks.%d.threshold_base" % i] for i in range(4)])
    dlt = x0 - theta
    for lo, hi in ((0, 64), (64, 192), (192, 448), (448, 960)):          # every block reaches both tails
        assert float(dlt[..., lo:hi].min()) < -15 and float(dlt[..., lo:hi].max()) > 15, (lo, float(dlt[..., lo:hi].min()), float(dlt[..., lo:hi].max()))
    outs = []
    for fd in fds:
        d_gpu, d_forced, _, flips, _ = U.fd_forward_forced(fd, sdd, patch)
        assert float((d_gpu - d_forced).abs().max()) <= TOL, float((d_gpu - d_forced).abs().max())
        assert fd.gate_violations() == 0, "gate counter fired on inputs where the reference's gate is closed"
        outs.append(d_gpu)
    assert float(outs[0].std()) > 1e-3
    knn = torch.empty((3, 24, 48, 32), dtype=torch.int32, device=U.dev())
    a = fds[0](patch.to(U.dev()), taps={"knn": knn})
    assert torch.equal(a, fds[1](patch.to(U.dev()), knn_force=knn))     # fused == per-stage, bit for bit, on this input class too
    # the reference-facing entry point: the numeric guards run at the end of upsample_seeds and must stay silent
    fn_ok = U.build_gpu_models(weights)[0]
    g3 = gen.Generator3D6(fn_ok, fds[0], U.dev(), k_neighbors=48, batch_size=64)
    cloud = T.sphere_cloud(1024, 0)
    out = g3.upsample_seeds(cloud, T.grid_queries(96, 4))
    assert out.shape[1] == 3 and np.isfinite(out).all() and fds[0].gate_violations() == 0


def test_gemm_modes_agree_and_stay_in_range(weights, monkeypatch):
    """The default split-f16 GEMMs against the exact-f32 MFMA kernels (SAPCU_GEMM=f32) on the same patches."""
    fn16, fd16, sdn, sdd = U.build_gpu_models(weights)
    fn16._engine(), fd16._engine()                      # handles are built lazily: build them before the switch
    monkeypatch.setenv("SAPCU_GEMM", "f32")
    fn32, fd32, _, _ = U.build_gpu_models(weights)
    fn16.knn_cache_mode = fn32.knn_cache_mode = "fresh"
    patch = U.sphere_patches(32, 48, skip=700).to(U.dev())
    n16, n32 = fn16(patch), fn32(patch)
    assert fn32.gemm_mode() == (False, 0) and fn16.gemm_mode() == (True, 0)
    assert (n16 - n32).abs().max() <= TOL
    knn = torch.empty((3, 32, 48, 32), dtype=torch.int32, device=U.dev())
    d32 = fd32(patch, taps={"knn": knn})
    d16 = fd16(patch, knn_force=knn)
    assert fd16.gemm_mode() == (True, 0)
    assert (d16 - d32).abs().max() <= TOL


def test_chunking_and_tiny_shapes(weights, monkeypatch):
    """Results do not depend on the internal chunk size (workspace tiling), batch of 1, patches of 5 points."""
    fn_a, fd_a, sdn, sdd = U.build_gpu_models(weights)
    fn_a._engine(), fd_a._engine()
    monkeypatch.setenv("SAPCU_CHUNK", "7")
    fn_b, fd_b, _, _ = U.build_gpu_models(weights)
    fn_a.knn_cache_mode = fn_b.knn_cache_mode = "fresh"
    patch = U.sphere_patches(20, 48, skip=900).to(U.dev())
    assert torch.equal(fn_a(patch), fn_b(patch))
    knn = torch.empty((3, 20, 48, 32), dtype=torch.int32, device=U.dev())
    da = fd_a(patch, taps={"knn": knn})
    assert torch.equal(da, fd_b(patch, knn_force=knn))
    assert torch.equal(fn_a(patch[:1]), fn_a(patch)[:1])
    small = U.sphere_patches(3, 5, skip=40)
    with torch.no_grad():
        ref_n = O.fn_forward(sdn, small, U.FN_HP)
    assert (fn_a(small.to(U.dev())).cpu() - ref_n).abs().max() <= TOL
    d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fd_a, sdd, small)
    assert (d_gpu - d_forced).abs().max() <= TOL


def test_fd_forced_tables_are_honoured(models):
    _, fd, _, sdd = models
    patch = U.sphere_patches(8, 48, skip=500)
    taps = {}
    with torch.no_grad():
        want = O.fd_forward(sdd, patch, U.FD_HP, taps=taps)
    force = torch.stack([taps["encoder.knn%d" % i] for i in (1, 2, 3)]).to(torch.int32).to(U.dev())
    got = fd(patch.to(U.dev()), knn_force=force).cpu()
    assert (got - want).abs().max() <= TOL


# ------------------------------------------------------------------------------- variants
def test_shape_and_time_step_variants(weights):
    g = golden("variants.npz")
    fn, fd, sdn, sdd = U.build_gpu_models(weights)
    fn.knn_cache_mode = "fresh"
    for M in (12, 100):
        p = torch.from_numpy(g["patch_M%d" % M])
        np.testing.assert_allclose(fn(p.to(U.dev())).cpu().numpy(), g["normals_M%d" % M], rtol=0, atol=TOL)
        d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fd, sdd, p)
        assert (d_gpu - d_forced).abs().max() <= TOL
    p = torch.from_numpy(g["patch_T"])
    for Tv in (6, 7):
        fnv, fdv, _, sddv = U.build_gpu_models(weights, {"time_steps_enc": Tv}, {"time_steps_enc": Tv})
        fnv.knn_cache_mode = "fresh"
        np.testing.assert_allclose(fnv(p.to(U.dev())).cpu().numpy(), g["normals_T%d" % Tv], rtol=0, atol=TOL)
        d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fdv, sddv, p, dict(U.FD_HP, time_steps_enc=Tv))
        assert (d_gpu - d_forced).abs().max() <= TOL
        assert fdv.gate_violations() == 0


# ------------------------------------------------------------------------------- end to end
def test_upsample_end_to_end_against_reference_run(models):
    """Generator3D6 (device) on the reference's own e2e case: sphere N=2048, 901 seeds from ./dense,
    batch 64 (two batch shapes -> the stale-cache path runs), k=48.

    fd is discontinuous in its input (feature-space kNN), and its input here depends on the normals, so
    the end-to-end comparison is made stage by stage on the first two batches (teacher forcing), and
    the full refined cloud is compared with the reference run as a distribution."""
    import sapcu_amd
    from sapcu_amd import testing as T
    from sapcu_amd import generation as gen_mod
    fn, fd, sdn, sdd = models
    g = golden("e2e_upsample.npz")
    seeds = g["seeds"]
    fn.knn_cache_mode = "reference"
    fn._knn_cache.clear()
    gen = sapcu_amd.Generator3D6(fn, fd, U.dev(), k_neighbors=48, dense_spacing=0.03, batch_size=64)
    cloud = T.sphere_cloud(2048, 0)
    c_dev, s_dev = _dev(cloud), _dev(seeds)
    with torch.no_grad():
        refined, normals, dists = gen.refine(c_dev, s_dev)
    refined, normals, dists = refined.cpu().numpy(), normals.cpu(), dists.cpu()
    # (1) stages on batches 0 and 1 (both 65 seeds: batch 1 replays batch 0's neighbour tables)
    chunks = G.split_batches(seeds.shape[0], 64)
    assert chunks[0][1] - chunks[0][0] == chunks[1][1] - chunks[1][0] == 65
    cache = None
    for (s, e) in chunks[:2]:
        q = seeds[s:e]
        idx = G.knn_bruteforce(cloud, q, 48)
        patch = torch.from_numpy(G.gather_centre(cloud, q, idx)).float()
        taps = {}
        with torch.no_grad():
            n_ref = torch.nn.functional.normalize(O.fn_forward(sdn, patch, U.FN_HP, knn_idx=cache, taps=taps), dim=-1)
        cache = cache or taps["knn_idx"]
        assert (normals[s:e] - n_ref).abs().max() <= TOL
        rot_gpu = gen_mod.gather_rotate(c_dev, _dev(q), _dev(idx), normals[s:e].to(U.dev())).cpu()
        rot_ref = torch.from_numpy(G.rotate_patches(G.gather_centre(cloud, q, idx), normals[s:e].numpy())).float()
        assert (rot_gpu - rot_ref).abs().max() <= 1e-9
        d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fd, sdd, rot_gpu)
        assert torch.equal(d_gpu, dists[s:e])
        assert (d_gpu - d_forced).abs().max() <= TOL
        assert np.array_equal(refined[s:e], G.displace(q, normals[s:e].numpy(), dists[s:e].numpy()))
    # (2) whole cloud vs the reference run
    err = np.abs(refined - g["unfiltered"]).max(axis=1)
    ok = err <= 2 * TOL
    print("e2e: %.1f%% of %d refined points within 2e-4 of the reference run; median %.3g, p90 %.3g (rest: fd neighbour flips)"
          % (100 * ok.mean(), err.size, np.median(err), np.quantile(err, 0.9)))
    assert ok.mean() >= 0.90 and np.median(err) <= 5e-5       # measured 0.93 / 1.2e-5 (round 1)
    # (3) outlier filter on the reference's own unfiltered cloud reproduces its keep set exactly
    keep = gen.outlier_filter(_dev(g["unfiltered"]))
    assert np.array_equal(g["unfiltered"][keep], g["filtered"])
    filtered = gen.upsample_seeds(cloud, seeds)
    assert filtered.dtype == np.float64 and filtered.shape[1] == 3 and abs(filtered.shape[0] - g["filtered"].shape[0]) <= 60
    # (4) the drop-in entry point: upsample(data[1,N,3]) generates the same seeds in process and gives the same cloud
    fn._knn_cache.clear()
    full = gen.upsample(cloud[None])
    fn._knn_cache.clear()
    assert np.array_equal(full, gen.upsample_seeds(cloud, seeds))


def _check_upsample_against_reference_run(models, cloud, seeds, unfiltered, filtered, spacing, tag, min_ok, stage_patches=16, k=48,
                                          batch_size=64, fn_hp=None, fd_hp=None):
    """Staged protocol of test_upsample_end_to_end_against_reference_run for any cloud: k 48, batch 64 (unless given), reference
    cache mode.  Returns the fraction of refined points within 2e-4 of the reference run."""
    import sapcu_amd
    from sapcu_amd import generation as gen_mod
    fn, fd, sdn, sdd = models
    fn_hp, fd_hp = fn_hp or U.FN_HP, fd_hp or U.FD_HP
    fn.knn_cache_mode = "reference"
    fn._knn_cache.clear()
    gen = sapcu_amd.Generator3D6(fn, fd, U.dev(), k_neighbors=k, dense_spacing=spacing, batch_size=batch_size)
    c_dev, s_dev = _dev(cloud), _dev(seeds)
    with torch.no_grad():
        refined, normals, dists = gen.refine(c_dev, s_dev)
    refined, normals, dists = refined.cpu().numpy(), normals.cpu(), dists.cpu()
    # (1) stages on the head of the first batch, teacher-forced (the cache is empty: batch 0 ranks its own neighbours; the
    # replay of cached tables by later batches is pinned by test_upsample_end_to_end_against_reference_run)
    e0 = min(stage_patches, G.split_batches(seeds.shape[0], batch_size)[0][1])
    q = seeds[:e0]
    idx = G.knn_bruteforce(cloud, q, k)
    patch = torch.from_numpy(G.gather_centre(cloud, q, idx)).float()
    with torch.no_grad():
        n_ref = torch.nn.functional.normalize(O.fn_forward(sdn, patch, fn_hp), dim=-1)
    assert (normals[:e0] - n_ref).abs().max() <= TOL, tag
    rot_gpu = gen_mod.gather_rotate(c_dev, _dev(q), _dev(idx), normals[:e0].to(U.dev())).cpu()
    rot_ref = torch.from_numpy(G.rotate_patches(G.gather_centre(cloud, q, idx), normals[:e0].numpy())).float()
    assert (rot_gpu - rot_ref).abs().max() <= 1e-9, tag
    d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fd, sdd, rot_gpu, fd_hp)
    assert torch.equal(d_gpu, dists[:e0]), tag
    assert (d_gpu - d_forced).abs().max() <= TOL, tag
    assert np.array_equal(refined[:e0], G.displace(q, normals[:e0].numpy(), dists[:e0].numpy())), tag
    # (2) the whole refined cloud against the reference run, as a distribution (the tail = fd neighbour flips)
    err = np.abs(refined - unfiltered).max(axis=1)
    ok = err <= 2 * TOL
    print("%s: %d seeds, %.1f%% of the refined points within 2e-4 of the reference run; median %.3g, p90 %.3g"
          % (tag, err.size, 100 * ok.mean(), np.median(err), np.quantile(err, 0.9)))
    assert ok.mean() >= min_ok and np.median(err) <= 5e-5, tag
    # (3) the outlier filter on the reference's own unfiltered cloud reproduces its keep set exactly
    keep = gen.outlier_filter(_dev(unfiltered))
    assert np.array_equal(unfiltered[keep], filtered), tag
    # (4) the drop-in entry point generates the same seeds in process and refines them to the same cloud
    fn._knn_cache.clear()
    full = gen.upsample(cloud[None])
    fn._knn_cache.clear()
    assert np.array_equal(full, gen.upsample_seeds(cloud, seeds)), tag
    assert abs(full.shape[0] - filtered.shape[0]) <= max(8, filtered.shape[0] // 12), tag
    gen.check_numeric_guards()
    return float(ok.mean())


def test_upsample_at_the_reference_defaults_against_reference_run(weights):
    """The configuration `generate.py` + `config/*.yaml` of the reference really run (BASELINE's M = 48, T = 4 is a benchmark
    choice): k_neighbors = 100 (generation.py:68), batch_size = 256 (generate.py:135), fn time_steps_enc = 6 (config/fn.yaml:41), fd
    time_steps_enc = 7 (config/fd.yaml:47).  e2e_default.npz = the reference's own Generator3D6.upsample run on the 2048-point sphere
    (901 seeds from its dense.cpp at spacing 0.03).  Device path: fn's fused edge chains at 100 points per patch, fd on the x0 path
    (per-stage front + fd_msc_kernel, general kernel: two groups of steps).  Staged protocol as for the other reference runs
    (teacher-forced stages on the head of the first batch: normals and forced-neighbour distances within 1e-4, rotation and
    displacement exact); the whole refined cloud as a distribution; keep set of the outlier filter exact; drop-in entry point
    consistent.  The distribution bar: a patch leaves the 2e-4 band when one of its feature-space neighbour rows flips against the
    reference's BLAS summation order, at a rate per ROW that does not depend on the patch size — the 48-point
    runs lose 4.5-7 % of their patches (144 rows each), so a 100-point patch (300 rows) is expected to lose 2.08 x that, 9.5-14.5 %:
    expected 0.855-0.905 within 2e-4 (measured 0.880, median error 6.9e-6); bar 0.80."""
    from sapcu_amd import testing as T
    g = golden("e2e_default.npz")
    models = U.build_gpu_models(weights, {"time_steps_enc": 6}, {"time_steps_enc": 7})
    assert models[1].fused_blocks(100) == 2 and models[0].fused_blocks(100) == 0b111
    frac = _check_upsample_against_reference_run(models, T.sphere_cloud(2048, 0), g["seeds"], g["unfiltered"], g["filtered"], 0.03,
                                                 "reference defaults (k 100, fn T 6, fd T 7, batch 256)", 0.80, stage_patches=8, k=100,
                                                 batch_size=256, fn_hp=dict(U.FN_HP, time_steps_enc=6), fd_hp=dict(U.FD_HP, time_steps_enc=7))
    assert frac >= 0.80


@pytest.mark.parametrize("shape", ["sphere", "torus", "cube", "cylinder", "two_spheres", "icosahedron"])
def test_shape_suite_upsample_against_reference_runs(models, shape):
    """BASELINE config 3 stand-in: six shapes of 2048 points — smooth, sharp-edged (cube), open
    (cylinder), creased (two-sphere union), a real scan-like cloud (the reference tree's Icosahedron.xyz) — against the
    reference's own Generator3D6.upsample runs (tests/golden/shape_suite.npz)."""
    from sapcu_amd import testing as T
    g = golden("shape_suite.npz")
    cloud = T.suite_cloud(shape, g)
    _check_upsample_against_reference_run(models, cloud, g[shape + "_seeds"], g[shape + "_unfiltered"], g[shape + "_filtered"],
                                          float(g[shape + "_spacing"]), "suite/" + shape, 0.90)       # measured 0.936-0.955


def test_arbitrary_scale_16x_against_reference_run(models):
    """BASELINE config 4 stand-in: 16x arbitrary scale = the body of generate.py:81-99 (normalise -> upsample -> denormalise ->
    farthest-point-sample to 16 N) on a 256-point cloud with a non-trivial bounding box, against the reference's own run
    (tests/golden/scale16.npz): staged refine parity, FPS indices exact on the reference's refined cloud, 16 N points out."""
    import sapcu_amd
    from sapcu_amd import pipeline, testing as T
    fn, fd, _, _ = models
    g = golden("scale16.npz")
    c = T.SCALE16_CASE
    raw = T.scale16_cloud()
    target = c["ratio"] * c["n"]
    cloud, loc, scale = pipeline.normalize_pointcloud(raw)
    assert np.array_equal(cloud, g["norm_cloud"]) and np.array_equal(loc, g["loc"]) and scale == float(g["scale"])
    _check_upsample_against_reference_run(models, cloud, g["seeds"], g["unfiltered"], g["filtered"], c["spacing"], "16x", 0.90)
    # FPS to 16 N on the reference's own refined, denormalised cloud: the index sequence is exact
    up = g["filtered"] * scale + loc
    idx = pipeline.farthest_point_sample(up, target, device=U.dev())
    assert np.array_equal(idx, g["fps_idx"])
    assert np.array_equal(up[idx], g["output"])
    # the whole driver body on the device path: 16 N points, every one a point of our own refined cloud
    fn.knn_cache_mode = "reference"
    fn._knn_cache.clear()
    gen = sapcu_amd.Generator3D6(fn, fd, U.dev(), k_neighbors=48, dense_spacing=c["spacing"], batch_size=64)
    out = pipeline.process_cloud(raw, gen, target)
    assert out.shape == (target, 3) and out.dtype == np.float64 and np.isfinite(out).all()
    fn._knn_cache.clear()
    mine = gen.upsample(cloud[None]) * scale + loc
    assert len({tuple(r) for r in out} - {tuple(r) for r in mine}) == 0
    assert len({tuple(r) for r in out}) == target                     # FPS never repeats a point while unsampled ones remain


def test_full_batch_4096_properties(models):
    """BASELINE size B=4096, M=48, T=4: batch-composition independence, determinism, unit normals."""
    from sapcu_amd import testing as T
    from sapcu_amd import generation as gen
    fn, fd, _, _ = models
    fn.knn_cache_mode = "fresh"
    cloud, q = _dev(T.sphere_cloud(5000, 0)), _dev(T.grid_queries(4096, 0))
    idx, _, patch = gen.knn_gather(cloud, q, 48)
    n1 = fn(patch)
    n2 = fn(patch)
    assert torch.equal(n1, n2)
    assert (n1.norm(dim=1) - 1).abs().max() < 1e-5
    sub = fn(patch[1000:1300])
    assert (sub - n1[1000:1300]).abs().max() <= 1e-6                # per-item results do not depend on the batch
    rot = gen.gather_rotate(cloud, q, idx, gen.l2_normalize3(n1))
    d1 = fd(rot)
    assert torch.equal(d1, fd(rot)) and bool((d1 > 0).all())
    assert (fd(rot[500:600]) - d1[500:600]).abs().max() <= 1e-6
    out = gen.displace(q, gen.l2_normalize3(n1), d1)
    assert torch.isfinite(out).all()
    assert fd.gate_violations() == 0


def test_headline_batch_rows_against_the_oracle(models):
    """BASELINE config 2 — the batch bench.py reports (sphere N = 5000 seed 0, the 4096 grid queries, M = 48, T = 4) — run as ONE
    device pass, then 48 of its rows (fixed seed) checked against the oracle stage by stage: outer-kNN indices and gathered
    patches exact, normals 1e-4, rotated patches to the last f32 digit, distances 1e-4 under the neighbour tables the device
    chose for those rows (forced-neighbour protocol), displaced points exact.  (the headline batch's parity no
    longer rests on sub-batch == full-batch transitivity.)"""
    from sapcu_amd import testing as T
    from sapcu_amd import generation as gen
    fn, fd, sdn, sdd = models
    fn.knn_cache_mode = "fresh"
    cloud_h, q_h = T.sphere_cloud(5000, 0), T.grid_queries(4096, 0)
    cloud, q = _dev(cloud_h), _dev(q_h)
    idx, _, patch = gen.knn_gather(cloud, q, 48)
    nrm = gen.l2_normalize3(fn(patch))
    rot = gen.gather_rotate(cloud, q, idx, nrm)
    knn = torch.empty((3, 4096, 48, 32), dtype=torch.int32, device=U.dev())
    dist = fd(rot, taps={"knn": knn})
    out = gen.displace(q, nrm, dist)
    torch.cuda.synchronize()
    rows = np.sort(np.random.default_rng(2024).choice(4096, 48, replace=False))
    rt = torch.as_tensor(rows, device=U.dev())
    ref_idx = G.knn_bruteforce(cloud_h, q_h[rows], 48)
    assert np.array_equal(idx[rt].cpu().numpy(), ref_idx)
    ref_patch = G.gather_centre(cloud_h, q_h[rows], ref_idx)
    assert np.array_equal(patch[rt].cpu().numpy(), ref_patch.astype(np.float32))
    with torch.no_grad():
        n_ref = torch.nn.functional.normalize(O.fn_forward(sdn, torch.from_numpy(ref_patch).float(), U.FN_HP), dim=-1)
    n_gpu = nrm[rt].cpu()
    e_n = float((n_gpu - n_ref).abs().max())
    ref_rot = G.rotate_patches(ref_patch, n_gpu.numpy()).astype(np.float32)
    rot_rows = rot[rt].cpu()
    assert float((rot_rows - torch.from_numpy(ref_rot)).abs().max()) <= 1e-9
    force = knn[:, rt].cpu().long()
    with torch.no_grad():
        d_ref = O.fd_forward(sdd, rot_rows, U.FD_HP, force_idx=[force[0], force[1], force[2]])
    d_gpu = dist[rt].cpu()
    e_d = float((d_gpu - d_ref).abs().max())
    print("headline batch, 48 sampled rows: normals max err %.2e, distances max err %.2e (forced tables)" % (e_n, e_d))
    assert e_n <= TOL and e_d <= TOL
    assert np.array_equal(out[rt].cpu().numpy(), G.displace(q_h[rows], n_gpu.numpy(), d_gpu.numpy()))
    assert fd.gate_violations() == 0


# ---------------------------------------------------------------------------------------------
# farthest-point sampling (generate.py:56-74) — index work: bit-exact
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_fps_matches_reference_indices_bit_exact():
    import sapcu_amd
    from sapcu_amd import pipeline
    from sapcu_amd import testing as T
    g = golden("fps.npz")
    for name in g["names"]:
        cloud, npoint = T.fps_case(str(name))
        idx = pipeline.farthest_point_sample(cloud, npoint)
        assert idx.dtype == np.int64 and idx.shape == (npoint,)
        np.testing.assert_array_equal(idx, g[name + "_idx"], err_msg=str(name))


@pytest.mark.gpu
def test_fps_against_oracle_sizes_around_the_kernel_variants():
    """one point per thread / 2 / 8 / 32 points per thread, ragged last workgroup, npoint in {0, 1, N}"""
    from oracle import fps_path as F
    import sapcu_amd
    from sapcu_amd import pipeline
    rng = np.random.default_rng(11)
    for n, npoint in ((1, 1), (2, 2), (255, 17), (256, 0), (257, 257), (65537, 300), (131073, 200), (300001, 64),
                      (600011, 48), (1200007, 24)):
        cloud = rng.standard_normal((n, 3)) * np.array([2.0, 1.0, 0.5])
        got = pipeline.farthest_point_sample(cloud, npoint)
        np.testing.assert_array_equal(got, F.farthest_point_sample(cloud, npoint), err_msg="n=%d" % n)
    with pytest.raises(Exception):
        pipeline.farthest_point_sample(np.zeros((8, 3)), 9)
    big = torch.zeros((256 * 8192 + 1 + 8192 * 64, 3), dtype=torch.float32, device=U.dev())   # beyond the resident grid
    with pytest.raises(sapcu_amd.SapcuError):
        pipeline.farthest_point_sample_device(big, 4)


@pytest.mark.gpu
def test_fps_properties_at_refined_cloud_size():
    """385 k points -> 8192: indices distinct, first = N//2, and the greedy invariant holds: the running
    minimum distance of each pick is non-increasing and every pick realises the maximum at its step (checked
    with torch on the device for a sample of steps)."""
    import sapcu_amd
    from sapcu_amd import pipeline
    rng = np.random.default_rng(5)
    n, npoint = 385582, 8192
    x = torch.from_numpy(rng.standard_normal((n, 3))).float().to(U.dev())
    idx = pipeline.farthest_point_sample_device(x, npoint)
    assert int(idx[0]) == n // 2 and idx.unique().numel() == npoint
    distance = torch.full((n,), 1e32, device=U.dev())
    last = float("inf")
    for i in range(64):
        distance = torch.minimum(distance, ((x - x[idx[i]]) ** 2).sum(-1))
        m = float(distance.max())
        assert m <= last and float(distance[idx[i + 1]]) == m
        last = m


@pytest.mark.gpu
def test_process_cloud_pipeline_end_to_end(weights):
    """generate.py:81-99: normalise -> upsample -> denormalise -> FPS, against the same steps assembled from the
    oracle pieces around OUR upsample result (the upsample itself is covered above)."""
    from oracle import fps_path as F
    import sapcu_amd
    from sapcu_amd import pipeline
    from sapcu_amd import testing as T
    fn_gpu, fd_gpu, _, _ = U.build_gpu_models(weights)
    fn_gpu.knn_cache_mode = "fresh"
    gen = sapcu_amd.Generator3D6(fn_gpu, fd_gpu, U.dev(), batch_size=256, dense_spacing=0.03)
    raw = T.sphere_cloud(2048, 0) * 12.5 + np.array([1.0, 2.0, -3.0])
    out = pipeline.process_cloud(raw, gen, 512)
    assert out.shape == (512, 3) and out.dtype == np.float64
    cloud, loc, scale = F.normalize_pointcloud(raw)
    up = np.array(gen.upsample(cloud[None])) * scale + loc
    np.testing.assert_array_equal(out, up[F.farthest_point_sample(up, 512)])


@pytest.mark.gpu
def test_generator_edge_cases(weights):
    """no seeds, one seed, a ragged last batch, fewer cloud points than neighbours (KDTree.query's ValueError)"""
    import sapcu_amd
    from sapcu_amd import testing as T
    fn_gpu, fd_gpu, sdn, sdd = U.build_gpu_models(weights)
    fn_gpu.knn_cache_mode = "fresh"
    gen = sapcu_amd.Generator3D6(fn_gpu, fd_gpu, U.dev(), k_neighbors=48, batch_size=7)
    cloud = T.sphere_cloud(500, 0)
    assert gen.upsample_seeds(cloud, np.zeros((0, 3))).shape == (0, 3)
    q = T.grid_queries(23, 0)
    kept, full = gen.upsample_seeds(cloud, q, return_unfiltered=True)          # batches of 7/8 queries (array_split), last ragged
    assert full.shape == (23, 3) and np.isfinite(full).all()
    one, full1 = gen.upsample_seeds(cloud, q[:1], return_unfiltered=True)
    assert full1.shape == (1, 3)
    # batch composition must not matter in 'fresh' mode: one big batch gives the same points
    gen_big = sapcu_amd.Generator3D6(fn_gpu, fd_gpu, U.dev(), k_neighbors=48, batch_size=4096)
    _, full_big = gen_big.upsample_seeds(cloud, q, return_unfiltered=True)
    np.testing.assert_allclose(full, full_big, rtol=0, atol=2e-4)              # fd feature-kNN near-ties may flip: distribution bar
    assert np.median(np.abs(full - full_big)) < 1e-6
    np.testing.assert_allclose(full1[0], full_big[0], rtol=0, atol=2e-4)
    with pytest.raises(ValueError):
        gen.upsample_seeds(cloud[:40], q)                                      # 40 points < 48 neighbours


@pytest.mark.gpu
def test_multi_scale_upsample_equals_chained_upsample_calls(weights):
    """SNNPointCloudGenerator.multi_scale_upsample (generation.py:191-220): num_passes passes, each feeding its refined cloud
    to the next — two passes on a 256-point cloud equal two chained upsample() calls, and one pass equals upsample()."""
    import sapcu_amd
    from sapcu_amd import testing as T
    from sapcu_amd.generation import SNNPointCloudGenerator
    fn_gpu, fd_gpu, _, _ = U.build_gpu_models(weights)
    fn_gpu.knn_cache_mode = "fresh"
    kw = dict(k_neighbors=48, batch_size=64, dense_spacing=0.03)
    gen = SNNPointCloudGenerator(fn_gpu, fd_gpu, U.dev(), upsampling_ratio=4, **kw)
    assert gen.upsampling_ratio == 4 and isinstance(gen, sapcu_amd.Generator3D6)
    cloud = T.sphere_cloud(256, 0)
    one = gen.multi_scale_upsample(cloud, num_passes=1)
    first = np.asarray(gen.upsample(cloud[None]))
    np.testing.assert_array_equal(one, first)
    assert 256 < first.shape[0] <= 5000 and first.shape[1] == 3          # a pass's output is the next pass's input cloud (dense.cpp: <= 5000 points)
    two = gen.multi_scale_upsample(cloud, num_passes=2)
    chained = np.asarray(gen.upsample(first[None]))
    np.testing.assert_array_equal(two, chained)
    assert two.dtype == np.float64 and np.isfinite(two).all()
    np.testing.assert_array_equal(gen.multi_scale_upsample(cloud[None], num_passes=1), first)      # [1,N,3] input as upsample() takes it
    assert gen.multi_scale_upsample(cloud, num_passes=0) is cloud


def test_fused_batches_equal_batch_by_batch(weights):
    """Generator3D6.refine runs consecutive reference batches of one shape as one device pass (fuse_queries) — in
    'reference' mode with the cached neighbour tables of the first batch of that shape, as fn's shape-keyed cache would.
    The refined cloud must equal the batch-by-batch run bit for bit, in both cache modes."""
    import sapcu_amd
    from sapcu_amd import testing as T
    cloud = _dev(T.sphere_cloud(800, 0))
    seeds = _dev(T.grid_queries(100, 0))                        # batch_size 8 -> 4 batches of 9, then 8 of 8
    for mode in ("reference", "fresh"):
        outs = []
        for fuse in (0, 4096, 20):
            fn, fd, _, _ = U.build_gpu_models(weights)
            fn.knn_cache_mode = mode
            gen = sapcu_amd.Generator3D6(fn, fd, U.dev(), k_neighbors=48, batch_size=8)
            gen.fuse_queries = fuse
            with torch.no_grad():
                outs.append(gen.refine(cloud, seeds))
            if mode == "reference":
                assert sorted(fn._knn_cache) == [(8, 48), (9, 48)]          # only the reference's own batch shapes
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert torch.equal(a, b), mode


# ---------------------------------------------------------------------------------------------
# row f-4, first piece: training-mode neuron loop (hard spikes forward, surrogate gradient backward)
# ---------------------------------------------------------------------------------------------
def test_training_neuron_loop_forward_backward_against_reference_run():
    from sapcu_amd import train
    g = golden("neuron_train.npz")
    for tag in g["tags"]:
        tag = str(tag)
        x = _dev(g[tag + "_x"]).requires_grad_(True)
        raw = [_dev(g[tag + "_raw"][i]).requires_grad_(True) for i in range(4)]
        out = train.lif_selfloop_train(x, *raw, steps=int(g[tag + "_T"]))
        assert torch.equal(out.detach().cpu(), torch.from_numpy(g[tag + "_spikes"])), tag       # 0/1 spikes: exact
        (out * _dev(g[tag + "_g"])).sum().backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[tag + "_gx"], rtol=2e-5, atol=1e-6, err_msg=tag)
        for p, key in zip(raw, ("_gmd", "_gta", "_grd", "_gtb")):
            np.testing.assert_allclose(p.grad.cpu().numpy(), g[tag + key], rtol=1e-4, atol=2e-5, err_msg=tag + key)


def test_training_neuron_loop_against_oracle_autograd_at_size():
    """[4096, 512] elements (2.1 M), T = 4: forward exact, gradients against torch autograd on the oracle; the parameter
    gradients are sums over 4096 rows — compared relative to the sum of absolute contributions; run twice: bit-identical
    (deterministic reduction)."""
    from oracle import train_path as TP
    from sapcu_amd import train
    rng = np.random.default_rng(3)
    rows, C = 4096, 512
    xh = rng.normal(0.6, 1.0, (rows, C)).astype(np.float32)
    gh = rng.normal(0.0, 1.0, (rows, C)).astype(np.float32)
    rawh = np.stack([rng.uniform(0.05, 1.1, C), rng.uniform(-0.02, 0.15, C), rng.uniform(0.05, 1.0, C), rng.normal(0.7, 0.4, C)]).astype(np.float32)
    xo = torch.from_numpy(xh).requires_grad_(True)
    ro = [torch.from_numpy(rawh[i]).requires_grad_(True) for i in range(4)]
    oo = TP.lif_selfloop_train(xo, *ro, steps=4)
    (oo * torch.from_numpy(gh)).sum().backward()
    runs = []
    for _ in range(2):
        x = _dev(xh).requires_grad_(True)
        raw = [_dev(rawh[i]).requires_grad_(True) for i in range(4)]
        out = train.lif_selfloop_train(x, *raw, steps=4)
        (out * _dev(gh)).sum().backward()
        runs.append([out.detach().cpu(), x.grad.cpu()] + [p.grad.cpu() for p in raw])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    out, gx, gp = runs[0][0], runs[0][1], runs[0][2:]
    assert torch.equal(out, oo.detach())
    np.testing.assert_allclose(gx.numpy(), xo.grad.numpy(), rtol=2e-5, atol=1e-6)
    for got, want in zip(gp, ro):
        scale = float(want.grad.abs().max()) + 1e-6
        assert float((got - want.grad).abs().max()) <= 2e-4 * scale + 1e-4, float((got - want.grad).abs().max())
    with pytest.raises(ValueError):
        train.lif_selfloop_train(torch.zeros(4, 4), *[torch.zeros(4)] * 4)      # CPU tensor: no CPU path


def test_weights_are_repacked_after_a_checkpoint_load(weights):
    """The packed device blob follows the module's parameters: after load_state_dict of other weights (CheckpointIO.load
    does exactly that) or an in-place edit, the next forward uses the new values; .eval()/.reset_states() keep it."""
    fn, fd, sdn, sdd = U.build_gpu_models(weights)
    fn.knn_cache_mode = "fresh"
    patch = U.sphere_patches(6, 48, skip=100).to(U.dev())
    n0, d0 = fn(patch).clone(), fd(patch).clone()
    h0 = fn._handle.value
    fn.eval(); fn.reset_states()
    assert torch.equal(fn(patch), n0) and fn._handle.value == h0            # nothing changed: same engine
    sd2 = {k: (v * 1.03 if v.dtype.is_floating_point and "running_var" not in k and "num_batches" not in k else v) for k, v in sdn.items()}
    fn.load_state_dict(sd2, strict=True)
    n1 = fn(patch)
    assert not torch.equal(n1, n0)
    fresh, _, _, _ = U.build_gpu_models(weights)
    fresh.load_state_dict(sd2, strict=True)
    fresh.knn_cache_mode = "fresh"
    assert torch.equal(fresh(patch), n1)
    with torch.no_grad():
        fd.distance_decoder.fc_distance.bias.add_(0.25)                         # in-place edit bumps the tensor version
    assert not torch.equal(fd(patch), d0)


def _rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_training_layer_conv_bn_neuron_forward_backward():
    """fn's basic layer in training mode on the HIP ops (exact-f32 GEMM, BatchNorm batch statistics, neuron loop; wgrad,
    BatchNorm backward, data-gradient GEMM) against the reference run (fixture) and, at 24 576 rows x 128 -> 256
    channels, against torch autograd on the oracle.  A pre-activation within rounding of its threshold may flip a hard
    spike: the flip rate is bounded and the gradients are compared in relative L2."""
    from oracle import train_path as TP
    from sapcu_amd import train
    g = golden("neuron_train.npz")
    B, cin, N = g["layer_x"].shape
    cout = g["layer_w"].shape[0]
    x = _dev(g["layer_x"]).permute(0, 2, 1).reshape(B * N, cin).clone().requires_grad_(True)
    prm = [_dev(g[k]).clone().requires_grad_(True) for k in ("layer_w", "layer_b", "layer_gamma", "layer_beta")]
    raw = [_dev(g["layer_raw"][i]).clone().requires_grad_(True) for i in range(4)]
    out = train.conv_bn_lif_train(x, *prm, *raw, steps=4)
    want = torch.from_numpy(g["layer_spikes"]).permute(0, 2, 1).reshape(B * N, cout)
    assert torch.equal(out.detach().cpu(), want)
    (out * _dev(g["layer_g"]).permute(0, 2, 1).reshape(B * N, cout)).sum().backward()
    gx = torch.from_numpy(g["layer_gx"]).permute(0, 2, 1).reshape(B * N, cin)
    assert _rel_l2(x.grad.cpu(), gx) <= 2e-4
    for p, key in zip(prm, ("layer_gw", "layer_gb", "layer_ggamma", "layer_gbeta")):
        ref = torch.from_numpy(g[key]).reshape(p.shape)
        assert (p.grad.cpu() - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max())), key
    for i, p in enumerate(raw):
        ref = torch.from_numpy(g["layer_graw"][i])
        assert (p.grad.cpu() - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max()))
    # larger, against the oracle's autograd
    rng = np.random.default_rng(9)
    rows, cin, cout = 24576, 128, 256
    xh = rng.normal(0, 1, (rows, cin)).astype(np.float32)
    wh = (rng.normal(0, 1, (cout, cin)) / np.sqrt(cin)).astype(np.float32)
    bh, gah, beh = rng.normal(0, 0.1, cout).astype(np.float32), rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0.4, 0.5, cout).astype(np.float32)
    rawh = np.stack([rng.uniform(0.05, 1.1, cout), rng.uniform(-0.02, 0.15, cout), rng.uniform(0.05, 1.0, cout), rng.normal(0.7, 0.4, cout)]).astype(np.float32)
    gh = rng.normal(0, 1, (rows, cout)).astype(np.float32)
    host = [torch.from_numpy(a).clone().requires_grad_(True) for a in (xh, wh, bh, gah, beh, rawh[0], rawh[1], rawh[2], rawh[3])]
    oo = TP.conv_bn_lif_train(*host, steps=4)
    (oo * torch.from_numpy(gh)).sum().backward()
    devt = [_dev(a).clone().requires_grad_(True) for a in (xh, wh, bh, gah, beh, rawh[0], rawh[1], rawh[2], rawh[3])]
    od = train.conv_bn_lif_train(*devt, steps=4)
    (od * _dev(gh)).sum().backward()
    flips = float((od.detach().cpu() != oo.detach()).float().mean())
    assert flips <= 2e-5, flips
    names = ["x", "w", "b", "gamma", "beta", "decay", "adapt", "rdecay", "theta0"]
    for nme, hd, dd in zip(names, host, devt):
        if nme == "b":                                    # the conv bias cancels in BatchNorm: its gradient is rounding noise
            assert float(dd.grad.abs().max()) <= 1e-2
            continue
        assert _rel_l2(dd.grad.cpu(), hd.grad) <= 5e-3 + 50 * flips, (nme, _rel_l2(dd.grad.cpu(), hd.grad))


def test_training_softmax_aggregate_forward_backward():
    """fn/snn_coder.py:379-389 as a differentiable op (forward = the inference kernel, backward = softmax_agg_bwd_kernel with a
    scatter-add into grad_v) against torch autograd on the oracle, for the three block shapes' k and head sizes."""
    from oracle import train_path as TP
    from sapcu_amd import train
    rng = np.random.default_rng(21)
    for b, m, kk, d, hd in ((3, 48, 24, 128, 16), (2, 48, 18, 256, 32), (2, 20, 12, 512, 64), (1, 5, 5, 64, 8)):
        pts = b * m
        ah = rng.normal(0, 2.0, (pts * kk, d)).astype(np.float32)
        ph = rng.random((pts * kk, d)).astype(np.float32)
        vh = rng.random((pts, d)).astype(np.float32)
        ih = rng.integers(0, m, size=pts * kk).astype(np.int32)
        gh = rng.normal(0, 1, (pts, d)).astype(np.float32)
        host = [torch.from_numpy(x).clone().requires_grad_(True) for x in (ah, ph, vh)]
        ro = TP.softmax_agg(*host, torch.from_numpy(ih), m, float(np.sqrt(hd)))
        (ro * torch.from_numpy(gh)).sum().backward()
        devt = [_dev(x).clone().requires_grad_(True) for x in (ah, ph, vh)]
        rd = train.softmax_agg(*devt, _dev(ih), m, float(np.sqrt(hd)))
        (rd * _dev(gh)).sum().backward()
        assert (rd.detach().cpu() - ro.detach()).abs().max() <= 2e-6
        for name, hh, dd in zip(("a", "pe", "v"), host, devt):
            assert (dd.grad.cpu() - hh.grad).abs().max() <= 1e-5 * max(1.0, float(hh.grad.abs().max())), (name, kk, d)


def test_training_transformer_block_forward_backward_against_reference_run():
    """Row f-4: one whole MultiHeadSNNTransformerBlock in training mode composed from the HIP training ops
    (sapcu_amd.train.transformer_block_train) against the reference block's own forward and autograd gradients."""
    from sapcu_amd import train
    g = golden("block_train.npz")
    p = {str(n): _dev(g["p:" + str(n)]).clone().requires_grad_(True) for n in g["names"]}
    feats = _dev(g["features"]).clone().requires_grad_(True)
    out = train.transformer_block_train(p, _dev(g["xyz"]), feats, _dev(g["knn_idx"]))
    assert (out.detach().cpu() - torch.from_numpy(g["out"])).abs().max() <= 2e-4
    (out * _dev(g["g"])).sum().backward()
    ref = torch.from_numpy(g["g_features"])
    assert _rel_l2(feats.grad.cpu(), ref) <= 2e-3
    floor = 5e-5 * max(float(np.abs(g["g:" + str(n)]).max()) for n in g["names"])
    for n in g["names"]:
        n = str(n)
        ref = torch.from_numpy(g["g:" + n])
        got = p[n].grad.cpu() if p[n].grad is not None else torch.zeros_like(ref)
        assert float((got - ref).abs().max()) <= 5e-3 * float(ref.abs().max()) + floor, (n, float((got - ref).abs().max()))


def test_training_group_max_and_linear_ops():
    """Row f-4 building blocks: max over each patch's points with gradient to the FIRST arg-max (adaptive_max_pool1d on
    {0,1} spikes ties constantly) and the Linear layer with HIP forward / weight / data gradients, against torch."""
    from sapcu_amd import train
    rng = np.random.default_rng(3)
    for groups, m, c in ((5, 32, 640), (3, 7, 96), (1, 1, 32)):
        x = torch.tensor((rng.uniform(size=(groups * m, c)) > 0.7).astype(np.float32)) * torch.tensor(rng.integers(1, 3, (groups * m, c)).astype(np.float32))
        go = torch.tensor(rng.normal(size=(groups, c)).astype(np.float32))
        xd = x.cuda().requires_grad_(True)
        yd = train.group_max(xd, m)
        yd.backward(go.cuda())
        xh = x.clone().requires_grad_(True)
        val, arg = xh.view(groups, m, c).max(dim=1)
        first = (xh.view(groups, m, c) == val.unsqueeze(1)).float().argmax(dim=1)           # first index of the maximum
        gh = torch.zeros(groups, m, c).scatter_(1, first.unsqueeze(1), go.unsqueeze(1)).view(groups * m, c)
        assert torch.equal(yd.detach().cpu(), val.detach())
        assert torch.equal(xd.grad.cpu(), gh)
    for rows, cin, cout in ((6, 2048, 1024), (6, 256, 3), (192, 3, 64), (40, 70, 33)):
        x = torch.tensor(rng.normal(size=(rows, cin)).astype(np.float32))
        w = torch.tensor((rng.normal(size=(cout, cin)) / np.sqrt(cin)).astype(np.float32))
        b = torch.tensor(rng.normal(size=(cout,)).astype(np.float32))
        go = torch.tensor(rng.normal(size=(rows, cout)).astype(np.float32))
        d = [v.cuda().requires_grad_(True) for v in (x, w, b)]
        h = [v.clone().requires_grad_(True) for v in (x, w, b)]
        yd = train.linear_train(*d)
        yh = torch.nn.functional.linear(*h)
        assert (yd.detach().cpu() - yh.detach()).abs().max() <= 2e-5 * max(1.0, float(yh.abs().max()))
        yd.backward(go.cuda())
        yh.backward(go)
        for dd, hh in zip(d, h):
            assert (dd.grad.cpu() - hh.grad).abs().max() <= 2e-5 * max(1.0, float(hh.grad.abs().max())), (rows, cin, cout)


def test_training_step_of_whole_fn_model_against_reference_run():
    """Row f-4: one training step of the whole fn model (train() mode, dropout off) composed from the HIP training ops —
    unit normals, loss and the gradient of every parameter against the reference's own run (tests/golden/fn_train.npz).
    The in-patch kNN tables come from sapcu_patch_knn and must equal the reference's."""
    from sapcu_amd import train
    from test_oracle_golden import _fn_train_params, check_fn_train_grads
    g = golden("fn_train.npz")
    p, names = _fn_train_params(g, "cuda")
    pts = _dev(g["points"])
    for i, k in enumerate((24, 18, 12)):
        idx = train.inpatch_knn(pts, k).cpu().numpy()
        assert np.array_equal(np.sort(idx, axis=2), np.sort(g["knn%d" % i], axis=2))
    normals = train.fn_train_forward(p, pts)
    assert (normals.detach().cpu() - torch.from_numpy(g["normals"])).abs().max() <= 2e-4
    loss, _ = train.angular_loss_with_consistency(normals, _dev(g["gt"]), None)      # the product's loss (no consistency term: [B, 3])
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-4
    loss.backward()
    check_fn_train_grads(g, p, names, 2e-2, 5e-5)


def test_training_softmax_aggregate_with_attention_dropout():
    """The attention dropout of train() mode (fn:383) inside the softmax-aggregate op: forward and the three gradients with a
    given keep mask against the same composition in torch autograd."""
    from sapcu_amd import train
    rng = np.random.default_rng(9)
    B, N, k, d, H = 3, 16, 6, 64, 8
    P = B * N
    a, pe = (torch.tensor(rng.normal(size=(P * k, d)).astype(np.float32)) for _ in range(2))
    v = torch.tensor(rng.normal(size=(P, d)).astype(np.float32))
    idx = torch.tensor(rng.integers(0, N, (P * k,)).astype(np.int32))
    go = torch.tensor(rng.normal(size=(P, d)).astype(np.float32))
    keep = train.dropout_keep((P * k, d), 0.3, "cuda")
    assert set(np.unique(keep.cpu().numpy()).round(5)) <= {0.0, np.float32(1 / 0.7).round(5)}
    assert 0.2 < float((keep == 0).float().mean()) < 0.4
    dv = [x.cuda().requires_grad_(True) for x in (a, pe, v)]
    out = train.softmax_agg(*dv, idx.cuda(), N, float((d // H) ** 0.5), keep)
    out.backward(go.cuda())
    hv = [x.clone().requires_grad_(True) for x in (a, pe, v)]
    w = torch.softmax(hv[0].view(P, k, d) / float((d // H) ** 0.5), dim=1) * keep.cpu().view(P, k, d)
    nb = (idx.view(P, k).to(torch.int64) + (torch.arange(P) // N * N).view(P, 1))
    ref = (w * (hv[2][nb] + hv[1].view(P, k, d))).sum(1)
    ref.backward(go)
    assert (out.detach().cpu() - ref.detach()).abs().max() <= 1e-5
    for dd, hh in zip(dv, hv):
        assert (dd.grad.cpu() - hh.grad).abs().max() <= 1e-5 * max(1.0, float(hh.grad.abs().max()))


def test_fn_trainer_step_against_reference_trainer_run():
    """Row f-4: sapcu_amd.fn_trainer.Trainer.train_step on the drop-in model in train() mode (dropout off) against the
    reference Trainer's own step: returned loss and confidence, every parameter update after global-norm clipping + SGD, and
    every BatchNorm running statistic (tests/golden/fn_trainer.npz); then the model goes back to eval() and infers."""
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T
    from test_oracle_golden import check_fn_trainer_updates
    g = golden("fn_trainer.npz")
    model = sapcu_amd.ImprovedSNNNormalEstimation(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8,
                                                  use_snn_decoder=False, decoder_dropout=0.1)
    model.load_state_dict(T.training_state_dict(model.state_dict(), int(g["seed"])), strict=True)
    model.attn_dropout = model.decoder_dropout = 0.0
    model.cuda()
    names = [str(n) for n in g["names"]]
    old = {n: prm.detach().clone() for n, prm in model.named_parameters()}
    opt = torch.optim.SGD(model.parameters(), lr=float(g["lr"]))
    tr = fn_trainer.Trainer(model, opt, device=torch.device("cuda"), grad_clip=float(g["grad_clip"]), grad_clip_type="norm")
    loss, loss_dict = tr.train_step({"input": torch.from_numpy(g["points"]), "normal": torch.from_numpy(g["gt"])})
    assert loss is not None and abs(loss - float(g["loss"])) <= 2e-4 and abs(loss_dict["confidence"] - float(g["confidence"])) <= 2e-4
    check_fn_trainer_updates(g, dict(model.named_parameters()), old, names, 3e-2, 1e-4)
    bufs = dict(model.named_buffers())
    for n in g["buffers"]:
        n = str(n)
        ref = torch.from_numpy(np.asarray(g["b:" + n]))
        got = bufs[n].cpu()
        if n.endswith("num_batches_tracked"):
            assert int(got) == int(ref) == 1
        else:
            assert (got - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max())), n
    assert all(prm.grad is None or float(prm.grad.abs().max()) == 0.0 for prm in model.parameters())      # zero_grad ran
    # with dropout on, a step still runs and is finite; afterwards eval() re-packs the updated parameters and infers
    model.attn_dropout, model.decoder_dropout = 0.1, 0.1
    loss2, _ = tr.train_step({"input": torch.from_numpy(g["points"]), "normal": torch.from_numpy(g["gt"])})
    assert loss2 is not None and np.isfinite(loss2)
    model.eval()
    with torch.no_grad():
        n_eval = model(torch.from_numpy(g["points"]).cuda())
    assert n_eval.shape == (2, 8, 3) and bool(torch.isfinite(n_eval).all())
    assert (n_eval.norm(dim=-1) - 1).abs().max() <= 1e-5


# ---------------------------------------------------------------------------------------------
# bf16 training (BASELINE config 5: "trainfn.py one epoch bf16"): the bf16-operand GEMMs against bf16-rounded references,
# then one epoch of the trainfn.py loop on a synthetic PU1K-shaped loader — loss curve of the bf16 run against the f32 HIP
# run and against the oracle's f32 restatement of the reference's train() forward + torch autograd on the CPU.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("r,k,n", [(1, 32, 3), (130, 64, 64), (3072, 192, 640), (5000, 512, 128), (36864, 128, 128), (257, 2048, 1024)])
def test_bf16_training_gemms_against_bf16_rounded_references(r, k, n):
    """sapcu_gemm_bf16 (forward / data gradient) and sapcu_conv1x1_wgrad_bf16 (weight gradient, row slabs) against float64
    products of the bf16-rounded operands: the only differences allowed are those of the f32 accumulation order."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(r + k + n)
    a = rng.normal(size=(r, k)).astype(np.float32)
    w = (rng.normal(size=(n, k)) / np.sqrt(k)).astype(np.float32)
    bias = rng.normal(size=n).astype(np.float32)
    rb = lambda x: torch.from_numpy(x).bfloat16().double().numpy()
    A, W, Bv = _dev(a), _dev(w), _dev(bias)
    C = torch.full((r, n), float("nan"), device=U.dev())
    _lib.check(lib.sapcu_gemm_bf16(_lib.ptr(A), r, k, k, _lib.ptr(W), n, _lib.ptr(Bv), _lib.ptr(C), n, _lib.current_stream()))
    want = rb(a) @ rb(w).T + bias.astype(np.float64)
    err = np.abs(C.cpu().numpy() - want).max()
    assert err <= 4e-6 * np.sqrt(k) * max(1.0, np.abs(want).max()), err
    # bf16 really is what was multiplied: the exact-f32 product differs by ~2^-9 relative per operand
    exact = a.astype(np.float64) @ w.astype(np.float64).T + bias
    if r * k * n > 100000:
        assert np.abs(C.cpu().numpy() - exact).max() > 20 * err
    # weight gradient: dw[n, k] = dy^T . x over r rows (512-row slabs), db = column sums in f32
    dy = (rng.normal(size=(r, n)) * 0.1).astype(np.float32)
    DY = _dev(dy)
    dw = torch.full((n, k), float("nan"), device=U.dev())
    db = torch.full((n,), float("nan"), device=U.dev())
    need = int(lib.sapcu_wgrad_bf16_workspace_bytes(r, n, k))
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=U.dev())
    _lib.check(lib.sapcu_conv1x1_wgrad_bf16(_lib.ptr(DY), n, _lib.ptr(A), k, r, n, k, _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), ws.numel(),
                                            _lib.current_stream()))
    want_w = rb(dy).T @ rb(a)
    assert np.abs(dw.cpu().numpy() - want_w).max() <= 4e-6 * np.sqrt(r) * max(1.0, np.abs(want_w).max())
    assert np.abs(db.cpu().numpy() - dy.astype(np.float64).sum(0)).max() <= 2e-6 * np.sqrt(r) * max(1.0, np.abs(dy).sum(0).max())
    assert lib.sapcu_conv1x1_wgrad_bf16(_lib.ptr(DY), n, _lib.ptr(A), k, r, n, k, _lib.ptr(dw), None, None, 0, None) == -2   # SAPCU_ERR_WORKSPACE


def test_training_epoch_bf16_tracks_the_f32_and_oracle_loss_curves():
    """One epoch of the trainfn.py:253-330 loop (fn_trainer.run_epoch) over a synthetic PU1K-shaped loader, AdamW + global-norm
    clipping + learning-rate warm-up as config/fn.yaml, dropout off.  Since round 4 a training step on the device is reproducible
    bit for bit (the backward's scatter-adds are fixed-order segmented sums), so the HIP f32 trajectory is ONE trajectory:
      * the epoch run batch by batch (one run_epoch call per batch) gives the same losses, bit for bit, as the epoch run at once;
      * f32 HIP vs the oracle, TEACHER-FORCED per step: before every step the oracle (CPU torch autograd restatement of the train()
        forward) gets the device model's current parameters and evaluates the same batch.  Hard spikes make the forward piecewise
        constant: the two agree to ~1e-6 unless a spike sits within rounding of its threshold, and one flipped spike moves a
        32-patch loss by up to ~0.03-0.08 (test_training_first_step_difference_is_spike_flips).  Teacher forcing keeps such a flip
        from compounding through the optimiser (the free-running comparison of rounds 2-3 needed bars of 0.12 and was still at
        the mercy of the CPU oracle's own thread-dependent sums: its step-2 loss was 1.7147 on one box and 1.7918 on another).
        Bars: step 1 (the loader's seed keeps it off every threshold) <= 1e-3; every step <= 0.1 (about three flips); mean <= 0.05.
        Measured on the first run of this form: 7e-7, 0.0, 0.0097, 0.020, 0.048, 0.033 — forward-only differences under identical
        parameters, i.e. zero, one or two spikes landing on the other side of a threshold in a 32-patch batch (an up-front
        "median <= 0.01" was refuted by that run: from the third update on most steps have a flip);
      * bf16 vs f32 HIP (different GEMM arithmetic from the first step on: the two runs flip different spikes and drift apart
        through the optimiser — a band, not an identity): per step <= 0.3, mean <= 0.12, epoch means within 0.06.
    Every loss finite; parameters of the two HIP runs within 20 x lr x steps."""
    import copy
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T
    from oracle import train_path as TP
    kw = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8, use_snn_decoder=False, decoder_dropout=0.1)
    sd = T.training_state_dict(sapcu_amd.ImprovedSNNNormalEstimation(**kw).state_dict(), 3)
    loader = lambda: fn_trainer.SyntheticPU1K(batches=6, batch_size=2, patches=16, points=12, seed=7)
    lr, clip, warm = 1.8e-4, 0.15, 4
    names = [n for n, _ in sapcu_amd.ImprovedSNNNormalEstimation(**kw).named_parameters()]

    def make(use_amp):
        model = sapcu_amd.ImprovedSNNNormalEstimation(**kw)
        model.load_state_dict(copy.deepcopy(sd), strict=True)
        model.attn_dropout = model.decoder_dropout = 0.0
        model.cuda()
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4, betas=(0.9, 0.999))
        return model, fn_trainer.Trainer(model, opt, device=torch.device("cuda"), use_amp=use_amp, grad_clip=clip, grad_clip_type="norm")

    curves, params, stats = {}, {}, {}
    for mode in ("f32", "bf16"):
        model, tr = make(mode == "bf16")
        it, losses, st = fn_trainer.run_epoch(tr, loader(), lr=lr, warmup_steps=warm, warmup_factor=0.01, state_reset_freq=25)
        assert it == 6 and len(losses) == 6 and st["skipped"] == 0 and all(np.isfinite(losses)), (mode, losses)
        curves[mode], stats[mode] = losses, st
        params[mode] = {n: q.detach().cpu().clone() for n, q in model.named_parameters()}

    def oracle_loss(p, batch):
        pts = batch["input"]
        B, NP, M, _ = pts.shape
        flat = pts.reshape(B * NP, M, 3)
        dist = ((flat[:, :, None, :] - flat[:, None, :, :]) ** 2).sum(-1)
        knn = [dist.topk(min(k, M), dim=-1, largest=False)[1] for k in (24, 18, 12)]
        with torch.no_grad():
            pred = torch.nn.functional.normalize(TP.fn_train_forward(p, flat, knn).view(B, NP, 3), dim=-1)
            loss, _ = TP.angular_loss_with_consistency(pred, torch.nn.functional.normalize(batch["normal"], dim=-1), pts.mean(dim=2))
        return float(loss)

    # the same epoch batch by batch, the oracle teacher-forced with the device model's parameters before every step
    model, tr = make(False)
    it, stepwise, oracle_losses = 0, [], []
    for batch in loader():
        snap = {n: q.detach().cpu().clone() for n, q in model.named_parameters()}
        it, l, _ = fn_trainer.run_epoch(tr, [batch], it=it, lr=lr, warmup_steps=warm, warmup_factor=0.01, state_reset_freq=25)
        stepwise.append(l[0])
        oracle_losses.append(oracle_loss(snap, batch))
    assert stepwise == curves["f32"], ("the device epoch is not reproducible", stepwise, curves["f32"])
    for n in names:
        assert torch.equal(params["f32"][n], dict(model.named_parameters())[n].detach().cpu()), n
    d_bf = [abs(a - b) for a, b in zip(curves["bf16"], curves["f32"])]
    d_or = [abs(a - b) for a, b in zip(curves["f32"], oracle_losses)]
    worst = max(float((params["f32"][n] - params["bf16"][n]).abs().max()) for n in names)
    print("epoch losses  f32: %s\n              bf16: %s\n  oracle (forced): %s\n  |bf16 - f32| max %.4f, |f32 - oracle| per step %s, parameter drift %.3g; "
          "%.1f / %.1f clouds/s (f32 / bf16)" % (np.round(curves["f32"], 4), np.round(curves["bf16"], 4), np.round(oracle_losses, 4),
                                                max(d_bf), np.array2string(np.array(d_or), precision=2), worst, stats["f32"]["clouds_per_s"],
                                                stats["bf16"]["clouds_per_s"]))
    assert max(d_bf) <= 0.3 and float(np.mean(d_bf)) <= 0.12, d_bf
    assert d_or[0] <= 1e-3 and max(d_or) <= 0.1 and float(np.mean(d_or)) <= 0.05, d_or
    assert abs(float(np.mean(curves["bf16"])) - float(np.mean(curves["f32"]))) <= 0.06
    assert worst <= 20 * lr * 6


def test_grouped_scatter_sum_is_deterministic_and_equals_index_add():
    """sapcu_scatter_add_rows_grouped (the backward of index_points on patch-structured indices, and — inside
    sapcu_softmax_agg_backward — the sum of grad_v): every destination row = the sum of its sources in ascending source order,
    stored, no atomics.  Against torch's index_add_ in float64 (1e-6 relative), against an explicit fixed-order float32 sum bit for
    bit, twice bit for bit, destinations nobody points to = 0, ragged shapes, and the out-of-group counter."""
    from sapcu_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(5)
    for groups, gs, k, d in ((9, 12, 12, 128), (3, 48, 24, 64), (2, 100, 12, 33), (1, 5, 3, 7)):
        gr = gs * k
        local = torch.stack([torch.stack([torch.randperm(gs, generator=g)[:k] for _ in range(gs)]) for _ in range(groups)])   # [groups, gs, k]
        local[:, :, 0] = 0 if gs > 1 else local[:, :, 0]               # one hot destination per group; some rows receive nothing
        index = (local + torch.arange(groups).view(-1, 1, 1) * gs).reshape(-1).to(torch.int64)
        gout = torch.randn((groups * gr, d), generator=g) * torch.logspace(-3, 3, groups * gr).view(-1, 1)
        ref64 = torch.zeros((groups * gs, d), dtype=torch.float64).index_add_(0, index, gout.double())
        ref32 = torch.zeros((groups * gs, d), dtype=torch.float32)
        for r in range(groups * gr):                                   # the kernel's order: ascending source row
            ref32[index[r]] += gout[r]
        outs = []
        gout_d, index_d = _dev(gout), _dev(index)                       # (kept alive: a temporary's block would be re-used at once)
        for _ in range(2):
            out = torch.full((groups * gs, d), float("nan"), device=U.dev())
            bad = torch.zeros(1, dtype=torch.int32, device=U.dev())
            _lib.check(lib.sapcu_scatter_add_rows_grouped(_lib.ptr(gout_d), _lib.ptr(index_d), groups * gr, d, _lib.ptr(out), d, groups * gs,
                                                          gs, gr, _lib.ptr(bad), _lib.current_stream()))
            outs.append(out.cpu())
            assert int(bad.item()) == 0
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], ref32), (groups, gs, k, d)
        scale = ref64.abs().max().item()
        assert float((outs[0].double() - ref64).abs().max()) <= 1e-5 * scale
        wrong = index.clone()
        wrong[0] = (int(index[0]) + gs) % (groups * gs) if groups > 1 else -1       # points into another group
        bad = torch.zeros(1, dtype=torch.int32, device=U.dev())
        out = torch.empty((groups * gs, d), device=U.dev())
        wrong_d = _dev(wrong)
        _lib.check(lib.sapcu_scatter_add_rows_grouped(_lib.ptr(gout_d), _lib.ptr(wrong_d), groups * gr, d, _lib.ptr(out), d, groups * gs,
                                                      gs, gr, _lib.ptr(bad), _lib.current_stream()))
        assert int(bad.item()) == 1
    assert lib.sapcu_scatter_add_rows_grouped(_lib.ptr(gout_d), _lib.ptr(index_d), 7, 7, _lib.ptr(out), 7, 5, 5, 3, None, _lib.current_stream()) != 0


def test_training_epoch_is_reproducible_bit_for_bit():
    """with the scatter-adds of the backward as fixed-order segmented sums (no float atomics left in a training
    step) two runs of the same epoch from the same state are the SAME run: every loss and every parameter bit for bit, f32 and
    bf16, and so is the HIP-graph replay of the step against itself."""
    import copy
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T
    kw = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8, use_snn_decoder=False, decoder_dropout=0.1)
    sd = T.training_state_dict(sapcu_amd.ImprovedSNNNormalEstimation(**kw).state_dict(), 3)
    lr = 1.8e-4
    for use_amp in (False, True):
        runs = []
        for rep in range(2):
            model = sapcu_amd.ImprovedSNNNormalEstimation(**kw)
            model.load_state_dict(copy.deepcopy(sd), strict=True)
            model.attn_dropout = model.decoder_dropout = 0.1                  # dropout ON: its masks come from a seeded generator
            model.cuda()
            torch.manual_seed(11)
            torch.cuda.manual_seed(11)
            opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4, betas=(0.9, 0.999))
            tr = fn_trainer.Trainer(model, opt, device=torch.device("cuda"), use_amp=use_amp, grad_clip=0.15, grad_clip_type="norm")
            loader = fn_trainer.SyntheticPU1K(batches=8, batch_size=4, patches=16, points=12, seed=5)      # seed 5: spikes ON thresholds
            it, losses, st = fn_trainer.run_epoch(tr, loader, lr=lr, warmup_steps=4, warmup_factor=0.01, state_reset_freq=25)
            assert it == 8 and st["skipped"] == 0 and all(np.isfinite(losses))
            runs.append((losses, {n: q.detach().cpu().clone() for n, q in model.named_parameters()},
                         {n: b.detach().cpu().clone() for n, b in model.named_buffers()}))
        assert runs[0][0] == runs[1][0], ("losses differ between two runs", use_amp, runs[0][0], runs[1][0])
        for n in runs[0][1]:
            assert torch.equal(runs[0][1][n], runs[1][1][n]), (use_amp, n)
        for n in runs[0][2]:
            assert torch.equal(runs[0][2][n], runs[1][2][n]), (use_amp, n)


def test_training_hard_spike_flips_sit_on_their_thresholds():
    """Why the f32 HIP training forward and the oracle's f32 restatement differ at the FIRST step of the epoch test although the
    parameters are identical: hard spikes.  Five neuron layers of that step — conv1 / snn_init and block 1's fc1 / snn1, w_qs / snn_q,
    w_ks / snn_k, w_vs / snn_v (conv + BatchNorm(batch statistics) + LIF x 4), each fed the ORACLE's input so that a difference cannot
    come from upstream — on the device and in the oracle: every spike that differs belongs to an element whose membrane passes
    within 1e-5 of its threshold at some step of the SAME computation in float64 (two f32 summation orders land on either side of
    it), and no element away from its threshold (margin > 1e-4) differs.  BatchNorm couples the patches of a batch, so downstream one
    flipped spike moves every patch a little: the whole-model comparison is the loss bar of the epoch test; this test pins the
    mechanism behind it."""
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T, train as TR
    from oracle import train_path as TP
    kw = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8, use_snn_decoder=False, decoder_dropout=0.1)
    sd = T.training_state_dict(sapcu_amd.ImprovedSNNNormalEstimation(**kw).state_dict(), 3)
    batch = next(iter(fn_trainer.SyntheticPU1K(batches=6, batch_size=2, patches=16, points=12, seed=5)))
    pts = batch["input"]
    x0 = pts.reshape(-1, 3)
    e = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}

    def layer(x, conv, bn, snn):
        """(device spikes, oracle spikes, float64 margin min_t |m_t - theta_t|) of one conv + BN(train) + LIF x 4 layer on input x"""
        W = e[conv + ".weight"].reshape(e[conv + ".weight"].shape[0], -1)
        args = [e[conv + ".bias"], e[bn + ".weight"], e[bn + ".bias"], e[snn + ".membrane_decay"], e[snn + ".threshold_adapt"],
                e[snn + ".refractory_decay"], e[snn + ".threshold_base"]]
        with torch.no_grad():
            dev_spk = TR.conv_bn_lif_train(TR._pad_channels(x.to(U.dev())), TR._pad_channels(W.to(U.dev())), *[t.to(U.dev()) for t in args],
                                           steps=4, eps=1e-5).cpu()
            or_spk = TP.conv_bn_lif_train(x, W, *args, steps=4, eps=1e-5)
            y = x.double() @ W.double().t() + args[0].double()                  # the same layer in float64 (fn/snn_coder.py:125-151)
            z = (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5) * args[1].double() + args[2].double()
            decay, adapt, rdecay = args[3].double().clamp(0.1, 0.99), args[4].double().clamp(0.001, 0.1), args[5].double().clamp(0.1, 0.95)
            th0 = args[6].double()
            m, th, r, inp = torch.zeros_like(z), th0.expand_as(z).clone(), torch.zeros_like(z), z
            margin = torch.full_like(z, float("inf"))
            for _ in range(4):
                inp = inp * (r <= 0).double()
                m = m * decay * (1 - r) + inp
                margin = torch.minimum(margin, (m - th).abs())
                sp = (m - th > 0).double()
                m = m * (1 - sp)
                r = r * rdecay + sp
                th = th0 + ((th + adapt * sp) - th0) * 0.95
                inp = sp
        return dev_spk, or_spk, margin

    total_flips = 0
    feat = None
    for name, conv, bn, snn, src in (("conv1", "conv1.0", "conv1.1", "snn_init", "x0"), ("trans1.fc1", "trans1.fc1.0", "trans1.fc1.1", "trans1.snn1", "feat"),
                                      ("trans1.w_qs", "trans1.w_qs.0", "trans1.w_qs.1", "trans1.snn_q", "x1"),
                                      ("trans1.w_ks", "trans1.w_ks.0", "trans1.w_ks.1", "trans1.snn_k", "x1"),
                                      ("trans1.w_vs", "trans1.w_vs.0", "trans1.w_vs.1", "trans1.snn_v", "x1")):
        x = {"x0": x0, "feat": feat, "x1": locals().get("x1")}[src]
        dev_spk, or_spk, margin = layer(x, conv, bn, snn)
        if name == "conv1":
            feat = or_spk                                                       # teacher forcing: the oracle's activations feed the next layer
        if name == "trans1.fc1":
            x1 = or_spk
        assert set(np.unique(dev_spk.numpy())) <= {0.0, 1.0} and dev_spk.shape == or_spk.shape
        flip = dev_spk != or_spk
        total_flips += int(flip.sum())
        print("%-12s %5d of %7d spikes differ (device vs oracle, same input); float64 threshold margin of the differing ones: max %.3g; "
              "elements within 1e-5 of a threshold: %d" % (name, int(flip.sum()), flip.numel(), float(margin[flip].max()) if flip.any() else 0.0,
                                                          int((margin < 1e-5).sum())))
        assert float(flip.float().mean()) <= 0.01, name
        assert not bool(flip[margin > 1e-4].any()), name + ": a spike away from its threshold differs — an arithmetic defect, not a rounding flip"
    print("differing spikes over the five teacher-forced layers: %d" % total_flips)


def test_training_first_step_difference_is_spike_flips():
    """The whole fn training forward (identical parameters, first batch of three loaders) on the device, in the oracle (f32) and in
    the oracle run in float64:
      * seed 7 (the epoch test's data): all three agree on every patch to 1e-4 (measured 6e-6) — the device forward has no
        arithmetic difference from the reference's forward where no spike sits on a threshold;
      * seed 5: the oracle disagrees WITH ITSELF between f32 and f64 on most patches (measured 31 of 32 beyond 1e-3, normals up to
        0.55 apart): that batch has spikes on their thresholds, and BatchNorm's batch statistics spread one flip over every patch —
        the device differs from the f32 oracle the same way (the 0.08 first-step loss gap round 2's epoch test saw on this data);
      * seed 6: the oracle's two precisions agree, the device lands a few spikes on the other side (its f32 sums run in another
        order than torch's): normals within 0.05, loss within 0.005 (measured 0.023 / 0.0007)."""
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T, train as TR
    from oracle import train_path as TP
    kw = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8, use_snn_decoder=False, decoder_dropout=0.1)
    sd = T.training_state_dict(sapcu_amd.ImprovedSNNNormalEstimation(**kw).state_dict(), 3)
    names = [n for n, _ in sapcu_amd.ImprovedSNNNormalEstimation(**kw).named_parameters()]
    res = {}
    for seed in (7, 5, 6):
        batch = next(iter(fn_trainer.SyntheticPU1K(batches=6, batch_size=2, patches=16, points=12, seed=seed)))
        pts = batch["input"]
        B, NP, M, _ = pts.shape
        flat = pts.reshape(B * NP, M, 3)
        dist = ((flat[:, :, None, :] - flat[:, None, :, :]) ** 2).sum(-1)
        knn = [dist.topk(min(k, M), dim=-1, largest=False)[1] for k in (24, 18, 12)]
        with torch.no_grad():
            n32 = TP.fn_train_forward({n: sd[n] for n in names}, flat, knn)
            n64 = TP.fn_train_forward({n: sd[n].double() for n in names}, flat.double(), knn).float()
            nd = TR.fn_train_forward({n: v.to(U.dev()) for n, v in sd.items()}, flat.to(U.dev()),
                                     knn=[k.to(torch.int32).to(U.dev()) for k in knn]).cpu()
        gt = torch.nn.functional.normalize(batch["normal"], dim=-1)
        loss = {nm: float(TP.angular_loss_with_consistency(torch.nn.functional.normalize(v.view(B, NP, 3), dim=-1), gt, pts.mean(dim=2))[0])
                for nm, v in (("dev", nd), ("o32", n32), ("o64", n64))}
        per = lambda a, b: (a - b).abs().max(1)[0]
        res[seed] = dict(dev_o32=per(nd, n32), o32_o64=per(n32, n64), loss=loss)
        print("seed %d: |dev - oracle32| max %.3g (%d of %d patches > 1e-3); |oracle32 - oracle64| max %.3g (%d patches > 1e-3); losses %s" % (
            seed, float(res[seed]["dev_o32"].max()), int((res[seed]["dev_o32"] > 1e-3).sum()), B * NP, float(res[seed]["o32_o64"].max()),
            int((res[seed]["o32_o64"] > 1e-3).sum()), {k: round(v, 5) for k, v in loss.items()}))
    assert float(res[7]["dev_o32"].max()) <= 1e-4 and float(res[7]["o32_o64"].max()) <= 1e-4
    assert abs(res[7]["loss"]["dev"] - res[7]["loss"]["o32"]) <= 1e-4
    assert int((res[5]["o32_o64"] > 1e-3).sum()) >= 16, "seed 5 was chosen because the oracle's own precisions disagree on it"
    assert float(res[6]["o32_o64"].max()) <= 1e-4 and float(res[6]["dev_o32"].max()) <= 0.05
    assert abs(res[6]["loss"]["dev"] - res[6]["loss"]["o32"]) <= 0.005


def test_training_epoch_of_50_batches_at_the_reference_batch_shape():
    """BASELINE config 5 at the reference's own batch shape (config/fn.yaml: 4 clouds x 64 patches x 12 points), 50 batches of the
    trainfn.py:253-330 loop on bf16 GEMM operands and on f32: no skipped batch, every loss finite and in the band the reference
    reports for this loss (Observations.md: 1.58-1.61), epoch means of the two arithmetics within 0.06, parameters moved."""
    import copy
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T
    kw = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=6, time_steps_dec=12, num_heads=8, use_snn_decoder=False, decoder_dropout=0.1)
    sd = T.training_state_dict(sapcu_amd.ImprovedSNNNormalEstimation(**kw).state_dict(), 3)
    lr, means = 1.8e-4, {}
    for mode in ("f32", "bf16"):
        model = sapcu_amd.ImprovedSNNNormalEstimation(**kw)
        model.load_state_dict(copy.deepcopy(sd), strict=True)
        model.cuda()
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4, betas=(0.9, 0.999))
        tr = fn_trainer.Trainer(model, opt, device=torch.device("cuda"), use_amp=(mode == "bf16"), grad_clip=0.15, grad_clip_type="norm")
        loader = fn_trainer.SyntheticPU1K(batches=50, batch_size=4, patches=64, points=12, seed=11)
        it, losses, st = fn_trainer.run_epoch(tr, loader, lr=lr, warmup_steps=20, warmup_factor=0.01, state_reset_freq=25)
        assert it == 50 and len(losses) == 50 and st["skipped"] == 0 and all(np.isfinite(losses)), (mode, losses)
        assert 1.2 <= min(losses) and max(losses) <= 2.0, (mode, min(losses), max(losses))
        moved = max(float((q.detach().cpu() - sd[n]).abs().max()) for n, q in model.named_parameters())
        assert 0 < moved <= 50 * lr * 4, moved
        means[mode] = float(np.mean(losses))
        print("%s: 50 batches of 4 x 64 x 12 in %.2f s = %.0f clouds/s, loss first %.4f mean %.4f last %.4f" % (
            mode, st["seconds"], st["clouds_per_s"], losses[0], means[mode], losses[-1]))
    assert abs(means["f32"] - means["bf16"]) <= 0.06, means


@pytest.mark.parametrize("r,k,n,lif,csplit", [(1024, 128, 128, 0, 0), (2048 + 77, 256, 256, 1, 1), (4096 + 3, 512, 512, 0, 0), (3000, 512, 512, 1, 0),
                                              (1500, 64, 384, 1, 1), (70000, 512, 512, 1, 1), (36864, 128, 128, 1, 1)])
def test_big_tile_gemm_is_bit_identical_to_the_ring_kernel(r, k, n, lif, csplit):
    """gemm_sf16_bt.hip (256-row tiles, epilogue in the MFMA waves) against gemm_sf16_ring.hip (a_split_rows = 2) through the C ABI
    on the same split-row operands: same products in the same order, so every output bit must agree — bias-only and
    neuron epilogues, f32 and split-row outputs, ragged last row tile, 1/2/3 column tiles."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(r + n)
    A = _dev(rng.normal(size=(r, k)).astype(np.float32))
    W = _dev((rng.normal(size=(n, k)) / np.sqrt(k)).astype(np.float32))
    Bv = _dev(rng.normal(size=n).astype(np.float32))
    L = _dev(np.stack([rng.uniform(0.05, 1.1, n), rng.uniform(0.0, 0.2, n), rng.uniform(0.05, 1.0, n), rng.normal(0.5, 0.3, n)]).astype(np.float32))
    As = torch.empty_like(A)
    _lib.check(lib.sapcu_to_split_rows(_lib.ptr(A), r, k, k, _lib.ptr(As), k, _lib.current_stream()))
    outs = []
    for a_split in (2, 1):                       # ring kernel only / the models' choice = big tile (sapcu.h: a_split_rows)
        C = torch.full((r, n), float("nan"), device=U.dev())
        ws = torch.zeros(4 * n * k + 16, dtype=torch.uint8, device=U.dev())
        _lib.check(lib.sapcu_gemm_f32(_lib.ptr(As), r, k, k, _lib.ptr(W), n, _lib.ptr(Bv), _lib.ptr(L) if lif else None, 4, _lib.ptr(C), n,
                                      _lib.ptr(ws), a_split, csplit, _lib.current_stream()))
        torch.cuda.synchronize()
        outs.append(C.cpu().view(torch.int32))
    assert not bool(torch.isnan(outs[1].view(torch.float32)).any()) or csplit
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("b,m,kk,d", [(4, 48, 12, 512), (3, 48, 18, 256), (5, 48, 24, 128), (37, 48, 12, 512)])
def test_big_tile_posenc_attention_gemm_is_bit_identical_to_the_ring_kernel(b, m, kk, d):
    """The production form of sapcu_posenc_gemm_f32 (split rows in, pe f32 + attn_in split rows out, q/k gathers) on the
    big-tile kernel (split_rows = 1) against the ring kernel (split_rows = 2): every bit of both outputs."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(b * 1000 + d)
    r = b * m * kk
    P1 = _dev(rng.random((r, d)).astype(np.float32))
    W, Bv = _dev((rng.normal(size=(d, d)) / np.sqrt(d)).astype(np.float32)), _dev(rng.normal(size=d).astype(np.float32))
    L = _dev(np.stack([rng.uniform(0.05, 1.1, d), rng.uniform(0.0, 0.2, d), rng.uniform(0.05, 1.0, d), rng.normal(0.5, 0.3, d)]).astype(np.float32))
    Q, I = _dev(rng.random((b * m, 3 * d)).astype(np.float32)), _dev(rng.integers(0, m, size=(b, m, kk)).astype(np.int32))
    P1s = torch.empty_like(P1)
    _lib.check(lib.sapcu_to_split_rows(_lib.ptr(P1), r, d, d, _lib.ptr(P1s), d, _lib.current_stream()))
    outs = []
    for split_rows in (2, 1):
        pe = torch.full((r, d), float("nan"), device=U.dev())
        att = torch.full((r, d), float("nan"), device=U.dev())
        tab = torch.empty((r, 2), dtype=torch.int32, device=U.dev())
        ws = torch.zeros(4 * d * d + 16, dtype=torch.uint8, device=U.dev())
        _lib.check(lib.sapcu_posenc_gemm_f32(_lib.ptr(P1s), r, d, _lib.ptr(W), _lib.ptr(Bv), _lib.ptr(L), 4, _lib.ptr(Q), _lib.ptr(I), kk, m,
                                             _lib.ptr(pe), _lib.ptr(att), _lib.ptr(tab), _lib.ptr(ws), split_rows, _lib.current_stream()))
        torch.cuda.synchronize()
        outs.append((pe.cpu().view(torch.int32), att.cpu().view(torch.int32)))
    assert not bool(torch.isnan(outs[1][0].view(torch.float32)).any())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("b,d,kk", [(6, 128, 24), (5, 256, 18), (1, 128, 24), (7, 512, 12)])
def test_fused_edge_chain_entry_against_the_oracle_primitives(b, d, kk):
    """sapcu_fn_edge_chain_f32 (the C ABI of fn_edge_chain.hip) on random operands against the oracle's building blocks in the
    reference's own tensor shapes ([b,C,N,k] 1x1 convolutions, T-step neuron loops, softmax over the neighbours, fn:355-389)."""
    from sapcu_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(d + kk + b)
    m, heads, T = 48, 8, 4
    P = b * m
    xyz = torch.from_numpy(rng.normal(0, 0.05, (b, m, 3)).astype(np.float32))
    idx = O.inpatch_knn(xyz.permute(0, 2, 1).contiguous(), kk)                     # [b,m,kk]
    qkv = torch.from_numpy(rng.random((P, 3 * d)).astype(np.float32))
    def lin(n, k, gain):
        return (torch.from_numpy((rng.uniform(-1, 1, (n, k)) * gain / np.sqrt(k)).astype(np.float32)),
                torch.from_numpy(rng.normal(0.6, 0.4, n).astype(np.float32)))
    def lif():
        return torch.from_numpy(np.stack([rng.uniform(0.05, 1.1, d), rng.uniform(0.0, 0.2, d), rng.uniform(0.05, 1.0, d),
                                          rng.normal(0.8, 0.3, d)]).astype(np.float32))
    wd, bd = lin(d, 3, 20.0)
    w1, b1 = lin(d, d, 2.0)
    w2, b2 = lin(d, d, 2.0)
    w3, b3 = lin(d, d, 4.0)
    ld, l1, l2 = lif(), lif(), lif()
    # reference in the model's own layout
    def npar(l):
        return {"decay": torch.clamp(l[0], 0.1, 0.99), "adapt": torch.clamp(l[1], 0.001, 0.1), "rdecay": torch.clamp(l[2], 0.1, 0.95), "theta0": l[3]}
    def conv(x, w, bias):
        return torch.nn.functional.conv2d(x, w[:, :, None, None], bias)
    with torch.no_grad():
        pos = xyz.permute(0, 2, 1)
        pos_diff = (pos.unsqueeze(-1) - O.gather_cols(pos, idx)).contiguous()
        q = qkv[:, :d].view(b, m, d).permute(0, 2, 1)
        kf = qkv[:, d:2 * d].view(b, m, d).permute(0, 2, 1).contiguous()
        v = qkv[:, 2 * d:].view(b, m, d).permute(0, 2, 1).contiguous()
        pe = O.neuron_selfloop(conv(pos_diff, wd, bd), npar(ld), T)
        pe = O.neuron_selfloop(conv(pe, w1, b1), npar(l1), T)
        a = q.unsqueeze(-1) - O.gather_cols(kf, idx) + pe
        a = O.neuron_selfloop(conv(a, w2, b2), npar(l2), T)
        a = torch.softmax(conv(a, w3, b3) / np.sqrt(d // heads), dim=-1)
        want = torch.einsum("bcnk,bcnk->bcn", a, O.gather_cols(v, idx) + pe).permute(0, 2, 1).reshape(P, d)
    need = lib.sapcu_fn_edge_chain_workspace_bytes(P, d, kk)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=U.dev())
    res = torch.full((P, d), float("nan"), device=U.dev())
    args = [_dev(x) for x in (xyz.reshape(P, 3), idx.reshape(-1).to(torch.int32), qkv, wd, bd, ld, w1, b1, l1, w2, b2, l2, w3, b3)]
    _lib.check(lib.sapcu_fn_edge_chain_f32(_lib.ptr(args[0]), _lib.ptr(args[1]), P, m, d, kk, *[_lib.ptr(t) for t in args[2:]],
                                           heads, T, _lib.ptr(res), _lib.ptr(ws), need, _lib.current_stream()))
    torch.cuda.synchronize()
    err = (res.cpu() - want).abs().max().item()
    print("edge chain d=%d kk=%d: max |device - oracle primitives| %.3g (|res| up to %.3g)" % (d, kk, err, want.abs().max()))
    assert err <= 2e-5 * max(1.0, float(want.abs().max()))
    assert lib.sapcu_fn_edge_chain_workspace_bytes(P, 512, 24) < 0                  # shapes the fused kernel does not take
    assert lib.sapcu_fn_edge_chain_f32(*([None] * 2), P, m, 512, 24, *([None] * 12), heads, T, None, None, 0, None) < 0


def test_fused_edge_chain_equals_the_unfused_chain_bit_for_bit(weights, monkeypatch):
    """fn_edge_chain.hip (blocks 1 and 2: pe1 -> fc_delta2 -> attn_in -> fc_gamma -> fc_gamma2 -> softmax-aggregate in one kernel,
    activations in LDS) against the five-kernel chain (a second handle created under SAPCU_CHAIN=0): identical block outputs and normals,
    bit for bit — full groups, a ragged last group (points not a multiple of 5 / 7), one patch, M = 100 (the reference's
    default patch size) and M = 20 (block 1's kk = 20 is not a shape the fused kernel takes: that block stays unfused).
    All three blocks (d = 128 / 256 / 512) run fused.  A third handle (SAPCU_CHAIN=wide) runs the fused kernel in the form it takes
    for q|k|v tensors of 4 GiB and more — 64-bit gather addresses instead of scalar base + 32-bit byte offsets — which no test
    shape reaches by size: same bits again."""
    fn, _, _, _ = U.build_gpu_models_under(weights, monkeypatch, {})
    fn_unfused, _, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_CHAIN": "0"})
    fn_wide, _, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_CHAIN": "wide"})
    assert fn.fused_blocks(48) == 0b111 and fn.fused_blocks(100) == 0b111 and fn_unfused.fused_blocks(48) == 0
    assert fn_wide.fused_blocks(48) == 0b111
    assert fn.fused_blocks(20) == 0b110                      # block 1's kk = min(24, 20) is not a shape the fused kernel takes
    for nq, mpts in ((64, 48), (37, 48), (1, 48), (9, 100), (11, 20)):
        patch = U.sphere_patches(nq, mpts, skip=1200).to(U.dev())
        outs = []
        for model in (fn, fn_unfused, fn_wide):
            taps = {k: torch.full((nq, mpts, 64), float("nan"), device=U.dev()) for k in ("block1", "block2", "block3")}
            n = model(patch, taps=taps)
            torch.cuda.synchronize()
            outs.append((n, taps))
        for k in ("block1", "block2", "block3"):
            assert not bool(torch.isnan(outs[0][1][k]).any()), (nq, mpts, k)
            assert torch.equal(outs[0][1][k], outs[1][1][k]) and torch.equal(outs[0][1][k], outs[2][1][k]), (nq, mpts, k)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0]), (nq, mpts)
    assert fn.gemm_mode() == (True, 0)


def test_fn_max_over_points_fused_into_the_gemm_is_bit_identical(weights, monkeypatch):
    """fn's conv_final + LIF + max over the patch's points (fn/snn_coder.py:465-472): (default) the GEMM's epilogue runs the neuron
    and takes the max by integer atomicMax on order-preserving keys — the [P, emb] activation is never written; SAPCU_FN_MAXFUSE=0:
    GEMM + rowgroup_max.  Identical pooled features and normals for full patches, M = 100, M = 5 (groups that do not align with the
    4-row register groups) and a single patch."""
    variants = [U.build_gpu_models_under(weights, monkeypatch, env)[0] for env in ({}, {"SAPCU_FN_MAXFUSE": "0"})]
    emb = variants[0].emb_dims
    for nq, mpts in ((40, 48), (7, 100), (70, 5), (1, 48)):
        patch = U.sphere_patches(nq, mpts, skip=900).to(U.dev())
        outs = []
        for fn in variants:
            taps = {"pooled": torch.full((nq, emb), float("nan"), device=U.dev())}
            n = fn(patch, taps=taps)
            torch.cuda.synchronize()
            outs.append((n, taps["pooled"]))
        assert not bool(torch.isnan(outs[0][1]).any()) and not bool(torch.isnan(outs[0][0]).any())
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]), (nq, mpts)


def test_fd_max_over_points_fused_into_the_gemm_is_bit_identical(weights, monkeypatch):
    """fd's multi_scale_conv in its three forms: (default) the neuron kernels write the spikes as split rows, the big-tile GEMM
    streams them by LDS-DMA and takes the max over the patch's points in its epilogue (integer atomicMax on order-preserving keys:
    the [T*P, 768] aggregate is never written); SAPCU_FD_SPLIT=0: f32 spikes, the f32-A GEMM with the same epilogue;
    SAPCU_FD_MAXFUSE=0: GEMM + rowgroup_max.  Identical pooled features, encodings and distances, for full patches, M = 100,
    M = 5 (groups that do not align with the 4-row register groups) and batches too small for the big-tile kernel."""
    variants = [U.build_gpu_models_under(weights, monkeypatch, dict(env, SAPCU_FD_FUSED="0"))[1]
                for env in ({}, {"SAPCU_FD_SPLIT": "0"}, {"SAPCU_FD_MAXFUSE": "0"})]
    for nq, mpts in ((40, 48), (7, 100), (70, 5), (5, 5), (1, 48)):
        patch = U.sphere_patches(nq, mpts, skip=1500).to(U.dev())
        kk = min(32, mpts)
        knn = torch.empty((3, nq, mpts, kk), dtype=torch.int32, device=U.dev())
        outs = []
        for fd in variants:
            taps = {"pooled": torch.full((4, nq, 768), float("nan"), device=U.dev()), "enc": torch.full((nq, 768), float("nan"), device=U.dev())}
            if not outs:
                taps["knn"] = knn
                d = fd(patch, taps=taps)
            else:
                d = fd(patch, taps=taps, knn_force=knn)          # same neighbour tables for all runs
            torch.cuda.synchronize()
            outs.append((d, taps["pooled"], taps["enc"]))
        assert not bool(torch.isnan(outs[0][1]).any())
        for o in outs[1:]:
            assert torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2]) and torch.equal(outs[0][0], o[0]), (nq, mpts)
    assert all(fd.gate_violations() == 0 for fd in variants)


def _fd_taps(nq, mpts, T, kk, emb=768):
    z = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float32, device=U.dev())
    return {"fused0": z(nq, mpts, 64), "spikes": z(T, nq, mpts, 960), "knn": torch.full((3, nq, mpts, kk), -1, dtype=torch.int32, device=U.dev()),
            "pooled": z(T, nq, emb), "enc": z(nq, emb)}


@pytest.mark.parametrize("T", [4, 6])
def test_fused_fd_encoder_equals_the_per_stage_path_bit_for_bit(weights, monkeypatch, T):
    """csrc/fd_encoder.hip (the whole encoder — blocks 0-3 and multi_scale_conv over all steps — in one LDS-resident kernel per patch;
    the [T, points, 960] spikes never reach HBM) against the per-stage kernels (a second handle created under SAPCU_FD_FUSED=0):
    scale-fusion output, the three feature-space neighbour tables, every spike of every step, pooled features, encoding and
    distance must agree BIT FOR BIT — full batches, one patch, small and odd patch sizes, and T = 6 (two groups of stacked steps).
    Checked in dependency order so that a failure names the first stage that differs."""
    over = {"time_steps_enc": T}
    _, fd, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_FD_FUSED": "1"}, None, over)
    _, fd_stage, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_FD_FUSED": "0"}, None, over)
    assert fd.fused_blocks(48) == 1 and fd.fused_blocks(5) == 1 and fd.fused_blocks(100) == 2 and fd_stage.fused_blocks(48) == 0 and fd_stage.fused_blocks(100) == 0
    for nq, mpts in ((64, 48), (37, 48), (1, 48), (9, 20), (70, 5), (5, 12), (3, 33)):
        patch = U.sphere_patches(nq, mpts, skip=1700).to(U.dev())
        kk = min(32, mpts)
        ta, tb = _fd_taps(nq, mpts, T, kk), _fd_taps(nq, mpts, T, kk)
        da, db = fd(patch, taps=ta), fd_stage(patch, taps=tb)
        torch.cuda.synchronize()
        tag = "nq=%d m=%d T=%d" % (nq, mpts, T)
        assert torch.equal(ta["fused0"], tb["fused0"]), tag + ": scale_fusion output"
        coff = (0, 64, 192, 448, 960)
        for l in range(4):
            if l:
                assert torch.equal(ta["knn"][l - 1], tb["knn"][l - 1]), tag + ": neighbour table of block %d" % l
            for t in range(T):
                a, b = ta["spikes"][t, :, :, coff[l]:coff[l + 1]], tb["spikes"][t, :, :, coff[l]:coff[l + 1]]
                assert not bool(torch.isnan(a).any()), tag + ": block %d step %d spikes not written" % (l, t)
                assert torch.equal(a, b), tag + ": block %d step %d spikes (max diff %g)" % (l, t, float((a - b).abs().max()))
        assert torch.equal(ta["pooled"], tb["pooled"]), tag + ": pooled (max diff %g)" % float((ta["pooled"] - tb["pooled"]).abs().max())
        assert torch.equal(ta["enc"], tb["enc"]) and torch.equal(da, db), tag
        # without taps (the production call) and with forced neighbour tables (the parity protocol)
        assert torch.equal(fd(patch), da), tag + ": tap-free call"
        assert torch.equal(fd(patch, knn_force=tb["knn"]), db), tag + ": forced tables"
    assert fd.gate_violations() == 0 and fd_stage.gate_violations() == 0 and fd.gemm_mode() == (True, 0)


@pytest.mark.parametrize("emb", [800, 896, 1024])
def test_fused_fd_encoder_with_a_ragged_last_column_sweep(monkeypatch, emb):
    """emb_dims above 768 whose second sweep of 24 column blocks ends in a partly owned wave (emb / 32 - 24
    not a multiple of 3: 800 -> block 24, 896 -> block 27, 1024 -> blocks 30, 31) — those pooled columns were never stored.
    Fused encoder against the per-stage kernels, bit for bit, with the outputs pre-filled with NaN."""
    import sapcu_amd
    from conftest import FD_KW
    from sapcu_amd import testing as T
    kw = dict(FD_KW, emb_dims=emb)
    tmpl = sapcu_amd.EnhancedSNNDistanceEstimation(**kw).state_dict()
    bn = {k: v for k, v in dict(golden("bn_calib_fd.npz")).items() if tuple(v.shape) == tuple(tmpl[k].shape)}
    sdd = T.conditioned_state_dict(tmpl, 0, bn_stats=bn)
    fds = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SAPCU_FD_FUSED", flag)
        fd = sapcu_amd.EnhancedSNNDistanceEstimation(**kw)
        fd.load_state_dict(sdd, strict=True)
        fd = fd.to(U.dev())
        fd._engine()
        monkeypatch.delenv("SAPCU_FD_FUSED")
        fds.append(fd)
    assert fds[0].fused_blocks(48) == 1 and fds[1].fused_blocks(48) == 0
    for nq, mpts in ((37, 48), (5, 12)):
        patch = U.sphere_patches(nq, mpts, skip=1900).to(U.dev())
        kk = min(32, mpts)
        ta, tb = _fd_taps(nq, mpts, 4, kk, emb), _fd_taps(nq, mpts, 4, kk, emb)
        del ta["spikes"], tb["spikes"], ta["fused0"], tb["fused0"]
        da = fds[0](patch, taps=ta)
        db = fds[1](patch, taps=tb, knn_force=ta["knn"])
        torch.cuda.synchronize()
        assert not bool(torch.isnan(ta["pooled"]).any()), "emb=%d: %d pooled columns never written" % (emb, int(torch.isnan(ta["pooled"][0, 0]).sum()))
        assert torch.equal(ta["pooled"], tb["pooled"]) and torch.equal(ta["enc"], tb["enc"]) and torch.equal(da, db), (emb, nq, mpts)
    assert fds[0].gate_violations() == 0 and fds[1].gate_violations() == 0


def test_fd_forward_on_patches_of_128_points(models):
    """the largest patch the entry points accept.  fd's EdgeConv + neuron kernel stages the patch's [m][128]
    tile AND its byte-sized neighbour table in LDS: 69 632 B at m = 128, k = 32 — above the default 64 KiB dynamic-LDS limit,
    which the launcher has to raise.  Distances against the oracle on the device's own neighbour tables (1e-4); fn on the same
    patches against the oracle."""
    fn, fd, sdn, sdd = models
    patch = U.sphere_patches(3, 128, skip=2300)
    d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fd, sdd, patch)
    assert float((d_gpu - d_forced).abs().max()) <= TOL and fd.gate_violations() == 0
    mode = fn.knn_cache_mode
    fn.knn_cache_mode = "fresh"
    try:
        with torch.no_grad():
            n_ref = O.fn_forward(sdn, patch, U.FN_HP)
        assert float((fn(patch.to(U.dev())).cpu() - n_ref).abs().max()) <= TOL
    finally:
        fn.knn_cache_mode = mode


@pytest.mark.parametrize("T", [4, 7, 5, 10])
def test_fd_x0_path_equals_the_spike_slab_path_bit_for_bit(weights, monkeypatch, T):
    """Round 4, patches of more than 48 points (the reference's default is 100, generation.py:68; config/fd.yaml runs T = 7): the
    per-stage kernels write the pre-activations x0 [points, 960] and the step-0 spikes only, and fd_msc_kernel (csrc/fd_encoder.hip)
    regenerates the spikes of all T steps on the CU for multi_scale_conv + the max over the points — instead of T spike slabs
    through HBM and the big-tile GEMM (handles created under SAPCU_FD_X0=0) or the whole old per-stage path (SAPCU_FD_FUSED=0).
    Every tap — scale fusion, neighbour tables, x0, every spike of every step, pooled, encoding — and the distances must agree
    BIT FOR BIT: 100, 128, 64, 49 and 77 points, one patch, T = 4 (two kernels: production / general) and the reference's T = 7
    (fd_msc8_kernel: eight step slots per point, one group), T = 5 and T = 10 (two groups of eight slots), with and without taps,
    free-running and under forced tables."""
    over = {"time_steps_enc": T}
    _, fd, _, sdd = U.build_gpu_models_under(weights, monkeypatch, {}, None, over)
    _, fd_slab, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_FD_X0": "0"}, None, over)
    _, fd_stage, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_FD_FUSED": "0"}, None, over)
    assert fd.fused_blocks(100) == 2 and fd.fused_blocks(49) == 2 and fd.fused_blocks(48) == 1
    assert fd_slab.fused_blocks(100) == 0 and fd_slab.fused_blocks(48) == 1 and fd_stage.fused_blocks(100) == 0
    for nq, mpts in ((9, 100), (3, 128), (5, 64), (1, 49), (4, 77)):
        patch = U.sphere_patches(nq, mpts, skip=2500).to(U.dev())
        kk = min(32, mpts)
        ta, tb = _fd_taps(nq, mpts, T, kk), _fd_taps(nq, mpts, T, kk)
        ta["x0"], tb["x0"] = torch.full((nq, mpts, 960), float("nan"), device=U.dev()), torch.full((nq, mpts, 960), float("nan"), device=U.dev())
        da, db = fd(patch, taps=ta), fd_slab(patch, taps=tb)
        torch.cuda.synchronize()
        tag = "nq=%d m=%d T=%d" % (nq, mpts, T)
        assert torch.equal(ta["fused0"], tb["fused0"]) and torch.equal(ta["knn"], tb["knn"]), tag
        assert not bool(torch.isnan(ta["x0"]).any()) and torch.equal(ta["x0"], tb["x0"]), tag + ": x0"
        for t in range(T):
            assert not bool(torch.isnan(ta["spikes"][t]).any()), tag + ": step %d spikes not written" % t
            assert torch.equal(ta["spikes"][t], tb["spikes"][t]), tag + ": step %d spikes (max diff %g)" % (t, float((ta["spikes"][t] - tb["spikes"][t]).abs().max()))
        assert torch.equal(ta["pooled"], tb["pooled"]), tag + ": pooled (max diff %g)" % float((ta["pooled"] - tb["pooled"]).abs().max())
        assert torch.equal(ta["enc"], tb["enc"]) and torch.equal(da, db), tag
        assert torch.equal(fd(patch), da), tag + ": tap-free call (production kernel at T = 4)"
        assert torch.equal(fd_slab(patch), da), tag + ": tap-free call, slab path"
        assert torch.equal(fd_stage(patch, knn_force=ta["knn"]), da), tag + ": old per-stage path"
        assert torch.equal(fd(patch, knn_force=ta["knn"]), da), tag + ": forced tables"
    # against the oracle on the device's tables (1e-4), at the reference's default patch size
    patch = U.sphere_patches(6, 100, skip=2600)
    d_gpu, d_forced, _, _, _ = U.fd_forward_forced(fd, sdd, patch, dict(U.FD_HP, time_steps_enc=T))
    assert float((d_gpu - d_forced).abs().max()) <= TOL
    assert fd.gate_violations() == 0 and fd_slab.gate_violations() == 0 and fd_stage.gate_violations() == 0


def test_exact_operation_order_build():
    """csrc/libsapcu_hip_exact.so (make exact: -DSAPCU_LIF_EXACT_ORDER, every neuron update in the reference's operation order —
    the variant INTEGRATION.md advertises) in a child process (the library is chosen at import: SAPCU_LIB_PATH): neuron unit
    against the reference vectors (1e-6), fn / fd against the oracle (1e-4), fused fd encoder == per-stage kernels bit for bit."""
    import subprocess
    import sys
    from sapcu_amd import _lib
    if not os.path.exists(_lib.EXACT_LIB_PATH):
        pytest.fail("%s is not built (make -C csrc exact; __graft_entry__.build() does)" % _lib.EXACT_LIB_PATH)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAPCU_LIB_PATH=_lib.EXACT_LIB_PATH)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "exact_order_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "EXACT_ORDER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    print(r.stdout.strip().splitlines()[-1])


def test_models_on_the_big_tile_kernel_equal_the_ring_kernel_bit_for_bit(weights, monkeypatch):
    """Whole fn and fd forwards with the split-row GEMMs on the big-tile kernel (default) against the ring kernel only
    (handles created under SAPCU_BT=0): identical normals and distances, bit for bit — including a batch whose last row tile is
    ragged (37 patches) and one below the big-tile threshold.  (fd on its per-stage kernels: the fused encoder has no split-row
    GEMM launches.)"""
    fn, fd, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_FD_FUSED": "0"})
    fn_ring, fd_ring, _, _ = U.build_gpu_models_under(weights, monkeypatch, {"SAPCU_BT": "0", "SAPCU_FD_FUSED": "0"})
    for nq in (64, 37, 1):
        patch = U.sphere_patches(nq, 48, skip=500).to(U.dev())
        n1, d1 = fn(patch), fd(patch)
        n0, d0 = fn_ring(patch), fd_ring(patch)
        assert torch.equal(n0, n1) and torch.equal(d0, d1), nq


def test_concurrent_forwards_of_one_handle_on_two_streams(weights):
    """include/sapcu.h: a handle is immutable after create, so forwards of ONE handle may run concurrently from different host
    threads on different streams, each with its own workspace (launch attributes are per device and set thread-safely, no
    forward reads the environment).  Two threads x two streams x 6 forwards each of fn and fd through the C ABI against the
    single-threaded results, bit for bit."""
    import ctypes
    import threading
    from sapcu_amd import _lib
    lib = _lib.load()
    fn, fd, _, _ = U.build_gpu_models(weights)
    fn.knn_cache_mode = "fresh"
    dev = U.dev()
    patches = [U.sphere_patches(48, 48, skip=2100 + 100 * i).to(dev) for i in range(2)]
    want = [(fn(p).clone(), fd(p).clone()) for p in patches]
    torch.cuda.synchronize()
    hn, hd = fn._engine(), fd._engine()
    errors, results = [], [None, None]

    def worker(i):
        try:
            st = torch.cuda.Stream(device=dev)
            p = patches[i]
            b, m = p.shape[0], p.shape[1]
            wsn = torch.empty(int(lib.sapcu_workspace_bytes(hn, b, m)), dtype=torch.uint8, device=dev)
            wsd = torch.empty(int(lib.sapcu_workspace_bytes(hd, b, m)), dtype=torch.uint8, device=dev)
            outs = []
            with torch.cuda.stream(st):             # the NaN fills of the outputs run on this thread's stream too
                for _ in range(6):
                    n = torch.full((b, 3), float("nan"), device=dev)
                    d = torch.full((b,), float("nan"), device=dev)
                    sp = ctypes.c_void_p(st.cuda_stream)
                    _lib.check(lib.sapcu_fn_forward(hn, _lib.ptr(p), b, m, None, None, _lib.ptr(n), _lib.ptr(wsn), wsn.numel(), None, sp))
                    _lib.check(lib.sapcu_fd_forward(hd, _lib.ptr(p), b, m, None, _lib.ptr(d), _lib.ptr(wsd), wsd.numel(), None, sp))
                    outs.append((n, d))
            st.synchronize()
            results[i] = outs
        except Exception as e:          # noqa: BLE001 - reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for n, d in results[i]:
            assert torch.equal(n, want[i][0]) and torch.equal(d, want[i][1]), i
    assert fd.gate_violations() == 0 and fn.gemm_mode() == (True, 0) and fd.gemm_mode() == (True, 0)


@pytest.mark.parametrize("use_amp", [False, True])
def test_graph_captured_training_step_equals_the_eager_step(use_amp):
    """fn_trainer.GraphedTrainStep (one optimisation step replayed as a HIP graph) against Trainer.train_step from the same
    initial state, dropout off: same first loss, finite losses and gradient norms on every replay (the scatter-add targets
    must be re-zeroed inside the graph), and parameters that stay close to the eager run's after four AdamW steps.
    use_amp=True (no scaler — a case the constructor accepts): both run the bf16 GEMMs, i.e. the captured step's first loss
    equals the eager bf16 step's and differs from the f32 step's (the graph used to capture the f32 GEMMs)."""
    import copy
    import sapcu_amd
    from sapcu_amd import fn_trainer, testing as T
    g = golden("fn_trainer.npz")
    kw = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8, use_snn_decoder=False, decoder_dropout=0.1)
    sd = T.training_state_dict(sapcu_amd.ImprovedSNNNormalEstimation(**kw).state_dict(), int(g["seed"]))
    data = {"input": torch.from_numpy(g["points"]), "normal": torch.from_numpy(g["gt"])}
    results = []
    for graphed in (False, True):
        model = sapcu_amd.ImprovedSNNNormalEstimation(**kw)
        model.load_state_dict(copy.deepcopy(sd), strict=True)
        model.attn_dropout = model.decoder_dropout = 0.0
        model.cuda()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True)
        tr = fn_trainer.Trainer(model, opt, device=torch.device("cuda"), grad_clip=0.15, grad_clip_type="norm", use_amp=use_amp)
        step = fn_trainer.GraphedTrainStep(tr, data, warmup=0) if graphed else tr.train_step     # (capturing runs no kernels)
        probe = torch.from_numpy(g["points"]).reshape(-1, g["points"].shape[-2], 3)[:8].cuda()
        model.eval()
        before = model(probe).clone()                    # packs the inference blob from the initial weights
        model.train()
        losses = [step(data)[0] for _ in range(4)]
        assert all(l is not None and np.isfinite(l) for l in losses), losses
        # eval after the steps must see the updated weights (a graph replay moves no version counter: the packed blob has to
        # be invalidated by the replay itself) = what a fresh model loaded from the current state_dict computes
        model.eval()
        after = model(probe).clone()
        twin = sapcu_amd.ImprovedSNNNormalEstimation(**kw)
        twin.load_state_dict(copy.deepcopy(model.state_dict()), strict=True)
        twin.attn_dropout = twin.decoder_dropout = 0.0
        twin.cuda().eval()
        assert torch.equal(after, twin(probe)) and not torch.equal(after, before), graphed
        model.train()
        results.append((losses, {n: p.detach().cpu().clone() for n, p in model.named_parameters()}))
    (le, pe), (lg, pg) = results
    # the first steps agree; later ones drift apart the way two eager runs do (float atomics in the scatter-adds reorder sums,
    # a hard spike flips, and the loss of a 16-patch batch moves by 0.1)
    assert abs(le[0] - lg[0]) <= 1e-5 and max(abs(a - b) for a, b in zip(le, lg)) <= 0.3, (le, lg)
    if use_amp:                                          # ... and it is the bf16 arithmetic that was captured
        model = sapcu_amd.ImprovedSNNNormalEstimation(**kw)
        model.load_state_dict(copy.deepcopy(sd), strict=True)
        model.attn_dropout = model.decoder_dropout = 0.0
        model.cuda().train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True)
        l32 = fn_trainer.Trainer(model, opt, device=torch.device("cuda"), grad_clip=0.15, grad_clip_type="norm").train_step(data)[0]
        assert abs(l32 - lg[0]) > 10 * abs(le[0] - lg[0]) and abs(l32 - lg[0]) > 1e-6, (l32, le[0], lg[0])
    worst = max(float((pe[n] - pg[n]).abs().max()) for n in pe)
    assert worst <= 1.2e-3, worst           # four AdamW steps of lr 1e-4: each moves a weight by at most ~lr, in either direction


# ---------------------------------------------------------------------------------------------
# N > 1 path on a one-GPU box: two ranks, both on cuda:0, gloo — the real Generator3D6.refine +
# upsample_sharded + gather_refined, and bench.py's own rank spawning.  (RCCL itself needs one GPU per rank: the driver's
# 8-GPU run.)  Child processes are started with subprocess (never an exec from this GPU-initialised process).
# ---------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.gpu
def test_sharded_upsample_two_ranks_on_one_gpu():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tests", "dist_rehearsal.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "REHEARSAL_OK ranks=2" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_rccl_world1_sharded_upsample_on_device_tensors():
    """RCCL contact on ONE GPU: a child rank with init_process_group("nccl", world_size=1, device_id=cuda:0) runs
    upsample_sharded + gather_refined on device tensors (the nccl branch of sapcu_amd/dist.py: f64 slabs stay on the device,
    all_gather_into_tensor over RCCL) and holds the gathered cloud to the single-process 'fresh' refine, bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_rehearsal.py"), "nccl"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "REHEARSAL_OK ranks=1 backend=nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_bench_gpus_1_takes_the_rccl_branch_when_asked():
    """SAPCU_BENCH_NCCL1=1 python bench.py --gpus 1: the N = 1 bench with the process group of the N > 1 runs (nccl, device_id)
    — barrier, the all-gather of the refined cloud and the max-over-ranks all-reduce all go through RCCL on one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAPCU_BENCH_REHEARSE")}
    env.update(SAPCU_BENCH_NCCL1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline", "--no-strong-leg"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["collective_backend"] == "nccl" and line["value"] > 0


@pytest.mark.gpu
def test_bench_gpus_2_as_a_plain_command_rehearsal():
    """`python bench.py --gpus 2` (no launcher around it) spawns its two ranks itself; on this one-GPU box that is a
    rehearsal (both ranks on cuda:0, gloo) — exit code 0 and ONE JSON line with n_gpus 2 and the strong-scaling leg."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SAPCU_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["scaling"] == "weak" and line["value"] > 0
    assert line["strong_scaling"]["seeds"] == 385582 and line["strong_scaling"]["n_gpus"] == 2


# ---------------------------------------------------------------------------------------------
# Hardware rounding facts the kernels rely on (round 3).  The parity tests above would catch a violation indirectly; these say
# which fact broke.
# ---------------------------------------------------------------------------------------------
def _build_and_run_micro(name, tmp_path):
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "micro", name + ".hip")
    exe = str(tmp_path / name)
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", src, "-o", exe], check=True, capture_output=True, timeout=600)
    return subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout


def test_f32_mfma_is_a_k_ascending_fma_chain(tmp_path):
    """The in-patch kNN scores are DEFINED as one channel-ascending f32 FMA chain per pair (oracle/snn_path.py); patch_knn_kernel
    and the fused fd encoder compute them with v_mfma_f32_16x16x4_f32.  That is exact only because the instruction adds its
    k products in ascending k with one IEEE rounding each (profiles/micro/mfma_f32_exact.hip)."""
    out = _build_and_run_micro("mfma_f32_exact", tmp_path)
    assert "32x32x2_f32, K = 256, 1024 outputs: differ from the k-ascending FMA chain 0," in out, out
    assert "16x16x4_f32, K = 256, 256 outputs: differ from the k-ascending FMA chain 0" in out, out


def test_one_16x16x32_f16_mfma_equals_two_chained_32x32x16(tmp_path):
    """fn_edge_chain.hip issues v_mfma_f32_16x16x32_f16, the unfused chain's ring kernel two chained 32x32x16 per product: their
    bit-identity (test_fused_edge_chain_equals_the_unfused_chain_bit_for_bit) rests on the two shapes rounding alike
    (profiles/micro/mfma_f16_shapes_bits.hip)."""
    out = _build_and_run_micro("mfma_f16_shapes_bits", tmp_path)
    assert "two chained 32x32x16 vs one 16x16x32 differ in 0 " in out, out
