"""Deterministic synthetic weights and inputs (no trained checkpoint ships with the reference).

Shared by the golden-fixture generator, the parity tests and ``bench.py`` so that every box
rebuilds bit-identical weights from ``(tensor name, shape, seed)`` alone — numpy's PCG64
stream is platform independent, torch's initialisers are not used.

Why "conditioned": under torch's default initialisation both reference networks are
input-insensitive (every pre-activation sits far below threshold 1.0), so
golden vectors from default weights would make any parity test pass trivially.  The recipe
below moves BatchNorm statistics so that pre-activations straddle the spike threshold.
"""
import zlib

import numpy as np
import torch

_NEURON_INIT = {
    "membrane_decay": 0.9, "threshold_adapt": 0.01, "refractory_decay": 0.5,
    "threshold_base": 1.0, "delta_T": 1.0, "theta_rh": 0.8,
}


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


BN_GAIN = 0.5                      # scale on BatchNorm gamma (keeps per-layer noise gain < 1)
FN_LOGIT_BIAS = (2.0, -1.0, 0.6)   # decoder.fc_out.bias: keeps LayerNorm(3) well conditioned


def conditioned_state_dict(template, seed=0, w_gain=1.0, bn_stats=None, bn_gain=BN_GAIN):
    """template: mapping name -> shape (or tensor).  Returns name -> torch tensor (cpu).

    ``bn_stats``: optional mapping ``<bn>.running_mean`` / ``<bn>.running_var`` -> array.  The
    golden fixtures carry statistics calibrated on 64 sphere patches (tests/golden/bn_calib_*.npz)
    so every BatchNorm output is ~N(beta, (gamma*bn_gain)^2): pre-activations straddle the spike
    threshold (the nets are input-sensitive) while fp32 rounding noise stays ~1e-5 at the outputs."""
    shapes = {k: tuple(v.shape) if hasattr(v, "shape") else tuple(v) for k, v in template.items()}
    out = {}
    for name, shape in shapes.items():
        r = _rng(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 1)[0] if "." in name else ""
        is_bn = (parent + ".running_mean") in shapes
        if leaf == "num_batches_tracked":
            out[name] = torch.zeros(shape, dtype=torch.int64)
            continue
        if leaf in _NEURON_INIT:
            a = _NEURON_INIT[leaf] * r.uniform(0.9, 1.1, shape)
        elif bn_stats is not None and name in bn_stats:
            a = np.asarray(bn_stats[name])
        elif leaf == "running_mean":
            a = r.normal(0.0, 0.05, shape)
        elif leaf == "running_var":
            a = r.uniform(0.005, 0.025, shape)
        elif is_bn and leaf == "weight":
            a = r.uniform(0.5, 1.5, shape) * bn_gain
        elif name == "decoder.fc_out.bias" and shape == (3,):
            a = np.asarray(FN_LOGIT_BIAS)
        elif is_bn and leaf == "bias":
            a = r.normal(0.6, 0.5, shape)
        elif parent.endswith("norm_out") or parent.endswith("attention.norm"):
            a = r.uniform(0.5, 1.5, shape) if leaf == "weight" else r.normal(0.0, 0.1, shape)
        elif parent.endswith("temporal_integration"):
            a = r.normal(1.0, 0.3, shape)
        elif len(shape) >= 2:                                  # conv / linear weight
            fan_in = int(np.prod(shape[1:]))
            b = w_gain / np.sqrt(fan_in)
            a = r.uniform(-b, b, shape)
        else:                                                   # conv / linear bias
            wshape = shapes.get(parent + ".weight")
            fan_in = int(np.prod(wshape[1:])) if wshape else shape[0]
            b = 1.0 / np.sqrt(fan_in)
            a = r.uniform(-b, b, shape)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return out


def training_state_dict(template, seed=0):
    """Parameters for TRAINING-mode cases (row f-4): conditioned_state_dict with the BatchNorm affine and the neuron
    parameters redrawn so that the HARD spikes of train() mode are not degenerate under batch statistics (BatchNorm output
    ~N(beta, gamma^2); thresholds ~N(0, 0.4^2): with the closed refractory gate a T-step self-loop neuron only keeps firing when
    its threshold is negative, so all-positive thresholds leave the net's output independent of its input)."""
    out = conditioned_state_dict(template, seed, bn_gain=1.0)
    for name, v in out.items():
        r = _rng(seed + 1000, name)
        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 1)[0] if "." in name else ""
        is_bn = (parent + ".running_mean") in out
        shape = tuple(v.shape)
        if is_bn and leaf == "weight":
            a = r.uniform(0.6, 1.4, shape)
        elif is_bn and leaf == "bias":
            a = r.normal(0.5, 0.4, shape)
        elif leaf == "membrane_decay":
            a = r.uniform(0.3, 0.95, shape)
        elif leaf == "threshold_adapt":
            a = r.uniform(0.005, 0.08, shape)
        elif leaf == "refractory_decay":
            a = r.uniform(0.2, 0.9, shape)
        elif leaf == "threshold_base":
            a = r.normal(0.0, 0.4, shape)
        else:
            continue
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return out


def sphere_cloud(n=5000, seed=0):
    """unit-normal directions * 0.5, rounded to 6 decimals, float64."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.round(v * 0.5, 6).astype(np.float64)


def grid_queries(count=4096, seed=0, spacing=0.004, band=(0.011, 0.015), radius=0.5):
    """Cell centres of the seed grid whose distance to the sphere lies in ``band`` (§8d).

    Candidates are drawn 200 000 cells at a time from one PCG64 stream, so the first rows are the same for every
    ``count`` (the 4096 of the benchmark are a prefix of the 32 768 an 8-rank run needs)."""
    rng = np.random.default_rng(seed)
    side = int(round(1.0 / spacing))
    got, total = [], 0
    for _ in range(4096):
        cells = rng.integers(0, side, (200000, 3))
        q = cells * spacing + spacing / 2 - 0.5
        off = np.abs(np.linalg.norm(q, axis=1) - radius)
        q = q[(off > band[0]) & (off < band[1])]
        got.append(q)
        total += q.shape[0]
        if total >= count:
            break
    if total < count:
        raise ValueError("not enough grid cells in the band: %d < %d" % (total, count))
    return np.ascontiguousarray(np.concatenate(got, axis=0)[:count], dtype=np.float64)


def analytic_cloud(kind, n=2048, seed=0):
    """Small suite of analytic shapes in [-0.5,0.5]^3 (stand-ins for the absent ShapeNet data)."""
    rng = np.random.default_rng(seed)
    if kind == "sphere":
        return sphere_cloud(n, seed)
    if kind == "torus":
        u, v = rng.uniform(0, 2 * np.pi, (2, n))
        R, r = 0.33, 0.14
        p = np.stack([(R + r * np.cos(v)) * np.cos(u), (R + r * np.cos(v)) * np.sin(u), r * np.sin(v)], 1)
    elif kind == "cube":
        p = rng.uniform(-0.45, 0.45, (n, 3))
        ax = rng.integers(0, 3, n)
        p[np.arange(n), ax] = np.where(rng.random(n) < 0.5, -0.45, 0.45)
    elif kind == "cylinder":
        u = rng.uniform(0, 2 * np.pi, n)
        p = np.stack([0.3 * np.cos(u), 0.3 * np.sin(u), rng.uniform(-0.45, 0.45, n)], 1)
    elif kind == "two_spheres":
        # union of two overlapping spheres (radius 0.3, centres +-0.18 on x): the outer surface only — a concave crease
        v = rng.normal(size=(4 * n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        side = np.where(rng.random(4 * n) < 0.5, -1.0, 1.0)
        p = v * 0.3
        p[:, 0] += 0.18 * side
        other = p.copy()
        other[:, 0] += 0.18 * side                      # coordinates relative to the OTHER sphere's centre (at -0.18*side)
        p = p[np.linalg.norm(other, axis=1) >= 0.3][:n]
        assert p.shape[0] == n
    else:
        raise ValueError(kind)
    return np.round(p, 6).astype(np.float64)


# ---------------------------------------------------------------------------------------------
# BASELINE configs 3 / 4 stand-ins: six shapes, N = 2048, run through the reference's
# Generator3D6.upsample by tests/golden/make_fixtures.py (shape_suite.npz); one 16x case through its generate.py body.
# (name, kind, cloud seed, dense_spacing).  "icosahedron" is the reference tree's one real cloud
# (external/SPU-PMD/evaluation_code/Icosahedron.xyz, 8192 points), subsampled — that cloud is stored in the fixture.
# ---------------------------------------------------------------------------------------------
SHAPE_SUITE = (("sphere", "sphere", 4, 0.032), ("torus", "torus", 1, 0.026), ("cube", "cube", 2, 0.030),
               ("cylinder", "cylinder", 5, 0.026), ("two_spheres", "two_spheres", 6, 0.028),
               ("icosahedron", None, 7, 0.034))
SHAPE_SUITE_N = 2048
SCALE16_CASE = dict(n=256, seed=8, spacing=0.0175, ratio=16, scale=37.5, loc=(3.0, -8.0, 0.25))


def suite_cloud(name, stored=None):
    """Cloud [2048,3] float64 of a SHAPE_SUITE entry (``stored``: the fixture file, needed for "icosahedron")."""
    for nm, kind, seed, _ in SHAPE_SUITE:
        if nm == name:
            if kind is None:
                return np.asarray(stored["%s_cloud" % name], dtype=np.float64)
            return analytic_cloud(kind, SHAPE_SUITE_N, seed)
    raise KeyError(name)


def scale16_cloud():
    c = SCALE16_CASE
    return sphere_cloud(c["n"], c["seed"]) * c["scale"] + np.array(c["loc"])


# ---------------------------------------------------------------------------------------------
# farthest-point-sampling cases (tests/golden/fps.npz holds the reference's indices for each)
# ---------------------------------------------------------------------------------------------
FPS_CASES = ("sphere2048", "torus1000_all", "dup200", "lattice216", "tiny5", "one", "big100k", "big400k")


def fps_case(name):
    """(cloud float64 [N,3], npoint) of a named FPS case; deterministic, so the inputs are not stored."""
    rng = np.random.default_rng([7, FPS_CASES.index(name)])
    if name == "sphere2048":          # denormalised like generate.py:90 leaves it
        return sphere_cloud(2048, 0) * 37.5 + np.array([3.0, -8.0, 0.25]), 256
    if name == "torus1000_all":       # npoint == N
        return analytic_cloud("torus", 1000, 1), 1000
    if name == "dup200":              # 5 copies of 40 points: ties at distance 0 once 40 are taken
        return np.repeat(rng.standard_normal((40, 3)), 5, axis=0), 60
    if name == "lattice216":          # exact ties at non-zero distance
        return np.stack(np.meshgrid(*[np.arange(6.0)] * 3, indexing="ij"), -1).reshape(-1, 3), 100
    if name == "tiny5":
        return rng.standard_normal((5, 3)), 3
    if name == "one":
        return rng.standard_normal((1, 3)), 1
    if name == "big100k":
        return rng.standard_normal((100000, 3)) * np.array([1.0, 0.5, 0.25]), 2048
    if name == "big400k":             # the size of a refined cloud (generation.py: ~385 k points)
        base = sphere_cloud(5000, 0)
        return base[rng.integers(0, 5000, 400000)] + 0.01 * rng.standard_normal((400000, 3)), 512
    raise KeyError(name)
