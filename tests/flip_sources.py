#!/usr/bin/env python3
"""Where do the free-running fd neighbour flips come from?  

fd picks 32 of 48 neighbours in 64/128/256-d soft-spike space from f32 scores; two correct f32 implementations break near-ties
differently, and one flipped neighbour moves the predicted distance far beyond 1e-4.  This script attributes the flips of the
device path to its arithmetic choices by running the SAME measurements under four builds / switches of the library:

    default            split-f16 GEMMs (3 x f16 MFMA), fused-FMA neuron arithmetic
    gemm_f32           SAPCU_GEMM=f32: exact-f32 MFMA GEMMs
    exact_order        csrc/libsapcu_hip_exact.so (-DSAPCU_LIF_EXACT_ORDER): neuron updates in the reference's operation order
    exact_order+f32    both

and printing, next to them, the reference against ITSELF (1 thread vs 8 threads; tests/golden/ref_vs_ref.npz, made by
tests/golden/make_fixtures.py --only-ref-vs-ref):

  * fd on 256 sphere patches: patches (of 256) whose feature-space neighbour sets differ from the oracle's free run, rows flipped,
    max |distance error| under the forced-neighbour protocol;
  * end to end on the eight reference runs (sphere-2048, six shapes, the 16x cloud): fraction of refined points within 2e-4.

usage (GPU box):  python tests/flip_sources.py            # all variants, markdown table on stdout
                  python tests/flip_sources.py --one      # this process's library / environment only, one JSON line
Lives under tests/ because it evaluates the oracle (test infrastructure); the product never imports it.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def measure():
    import torch
    import sapcu_amd
    from sapcu_amd import testing as T, _lib
    import gpu_utils as U
    from conftest import golden

    class W:                                            # the session fixture of conftest, without pytest
        def __call__(self, kind, **over):
            from conftest import FD_KW, FN_KW
            if kind == "fn":
                m = sapcu_amd.ImprovedSNNNormalEstimation(**dict(FN_KW, **over))
                bn = dict(golden("bn_calib_fn.npz"))
            else:
                m = sapcu_amd.EnhancedSNNDistanceEstimation(**dict(FD_KW, **over))
                bn = dict(golden("bn_calib_fd.npz"))
            return T.conditioned_state_dict(m.state_dict(), 0, bn_stats=bn)

    fn, fd, sdn, sdd = U.build_gpu_models(W())
    out = {"lib": os.path.basename(_lib.LIB_PATH), "SAPCU_GEMM": os.environ.get("SAPCU_GEMM", "")}
    # (i) fd, 256 patches, forced-neighbour protocol + free-running flip count (the shapes of test_fd_forward_256_patches...)
    patch = U.sphere_patches(256, 48, skip=200)
    d_gpu, d_forced, d_free, flips, _ = U.fd_forward_forced(fd, sdd, patch)
    flip_patches = (flips[0] | flips[1] | flips[2]).any(-1)
    out["fd256_flip_patches"] = int(flip_patches.sum())
    out["fd256_flip_rows"] = int(sum(int(f.sum()) for f in flips))
    out["fd256_max_err_forced"] = float((d_gpu - d_forced).abs().max())
    out["fd256_within_2e4_free"] = float(((d_gpu - d_free).abs() <= 2e-4).float().mean())
    # (ii) end to end against the reference runs
    dev = U.dev()

    def frac(cloud, seeds, unfiltered, spacing):
        fn.knn_cache_mode = "reference"
        fn._knn_cache.clear()
        gen = sapcu_amd.Generator3D6(fn, fd, dev, k_neighbors=48, dense_spacing=spacing, batch_size=64)
        with torch.no_grad():
            refined, _, _ = gen.refine(torch.as_tensor(cloud, device=dev), torch.as_tensor(seeds, device=dev))
        err = np.abs(refined.cpu().numpy() - unfiltered).max(axis=1)
        return float((err <= 2e-4).mean()), int(err.size)

    runs = {}
    g = golden("e2e_upsample.npz")
    runs["sphere2048"] = frac(T.sphere_cloud(2048, 0), g["seeds"], g["unfiltered"], 0.03)
    g = golden("shape_suite.npz")
    for shape in ("sphere", "torus", "cube", "cylinder", "two_spheres", "icosahedron"):
        runs["suite/" + shape] = frac(T.suite_cloud(shape, g), g[shape + "_seeds"], g[shape + "_unfiltered"], float(g[shape + "_spacing"]))
    g = golden("scale16.npz")
    runs["16x"] = frac(g["norm_cloud"], g["seeds"], g["unfiltered"], T.SCALE16_CASE["spacing"])
    out["e2e"] = {k: round(v[0], 4) for k, v in runs.items()}
    pts = sum(v[1] for v in runs.values())
    out["e2e_within_2e4_pooled"] = round(sum(v[0] * v[1] for v in runs.values()) / pts, 4)
    out["e2e_min"], out["e2e_max"] = min(v[0] for v in runs.values()), max(v[0] for v in runs.values())
    out["gate_violations"] = fd.gate_violations()
    return out


def main():
    if "--one" in sys.argv:
        print("FLIP_SOURCES " + json.dumps(measure()), flush=True)
        return
    exact = os.path.join(ROOT, "sapcu_amd.py")
    pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd") and os.path.isdir(os.path.join(ROOT, d))][0]
    exact = os.path.join(ROOT, pkg, "csrc", "libsapcu_hip_exact.so")
    variants = [("default", {}), ("gemm_f32", {"SAPCU_GEMM": "f32"})]
    if os.path.exists(exact):
        variants += [("exact_order", {"SAPCU_LIB_PATH": exact}), ("exact_order+f32", {"SAPCU_LIB_PATH": exact, "SAPCU_GEMM": "f32"})]
    rows = []
    for name, env in variants:
        e = {k: v for k, v in os.environ.items() if k not in ("SAPCU_GEMM", "SAPCU_LIB_PATH")}
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], capture_output=True, text=True, env=e, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("FLIP_SOURCES ")]
        if r.returncode != 0 or not line:
            print("variant %s failed:\n%s\n%s" % (name, r.stdout[-2000:], r.stderr[-3000:]), file=sys.stderr)
            raise SystemExit(1)
        rows.append((name, json.loads(line[0][len("FLIP_SOURCES "):])))
    print("| build / switch | fd: flipped patches of 256 | flipped rows of 36 864 | max err, forced neighbours | e2e within 2e-4 (8 clouds pooled) | min .. max over the clouds |")
    print("|---|---:|---:|---:|---:|---:|")
    for name, m in rows:
        print("| %s | %d | %d | %.2e | %.4f | %.3f .. %.3f |" % (name, m["fd256_flip_patches"], m["fd256_flip_rows"], m["fd256_max_err_forced"],
                                                             m["e2e_within_2e4_pooled"], m["e2e_min"], m["e2e_max"]))
    p = os.path.join(HERE, "golden", "ref_vs_ref.npz")
    if os.path.exists(p):
        g = np.load(p)
        print("| reference, 1 thread vs 8 threads (the same reference code against itself) | %d | %d | (distances: %.4f within 2e-4) | %.4f (sphere-2048) | — |"
              % (int(g["fd256_flip_patches"]), int(g["fd256_flip_rows"]), float(g["fd256_within_2e4"]), float(g["e2e_within_2e4"])))
    print()
    print("per cloud: " + json.dumps({name: m["e2e"] for name, m in rows}))


if __name__ == "__main__":
    main()
