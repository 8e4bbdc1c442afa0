#!/usr/bin/env python3
"""Condense rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel averages (JSON).

Units and corrections follow MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide (16 B/lane) coalesced stream, so `fetch_bytes_corrected = 2 * fetch_bytes_raw`;
WRITE_SIZE is exact for 4-16 B/lane streaming stores."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256():
    """Hash of the kernel sources the counters were taken on (bench.py refuses a PMC file whose hash is not the tree's)."""
    d = os.path.join(ROOT, "c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void sapcu::", "").replace("sapcu::", "")
        agg[name].append(float(r["Counter_Value"]) * 1024.0)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def per_kernel_raw(d, counter):
    """like per_kernel, but the counter's own unit (cycles), averaged per launch"""
    fs = glob.glob(d + "/*/*_counter_collection.csv")
    if not fs:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void sapcu::", "").replace("sapcu::", "")
        agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main(fetch_dir, write_dir, out, mfma_dir=None, steps_profiled=None):
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    # third pass (optional): matrix-pipe busy cycles (= 32 per v_mfma_f32_32x32x16, summed over the chip's SIMDs) and
    # GRBM_GUI_ACTIVE (summed over the 8 XCDs) -> busy fraction = busy / (gui_active / 8 * 1024 SIMDs)
    mb = per_kernel_raw(mfma_dir, "SQ_VALU_MFMA_BUSY_CYCLES") if mfma_dir else {}
    ga = per_kernel_raw(mfma_dir, "GRBM_GUI_ACTIVE") if mfma_dir else {}
    res = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith(("gemm", "fn_", "fd_", "patch_", "knn_", "rowgroup", "edge_", "gather", "displace", "l2_", "to_split")):
            continue
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        res[k] = {"launches": max(nf, nw), "fetch_bytes_raw": round(f), "fetch_bytes_corrected": round(2 * f),
                  "write_bytes": round(w), "hbm_bytes_per_launch": round(2 * f + w)}
        if k in mb and ga.get(k):
            res[k]["mfma_busy_cycles"] = round(mb[k])
            res[k]["gui_active_sum_xcd"] = round(ga[k])
            res[k]["mfma_busy_frac"] = round(mb[k] / (ga[k] / 8.0 * 1024.0), 4)
    # the positional-encoding GEMM runs as two instantiations of the big-tile kernel (256- and 128-column tiles): one
    # launch-weighted record for the family, which is what bench.py's roofline leg times
    fam = [v for k, v in res.items() if k.startswith("gemm_bt_kernel<6,")]
    if fam:
        n = sum(v["launches"] for v in fam)
        comb = {"launches": n, "members": sorted(k for k in res if k.startswith("gemm_bt_kernel<6,"))}
        for key in ("fetch_bytes_raw", "fetch_bytes_corrected", "write_bytes", "hbm_bytes_per_launch"):
            comb[key] = round(sum(v[key] * v["launches"] for v in fam) / n)
        if all("mfma_busy_cycles" in v for v in fam):
            busy = sum(v["mfma_busy_cycles"] * v["launches"] for v in fam)
            act = sum(v["gui_active_sum_xcd"] * v["launches"] for v in fam)
            comb["mfma_busy_frac"] = round(busy / (act / 8.0 * 1024.0), 4)
        res["gemm_bt_kernel<6>"] = comb
    # whole step: sum over the kernels of (bytes per launch x launches) / steps profiled (warmup + timed; the PMC runs use
    # --no-roofline --no-strong-leg, so every launch belongs to a step)
    if steps_profiled:
        n = float(steps_profiled)
        tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in res.items() if "members" not in v)
        res["_step"] = {"steps_profiled": int(n), "hbm_bytes_per_step": round(tot / n),
                        "launches_per_step": {k: v["launches"] / n for k, v in res.items() if "members" not in v and not k.startswith("_")}}
    res["_meta"] = {"csrc_sha256": csrc_sha256(),
                    "commit": os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip() or None,
                    "units": "bytes; FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, separate --pmc passes"}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: res[k] for k in res if k.startswith("_")}, indent=1)[:2000])


if __name__ == "__main__":
    main(*sys.argv[1:6])
