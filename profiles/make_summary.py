#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a per-kernel table (markdown)."""
import collections
import csv
import glob
import sys


def main(d, title):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void sapcu::", "").replace("sapcu::", "")
        agg[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print("## %s\n" % title)
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("| `%s` | %d | %.2f | %.1f | %.1f |" % (k, len(v), sum(v) / 1e3, sum(v) / len(v), 100 * sum(v) / tot))
    print("\ntotal kernel time %.1f ms over %d dispatches\n" % (tot / 1e3, len(rows)))
    # per-launch durations (us, dispatch order) of every kernel that is launched with more than one shape per step and matters
    # (GEMMs: the three fn blocks alternate d = 128, 256, 512; in-patch kNN: xyz and 64 / 128 / 256-d feature space; ...)
    seq = {k: [round(x, 1) for x in v] for k, v in agg.items()
           if len(v) <= 64 and sum(v) / len(v) >= 20.0 and not k.startswith(("void at::", "__amd"))}
    if len(sys.argv) > 3:
        import json
        json.dump(seq, open(sys.argv[3], "w"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
