"""List the dispatches of one bench step in order (rocprofv3 kernel trace csv) — which kernels surround the tiny copyBuffer launches."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void sapcu::", "").replace("sapcu::", "") for r in rows]
# last step = from the last knn_outer_kernel on
last = max(i for i, n in enumerate(names) if n.startswith("knn_outer"))
seq = names[last:]
out, prev, cnt = [], None, 0
for n in seq:
    n = n[:60]
    if n == prev: cnt += 1
    else:
        if prev: out.append("%s x%d" % (prev, cnt))
        prev, cnt = n, 1
out.append("%s x%d" % (prev, cnt))
print("\n".join(out))
t0 = int(rows[last]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[last:])
print("step wall %.2f ms, kernel busy %.2f ms, dispatches %d" % ((t1 - t0) / 1e6, busy / 1e6, len(seq)))
