"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV: aggregates the
kernels between the last two optimizer steps (sgd_clip_kernel launches; multi_tensor_apply clusters in revisions before the fused optimiser)."""
import csv, glob, collections, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name'] or 'sgd_clip_kernel' in r['Kernel_Name']]
cl = []
for i in opt:
    t = int(rows[i]['Start_Timestamp'])
    if not cl or t - cl[-1][1] > 50e6:
        cl.append([t, t, i, i])
    else:
        cl[-1][1] = t; cl[-1][3] = i
print("optimizer clusters", len(cl))
a = cl[-2][3] + 1; b = cl[-1][3] + 1
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
print("step wall ms %.1f  kernels %d" % ((t1 - t0) / 1e6, len(seg)))
agg = collections.defaultdict(lambda: [0, 0]); busy = 0
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = r['Kernel_Name'][:100]
    agg[n][0] += d; agg[n][1] += 1; busy += d
print("sum kernel ms %.1f" % (busy / 1e6))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:top]:
    print("%7.2f ms %5d  avg %8.1f us  %s" % (d / 1e6, c, d / c / 1e3, n))
