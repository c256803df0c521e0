"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV: aggregates the
kernels between the last two optimizer steps (sgd_clip_kernel launches; multi_tensor_apply clusters in revisions before the fused optimiser)."""
import csv, glob, collections, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'sgd_clip_kernel' in r['Kernel_Name']]
if not opt:     # revisions before the fused optimiser: torch.optim.SGD's foreach kernels
    opt = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name']]
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

# idle gaps of the step: time between the end of a kernel and the start of the next one
gaps = []
prev_end = int(seg[0]['End_Timestamp'])
for a_, b_ in zip(seg[:-1], seg[1:]):
    e = max(prev_end, int(a_['End_Timestamp']))
    g_ = int(b_['Start_Timestamp']) - e
    prev_end = e
    if g_ > 0:
        gaps.append((g_, a_['Kernel_Name'][:60], b_['Kernel_Name'][:60]))
tot_gap = sum(g_ for g_, _, _ in gaps)
print("idle between kernels: %.2f ms in %d gaps; > 20 us: %.2f ms" % (tot_gap / 1e6, len(gaps),
      sum(g_ for g_, _, _ in gaps if g_ > 20000) / 1e6))
for g_, a_, b_ in sorted(gaps, reverse=True)[:25]:
    print("  %7.1f us  after %-60s before %s" % (g_ / 1e3, a_, b_))
