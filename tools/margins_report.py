"""Parse the reports the multi-step tests print (pytest -s logs) and show, per asserted quantity, the worst value over
the runs next to its bound: tools/margins_report.py gpurun_out/r5_w_margins_run*.log"""
import re, sys, collections
five = collections.defaultdict(list)      # (scene-ish index, step, key) -> values
ten = collections.defaultdict(list)
worst5 = collections.defaultdict(list)
for path in sys.argv[1:]:
    txt = open(path).read()
    blocks5 = 0
    for line in txt.splitlines():
        m = re.match(r".*step (\d): \|dloss\| hip (\S+) aten (\S+) .*norm err hip (\S+) aten (\S+)  cumulative update err: backbone hip (\S+) \((\S+)\) aten (\S+), other hip (\S+) \((\S+)\) aten (\S+)", line)
        if m:
            k = int(m.group(1))
            if k == 0:
                blocks5 += 1
            sc = (blocks5 - 1) % 2
            for key, idx in (("dl", 2), ("dnorm", 4), ("deep", 6), ("rest", 9)):
                five[(sc, k, key)].append(float(m.group(idx)))
            continue
        m = re.match(r".*worst step, (\w+): hip (\S+) aten (\S+) \(ratio (\S+)\)", line)
        if m:
            worst5[m.group(1)].append((float(m.group(2)), float(m.group(3))))
            continue
        m = re.match(r".*step (\d): loss \S+ / \S+  \|dparts\| hip (\S+) control (\S+)  min cos hip (\S+) \((\S+)\) control (\S+)  max drift hip (\S+) \((\S+)\) control (\S+)  worst hip/control drift ratio (\S+)", line)
        if m:
            k = int(m.group(1))
            ten[(k, "dl")].append(float(m.group(2))); ten[(k, "dc")].append(float(m.group(3)))
            ten[(k, "1-cos hip")].append(1 - float(m.group(4))); ten[(k, "1-cos ctl")].append(1 - float(m.group(6)))
            ten[(k, "ratio")].append(float(m.group(10)))
print("five-step test: worst over runs per (scene, step)   [bounds LOSS 1e-4,1e-3,2.5e-3,5e-2,5e-2 | NORM 2e-3,1e-2.. | DEEP 2e-2,.15,.25,.6,.6 | REST 2e-3,.05,.06,.1,.12]")
for sc in (0, 1):
    for k in range(5):
        print("  scene %d step %d:" % (sc, k), "  ".join("%s max %.2e (n=%d)" % (key, max(five[(sc, k, key)]), len(five[(sc, k, key)])) for key in ("dl", "dnorm", "deep", "rest") if five[(sc, k, key)]))
print("five-step control (worst step hip vs max(5 x aten, floor)): floors dl 2.3e-2 dnorm 8.3e-3 deep 0.14 rest 9.2e-2")
for key, v in worst5.items():
    print("  %s: hip max %.2e; worst hip/(5*aten) %.2f; n=%d" % (key, max(h for h, c in v), max(h / (5 * c) for h, c in v), len(v)))
print("ten-step test")
for k in range(10):
    if ten[(k, "dl")]:
        print("  step %d: dl max %.2e  dc max %.2e  (1-cos hip)/(1.5(1-cos ctl)+0.003) n/a  worst drift ratio %.2f  1-cos hip max %.4f ctl min %.4f" % (
            k, max(ten[(k, "dl")]), max(ten[(k, "dc")]), max(ten[(k, "ratio")]), max(ten[(k, "1-cos hip")]), min(ten[(k, "1-cos ctl")])))
