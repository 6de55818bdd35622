import re, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for line in open(sys.argv[1]):
    m = re.match(r"\[pcabo batch\] gang (\d+): (\d+) runs, (\d+) launches \(([\d.]+) entries each\), host steps ([\d.]+) ms, launch calls ([\d.]+) ms, waiting ([\d.]+) ms of ([\d.]+) ms", line)
    if m:
        g = int(m.group(1)); a = agg[g]
        a[0] += int(m.group(3)); a[1] += float(m.group(5)); a[2] += float(m.group(6)); a[3] += float(m.group(7)); a[4] += float(m.group(8))
for g in sorted(agg):
    a = agg[g]
    print(f"gang {g}: launches {a[0]}, per launch: wait {1e3*a[3]/a[0]:.1f} us, steps {1e3*a[1]/a[0]:.1f}, launch {1e3*a[2]/a[0]:.1f}; total {a[4]/330:.2f} ms/iter")
