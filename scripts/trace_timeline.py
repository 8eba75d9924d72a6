# prints the kernel timeline of the last train step in a rocprofv3 kernel trace (eager bench run): start offset, duration,
# gap to the previous kernel's end (negative = overlapped), name.   usage: python scripts/trace_timeline.py gpurun_out/<name> [min_us]
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['End_Timestamp'])
prev = t0
small = 0.0; nsmall = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d = (e - s) / 1e3
    if d >= min_us:
        print('%9.1f %8.1f gap %7.1f  %s' % ((s - t0) / 1e3, d, (s - prev) / 1e3, r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70]))
    else:
        small += d; nsmall += 1
    prev = max(prev, e)
print('step span %.1f us; %d kernels under %.1f us totalling %.1f us' % ((prev - t0) / 1e3, nsmall, min_us, small))
