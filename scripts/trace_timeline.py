# prints the dispatch timeline of the LAST train step in a rocprofv3 --kernel-trace CSV: start offset, duration, name
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = from the last 'l2_norm' back to the previous one
idx = [i for i, r in enumerate(rows) if 'clip_adam' in r['Kernel_Name'] or 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][-60:]
    print('%9.1f us  dur %8.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = max(prev_end, e)
print('step span %.1f us' % ((prev_end - t0) / 1e3))
