# prints the dispatch timeline of the LAST train step in a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous end, queue, name
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step = from one counter_add launch (the last launch of a train step) to the next
idx = [i for i, r in enumerate(rows) if 'counter_add' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
def nm(s):
    s = re.sub(r'\(anonymous namespace\)::', '', s)
    s = re.sub(r'^void ', '', s)
    m = re.match(r'([\w:]+(<[^(]*>)?)', s)
    return (m.group(1) if m else s)[:70]
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  dur %8.1f  gap %7.1f  q%s g%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id','?'), r.get('Grid_Size','?'), nm(r['Kernel_Name'])))
    prev_end = max(prev_end, e)
print('step span %.1f us' % ((prev_end - t0) / 1e3))
