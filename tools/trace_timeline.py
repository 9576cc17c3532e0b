"""Timeline of one slice of a rocprofv3 kernel trace: per dispatch start offset, duration and the gap since the
previous dispatch ended (same queue order).  usage: trace_timeline.py <dir> <first> <count>"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
first, count = int(sys.argv[2]), int(sys.argv[3])
t0 = int(rows[first]['Start_Timestamp'])
prev_end = None
for r in rows[first:first + count]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void fk::', '')[:40]
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print('%9.1f us  dur %7.1f  gap %6.1f  q%s  %-40s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get('Queue_Id', '?'), name, g))
    prev_end = e
