import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + '/*/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void fk::','')
    g = (int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[(name, g)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot = sum(sum(v) for v in agg.values())
print('total kernel time us', tot)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    print('%-36s grid %-14s n=%4d avg %8.1f us  total %8.1f us (%.1f%%)' % (k[0][:36], k[1], len(v), sum(v)/len(v), sum(v), 100*sum(v)/tot))
# class-wide mean the bench's roofline uses: every k_keyswitch* dispatch, per launch of the class
# (a limb-parallel op = stage-1 kernel + its normalisation pass = one launch)
ks = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'k_keyswitch' in r['Kernel_Name']]
n = sum(1 for nm, _ in ks if 'k_keyswitch_norm' not in nm)
if n:
    print('keyswitch class: %d launches, mean %.2f us per launch' % (n, sum(d for _, d in ks) / n))
