import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    if 'rocclr' in k: continue
    print(k, 'dur_us=%.1f' % (sum(dur[k]) / len(dur[k]) / 1e3), {c: round(sum(x) / len(x)) for c, x in v.items()})
