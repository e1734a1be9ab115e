import csv,sys,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    if 'attn' not in k: continue
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} n={len(vals)} mean={sum(vals)/len(vals):.4e}")
