"""Sum rocprofv3 counter_collection rows per kernel and counter: python scripts/pmc_sum.py <counter_collection.csv> [name filter ...]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if len(sys.argv) > 2 and not any(f in k for f in sys.argv[2:]): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); calls[(k, r['Counter_Name'])] += 1
for k, cs in acc.items():
    print(k[:100])
    for c, v in sorted(cs.items()): print(f'   {c:32s} {v / calls[(k, c)]:16.0f} per launch ({calls[(k, c)]} launches)')
