import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    if r['Name'].startswith('ov_') or 'ov_' in r['Name'][:12]:
        n = r['Name'].split('(')[0][:70]
        print(f"{n:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f}")
