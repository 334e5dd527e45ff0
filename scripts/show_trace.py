import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_zero_f64' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
tot = 0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += en - st
    print("%8.1f  dur %6.1f  grid %-14s %s" % ((st - t0) / 1e3, (en - st) / 1e3, r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'] + 'x' + r['Grid_Size_Z'], r['Kernel_Name'][:64]))
print("kernels", e - s, "sum dur %.1f us" % (tot / 1e3), "span %.1f us" % ((int(rows[e]['Start_Timestamp']) - t0) / 1e3))
