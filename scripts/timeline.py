import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_init_state')][-1]
t0 = int(rows[idx]['Start_Timestamp'])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for r in rows[idx:idx + n]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:44]}")
