"""Durations of k_lin_gram launches in a rocprofv3 --kernel-trace CSV: the first launch of every step has all windows active
(the state is restored at the start of a step), so the LONGEST launches are the comparable ones across diagnostic variants."""
import csv, glob, sys
for d in sys.argv[1:]:
    fs = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if not fs: print(d, 'no trace'); continue
    by = {}
    for r in csv.DictReader(open(fs[0])):
        n = r['Kernel_Name'].split('(')[0]
        by.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = []
    for n, v in by.items():
        if any(k in n for k in ('k_lin_gram', 'k_rank1_mfma', 'k_build_solve_st', 'k_dogleg')):
            v2 = sorted(v, reverse=True)
            top = v2[:max(1, len(v2) // 10)]
            out.append(f"{n[:40]}: n={len(v)} mean={sum(v)/len(v):.1f} top10%={sum(top)/len(top):.1f} max={v2[0]:.1f}")
    print(d.rstrip('/').split('/')[-1], ' | '.join(sorted(out)))
