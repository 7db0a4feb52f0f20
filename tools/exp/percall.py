import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
for key in sys.argv[2:]:
    d = [(e - s) / 1e3 for s, e, n in rows if key in n]
    print(key, len(d), " ".join(f"{x:.0f}" for x in d[-12:]))
