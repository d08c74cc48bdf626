import csv, glob
f = glob.glob("gpurun_out/b1prof/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("GPU busy total %.3f s over %d launches (2 images: warm-up + timed)" % (tot / 1e9, calls))
for r in rows[:14]:
    print("%6.2f%% %7d %8.1f us  %s" % (float(r["TotalDurationNs"]) / tot * 100, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
