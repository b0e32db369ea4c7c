import csv, sys
for r in csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kstats/kernel_stats.csv")):
    n = r["Name"].replace("besst::(anonymous namespace)::", "")
    if n.startswith("void at::") or "rocprim" in n or n.startswith("__amd") or n.startswith("at::"): continue
    print("%-60s %6s %10.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
