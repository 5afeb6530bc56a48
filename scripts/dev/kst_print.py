import csv,glob,sys
f=glob.glob("/tmp/kst/**/k_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[1]) if len(sys.argv)>1 else 6]:
    if "selftest" in r["Name"]: continue
    print(f'{r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50]:50s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.2f} min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f} us')
