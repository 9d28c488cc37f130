import sys, csv, glob, os
from collections import defaultdict
root=sys.argv[1]
f=glob.glob(os.path.join(root,"**","*kernel_stats.csv"),recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("dpmm::","")
    print(f"{float(r['AverageNs'])/1e3:10.1f} us  calls {r['Calls']:>5s}  {n[:60]}")
