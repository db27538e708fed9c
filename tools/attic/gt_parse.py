import csv, glob
rows=[]
for f in glob.glob("gpurun_out/ks_gt/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "gemm_tokens_mfma_kernel" in n or n.startswith("Cijk"):
            rows.append((int(r["Start_Timestamp"]), "MFMA" if "gemm_tokens" in n else n[5:14], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
rows.sort()
import itertools
out=[(n, round(d,1)) for _,n,d in rows]
for i in range(0, len(out), 15): print(out[i:i+15])
