import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("k_lines","k_update"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[200:1200]
import statistics as st
g1=[];g2=[];d1=[];d2=[]
for a,b in zip(rows,rows[1:]):
    gap=int(b["Start_Timestamp"])-int(a["End_Timestamp"])
    if a["Kernel_Name"].startswith("k_lines") and b["Kernel_Name"].startswith("k_update"): g1.append(gap); d1.append(int(a["End_Timestamp"])-int(a["Start_Timestamp"]))
    if a["Kernel_Name"].startswith("k_update") and b["Kernel_Name"].startswith("k_lines"): g2.append(gap); d2.append(int(a["End_Timestamp"])-int(a["Start_Timestamp"]))
print("k_lines dur med",st.median(d1),"gap lines->update med",st.median(g1),"k_update dur med",st.median(d2),"gap update->lines med",st.median(g2), "p90 gaps", sorted(g1)[int(.9*len(g1))], sorted(g2)[int(.9*len(g2))])
