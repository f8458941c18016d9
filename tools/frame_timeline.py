"""print the kernels of one steady-state frame of a rocprofv3 kernel trace (tools/prof.sh TAG -> gpurun_out/prof_TAG/TAG_kernel_trace.csv):
stream, start and duration in us -- what overlaps what, and what the frame's critical path is"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_composite_merged" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a + 1]["Start_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(r["Kernel_Name"][:44].ljust(44), "stream", r.get("Stream_Id", "?"), "start", str(round((s - t0) / 1e3, 1)).rjust(7), "dur", str(round((e - s) / 1e3, 1)).rjust(7))
print("frame us", round((int(rows[b]["End_Timestamp"]) - t0) / 1e3, 1))
