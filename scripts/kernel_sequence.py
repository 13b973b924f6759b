"""Developer probe: prints the kernel sequence (name, duration, gap to the previous kernel) of the
LAST `count` kernels of a rocprofv3 --kernel-trace csv."""
import csv
import sys

path, count = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-count:]
prev = None
for r in rows:
    start, end = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (start - prev) / 1e3 if prev is not None else 0.0
    name = r['Kernel_Name'].replace('tonic::', '').replace('(anonymous namespace)::', '')[:60]
    print(f'{(end - start) / 1e3:8.2f} us  gap {gap:6.2f}  grid {r.get("Grid_Size_X", "?"):>6} x {r.get("Grid_Size_Y", "?")} x {r.get("Grid_Size_Z", "?")}  {name}')
    prev = end
