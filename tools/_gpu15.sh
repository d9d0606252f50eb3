#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/t8prof; mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python tools/t8_bench.py --iters 2 > $out/log.txt 2>&1
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 2*... print the tail: one iteration of each call
sel=[r for r in rows if 'conv3d_bf16_kernel' in r['Kernel_Name'] or 'splitk' in r['Kernel_Name'] or 'wgrad' in r['Kernel_Name'] or 'reduce' in r['Kernel_Name']]
for r in sel[:60]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
    print(f"{d:9.1f} us  grid {r.get('Grid_Size_X','?'):>8}  wg {r.get('Workgroup_Size_X','?')}  {r['Kernel_Name'][:90]}")
PY
rm -rf $out/kt
