#!/bin/bash
# HBM-side traffic of the roofline GEMM ([4096,624] x [624,256]): FETCH_SIZE and WRITE_SIZE in separate passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/gemm_traffic; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/f -o g -- python tools/gemm_bench.py 0 f32 > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/w -o g -- python tools/gemm_bench.py 0 f32 > $O/w.log 2>&1
python - <<'PY'
import csv, json
def avg(path, counter):
  v = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter and 'gemm_f32_kernel' in r['Kernel_Name']]
  return sum(v) / len(v), len(v)
f, n = avg('gpurun_out/gemm_traffic/f/g_counter_collection.csv', 'FETCH_SIZE')
w, _ = avg('gpurun_out/gemm_traffic/w/g_counter_collection.csv', 'WRITE_SIZE')
out = {'kernel': 'er::gemm_f32_kernel<NN> 4096x256x624', 'launches': n, 'fetch_KiB': f, 'write_KiB': w,
       'bytes_per_launch': (2.0 * f + w) * 1024.0,
       'algorithmic_bytes_per_launch': 4.0 * (4096 * 624 + 624 * 256 + 4096 * 256)}
print(json.dumps(out))
json.dump(out, open('gpurun_out/gemm_traffic/summary.json', 'w'), indent=1)
PY
rm -f $O/*/*kernel_trace.csv
