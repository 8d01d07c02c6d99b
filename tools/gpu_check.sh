set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out; rm -rf gpurun_out/tailp
timeout -s KILL 120 rocprofv3 --kernel-trace -d $R/gpurun_out/tailp -o t --output-format csv -- python3 tools/tail_probe.py > gpurun_out/tailp.log 2>&1
python3 - <<'P'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/tailp/t_kernel_trace.csv')) if 'k_tail' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
for k in range(0,len(d),20): print(k//20, sum(d[k+5:k+20])/15)
P
