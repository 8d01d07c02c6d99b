export TMPDIR=/tmp
for T in 1024 512 256 128; do
  echo "== threads $T"
  MGX_NO_MID=1 MGX_TAIL_THREADS=$T timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl$T -o t -- python3 tools/tail_levels.py > /dev/null 2> gpurun_out/tl.err
  python3 tools/tail_levels.py --read $(find gpurun_out/tl$T -name "*kernel_trace.csv" | sed -n 1p)
  rm -rf gpurun_out/tl$T
done
