for rep in 1 2; do
for lib in "" variants/libfsr1_rcas8.so; do
 for math in strict f; do
  FSR1_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --math $math --steps 2000 --warmup 200 --no-cpu-baseline --no-also --no-latency --no-parity --no-submit-ceiling --no-cold-rcas --no-telemetry-window 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=$lib math=$math', d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('value'))" | tee -a gpurun_out/r6g1_rcas_rows_overlapped.log
 done
done
done
