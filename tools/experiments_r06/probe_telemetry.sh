for c in /sys/class/drm/card*/device; do echo "== $c"; cat $c/vendor 2>/dev/null; ls $c/hwmon/hwmon*/ 2>/dev/null | tr '\n' ' '; echo; cat $c/hwmon/hwmon*/power1_cap $c/hwmon/hwmon*/power1_cap_max $c/hwmon/hwmon*/power1_cap_default 2>/dev/null; cat $c/current_compute_partition $c/available_compute_partition $c/current_memory_partition 2>/dev/null; cat $c/numa_node 2>/dev/null; cat $c/local_cpulist 2>/dev/null; ls $c | tr '\n' ' '; echo; python3 - $c <<'PY'
import sys,struct
try:
    b=open(sys.argv[1]+'/gpu_metrics','rb').read()
    print('gpu_metrics bytes',len(b),'size',struct.unpack_from('<H',b,0)[0],'format',b[2],'content',b[3])
except Exception as e: print('gpu_metrics',e)
PY
done
which amd-smi rocm-smi; timeout 20 amd-smi metric --help 2>&1 | head -40; timeout 30 amd-smi metric -g 0 --throttle --power --clock --json 2>&1 | head -80; nproc; lscpu | grep -i "numa\|socket\|model name" | head
