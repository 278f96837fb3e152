# A/B of fill block sizes (GIQL_FILL_NT builds): headline bench + cfg2 sparse / dense
for rep in 1 2; do for lib in libgiql_hip.so libgiql_hip_f512.so libgiql_hip_f1024.so; do echo -n "$lib "; GIQL_HIP_LIB=$PWD/giql_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['phase_ms']['fill'])"; done; done
for lib in libgiql_hip.so libgiql_hip_f512.so libgiql_hip_f1024.so; do echo "$lib"; GIQL_HIP_LIB=$PWD/giql_amd/$lib timeout -k 10 200 python tools/bench_ops.py --only cfg2 2>&1 | grep cfg2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' ', d['case'], d['ms'], d['phase_ms'].get('fill'))"; done
