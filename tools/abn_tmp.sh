for i in 1 2 3; do for K in 0 1; do
HIP_FORCE_DEV_KERNARG=$K python3 bench.py --no-sweep --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('DEV_KERNARG=$K', round(d['ms_per_step']*1e3,2), 'us/step')"
done; done
python3 bench.py --no-sweep --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step']*1e3,2), 'us/step')"
