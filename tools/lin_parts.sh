python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for m in 1 2 4 8 16 31; do echo "parts=$m"; PPO_BA_LIN_PARTS=$m python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['launch_ms']*1e3,'us')"; done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['launch_ms']*1e3, j['roofline']['frac'], {k:round(v['us'],1) for k,v in j['config']['kernels'].items()})"
