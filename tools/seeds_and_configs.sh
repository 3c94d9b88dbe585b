#!/bin/bash
# development aid: the headline figure over seeds 1-5 of config 4, the other BASELINE configurations WITH the CPU restatement timed beside
# each of them (1 thread and the OpenMP build), and config 4 with corner edges
P='import sys,json; j=json.loads(sys.stdin.read()); c=j["config"]; b=j.get("cpu_baseline"); o=c.get("cpu_secondary_openmp") or {}; print("%-22s %8.1f LM iters/s  %.3f ms/iter  %4.1f iters %4.1f attempts per call  lin %.1f us  call(create+run+read) %.1f ms%s" % (sys.argv[1], j["value"], c["ms_per_lm_iter"], c["lm_iters_per_step"], c["solve_attempts_per_step"], j["roofline"]["launch_ms"]*1e3, c["steady_state_call_ms"].get("total", 0), ("  | CPU port 1 thread %.2f LM iters/s (%.1f ms per call); OpenMP x%s %s LM iters/s" % (b["value"], b["ms_per_call"], o.get("cores", "-"), ("%.2f" % o["value"]) if "value" in o else "-")) if b else ""))'
for s in 1 2 3 4 5; do python bench.py --no-cpu-baseline --no-extras --seed $s 2>/dev/null | python -c "$P" "cfg4 seed $s"; done
for c in cfg2 cfg3 cfg4; do python bench.py --no-extras --config $c 2>/dev/null | python -c "$P" "$c seed 1"; done
python bench.py --no-cpu-baseline --no-extras --with-corners 2>/dev/null | python -c "$P" "cfg4 + corner edges"
