cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 3061 3083 3201; do timeout 300 python tools/strict_sweep.py 1 $s 2>&1 | head -2 | cut -c1-700; done
timeout 300 python -m pytest tests/test_lm_control.py -m gpu -x -q --timeout 100 -k "decision" 2>&1 | tail -2
