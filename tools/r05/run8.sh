cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_lm_control.py -m gpu -x -q --timeout 100 -k "decision" 2>&1 | grep -E "^E |assert" | head -20
