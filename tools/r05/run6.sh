cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in x p1x p2x p3x; do echo "== build/chol_check_$v (x: product schedule; p1x: PAIR_BOUND=1; p2x: PAIR_BOUND=2; p3x: =3)"; timeout 120 build/chol_check_$v panels 2>&1 | grep -E "^n ( 594|1194|1644)"; done
done | tee gpurun_out/r05_chol_pairs.txt
