# sweep of the whole-sample driver's host knobs on the GPU box:  bash tools/e2e_sweep.sh
cd /tmp && export TMPDIR=/tmp
for inf in 2 8 32; do for ft in 4 6; do
  echo "inflate threads $inf, fetch threads $ft"
  C3R_FETCH_INFLATE=$inf timeout 600 python $GRAFT_REPO_ROOT/tools/sample_e2e.py --contigs 22 --scale 0.25 --repeat 3 --fetch_threads $ft 2>&1 | grep "run [12]"
done; done
