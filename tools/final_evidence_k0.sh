H=$1; O=gpurun_out/r6_final2; mkdir -p $O
run() { f=$O/$1; note=$2; shift 2; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# commit $H: ${envs[*]} $*   $note" > $f; env "${envs[@]}" "$@" 2>&1 | tail -5 >> $f; }
run fuzz_soak.txt "(seeds 60000..: 12x the CI seed count, none shared with CI or earlier soaks; final kernels incl. K0's counter layout)" C3R_FUZZ_BASE=60000 C3R_FUZZ_SCALE=12 -- python -m pytest tests/test_gpu_fuzz.py -m gpu -q
run full_contig_check.txt "" C3R_X=0 -- python tests/evidence/full_contig_check.py
run full_contig_check_config3.txt "" C3R_X=0 -- python tests/evidence/full_contig_check.py --config3
( echo "# commit $H: tools/fault_soak2.sh, short processes, one MI355X"; bash tools/fault_soak2.sh 40 python tools/step_time.py stress 3; bash tools/fault_soak2.sh 30 python tools/step_time.py cap 3; bash tools/fault_soak2.sh 30 python tools/step_time.py real 3; bash tools/fault_soak2.sh 20 python tools/step_time.py nocap 3; STEP_UNPINNED=1 bash tools/fault_soak2.sh 20 python tools/step_time.py cap 3; bash tools/fault_soak2.sh 20 python tools/deep_phases.py stress 0 ) > $O/fault_soak.txt 2>&1
grep -H "passed\|failed\|FULL CONTIG\|rounds\|FAIL" $O/*.txt
