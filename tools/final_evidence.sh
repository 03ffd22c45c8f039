#!/bin/bash
# The end-of-round evidence runs at one commit, one MI355X (~30 min):  bash tools/final_evidence.sh <commit> <out dir>
# fuzz soaks (fresh seeds: 12x the CI count; 6x with every span through k_fused_deep; 6x with every span cut into slices as a giant span) and the
# whole-contig checks (18 channels, 30 channels, 18 through the deep kernel, 18 with every span a giant span)
H=$1; O=$2; mkdir -p $O
GI="C3R_SPLIT_CUS=1000000000 C3R_SPLIT_MIN=1 C3R_GIANT=1 C3R_DEEP_MIN=1 C3R_EVWG=1 C3R_SPLIT_SLICE=64"
run() {   # <file> <note> <env...> -- <command...>
  f=$O/$1; note=$2; shift 2; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# commit $H: ${envs[*]} $*   $note" > $f
  env "${envs[@]}" "$@" 2>&1 | tail -5 >> $f
}
run fuzz_soak.txt "(seeds 40000..: 12x the CI seed count, none shared with CI or earlier soaks)" C3R_FUZZ_BASE=40000 C3R_FUZZ_SCALE=12 -- python -m pytest tests/test_gpu_fuzz.py -m gpu -q
run fuzz_soak_deep_kernel.txt "(EVERY span through k_fused_deep)" C3R_DEEP_MIN=1 C3R_EVWG=1 C3R_FUZZ_BASE=46000 C3R_FUZZ_SCALE=6 -- python -m pytest tests/test_gpu_fuzz.py -m gpu -q
run fuzz_soak_giant_spans.txt "(EVERY span a giant span, cut into 64-record slices for k_deep_walk / k_deep_alleles)" $GI C3R_FUZZ_BASE=52000 C3R_FUZZ_SCALE=6 -- python -m pytest tests/test_gpu_fuzz.py -m gpu -q
run full_contig_check.txt "" C3R_X=0 -- python tests/evidence/full_contig_check.py
run full_contig_check_deep_kernel.txt "(every span through k_fused_deep)" C3R_DEEP_MIN=1 C3R_EVWG=1 -- python tests/evidence/full_contig_check.py
run full_contig_check_giant_spans.txt "(every span a giant span while slots last: 256 per scan)" $GI -- python tests/evidence/full_contig_check.py
run full_contig_check_config3.txt "" C3R_X=0 -- python tests/evidence/full_contig_check.py --config3
grep -H "passed\|failed\|FULL CONTIG\|Error\|error" $O/*.txt
