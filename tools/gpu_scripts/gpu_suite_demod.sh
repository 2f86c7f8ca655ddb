#!/bin/bash
# GPU parity suite, then the demodulator timing of the product build (3 short bench runs)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
bash $R/tools/gpu_scripts/gpu_suite.sh || exit 1
bash $R/tools/gpu_scripts/gpu_demod_ab.sh - "$@" -
