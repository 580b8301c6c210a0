#!/bin/bash
# Tuning build of libme_amd.so: adds the phase-counter, timing-ablation (INVALID results) and LDS-DMA experiment
# kernels that the default build leaves out (csrc/me_amd_debug.h).  The tuning scripts (tune_conv*.py, phase_timing*.py,
# check_variant.py, prof_conv.py with VAR != 0) need it; tests and bench.py do not.
cd "$(dirname "$0")/.." && ME_AMD_EXTRA_HIPCC_FLAGS="-DME_DEBUG_VARIANTS" python -m minkowskiengine_amd.build "$@"
