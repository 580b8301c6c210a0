#!/bin/bash
# Tagged libraries with the timing ablations of k_conv_tile_bf16 (INVALID results; see conv_bf16.hip) — only
# conv_bf16.o is recompiled, the other objects come from the default build.  Then on a GPU:
#   for tag in "" a1 ...; do ME_AMD_HOST=python ME_AMD_LIB_TAG=$tag python scripts/unet_layers.py; done
#   ME_AMD_HOST=python ME_AMD_LIB_TAG=tim python scripts/bf16_phase_timing.py     (phase counters, valid results)
cd "$(dirname "$0")/.." || exit 1
python -m minkowskiengine_amd.build > /dev/null || exit 1
B=minkowskiengine_amd/csrc/build
for v in "noaread:-DME_ABL_NO_AREAD" "noacc:-DME_ABL_NO_ACC" "noboth:-DME_ABL_NO_AREAD -DME_ABL_NO_ACC" \
         "a1:-DME_ABL_NO_AREAD -DME_ABL_NO_ACC -DME_ABL_NO_MFMA" "a2:-DME_ABL_NO_AREAD -DME_ABL_NO_ACC -DME_ABL_NO_STAGE" \
         "a3:-DME_ABL_NO_GATHER" "a4:-DME_ABL_NO_WLOAD" "a6:-DME_ABL_NO_GATHER -DME_ABL_NO_WLOAD" \
         "tim:-DME_BF16_TIMING" \
         "a5:-DME_ABL_NO_AREAD -DME_ABL_NO_ACC -DME_ABL_NO_MFMA -DME_ABL_NO_STAGE -DME_ABL_NO_GATHER -DME_ABL_NO_WLOAD"; do
  tag=${v%%:*}; fl=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-lambda-capture $fl \
      -c minkowskiengine_amd/csrc/conv_bf16.hip -o /tmp/conv_bf16_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/conv.o /tmp/conv_bf16_$tag.o $B/conv_f32x3.o $B/coords.o \
      $B/norm.o $B/pack.o $B/pool.o -o minkowskiengine_amd/libme_amd_$tag.so ) &
done
wait
ls minkowskiengine_amd/libme_amd_*.so
