#!/bin/bash
# Build csrc/ into build_ab/libcss_hip_<NAME>.so (A/B timing: CSS_HIP_LIB=build_ab/libcss_hip_<NAME>.so).
# usage: tools/build_variant.sh NAME [extra hipcc flags ...]     (CSS_SRC_ROOT=<another checkout> builds that tree's csrc)
set -e
NAME=$1
shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${CSS_SRC_ROOT:-$ROOT}
OUT=$ROOT/build_ab/$NAME
mkdir -p $OUT
for f in css_core css_index css_encoder css_tokenizer; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" \
      -c $SRC/claude_semantic_search_amd/csrc/$f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/libcss_hip_$NAME.so $OUT/css_core.o $OUT/css_index.o $OUT/css_encoder.o $OUT/css_tokenizer.o
echo built $ROOT/build_ab/libcss_hip_$NAME.so
