#!/bin/bash
# bench/variants.sh OUT SCRIPT ARGS... : runs a benchmark script with the in-tree library and with every variant library
# in bench/micro/bin/libdtf_hip_*.so (tuning experiments built with other -D flags)
out=$1; shift
mkdir -p $(dirname $out)
: > $out
echo "== in-tree" >> $out
timeout -k 10 300 python -u "$@" >> $out 2>&1 || exit 1
for lib in bench/micro/bin/libdtf_hip_*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib" >> $out
  DTF_HIP_LIB=$PWD/$lib timeout -k 10 300 python -u "$@" >> $out 2>&1 || exit 1
done
grep -v amdgpu.ids $out
