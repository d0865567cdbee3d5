#!/bin/bash
# build a variant of the library for an A/B run: tools/build_variant.sh <out.so> [-DNAME=value ...]
# (objects under build/ab/<name>/; select the result with NAQS_HIP_LIB, e.g. through tools/gpu_ab.sh)
set -e
out=$1; shift
name=$(basename "$out" .so)
src=${NAQS_SRC:-naqs-for-quantum-chemistry_amd/csrc}
obj=build/ab/$name
mkdir -p "$obj"
echo "#define NAQS_SRC_HASH \"$(cat $src/*.hip $src/*.hpp | sha256sum | cut -c1-12)-var\"" > "$obj/naqs_src_hash.h"
pids=()
for f in naqs_hip naqs_logpsi naqs_sample naqs_grad naqs_phase_grad; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -I"$src" -Iinclude -I"$obj" "$@" -c -o "$obj/$f.o" "$src/$f.hip" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj"/*.o
echo "built $out"
