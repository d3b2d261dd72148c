#!/bin/bash
# Diagnostic: build superdsm_amd/libsdsm_hip_<name>.so (or the product library for name = "main") with extra compiler flags, one object
# per source (objects of unchanged sources are reused: build/<name>/*.o).   usage: bash tools/build_variant.sh <name> [-DFOO=1 ...]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
src=${SDSM_SRC:-$root/superdsm_amd/csrc}
out=$root/build/$name; mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics $*"
echo "$flags" > "$out/flags.new"
if ! cmp -s "$out/flags.new" "$out/flags" 2>/dev/null; then rm -f "$out"/*.o; mv "$out/flags.new" "$out/flags"; fi
pids=()
for f in sdsm_api.hip sdsm_prepare.hip sdsm_setup.hip sdsm_solve.hip sdsm_post.hip sdsm_host.cpp; do
  o=$out/${f%.*}.o
  if [ ! -f "$o" ] || [ "$src/$f" -nt "$o" ] || [ "$src/sdsm_common.h" -nt "$o" ] || [ "$root/include/sdsm.h" -nt "$o" ] || { [ -f "$src/sdsm_logtab.h" ] && [ "$src/sdsm_logtab.h" -nt "$o" ]; }; then
    /opt/rocm/bin/hipcc $flags -c "$src/$f" -o "$o" & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
lib=$root/superdsm_amd/libsdsm_hip_$name.so
[ "$name" = main ] && lib=$root/superdsm_amd/libsdsm_hip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$lib" "$out"/*.o
echo "built $lib"
