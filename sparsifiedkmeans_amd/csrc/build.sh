#!/bin/bash
# Builds libspkm.so (gfx950 only) in-tree.  HIP runtime symbols are left undefined on purpose:
# the loader (sparsifiedkmeans_amd/_lib.py) binds them to the libamdhip64 that is already in the
# process (PyTorch's bundled copy when torch is imported, /opt/rocm otherwise), so that device
# pointers handed over from torch tensors belong to the same runtime.
# Six translation units, compiled in parallel: api.hip (contexts, shards, operators, FWHT, RCCL), api_lloyd.hip (the Lloyd
# engine and the kernels it launches) and the four instantiation sets of the 4-lanes-per-point screen kernel (screen_quad.hip).
set -euo pipefail
cd "$(dirname "$0")"
python3 gen_assign_steps.py
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
pids=()
$HIPCC $FLAGS -c api.hip -o api.o "$@" & pids+=($!)
$HIPCC $FLAGS -c api_lloyd.hip -o api_lloyd.o "$@" & pids+=($!)
for bits in 16 32; do
  for pts in 0 1; do
    $HIPCC $FLAGS -DSPKM_SQ_IRBITS=$bits -DSPKM_SQ_PTS=$pts -c screen_quad.hip -o sq_${bits}_${pts}.o "$@" & pids+=($!)
  done
done
rc=0
for pid in "${pids[@]}"; do wait "$pid" || rc=1; done
[ $rc -eq 0 ] || { echo "build.sh: a translation unit failed to compile" >&2; exit 1; }
# SPKM_BUILD_OUT: write the library somewhere else (compile checks while a GPU job may be snapshotting the tree)
OUT=${SPKM_BUILD_OUT:-../libspkm.so}
g++ -shared -o "$OUT" api.o api_lloyd.o sq_16_0.o sq_16_1.o sq_32_0.o sq_32_1.o -Wl,-z,undefs
rm -f api.o api_lloyd.o sq_16_0.o sq_16_1.o sq_32_0.o sq_32_1.o
# (HIP runtime symbols stay undefined on purpose; anything else undefined is a kernel or helper that no translation unit defines)
if nm -u "$OUT" | grep -v " hip\| __hip\|GLIBC\|CXXABI\|GCC_\|_ITM_\|__gmon\|__cxa\|dlopen\|dlsym\|dlerror\|dlclose" | grep -q " U "; then
  echo "build.sh: unexpected undefined symbols:" >&2; nm -u "$OUT" | grep -v " hip\| __hip\|GLIBC\|CXXABI\|GCC_\|_ITM_\|__gmon\|__cxa\|dlopen\|dlsym\|dlerror\|dlclose" >&2; exit 1
fi
