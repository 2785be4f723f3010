#!/bin/bash
# Builds libspkm.so (gfx950 only) in-tree.  HIP runtime symbols are left undefined on purpose:
# the loader (sparsifiedkmeans_amd/_lib.py) binds them to the libamdhip64 that is already in the
# process (PyTorch's bundled copy when torch is imported, /opt/rocm otherwise), so that device
# pointers handed over from torch tensors belong to the same runtime.
set -euo pipefail
cd "$(dirname "$0")"
python3 gen_assign_steps.py
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function \
    -c api.hip -o api.o "$@"
g++ -shared -o ../libspkm.so api.o -Wl,-z,undefs
rm -f api.o
