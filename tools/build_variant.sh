#!/bin/bash
# experiment builds: tools/build_variant.sh NAME [-DFLAG ...]  ->  build_tmp/libspkm_NAME.so (never shipped; a GPU-box
# script copies it over sparsifiedkmeans_amd/libspkm.so inside the box's scratch copy of the repo)
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $root/sparsifiedkmeans_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -c api.hip -o $root/build_tmp/api_$name.o "$@"
g++ -shared -o $root/build_tmp/libspkm_$name.so $root/build_tmp/api_$name.o -Wl,-z,undefs
rm -f $root/build_tmp/api_$name.o
