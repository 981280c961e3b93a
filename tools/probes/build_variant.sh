#!/bin/bash
# build_variant.sh <name> [-DFLAG ...]: a complete library with gn.hip compiled with the extra flags, into
# tools/probes/_build/lib_<name>.so (git-ignored; it travels to the GPU box with gpurun).  For A/B runs only.
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p tools/probes/_build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off "$@" -c dex-ct-sim_amd/csrc/${SRC:-gn}.hip -o tools/probes/_build/${SRC:-gn}_$name.o
objs=$(ls dex-ct-sim_amd/csrc/_build/*.o | grep -v "/${SRC:-gn}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/_build/lib_$name.so tools/probes/_build/${SRC:-gn}_$name.o $objs -ldl
echo built tools/probes/_build/lib_$name.so
