#!/bin/bash
# tools/build_rev.sh <git revision> <name>  -> optical-flow-2d-data-generation_amd/lib/libofdg_<name>.so built from that revision's
# sources (same-box A/B against the working tree: OFDG_LIB=.../libofdg_<name>.so, see tools/ab.sh)
set -e
rev=$1; name=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
pkg=optical-flow-2d-data-generation_amd
tmp=${TMPDIR:-/tmp}/ofdg_rev_$name   # (outside the package: the snapshot that travels to the GPU box holds no second copy of the sources)
rm -rf $tmp; mkdir -p $tmp
git -C $root archive $rev $pkg/csrc include | tar -x -C $tmp
cd $tmp/$pkg
objs=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16 -c csrc/ofdg_api.hip -o ofdg_api.o
for f in realize sampler_ref layer warpfields comm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c csrc/$f.cpp -o $f.o; objs="$objs $f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $root/$pkg/lib/libofdg_$name.so ofdg_api.o $objs -ldl
echo "built $pkg/lib/libofdg_$name.so from $rev"
