#!/bin/bash
# tools/build_isa_patched.sh <name> [extra hipcc flags...] -> lib/libofdg_<name>.so whose device code went through an assembly
# pass: v_cndmask_b32 in its VOP2 form (implicit VCC) re-encoded as VOP3 (tools/microbench/valu_rates.hip: the VOP2 form
# issues 4 - 7 x slower unless it directly follows the v_cmp that wrote VCC).
set -e
name=$1; shift
cd "$(dirname "$0")/../optical-flow-2d-data-generation_amd"
B=build/isa_$name; mkdir -p $B
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $FLAGS "$@" --cuda-device-only -S csrc/ofdg_api.hip -o $B/dev.s
if [ "${PATCH:-1}" = 1 ]; then sed -E 's/^(\s*)v_cndmask_b32_e32 (.*), vcc\s*$/\1v_cndmask_b32_e64 \2, vcc/' $B/dev.s > $B/dev_p.s; else cp $B/dev.s $B/dev_p.s; fi
echo "cndmask e32 -> e64: $(grep -c 'v_cndmask_b32_e32' $B/dev.s) before, $(grep -c 'v_cndmask_b32_e32' $B/dev_p.s) after"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/dev_p.s -o $B/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/dev.out $B/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$B/dev.out -output=$B/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/dev.hipfb -c csrc/ofdg_api.hip -o $B/ofdg_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o lib/libofdg_$name.so $B/ofdg_api.o build/realize.o build/sampler_ref.o build/layer.o build/warpfields.o build/comm.o -ldl
echo "built lib/libofdg_$name.so"
