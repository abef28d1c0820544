#!/usr/bin/env bash
# scripts/disasm.sh <object file> <kernel name substring>   -> gfx950 ISA of that kernel on stdout
L=/opt/rocm/lib/llvm/bin; d=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$d/fat.bin "$1" 2>/dev/null
$L/clang-offload-bundler --unbundle --type=o --input=$d/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$d/k.co 2>/dev/null
$L/llvm-objdump -d $d/k.co | awk -v k="$2" '$0 ~ "^[0-9a-f]+ <.*"k {p=1} p {print} p && /s_endpgm/ {exit}'
rm -rf $d
