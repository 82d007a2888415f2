#!/bin/bash
# usage: tools/disasm.sh conv|conv5|nnops ... -> /tmp/dis/<name>.s (gfx950 ISA of the built object)
set -e
L=/opt/rocm/lib/llvm/bin
mkdir -p /tmp/dis
for n in "$@"; do
  cp "$(dirname "$0")/../multibox_amd/csrc/_obj/$n.o" /tmp/dis/$n.o
  $L/llvm-objcopy --dump-section .hip_fatbin=/tmp/dis/$n.fatbin /tmp/dis/$n.o
  $L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/dis/$n.fatbin --output=/tmp/dis/$n.co --unbundle
  $L/llvm-objdump -d /tmp/dis/$n.co > /tmp/dis/$n.s
  echo /tmp/dis/$n.s
done
