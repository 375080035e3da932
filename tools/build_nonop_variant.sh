#!/bin/bash
# Experiment: the C-ABI library with the compiler's `s_nop 0` after packed instructions removed from the given kernel sources
# (tools/strip_nops.py), as tools/variants/lib_<name>.so, for A/B timing + parity on one GPU box.
# usage: tools/build_nonop_variant.sh <name> <file.hip> [file.hip ...]   (files relative to paragraph_amd/csrc)
set -e
N=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d); L=/opt/rocm/lib/llvm/bin
mkdir -p $R/tools/variants
cd $R/paragraph_amd/csrc
OBJS=""
for f in *.hip; do
  if [[ " $* " == *" $f "* ]]; then
    b=${f%.hip}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o $T/$b.s $f 2>/dev/null
    python3 $R/tools/strip_nops.py $T/$b.s $T/$b.nonop.s
    $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/$b.nonop.s -o $T/$b.dev.o
    $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/$b.hsaco $T/$b.dev.o
    $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/$b.hsaco -output=$T/$b.hipfb
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/$b.hipfb -c $f -o $T/$b.o 2>/dev/null
  else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o $T/${f%.hip}.o 2>/dev/null
  fi
  OBJS="$OBJS $T/${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/tools/variants/lib_$N.so $OBJS
rm -rf $T; ls -la $R/tools/variants/lib_$N.so
