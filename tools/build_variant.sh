#!/bin/bash
# Builds the C-ABI library of another commit (or of the working tree: "WORK") as tools/variants/lib_<name>.so for A/B timing on
# ONE GPU box (boxes differ by a few percent).  usage: tools/build_variant.sh <commit|WORK> <name>
set -e
C=$1; N=$2; shift 2; X="$@"; R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
mkdir -p $R/tools/variants $T/paragraph_amd/csrc $T/include
if [ "$C" = WORK ]; then cp $R/paragraph_amd/csrc/* $T/paragraph_amd/csrc/; cp $R/include/*.h $T/include/
else for f in $(git -C $R ls-tree --name-only $C paragraph_amd/csrc/ include/); do git -C $R show $C:$f > $T/$f; done; fi
cd $T/paragraph_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $X -fPIC -shared -o $R/tools/variants/lib_$N.so *.hip
rm -rf $T; ls -la $R/tools/variants/lib_$N.so
