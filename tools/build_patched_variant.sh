#!/bin/bash
# Timing-only experiments: builds the working tree's C-ABI library with a python patch applied to a COPY of the kernel
# sources (the experiment switches never enter the tree) as tools/variants/lib_<name>.so.
# usage: tools/build_patched_variant.sh <name> <patch.py>    (patch.py: reads / rewrites files under $1 = the copy of csrc)
set -e
N=$1; P=$2; R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
mkdir -p $R/tools/variants $T/paragraph_amd/csrc $T/include
cp $R/paragraph_amd/csrc/*.h $R/paragraph_amd/csrc/*.hip $T/paragraph_amd/csrc/; cp $R/include/*.h $T/include/
python3 $P $T/paragraph_amd/csrc
cd $T/paragraph_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $R/tools/variants/lib_$N.so *.hip 2>&1 | grep -E "error" -A4 || true
rm -rf $T; ls -la $R/tools/variants/lib_$N.so
