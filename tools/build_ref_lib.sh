#!/bin/bash
# Build the kernel library of another revision (default HEAD) into ab_ref/libstonk_hip.so, for an A/B of two builds on one
# box: STONK_HIP_LIB=ab_ref/libstonk_hip.so python tools/...  (ab_ref/ is not tracked; the .so travels with gpurun).
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" stonkgs_amd/csrc include Makefile tools/gen_gemm_a4.py | tar -x -C "$tmp"
make -C "$tmp" -j8 >/dev/null
mkdir -p "$root/ab_ref"
cp "$tmp/stonkgs_amd/csrc/libstonk_hip.so" "$root/ab_ref/libstonk_hip.so"
rm -rf "$tmp"
echo "ab_ref/libstonk_hip.so <- $rev"
