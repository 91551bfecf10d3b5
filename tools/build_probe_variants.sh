#!/bin/bash
# Development: libmira_gpu.so built from the WORKING TREE with extra preprocessor flags into tools/_variants/<name>.so
# (timing probes such as -DMIRA_PROBE_NO_TREE: the results of such a build are wrong on purpose).
# usage: tools/build_probe_variants.sh <name> "<flags>"
set -e
name=${1:?name}; flags=${2:?flags}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/mira_amd" "$tmp/include"
cp -r "$root/mira_amd/csrc" "$tmp/mira_amd/csrc"; cp "$root/include/mira_gpu.h" "$tmp/include/"
rm -f "$tmp"/mira_amd/csrc/*.o "$tmp"/mira_amd/csrc/*.so
make -s -C "$tmp/mira_amd/csrc" -j6 EXTRA="$flags" msm_bn256.o msm_grumpkin.o capi.o ntt.o fold.o graph.o libmira_gpu.so 2>&1 | grep -E " error |Stop" || true
mkdir -p "$root/tools/_variants"
cp "$tmp/mira_amd/csrc/libmira_gpu.so" "$root/tools/_variants/$name.so"
rm -rf "$tmp"
echo "built tools/_variants/$name.so with $flags"
