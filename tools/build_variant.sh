#!/bin/bash
# Development: build libmira_gpu.so from the COMMITTED sources (HEAD) into tools/_variants/<name>.so, for
# same-box A/B runs against the working tree's library (probes take MIRA_PROBE_LIB).
# usage: tools/build_variant.sh <name> [git-ref]
set -e
name=${1:?name}; ref=${2:-HEAD}
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" mira_amd/csrc include | tar -x -C "$tmp"
make -s -C "$tmp/mira_amd/csrc" -j6 2>&1 | grep -E " error |Stop" || true
mkdir -p "$root/tools/_variants"
cp "$tmp/mira_amd/csrc/libmira_gpu.so" "$root/tools/_variants/$name.so"
rm -rf "$tmp"
echo "built tools/_variants/$name.so from $ref"
