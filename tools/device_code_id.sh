#!/bin/bash
# Developer tool: md5 of the gfx950 code object inside libm3t_hip.so (the device code of every kernel, nothing of the
# host side).  profiles/README.md quotes it for the library the round's profiles were collected with: host-side
# commits after the collection leave it unchanged.   usage: bash tools/device_code_id.sh [library]
set -e
lib=${1:-$(dirname "$0")/../3dobjecttracking_amd/csrc/libm3t_hip.so}
lib=$(readlink -f "$lib")
bin=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
"$bin/llvm-objcopy" --dump-section .hip_fatbin="$tmp/fat.bin" "$lib"
"$bin/clang-offload-bundler" --unbundle --type=o --input="$tmp/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
  --output="$tmp/dev.co"
md5sum "$tmp/dev.co" | cut -c1-32
