#!/bin/bash
# Developer tool: identity of the device code inside libm3t_hip.so (nothing of the host side).
#   bash tools/device_code_id.sh [library]            md5 of the gfx950 code object
#   bash tools/device_code_id.sh --kernels [library]  md5 of every kernel's disassembly (addresses stripped), one per line
# profiles/README.md quotes these for the library the round's profiles were collected with: commits after the collection
# that leave a kernel's line unchanged did not touch that kernel.
set -e
mode=object
if [ "$1" = "--kernels" ]; then mode=kernels; shift; fi
lib=${1:-$(dirname "$0")/../3dobjecttracking_amd/csrc/libm3t_hip.so}
lib=$(readlink -f "$lib")
bin=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
"$bin/llvm-objcopy" --dump-section .hip_fatbin="$tmp/fat.bin" "$lib"
"$bin/clang-offload-bundler" --unbundle --type=o --input="$tmp/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
  --output="$tmp/dev.co"
if [ $mode = object ]; then
  md5sum "$tmp/dev.co" | cut -c1-32
else
  "$bin/llvm-objdump" -d --no-show-raw-insn --no-leading-addr "$tmp/dev.co" > "$tmp/dev.s"
  python3 - "$tmp/dev.s" <<'PY'
import hashlib, re, sys
name, body, out, after_getpc = None, [], {}, 0
for line in open(sys.argv[1]):
    m = re.match(r"^<([^>]+)>:", line) or re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
    if m:
        if name: out[name] = body
        name, body = m.group(1), []
    elif name is not None:
        # branch targets are printed as absolute addresses: keep the instruction, drop the address comment; the
        # literal added to a fresh s_getpc_b64 (a pc-relative address of a function or table elsewhere in the code
        # object) moves with the layout of the object and says nothing about this kernel's code
        text = re.sub(r"\s*//.*$", "", line.rstrip())
        if "s_getpc_b64" in text:
            after_getpc = 3
        elif after_getpc > 0:
            after_getpc -= 1
            text = re.sub(r"^(\s*s_addc?_u32 s\d+, s\d+, )0x[0-9a-f]+$", r"\1<pc-relative>", text)
        body.append(text)
if name: out[name] = body
for k in sorted(out):
    if k.startswith("__") or k.startswith("L"):  # local labels
        continue
    print("%s  %6d lines  %s" % (hashlib.md5("\n".join(out[k]).encode()).hexdigest(), len(out[k]), k))
PY
fi
