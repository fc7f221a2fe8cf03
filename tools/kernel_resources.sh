#!/bin/bash
# kernel_resources.sh OBJECT... -- per-kernel VGPR / AGPR / scratch / LDS of the gfx950 code object embedded in each
# host object (or the linked .so): the "does it spill" check run after every kernel change.
set -e
LLVM=/opt/rocm/lib/llvm/bin
for obj in "$@"; do
  tmp=$(mktemp -d)
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" "$tmp/fatbin"
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fatbin" --output="$tmp/co" --unbundle
  $LLVM/llvm-readelf --notes "$tmp/co" | awk -v obj="$(basename "$obj")" '
    /\.agpr_count:/ {a=$NF} /\.group_segment_fixed_size:/ {l=$NF} /\.name:/ {n=$NF} /\.private_segment_fixed_size:/ {s=$NF}
    /\.vgpr_count:/ {v=$NF; printf "%s scratch=%s vgpr=%s agpr=%s lds=%s %s\n", obj, s, v, a, l, n}' | c++filt | sed "s/(mcp::SimParams[^)]*)//"
  rm -rf "$tmp"
done
