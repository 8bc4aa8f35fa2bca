#!/bin/bash
# Registers, LDS, scratch and spill counts of every kernel (device-only compile to assembly; no GPU needed).
#   bash scripts/kernel_resources.sh [file.hip ...]
cd "$(dirname "$0")/.." || exit 1
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(pc-gnn_amd/csrc/*.hip)
for f in "${files[@]}"; do
  s=/tmp/$(basename "$f" .hip).s
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipc-gnn_amd/csrc --offload-device-only -S -o "$s" "$f" 2>/dev/null || { echo "compile failed: $f"; continue; }
  python3 - "$s" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    print("%-70s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill v%s s%s" % (g("name")[:70], g("vgpr_count"), g("agpr_count"),
          g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
PY
done
