#!/bin/bash
# Per-kernel VGPR / SGPR / scratch / LDS of the built library (development aid): compiles the kernels to assembly and
# prints the .amdhsa metadata of every k1_pairs instantiation (and K0).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=${1:-/tmp/icikt_kernels.s}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only -I "$ROOT/include" -I "$ROOT/icikendalltau_amd/csrc" \
  "$ROOT/icikendalltau_amd/csrc/icikt_kernels.hip" -o "$OUT" $EXTRA || exit 1
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if "k1_pairs" not in name and "k0_" not in name: continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    short = re.sub(r"_ZN5icikt", "", name)
    print(f"{short[:48]:48s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s} accum_off {g('accum_offset')}")
PY
