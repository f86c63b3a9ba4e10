#!/usr/bin/env bash
# Build libsisr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libsisr_hip.so
SRCS=(conv3x3_mfma.hip wgrad3x3_mfma.hip conv3x3_small.hip attention.hip misc.hip han.hip san.hip degrade.hip diag.hip)
newest=$(ls -t "${SRCS[@]}" sisr_common.h build.sh | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then exit 0; fi
mkdir -p ../_build
objs=()
pids=()
for s in "${SRCS[@]}"; do
  o=../_build/${s%.hip}.o
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ sisr_common.h -nt "$o" ]; then
    "$HIPCC" -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $OUT"
