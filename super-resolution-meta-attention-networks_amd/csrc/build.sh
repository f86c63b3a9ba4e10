#!/usr/bin/env bash
# Build libsisr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
#   build.sh        the product library (no diagnostic code, no process-wide state)
#   build.sh diag   additionally libsisr_hip_diag.so: the same sources with -DSISR_DIAG (ablation / phase-stamp kernel
#                   builds, occupancy query, MFMA-peak probe) for tools/conv_phases.py, tools/x3_phases.py, tools/mfma_peak.py
#                   -- select it with SISR_HIP_LIB=.../libsisr_hip_diag.so
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS=(conv3x3_mfma.hip wgrad3x3_mfma.hip conv3x3_small.hip attention.hip misc.hip han.hip san.hip degrade.hip sft.hip nonlocal.hip sparnet.hip)

build() {  # $1 = output, $2 = object dir, $3... = extra flags / sources
  local out=$1 odir=$2; shift 2
  local flags=() srcs=()
  for a in "$@"; do case "$a" in -D*) flags+=("$a");; *) srcs+=("$a");; esac; done
  local newest
  newest=$(ls -t "${srcs[@]}" sisr_common.h ca_gate.h conv_rgb_out.h build.sh | head -1)
  if [ -f "$out" ] && [ "$out" -nt "$newest" ]; then return 0; fi
  mkdir -p "$odir"
  local objs=() pids=()
  for s in "${srcs[@]}"; do
    local o=$odir/${s%.hip}.o
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ sisr_common.h -nt "$o" ] || [ ca_gate.h -nt "$o" ] || [ conv_rgb_out.h -nt "$o" ] || [ build.sh -nt "$o" ]; then
      "$HIPCC" -O3 -std=c++17 --offload-arch=gfx950 -fPIC "${flags[@]}" -c "$s" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
  echo "built $out"
}

build ../libsisr_hip.so ../_build "${SRCS[@]}"
if [ "${1:-}" = "diag" ]; then
  build ../libsisr_hip_diag.so ../_build_diag -DSISR_DIAG "${SRCS[@]}" diag.hip
fi
