#!/bin/bash
# Reproduce the audit of profiles/r02_epilogue_fence_audit.md (the store-data hazard behind round 1's "fence").
#   tools/epilogue_fence_audit.sh isa     # CPU: hazard scan of the shipped ISA and of the build without the wait state
#   tools/epilogue_fence_audit.sh build   # CPU: gpurun_out/libopenglottal_hip_nonop.so = the library WITHOUT the wait state (-DOG_STORE_NOP=0)
#   tools/epilogue_fence_audit.sh probe   # GPU box: one correctness probe of each (tools/store_hazard_probe.py)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRC="$HERE/openglottal_amd/csrc/og_api.hip"
case "${1:-isa}" in
isa)
    T="$(mktemp -d)"
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$SRC" -o "$T/shipped.s" 2>/dev/null
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DOG_STORE_NOP=0 "$SRC" -o "$T/nonop.s" 2>/dev/null
    echo "== shipped (one wait state behind every 16-byte buffer store)"; python3 "$HERE/tools/isa_store_hazard.py" "$T/shipped.s" || true
    echo "== -DOG_STORE_NOP=0"; python3 "$HERE/tools/isa_store_hazard.py" "$T/nonop.s" || true
    ;;
build)
    bash "$HERE/openglottal_amd/csrc/build.sh" -DOG_STORE_NOP=0 -o "$HERE/gpurun_out/libopenglottal_hip_nonop.so" > /dev/null 2>&1
    ls -la "$HERE"/gpurun_out/libopenglottal_hip_nonop.so
    ;;
probe)
    OPENGLOTTAL_HIP_ALLOW_AUDIT_BUILD=1 OPENGLOTTAL_HIP_LIB="$HERE/gpurun_out/libopenglottal_hip_nonop.so" timeout -k 10 120 python3 "$HERE/tools/store_hazard_probe.py" 2>&1 | grep -v amdgpu.ids || true
    timeout -k 10 120 python3 "$HERE/tools/store_hazard_probe.py" 2>&1 | grep -v amdgpu.ids
    ;;
esac
