#!/bin/bash
# Build libopenglottal_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libopenglottal_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -o "$OUT" "$HERE/og_api.hip" "$@"
echo "built $OUT"
