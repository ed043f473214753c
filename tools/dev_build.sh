#!/bin/bash
# Incremental build of gcn_amd/lib/libccnn_hip.so for development: one object per source
# (rebuilt only when the source or a header changed), sources compiled in parallel, then linked.
# __graft_entry__.build() is the build of record (same flags, its objects under build/obj).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/gcn_amd/csrc"
OBJ="${CCNN_OBJ_DIR:-/tmp/ccnn_obj}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall $CCNN_EXTRA_FLAGS"
mkdir -p "$OBJ" "$ROOT/gcn_amd/lib"
SRCS="gather.hip forward.hip mbgd.hip tile.hip sgd.hip comm.hip host_prep.cpp profile.cpp"
newest_hdr=$(ls -t "$CSRC"/*.h "$ROOT/include/ccnn.h" | head -1)
pids=()
for s in $SRCS; do
    o="$OBJ/${s%.*}.o"
    if [ ! -f "$o" ] || [ "$CSRC/$s" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
        ( $HIPCC $FLAGS -c "$CSRC/$s" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gcn_amd/lib/libccnn_hip.so" $(for s in $SRCS; do echo "$OBJ/${s%.*}.o"; done)
echo "built $ROOT/gcn_amd/lib/libccnn_hip.so"
