#!/bin/bash
# Build librankaae_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so it
# travels to the GPU box with the snapshot.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
OBJS=""
for f in raae_*.hip; do
  o="${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ raae_common.h -nt "$o" ] || [ ../../include/rankaae_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    $HIPCC $FLAGS -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../librankaae_hip.so $OBJS
echo "built $(realpath ../librankaae_hip.so)"
