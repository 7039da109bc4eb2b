#!/bin/bash
# Build librankaae_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so it
# travels to the GPU box with the snapshot.
set -e
export LC_ALL=C
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $RAAE_EXTRA_FLAGS"
# every object depends on every header / .inc; the digest of all sources is baked into the library
# (raae_source_digest) so that the Python loader can refuse a stale build.
DIGEST=$(cat ../../include/rankaae_hip.h raae_*.h raae_*.inc raae_*.hip | sha256sum | cut -c1-16)
# objects are rebuilt when the sources OR the extra flags change (the digest baked into the library is of the sources)
if [ "$(cat .digest 2>/dev/null)" != "$DIGEST $RAAE_EXTRA_FLAGS" ]; then rm -f raae_*.o; echo "$DIGEST $RAAE_EXTRA_FLAGS" > .digest; fi
FLAGS="$FLAGS -DRAAE_SOURCE_DIGEST=\"$DIGEST\""
OBJS=""
for f in raae_*.hip; do
  o="${f%.hip}.o"
  if [ ! -f "$o" ]; then
    echo "hipcc $f"
    $HIPCC $FLAGS -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../librankaae_hip.so $OBJS
echo "built $(realpath ../librankaae_hip.so)"
