#!/bin/bash
# builds variant libraries lib/libpgcn_hip.<tag>.so with extra -D flags:  ab_build.sh tag "-DX=1" ...
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
PKG="$HERE/scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
tag="$1"; shift
PGCN_EXTRA_FLAGS="$*" bash "$PKG/csrc/build.sh" > /dev/null
cp "$PKG/lib/libpgcn_hip.so" "$PKG/lib/libpgcn_hip.$tag.so"
echo "built variant $tag ($*)"
