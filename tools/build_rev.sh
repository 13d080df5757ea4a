#!/bin/bash
# Build the library of another git revision next to the product, for same-box A/B runs (tools/ab.sh):
#   bash tools/build_rev.sh <rev> [outdir-name, default lib_prev]   -> rustradio_amd/<outdir>/librustradio_amd.so
set -e
REV=${1:-HEAD}; NAME=${2:-lib_prev}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/rr_rev_$NAME; rm -rf $T; mkdir -p $T
git -C "$ROOT" archive "$REV" rustradio_amd/csrc include | tar -x -C $T
make -C $T/rustradio_amd/csrc -j8 OUT="$ROOT/rustradio_amd/$NAME" > /dev/null
ls -la "$ROOT/rustradio_amd/$NAME/librustradio_amd.so"
