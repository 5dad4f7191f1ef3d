#!/bin/sh
# Compile the reference's own region layer (yolo3_frame_test_public/region_layer.{c,h}) from
# where it lies under /root/reference into oracle/_ref/libregion_ref.so.
#
# No reference source is copied or written anywhere and no stand-in header is created: the two
# files are streamed to gcc on stdin.  The only edit made on the fly is dropping two #include
# lines — `#include "kpu.h"` (region_layer.h:4; the un-vendored kendryte SDK header, from which
# region_layer.{c,h} use no symbol) and the .c file's include of its own header,
# which is already inlined by the concatenation.  Every compiled function body is the
# reference's, unmodified.  Flags match a plain x86-64 build (gnu99, -O2, no FMA contraction).
set -e
REF=${REFERENCE_ROOT:-/root/reference}/yolo3_frame_test_public
HERE=$(cd "$(dirname "$0")" && pwd)
if [ ! -f "$REF/region_layer.c" ]; then
    echo "build_ref: $REF not present (GPU box) - keeping prebuilt oracle/_ref if any"
    exit 0
fi
mkdir -p "$HERE/_ref"
{ grep -v '#include "kpu.h"' "$REF/region_layer.h"; grep -v '#include "region_layer.h"' "$REF/region_layer.c"; } \
  | gcc -x c -std=gnu99 -O2 -ffp-contract=off -w -shared -fPIC -o "$HERE/_ref/libregion_ref.so" - -lm
echo "build_ref: built $HERE/_ref/libregion_ref.so"
