#!/bin/bash
# Builds libgfship.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: results must match the reference's non-FMA x86-64 arithmetic bit for bit.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$HERE/../../include -I$HERE -Wall -Wno-unused-function"
OBJS=""
for f in "$HERE"/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  stale=0
  [ -f "$o" ] || stale=1
  # every header of this directory (relax_skew.hpp, gfship_internal.hpp, tree.hpp ...) and the C ABI
  for d in "$f" "$HERE"/*.hpp "$HERE/../../include/gfship.h"; do
    [ "$d" -nt "$o" ] && stale=1
  done
  if [ $stale = 1 ]; then
    "$HIPCC" $FLAGS -c "$f" -o "$o" &
  fi
  OBJS="$OBJS $o"
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgfship.so" $OBJS -L/opt/rocm/lib -lhipfft -ldl
echo "built $OUT/libgfship.so"
# host front end: the reference's gerris2D / gerris3D command, on libgfship (C ABI only)
BIN="$HERE/../bin"
mkdir -p "$BIN"
CXX="${CXX:-g++}"
if [ ! -f "$BIN/gfship2D" ] || [ "$HERE/host/gfsrun.cpp" -nt "$BIN/gfship2D" ] || [ "$HERE/host/gfs_text.hpp" -nt "$BIN/gfship2D" ] || [ "$HERE/host/gfs_snapshot.hpp" -nt "$BIN/gfship2D" ] || [ "$HERE/host/gfs_function.hpp" -nt "$BIN/gfship2D" ] || [ "$HERE/../../include/gfship.h" -nt "$BIN/gfship2D" ]; then
  "$CXX" -O2 -std=c++17 -Wall -I"$HERE/../../include" -I"$HERE/host" "$HERE/host/gfsrun.cpp" \
    -o "$BIN/gfship2D" -L"$OUT" -lgfship -ldl -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib
  cp "$BIN/gfship2D" "$BIN/gfship3D"
fi
if [ ! -f "$BIN/gfshipcompare2D" ] || [ "$HERE/host/gfscompare.cpp" -nt "$BIN/gfshipcompare2D" ] || [ "$HERE/host/gfs_snapshot.hpp" -nt "$BIN/gfshipcompare2D" ] || [ "$HERE/host/gfs_text.hpp" -nt "$BIN/gfshipcompare2D" ]; then
  "$CXX" -O2 -std=c++17 -Wall -I"$HERE/host" "$HERE/host/gfscompare.cpp" -o "$BIN/gfshipcompare2D"
  cp "$BIN/gfshipcompare2D" "$BIN/gfshipcompare3D"
fi
echo "built $BIN/gfship2D $BIN/gfship3D $BIN/gfshipcompare2D $BIN/gfshipcompare3D"
