#!/bin/bash
# tools/build_variant.sh NAME "-DSK_D=16 -DSK_DH=8": a libgfship variant with other compile-time
# parameters of the pipelined sweep, as gerris-fft-particles_amd/lib/libgfship_NAME.so (select it with
# GFSHIP_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/gerris-fft-particles_amd/csrc"; OUT="$ROOT/gerris-fft-particles_amd/lib"
NAME="$1"; EXTRA="$2"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I$ROOT/include -I$SRC -Wall -Wno-unused-function $EXTRA"
OBJS=""
for f in "$SRC"/*.hip; do
  b="$(basename "${f%.hip}")"
  case "$b" in
    relax_skew|relax_skew_loop|relax_patch_loop|timestep_kernels|tree) /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$OUT/${b}_$NAME.o" & OBJS="$OBJS $OUT/${b}_$NAME.o" ;;
    *) OBJS="$OBJS $OUT/$b.o" ;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgfship_$NAME.so" $OBJS -L/opt/rocm/lib -lhipfft -ldl
echo "built $OUT/libgfship_$NAME.so"
