#!/bin/bash
# build_variant.sh NAME FILE 'SED-EXPRESSION' [FILE 'SED-EXPRESSION' ...] - builds varscot_amd/libvsc_NAME.so from a copy of
# varscot_amd/csrc with the given edits (A/B experiments: `tools/gpu.sh ab c3 varscot_amd/libvarscot_hip.so
# varscot_amd/libvsc_NAME.so`).  The tree is not touched; the variant libraries are git-ignored (*.so).
set -e
NAME=${1:?name}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/vsc_variant.XXXXXX)
cp -r "$ROOT/varscot_amd/csrc/." "$TMP/"
while [ $# -ge 2 ]; do
    before=$(md5sum "$TMP/$1" | cut -d' ' -f1)
    sed -i -E "$2" "$TMP/$1"
    [ "$before" != "$(md5sum "$TMP/$1" | cut -d' ' -f1)" ] || { echo "build_variant.sh: '$2' changed nothing in $1" >&2; exit 1; }
    shift; shift
done
cd "$TMP"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $VARIANT_FLAGS -I"$ROOT/include" -I"$TMP" -x hip vsc_kernels.hip -x hip vsc_seed.hip \
    -x hip vsc_sort.hip -x hip vsc_api.cpp -x hip vsc_pack.cpp -x hip vsc_windows.cpp -x hip vsc_multi.cpp -pthread -ldl -shared \
    -o "$ROOT/varscot_amd/libvsc_$NAME.so"
rm -rf "$TMP"
echo "built varscot_amd/libvsc_$NAME.so"
