#!/bin/bash
# usage (on the GPU box): tools/with_variant.sh <name> <command...>   -- runs the command with tools/_var/libstattn_<name>.so in the
# place of the product library, and puts the product library back afterwards (also when the command fails)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/video-description-with-spatial-temporal-attention_amd
n=$1; shift
[ -f $ROOT/tools/_var/libstattn_$n.so ] || { echo "no variant $n"; exit 2; }
cp $PKG/libstattn.so $ROOT/tools/_var/_product.so
cp $ROOT/tools/_var/libstattn_$n.so $PKG/libstattn.so
"$@"; rc=$?
cp $ROOT/tools/_var/_product.so $PKG/libstattn.so
exit $rc
