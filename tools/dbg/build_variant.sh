#!/bin/bash
# another build of the library with extra compile flags -> tools/dbg/lib/libmkpileup_<name>.so; used through MKP_LIB_PATH (tools/dbg/ab.sh)
# usage: tools/dbg/build_variant.sh wb12k "-DMKP_SLOT_WB=12288u"
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../.."; B=/tmp/mkp_variant_$NAME; rm -rf $B; mkdir -p $B/modkit_amd tools/dbg/lib
cp -r modkit_amd/csrc $B/modkit_amd/; cp -r include $B/; rm -f $B/modkit_amd/csrc/*.o $B/modkit_amd/csrc/*.so
make -s -C $B/modkit_amd/csrc -j8 libmkpileup.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I../../include $EXTRA"
cp $B/modkit_amd/csrc/libmkpileup.so tools/dbg/lib/libmkpileup_$NAME.so; ls -la tools/dbg/lib/libmkpileup_$NAME.so
