#!/bin/bash
# the -DMKP_DEBUG build of the library (MKP_DEBUG_SKIP ablations) -> tools/dbg/lib/libmkpileup_debug.so; used through MKP_LIB_PATH
set -e
cd "$(dirname "$0")/../.."; B=/tmp/mkp_debug_build; rm -rf $B; mkdir -p $B/modkit_amd tools/dbg/lib
cp -r modkit_amd/csrc $B/modkit_amd/; cp -r include $B/; rm -f $B/modkit_amd/csrc/*.o $B/modkit_amd/csrc/*.so
make -s -C $B/modkit_amd/csrc -j8 libmkpileup.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I../../include -DMKP_DEBUG"
cp $B/modkit_amd/csrc/libmkpileup.so tools/dbg/lib/libmkpileup_debug.so; ls -la tools/dbg/lib/libmkpileup_debug.so
