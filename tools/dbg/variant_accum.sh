cp modkit_amd/csrc/libmkpileup.so /tmp/orig.so
for T in 0 960; do echo "main skiprows tile=$T"; MKP_DEBUG_SKIP=4 MKP_TILE=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"tiles": [0-9]*\|"kernel_ms": {[^}]*}'; done
cp tools/dbg/variants/lib_accum.so modkit_amd/csrc/libmkpileup.so
for T in 960 0; do echo "accum-only 8 waves/SIMD tile=$T"; MKP_TILE=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"tiles": [0-9]*\|"kernel_ms": {[^}]*}'; done
cp /tmp/orig.so modkit_amd/csrc/libmkpileup.so
