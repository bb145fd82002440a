cp modkit_amd/csrc/libmkpileup.so /tmp/orig.so
for U in 2 8; do cp tools/dbg/variants/lib_u$U.so modkit_amd/csrc/libmkpileup.so; echo "unroll=$U"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"kernel_ms": {[^}]*}'; done
cp /tmp/orig.so modkit_amd/csrc/libmkpileup.so
echo "unroll=4 tiles"; for T in 1024 1536; do MKP_TILE=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"tiles": [0-9]*\|"kernel_ms": {[^}]*}'; done
