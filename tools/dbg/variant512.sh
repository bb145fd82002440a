cp modkit_amd/csrc/libmkpileup.so /tmp/orig.so
for T in 0 1024 1536; do echo "1024thr tile=$T"; MKP_TILE=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"tiles": [0-9]*\|"kernel_ms": {[^}]*}'; done
cp tools/dbg/variants/lib512.so modkit_amd/csrc/libmkpileup.so
for T in 960 1024 1536 0; do echo "512thr tile=$T"; MKP_TILE=$T python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"tiles": [0-9]*\|"kernel_ms": {[^}]*}'; done
cp /tmp/orig.so modkit_amd/csrc/libmkpileup.so
