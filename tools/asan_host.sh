#!/bin/bash
# AddressSanitizer + UBSan build of the host side (ingest, focus builder, packer, driver) and a dry run
# (`--plan-only`: no device) over the reference's fixture BAMs.  Usage: tools/asan_host.sh [more.bam ...]
set -e
cd "$(dirname "$0")/../modkit_amd/csrc"
make -s mkp_kernels.o mkp_slots.o mkp_inflate_wave4.o mkp_ingest.o
H="/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
$H -c mkp_api.cpp -o /tmp/asan_api.o; $H -c mkp_driver.cpp -o /tmp/asan_drv.o; $H -DMKP_INGEST_HOST_SHIM -c mkp_ingest_host.cpp -o /tmp/asan_ing.o; gcc -c -O1 -I../../include mkpileup_cli.c -o /tmp/asan_cli.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined -o /tmp/mkpileup_asan /tmp/asan_cli.o /tmp/asan_api.o /tmp/asan_drv.o /tmp/asan_ing.o mkp_kernels.o mkp_slots.o mkp_inflate_wave4.o mkp_ingest.o -lz -lpthread -ldl
F=../../tests/golden/modkit_fixtures
for b in $F/*.bam "$@"; do
  for fl in "" "--cpg --ref $F/CGI_ladder_3.6kb_ref.fa -i 37" "--cpg --combine-strands --ref $F/CGI_ladder_3.6kb_ref.fa -i 41" "--include-bed $F/CGI_ladder_3.6kb_ref_include_positions.bed"; do
    ASAN_OPTIONS=detect_leaks=0 /tmp/mkpileup_asan pileup $b /tmp/asan_plan.tsv --plan-only $fl 2>&1 | grep -E "ERROR: AddressSanitizer|runtime error" && exit 1
  done
done
# the all-cores packer (per-thread pieces + ShardHost::append_all), forced on for every batch size
for b in $F/*.bam "$@"; do
  ASAN_OPTIONS=detect_leaks=0 /tmp/mkpileup_asan pileup $b /tmp/asan_plan.tsv --plan-only --plan-pack-min 0 2>&1 | grep -E "ERROR: AddressSanitizer|runtime error" && exit 1
done
echo "asan/ubsan: clean"
