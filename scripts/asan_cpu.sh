#!/bin/bash
# CPU-side sanitizer pass (ASan + UBSan): the oracle behind its Python tests, and the C++ mirror's host-only
# code (YAML loader, .npz writer/reader) as a sanitized executable.  GPU sanitizers do not exist on this pool.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/oracle asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
echo "== oracle under ASan/UBSan =="
LD_PRELOAD="$ASAN_LIB $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  FDM_REF_LIB=$R/oracle/_build/libfdm_ref_asan.so \
  python -m pytest $R/tests/test_oracle_reference_spec.py $R/tests/test_oracle_grid.py $R/tests/test_oracle_post_spec.py \
     $R/tests/test_oracle_raycast_spec.py $R/tests/test_oracle_egress_spec.py $R/tests/test_oracle_ingest_spec.py \
     $R/tests/test_golden.py -q -m "not gpu" -p no:cacheprovider
echo "== C++ mirror host-only code under ASan/UBSan =="
make -s -C $R/fastdem_amd/cpp asan
FDM_CONFIG_DIR=$R/fastdem_amd/config ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 \
  $R/fastdem_amd/cpp/build/fdm_cpp_tests_asan ConfigLoad.
echo "== nanopcl::PointCloud4 (host-only groups) under ASan/UBSan =="
for G in ConstructorsAdd ResizeReserve PointsIsOne MetadataAnd ChannelsFollow ExtractErase IndexRange; do ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 $R/fastdem_amd/cpp/build/fdm_cpp_tests_asan PointCloud4.$G | tail -1; done
echo "== libfdm_halo host code (tile plan, route plan) under ASan/UBSan =="
make -s -C $R/fastdem_amd/csrc asan
LD_PRELOAD="$ASAN_LIB $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  FDM_HALO_LIB=$R/fastdem_amd/lib/libfdm_halo_asan.so \
  python -m pytest $R/tests/test_halo_capi.py -q -m "not gpu" -p no:cacheprovider
