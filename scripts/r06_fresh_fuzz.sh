#!/bin/bash
# The LIVE differential tests (product parsers against the reference's own libraries of oracle/_ref) on streams no earlier run has seen:
# LILLIPUT_FUZZ_SEED_OFFSET moves their seeds (tests/conftest.py fresh_seed). With scripts/r06_asan.sh build done first, the product side
# is the AddressSanitizer + UBSan build.   scripts/r06_fresh_fuzz.sh FROM TO   (profiles/r06_sanitizers.md)
R=$(cd $(dirname $0)/.. && pwd); cd $R
T="tests/test_pxm.py::test_damaged_files_decode_like_the_reference_decoder_live tests/test_bmp.py::test_arbitrary_rle_streams_and_bit_field_masks_live tests/test_png.py::test_accept_reject_matches_libpng_live tests/test_gif.py::test_host_reader_matches_giflib_live tests/test_meta.py::test_readers_match_the_reference_libraries_live tests/test_webp.py::test_decoder_matches_the_reference_library_live tests/test_inflate.py::test_mutated_streams_never_accept_what_zlib_rejects"
for k in $(seq $1 $2); do
  if [ -f /tmp/asan_build/liblilliput_hip_asan.so ]; then
    LILLIPUT_FUZZ_SEED_OFFSET=$((k * 1000)) bash scripts/r06_asan.sh cpu $T 2>&1 | tail -4 | sed "s/^/offset $((k * 1000)) (asan): /"
  else
    LILLIPUT_FUZZ_SEED_OFFSET=$((k * 1000)) python -m pytest $T -q -p no:cacheprovider 2>&1 | tail -4 | sed "s/^/offset $((k * 1000)): /"
  fi
done
