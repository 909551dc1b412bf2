#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the library (the parsers, the ABI layer, engine / batch / coalescer host code;
# the device code objects are the shipped ones -- device ASan is not available on this pool) and the CPU test suite on it.
#   scripts/r06_asan.sh build        -> /tmp/asan_build/liblilliput_hip_asan.so
#   scripts/r06_asan.sh cpu [tests]  -> pytest -m "not gpu" with LILLIPUT_HIP_LIB = that library (profiles/r06_sanitizers.md)
#   scripts/r06_asan.sh ubsan        -> /tmp/asan_build/liblilliput_hip_ubsan.so (UBSan only: the variant that runs on the GPU box -- ROCm's ASan runtime
#                                       intercepts hsa_amd_memory_pool_allocate and needs the instrumented runtime of /opt/rocm/lib/asan, absent from this image)
#   scripts/r06_asan.sh gpu [tests]  -> pytest -m gpu with lilliput_amd/liblilliput_hip_ubsan.so (copy it there for ONE gpurun call, remove it afterwards)
set -e
R=$(cd $(dirname $0)/.. && pwd); B=/tmp/asan_build
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
case "$1" in
build)
  rm -rf $B && mkdir -p $B/lilliput_amd $B/include && cp -r $R/lilliput_amd/csrc $B/lilliput_amd/ && cp $R/include/*.h $B/include/
  cd $B/lilliput_amd/csrc && ls *.o 2>/dev/null | grep -v lp_kernels | xargs rm -f
  SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
  FLAGS="-O1 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off"
  for f in *.cpp; do
    ( /opt/rocm/bin/hipcc $FLAGS $SAN -x hip -c $f -o ${f%.cpp}.o 2> ${f%.cpp}.log || echo "FAILED $f" ) &
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
  done
  wait
  for f in lp_kernels_*.hip; do [ -f ${f%.hip}.o ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c $f -o ${f%.hip}.o; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -shared-libsan -o $B/liblilliput_hip_asan.so *.o -lz -l:libwebp.so.7
  ls -la $B/liblilliput_hip_asan.so ;;
ubsan)
  cd $B/lilliput_amd/csrc && mkdir -p ub
  SAN="-fsanitize=undefined,float-cast-overflow -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
  FLAGS="-O1 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off"
  for f in *.cpp; do
    ( /opt/rocm/bin/hipcc $FLAGS $SAN -x hip -c $f -o ub/${f%.cpp}.o 2> ub/${f%.cpp}.log || echo "FAILED $f" ) &
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
  done
  wait
  cp lp_kernels_*.o ub/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -shared-libsan -o $B/liblilliput_hip_ubsan.so ub/*.o -lz -l:libwebp.so.7
  ls -la $B/liblilliput_hip_ubsan.so ;;
cpu|gpu)
  M="not gpu"; L=$B/liblilliput_hip_asan.so
  if [ "$1" = gpu ]; then M=gpu; L=$R/lilliput_amd/liblilliput_hip_ubsan.so; RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so); fi
  shift
  cd $R
  # reports go to files: a report that ends the process inside a test would be lost with pytest's captured stderr
  O=${ASAN_OUT:-/tmp/asan_build}/reports; mkdir -p $O
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:log_path=$O/asan UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$O/ubsan LILLIPUT_HIP_LIB=$L \
    python -m pytest ${@:-tests} -q -m "$M" -p no:cacheprovider ;;
*) echo "usage: $0 build | cpu [tests] | gpu [tests]"; exit 2 ;;
esac
