#!/bin/bash
# SURVEY section 5's sanitizer row, as far as this pool allows it (GPU AddressSanitizer / xnack+ code objects are refused by gpurun): the HOST side of
# libdpig_hip.so -- descriptor validation, SAME-padding geometry, split / Winograd / large-tile planners, workspace sizing, the job planner,
# error paths: everything a C-ABI call does before its launch -- built with AddressSanitizer + UndefinedBehaviorSanitizer
# (-fsanitize=address,undefined -fno-gpu-sanitize: device code is compiled as always), then the CPU test files that drive those paths
# through ctypes run against it.
#   bash scripts/build_asan.sh            ->  /tmp/dpig_asan/libdpig_hip_asan.so + the test run (profiles/r06_asan_ubsan_host.txt holds a record)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
O=${DPIG_ASAN_DIR:-/tmp/dpig_asan}; mkdir -p $O      # OUTSIDE the repository: gpurun refuses a snapshot that carries sanitizer-built objects
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined -I include -I disentangled-person-image-generation_amd/csrc"
pids=()
for f in disentangled-person-image-generation_amd/csrc/*.hip; do
  o=$O/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ include/dpig_hip.h -nt $o ]; then hipcc $FLAGS -c $f -o $o & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $O/libdpig_hip_asan.so $O/*.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
echo "built $O/libdpig_hip_asan.so; preloading $RT"
# (python itself is not instrumented: the runtime is preloaded; leak detection off -- the interpreter's own allocations are not ours)
DPIG_LIB_PATH=$O/libdpig_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_abi.py tests/test_wino_plan.py tests/test_host_logic.py tests/test_host_sweep.py -q -m "not gpu" -p no:cacheprovider "$@"
