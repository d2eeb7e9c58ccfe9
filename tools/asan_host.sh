#!/bin/bash
# CPU (no GPU needed): the HOST side of libmcalf_hip.so built with AddressSanitizer + UndefinedBehaviorSanitizer, and the part of the C ABI
# that is reachable without a device -- argument checks of every entry, mcalf_stream_partition, the configuration snapshot -- run under them.
# (GPU AddressSanitizer is not available on this pool.)   tools/asan_host.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd); out=${TMPDIR:-/tmp}/mcalf_asan; mkdir -p "$out"
cd "$root/mc-alf_amd/csrc"
[ -f obj/kernels.o ] || python3 ../build.py > /dev/null
clangxx=/opt/rocm/lib/llvm/bin/clang++
for f in host_abi host_stream host_config host_multi broker comm; do
  $clangxx -x c++ -O1 -g -std=c++17 -fPIC -fvisibility=hidden -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -DMCALF_SRC_HASH='"asan"' \
    -fsanitize=address,undefined -fno-omit-frame-pointer -c $f.cpp -o "$out/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o "$out/libmcalf_asan.so" obj/kernels.o "$out"/*.o -ldl
rt=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan*x86_64*.so' | head -1)
cd "$root"
MCALF_HIP_LIB="$out/libmcalf_asan.so" LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 -m pytest tests/test_abi_symbols.py tests/test_host_logic.py tests/test_adapters.py tests/test_host_analysis.py -q -m "not gpu" -k "not carries_no_failure"
