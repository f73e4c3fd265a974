# Host-side AddressSanitizer over the CPU-runnable part of the C ABI (argument checking, state machine, error paths, host helpers):
# `make asan` compiles the HOST code with ASan (-Xarch_host -fsanitize=address; device code as always) and tests/test_abi.py +
# the host-gather / unshuffle tests run against it.  On a GPU box the same build cannot be used: the ROCm build of the ASan
# runtime intercepts hsa_amd_memory_pool_allocate for GPU-ASan (which this pool does not offer) and aborts at the first device
# allocation ("out of memory: allocator is trying to allocate 0x400000 bytes", measured in round 4).
set -e
cd "$(dirname "$0")/.."
make -C nemoflux_amd/csrc asan -j8 -s
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$RT NEMOFLUX_AMD_LIB=$PWD/build/asan/libnemoflux_amd_asan.so \
  python -m pytest tests/test_abi.py tests/test_hdf5min.py -x -q -p no:cacheprovider
