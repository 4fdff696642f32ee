#!/bin/bash
# AddressSanitizer + UBSan on the HOST code of the library only (device code objects stay uninstrumented; GPU sanitizers are not used)
set -e
cd "$(dirname "$0")/../.."
S=ray-tracing_amd/csrc
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -ffp-contract=off -munsafe-fp-atomics \
    -Wno-unused-function $S/rt_amd.hip $S/rt_build.hip $S/rt_scene.cpp tools/sanitize/asan_host.cpp -o tools/sanitize/asan_host -lpthread
echo built tools/sanitize/asan_host
