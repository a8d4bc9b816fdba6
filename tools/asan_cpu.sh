#!/bin/bash
# The host libraries (liblentil_host.so, liblentil_bridge.so, lentil.so and the stand-in renderer) rebuilt with
# AddressSanitizer + UBSan, the CPU test suite run against them, then the normal builds restored.  No GPU needed
# (GPU sanitizers are not available on the pool).  usage: tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
g++ -std=c++17 $SAN -ffp-contract=off -fPIC -shared -fvisibility=hidden -o pota_amd/liblentil_host.so pota_amd/csrc/host/lentil_host.cpp
g++ -std=c++17 $SAN -fPIC -shared -fvisibility=hidden -I include -o pota_amd/liblentil_bridge.so pota_amd/csrc/host/lentil_bridge.cpp \
    -L pota_amd -llentil_hip -Wl,-rpath,'$ORIGIN' -lpthread
g++ -std=c++17 $SAN -fPIC -shared -fvisibility=hidden -o tests/fake_arnold/libai_fake.so tests/fake_arnold/fake_arnold.cpp -ldl -lpthread
g++ -std=c++17 $SAN -fPIC -shared -fvisibility=hidden -I include -o pota_amd/lentil.so pota_amd/csrc/plugin/*.cpp -L pota_amd \
    -llentil_bridge -llentil_host -llentil_hip -Wl,-rpath,'$ORIGIN' -lpthread -I tests/fake_arnold -L tests/fake_arnold -lai_fake \
    -Wl,-rpath,'$ORIGIN/../tests/fake_arnold'
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 \
    UBSAN_OPTIONS=print_stacktrace=1 python -m pytest tests -q -s -m "not gpu" > /tmp/lentil_asan.log 2>&1 || true
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/lentil_asan.log)"; tail -1 /tmp/lentil_asan.log
python -c "import __graft_entry__ as g; g.build(force=True)"
