#!/bin/bash
# usage: build.sh <asan|tsan> <outdir> - host half of bspy_amd/csrc/bsk_api.hip + stub runtime + driver under a sanitizer
set -e
kind=$1; out=$2
here=$(cd "$(dirname "$0")" && pwd); src=$here/../../bspy_amd/csrc
case $kind in asan) SAN="-fsanitize=address -fno-omit-frame-pointer";; tsan) SAN="-fsanitize=thread";; *) exit 1;; esac
mkdir -p $out
HIPCC=${HIPCC:-hipcc}
[ -f $out/bsk_api.o -a $out/bsk_api.o -nt $src/bsk_api.hip ] || $HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 --cuda-host-only $SAN -Wno-unused-function -c $src/bsk_api.hip -o $out/bsk_api.o
[ -f $out/bsk_multi.o -a $out/bsk_multi.o -nt $src/bsk_multi.hip ] || $HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 --cuda-host-only $SAN -Wno-unused-function -c $src/bsk_multi.hip -o $out/bsk_multi.o
# stub librccl.so.1 for the gathering multi-device calls (found through LD_LIBRARY_PATH=<outdir>)
$HIPCC -O1 -g -std=c++17 $SAN -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -fPIC -c $here/rccl_stub.cpp -o $out/rccl_stub.o
/opt/rocm/lib/llvm/bin/clang++ $SAN -shared -Wl,-soname,librccl.so.1 -o $out/librccl.so.1 $out/rccl_stub.o
# the fat-binary symbol the host object refers to is named after a hash of the unit: define it in the stub
fat=$(nm -u $out/bsk_api.o | awk '/__hip_fatbin_/ {print $2}' | head -1)
echo "extern \"C\" { extern const char ${fat:-__hip_fatbin_unused}[16]; const char ${fat:-__hip_fatbin_unused}[16] = {0}; }" > $out/fatbin.cpp
$HIPCC -O1 -g -std=c++17 $SAN -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $here/hip_stub.cpp -o $out/hip_stub.o
$HIPCC -O1 -g -std=c++17 $SAN -x hip --offload-arch=gfx950 --cuda-host-only -Wno-unused-function -c $here/tu_stubs.cpp -o $out/tu_stubs.o
$HIPCC -O1 -g -std=c++17 $SAN -x c++ -c $out/fatbin.cpp -o $out/fatbin.o
$HIPCC -O1 -g -std=c++17 $SAN -x c++ -c $here/driver.cpp -o $out/driver.o
/opt/rocm/lib/llvm/bin/clang++ $SAN -o $out/driver $out/driver.o $out/bsk_api.o $out/bsk_multi.o $out/hip_stub.o $out/tu_stubs.o $out/fatbin.o -ldl -lpthread
