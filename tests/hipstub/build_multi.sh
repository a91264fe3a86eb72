#!/bin/bash
# usage: build_multi.sh <asan|tsan> <outdir> - bspy_amd/csrc/bsk_multi.hip (host-only, unchanged) + the stub HIP runtime
# with HIPSTUB_DEVICES fake devices + the stub librccl.so.1 + multi_driver.cpp (stand-in single-device layer).
# Run:  HIPSTUB_DEVICES=8 LD_LIBRARY_PATH=<outdir> <outdir>/multi_driver
set -e
kind=$1; out=$2
here=$(cd "$(dirname "$0")" && pwd); src=$here/../../bspy_amd/csrc
case $kind in asan) SAN="-fsanitize=address -fno-omit-frame-pointer";; tsan) SAN="-fsanitize=thread";; *) exit 1;; esac
mkdir -p $out
HIPCC=${HIPCC:-hipcc}
$HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 --cuda-host-only $SAN -Wno-unused-function -c $src/bsk_multi.hip -o $out/bsk_multi.o
$HIPCC -O1 -g -std=c++17 $SAN -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $here/hip_stub.cpp -o $out/hip_stub.o
$HIPCC -O1 -g -std=c++17 $SAN -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -fPIC -c $here/rccl_stub.cpp -o $out/rccl_stub.o
$HIPCC -O1 -g -std=c++17 $SAN -x hip --offload-arch=gfx950 --cuda-host-only -Wno-unused-function -c $here/multi_driver.cpp -o $out/multi_driver.o
/opt/rocm/lib/llvm/bin/clang++ $SAN -shared -Wl,-soname,librccl.so.1 -o $out/librccl.so.1 $out/rccl_stub.o
/opt/rocm/lib/llvm/bin/clang++ $SAN -o $out/multi_driver $out/multi_driver.o $out/bsk_multi.o $out/hip_stub.o -L$out -l:librccl.so.1 -ldl -lpthread
