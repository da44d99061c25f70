#!/bin/bash
# builds the test suite's in-process stand-in for librccl (see mock_rccl.cpp); host code only
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
  "$HERE/mock_rccl.cpp" -o "$HERE/librccl_mock.so" -L/opt/rocm/lib -lamdhip64 -lpthread
echo "built $HERE/librccl_mock.so"
